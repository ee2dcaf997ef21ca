"""cv2 stand-in (TEST INFRASTRUCTURE): misc_functions.py and data/Imagenet.py import cv2 at module level; the code paths
the tests run never call it.  Any attribute access fails loudly."""


def __getattr__(name):
    raise AttributeError(f"stub cv2 has no {name!r}: the scripts under test must not reach OpenCV")
