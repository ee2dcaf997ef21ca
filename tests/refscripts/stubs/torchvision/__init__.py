"""Minimal torchvision stand-in (TEST INFRASTRUCTURE, see tests/refscripts/README.md)."""
from . import datasets, transforms  # noqa: F401

__version__ = "0.0-stub"
