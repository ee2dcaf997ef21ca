"""skimage stand-in (TEST INFRASTRUCTURE): utils/render.py imports skimage.io / .feature / .filters at module level;
rendering (--save-img) is not exercised."""
