"""imageio stand-in (TEST INFRASTRUCTURE): imsave / imwrite through PIL (imagenet_seg_eval.py --save-img only)."""
import numpy as np
from PIL import Image


def imwrite(path, arr, **kwargs):
    Image.fromarray(np.asarray(arr)).save(path)


imsave = imwrite
