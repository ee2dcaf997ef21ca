"""transforms.Compose / Resize / ToTensor / Normalize over PIL images and torch tensors -- the four the reference's
evaluation scripts build (imagenet_seg_eval.py:120-129, generate_visualizations.py:186-189, misc_functions.py:16-27)."""
import numpy as np
import torch
from PIL import Image


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class Resize:
    def __init__(self, size, interpolation=Image.BILINEAR):
        self.size = (size, size) if isinstance(size, int) else tuple(size)      # (h, w)
        self.interpolation = interpolation

    def __call__(self, img):
        h, w = self.size
        return img.resize((w, h), self.interpolation)


class ToTensor:
    def __call__(self, img):
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.float().div(255.0) if t.dtype == torch.uint8 else t.float()


class Normalize:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = mean, std

    def __call__(self, tensor):
        mean = torch.as_tensor(self.mean, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
        return (tensor - mean) / std
