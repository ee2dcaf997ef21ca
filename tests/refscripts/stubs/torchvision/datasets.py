"""datasets.ImageNet(root, split='val', download=False, transform=...) over a folder of images:
``<root>/<split>/*.png`` in sorted order, class indices in ``<root>/<split>/targets.txt`` (one per line).
Items are (transformed image, target) like torchvision's (generate_visualizations.py:191)."""
import glob
import os

from PIL import Image


class ImageNet:
    def __init__(self, root, split="val", download=None, transform=None, target_transform=None, **kwargs):
        self.root, self.split, self.transform, self.target_transform = root, split, transform, target_transform
        d = os.path.join(root, split)
        paths = sorted(glob.glob(os.path.join(d, "*.png")))
        with open(os.path.join(d, "targets.txt")) as f:
            targets = [int(line) for line in f.read().split()]
        if len(paths) != len(targets) or not paths:
            raise RuntimeError(f"stub ImageNet: {len(paths)} images and {len(targets)} targets under {d}")
        self.samples = list(zip(paths, targets))
        self.loader = lambda p: Image.open(p).convert("RGB")

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        path, target = self.samples[index]
        sample = self.loader(path)
        if self.transform is not None:
            sample = self.transform(sample)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return sample, target
