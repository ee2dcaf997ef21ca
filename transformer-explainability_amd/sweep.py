"""Saliency sweep + result store, SURVEY.md section 8(f) row 2 / 8(d) config 5: mirror of
baselines/ViT/generate_visualizations.py:27-100 (``compute_saliency_and_save``) and of the reader
dataset/expl_hdf5.py:8-31 (``ImagenetResults``).

Per loader batch the reference explains the images with the selected method, up-samples the patch map x16 (bilinear),
min-max normalises it and appends image / target / map to three resizable gzip datasets of ``results.hdf5``.  Here:

  * the explanation methods are the generators of this package (same dispatch table, :67-93), the up-sampling +
    normalisation is ONE te_heatmap_f32 launch per batch (per-map min-max: a batch is B independent batch-1 problems,
    the reference's default --batch-size is 1);
  * the sweep shards over ranks by contiguous blocks of the dataset (``parallel.shard_range``) -- every image is an
    independent problem and every rank appends to its own shard files, so the data path has no collective at all;
  * storage: the three datasets ``vis`` [N,1,H,W] f32, ``image`` [N,3,H,W] f32, ``target`` [N] i32 of the reference's
    layout.  Backend "hdf5" writes the reference's own results.hdf5 (single rank; needs h5py, which this image does not
    ship -- untested here); backend "npy" writes one pre-sized .npy memmap per dataset and shard under
    ``<method_dir>/results/`` (readable with numpy alone).  ``ImagenetResults`` reads either.
"""
from __future__ import annotations

import glob
import json
import os
import re

import numpy as np
import torch

from . import ops, parallel

METHODS = ('rollout', 'lrp', 'transformer_attribution', 'full_lrp', 'lrp_last_layer', 'attn_last_layer',
           'attn_gradcam')                                          # generate_visualizations.py:110-112


def _have_h5py():
    try:
        import h5py  # noqa: F401
        return True
    except ImportError:
        return False


class ResultsStore:
    """Append-only writer of the (vis, image, target) datasets for the samples [lo, hi) of a sweep."""

    def __init__(self, method_dir, num_samples, image_shape=(3, 224, 224), vis_shape=(1, 224, 224), lo=0, hi=None,
                 backend="auto"):
        hi = num_samples if hi is None else hi
        self.lo, self.hi, self.count = lo, hi, 0
        self.backend = ("hdf5" if _have_h5py() else "npy") if backend == "auto" else backend
        os.makedirs(method_dir, exist_ok=True)
        n = hi - lo
        if self.backend == "hdf5":
            if (lo, hi) != (0, num_samples):
                raise ValueError("the hdf5 backend writes one results.hdf5: use it from a single rank")
            import h5py
            self._f = h5py.File(os.path.join(method_dir, "results.hdf5"), "a")
            mk = self._f.create_dataset                                         # generate_visualizations.py:29-44
            self._d = {"vis": mk("vis", (n, *vis_shape), maxshape=(None, *vis_shape), dtype=np.float32, compression="gzip"),
                       "image": mk("image", (n, *image_shape), maxshape=(None, *image_shape), dtype=np.float32,
                                   compression="gzip"),
                       "target": mk("target", (n,), maxshape=(None,), dtype=np.int32, compression="gzip")}
        elif self.backend == "npy":
            d = os.path.join(method_dir, "results")
            os.makedirs(d, exist_ok=True)
            self._meta = os.path.join(d, f"shard.{lo:09d}-{hi:09d}.json")
            mm = np.lib.format.open_memmap
            tag = f"{lo:09d}-{hi:09d}"
            self._d = {"vis": mm(os.path.join(d, f"vis.{tag}.npy"), "w+", np.float32, (n, *vis_shape)),
                       "image": mm(os.path.join(d, f"image.{tag}.npy"), "w+", np.float32, (n, *image_shape)),
                       "target": mm(os.path.join(d, f"target.{tag}.npy"), "w+", np.int32, (n,))}
        else:
            raise ValueError(f"unknown results backend {backend!r}")

    def append(self, image, target, vis):
        b = image.shape[0]
        if self.count + b > self.hi - self.lo:
            raise ValueError("more samples appended than the shard was sized for")
        sl = slice(self.count, self.count + b)
        self._d["image"][sl] = image.detach().cpu().numpy()
        self._d["target"][sl] = target.detach().cpu().numpy().astype(np.int32)
        self._d["vis"][sl] = vis.detach().cpu().numpy()
        self.count += b

    def close(self):
        if self.backend == "hdf5":
            for d in self._d.values():
                d.resize(self.count, axis=0)
            self._f.close()
        else:
            for d in self._d.values():
                d.flush()
            with open(self._meta, "w") as f:
                json.dump({"lo": self.lo, "hi": self.hi, "count": self.count}, f)
        self._d = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class ImagenetResults(torch.utils.data.Dataset):
    """dataset/expl_hdf5.py:8-31: item -> (image [3,H,W] f32, vis [1,H,W] f32, target int64)."""

    def __init__(self, path):
        super().__init__()
        self.path = os.path.join(path, "results.hdf5")
        self._h5 = None
        self._shards = None
        if os.path.exists(self.path):
            import h5py
            with h5py.File(self.path, "r") as f:
                self.data_length = len(f["/image"])
            return
        metas = sorted(glob.glob(os.path.join(path, "results", "shard.*.json")))
        if not metas:
            raise FileNotFoundError(f"no results.hdf5 and no results/shard.*.json under {path}")
        self._shards, self._starts, total = [], [], 0
        for m in metas:
            with open(m) as f:
                info = json.load(f)
            tag = re.search(r"shard\.(\d+-\d+)\.json$", m).group(1)
            d = os.path.dirname(m)
            arrs = {k: np.load(os.path.join(d, f"{k}.{tag}.npy"), mmap_mode="r") for k in ("image", "vis", "target")}
            self._shards.append((info["count"], arrs))
            self._starts.append(total)
            total += info["count"]
        self.data_length = total

    def __len__(self):
        return self.data_length

    def __getitem__(self, item):
        if self._shards is None:
            if self._h5 is None:
                import h5py
                self._h5 = h5py.File(self.path, "r")
            d, i = self._h5, item
        else:
            if item < 0:
                item += self.data_length
            s = int(np.searchsorted(self._starts, item, side="right")) - 1
            if not (0 <= item < self.data_length):
                raise IndexError(item)
            d, i = self._shards[s][1], item - self._starts[s]
        return (torch.tensor(np.asarray(d["image"][i])), torch.tensor(np.asarray(d["vis"][i])),
                torch.tensor(np.asarray(d["target"][i])).long())


def normalize(tensor, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """generate_visualizations.py:18-24 (out of place)."""
    mean = torch.as_tensor(mean, dtype=tensor.dtype, device=tensor.device)
    std = torch.as_tensor(std, dtype=tensor.dtype, device=tensor.device)
    return (tensor - mean[None, :, None, None]) / std[None, :, None, None]


class SaliencySweep:
    """``compute_saliency_and_save`` (generate_visualizations.py:27-100).  ``lrp`` / ``orig_lrp`` / ``baselines`` are
    the three generator objects the script builds (:172-183); only the one the method needs must be given."""

    def __init__(self, method, lrp=None, orig_lrp=None, baselines=None, vis_class="top", is_ablation=False,
                 device=None):
        if method not in METHODS:
            raise ValueError(f"method must be one of {METHODS}")
        self.method, self.vis_class, self.is_ablation = method, vis_class, bool(is_ablation)
        self.lrp, self.orig_lrp, self.baselines, self.device = lrp, orig_lrp, baselines, device

    def explain(self, data, target=None, return_maps=False):
        """One batch of normalised images -> min-max normalised maps [B,1,H,W] at image resolution (:60-98);
        return_maps: also the patch-level maps [B, g*g] they were up-sampled from."""
        index = target if self.vis_class == "target" else None                   # :62-64
        m = self.method
        if m == "rollout":
            res = self.baselines.generate_rollout(data, start_layer=1)
        elif m == "lrp":
            res = self.lrp.generate_LRP(data, start_layer=1, index=index)
        elif m == "transformer_attribution":
            res = self.lrp.generate_LRP(data, start_layer=1, method="grad", index=index)
        elif m == "full_lrp":
            res = self.orig_lrp.generate_LRP(data, method="full", index=index)
        elif m == "lrp_last_layer":
            res = self.orig_lrp.generate_LRP(data, method="last_layer", is_ablation=self.is_ablation, index=index)
        elif m == "attn_last_layer":
            res = self.lrp.generate_LRP(data, method="last_layer_attn", is_ablation=self.is_ablation)
        else:
            res = self.baselines.generate_cam_attn(data, index=index)
        B, H = data.shape[0], data.shape[-1]
        res = res.detach().reshape(B, -1)
        g = int(round(res.shape[1] ** 0.5))
        if g == H:                                   # full_lrp is already at pixel resolution (:95): min-max only
            lo, hi = res.amin(dim=1, keepdim=True), res.amax(dim=1, keepdim=True)
            heat = ((res - lo) / (hi - lo)).reshape(B, 1, H, H)
        else:
            heat = ops.heatmap(res, scale=H // g, normalise=True)   # :96-97: bilinear x16 + min-max, one launch
        return (heat, res) if return_maps else heat

    def run(self, loader_batches, store, rank=0, world=1):
        """loader_batches: iterable of (data [B,3,H,W] in [0,1], target [B]) covering THIS rank's samples in order."""
        for data, target in loader_batches:
            dev = self.device if self.device is not None else data.device
            vis = self.explain(normalize(data.to(dev)), target.to(dev))
            if torch.is_tensor(vis) and vis.is_cuda:
                # never store maps of a step whose x6 hand-over failed: check BEFORE the append (it synchronises, which
                # the store's host copy would do anyway), so a failed batch is neither written nor counted
                ops.x6_raise_if_failed(vis.device)
            store.append(data, target, vis)
        return store


def shard_batches(dataset, batch_size, rank=0, world=1):
    """The rank's contiguous block of ``dataset`` (items (image, target)) in batches, plus its [lo, hi)."""
    lo, hi = parallel.shard_range(len(dataset), rank, world)

    def gen():
        for s in range(lo, hi, batch_size):
            items = [dataset[i] for i in range(s, min(s + batch_size, hi))]
            yield (torch.stack([torch.as_tensor(im) for im, _ in items]),
                   torch.as_tensor([int(t) for _, t in items]))
    return gen(), lo, hi
