"""Segmentation test of a relevance map, SURVEY.md section 8(f) row 2: mirror of ``eval_batch`` and the running totals
of baselines/ViT/imagenet_seg_eval.py:170-311 with the metrics of utils/metrices.py (pixel accuracy :135-151,
intersection / union :154-178, average precision :81-99, F1 :26-38), without the dataset / saver / image dumps.

The map -> (heat, foreground mask) step (:214-222: bilinear x16, min-max, mean threshold) is the te_heatmap_f32 kernel;
the metrics are a handful of reductions and one sort over 2*H*W scores per image, evaluated on the device for the
whole batch (the reference does them one image at a time in numpy / scikit-learn after a device-to-host copy each).

Semantics kept from the reference, including its quirks:
  * a batch is B independent images (the script runs with batch_size = 1, :31);
  * average precision is scikit-learn's step-wise sum over DISTINCT score thresholds of the 2*H*W scores
    (1 - heat for class 0, heat for class 1) against the one-hot labels;
  * ``get_f1_scores(output[0, 1], labels[0])`` receives a 2-D [H, W] mask, so its "batch" loop runs over the image ROWS:
    the F1 the script averages is the mean of the per-row F1 scores (an all-negative row counts 0);
  * labels < 0 are ignored by pixel accuracy / IoU / AP (the F1 path of the reference cannot handle them).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def foreground_split(maps, scale=16):
    """imagenet_seg_eval.py:214-232: patch maps [B, g*g] -> (heat [B,H,W] in [0,1], mask [B,H,W] = heat > mean), NaN -> 0."""
    heat, mask = ops.heatmap(maps, scale=scale, normalise=True, with_mask=True)
    heat, mask = heat[:, 0], mask[:, 0]
    nan = torch.isnan(heat)
    return torch.where(nan, torch.zeros_like(heat), heat), torch.where(nan, torch.zeros_like(mask), mask)


def pixel_accuracy(mask, labels):
    """utils/metrices.py:135-151 per image: (correct, labeled) int64 [B]."""
    valid = labels >= 0
    correct = ((mask.long() == labels) & valid).flatten(1).sum(1)
    return correct, valid.flatten(1).sum(1)


def intersection_union(mask, labels, nclass=2):
    """utils/metrices.py:154-178 per image: (inter, union) int64 [B, nclass]."""
    valid = labels >= 0
    pred = (mask.long() + 1) * valid
    tgt = labels.long() + 1
    inter = pred * (pred == tgt)
    cls = torch.arange(1, nclass + 1, device=mask.device).view(1, -1, 1)
    a_inter = (inter.flatten(1).unsqueeze(1) == cls).sum(-1)
    a_pred = (pred.flatten(1).unsqueeze(1) == cls).sum(-1)
    a_lab = (tgt.flatten(1).unsqueeze(1) == cls).sum(-1)
    return a_inter, a_pred + a_lab - a_inter


def average_precision(heat, labels):
    """utils/metrices.py:81-99 (sklearn.average_precision_score over the 2*H*W class scores) per image: float64 [B]."""
    B = heat.shape[0]
    h = heat.flatten(1).float()
    lab = labels.flatten(1)
    scores = torch.cat([1.0 - h, h], 1).double()                          # class 0 / class 1 scores (fp32 as :225-226)
    truth = torch.cat([lab.clamp(min=0) == 0, lab.clamp(min=0) == 1], 1)  # one-hot of clamp(label, 0)
    valid = torch.cat([lab >= 0, lab >= 0], 1)
    out = torch.zeros(B, dtype=torch.float64, device=heat.device)
    for b in range(B):                                                    # ragged after the ignore mask
        s, t = scores[b][valid[b]], truth[b][valid[b]].double()
        npos = t.sum()
        if s.numel() == 0 or float(npos) == 0.0:
            continue                                                      # nan_to_num(nan) -> 0 in the reference
        order = torch.argsort(s, descending=True, stable=True)
        s, t = s[order], t[order]
        tp = torch.cumsum(t, 0)
        last = torch.ones_like(s, dtype=torch.bool)
        last[:-1] = s[1:] != s[:-1]                                       # last element of every run of equal scores
        tp_d = tp[last]
        n_d = (torch.nonzero(last).flatten() + 1).double()
        precision, recall = tp_d / n_d, tp_d / npos
        prev = torch.cat([torch.zeros(1, dtype=torch.float64, device=s.device), recall[:-1]])
        out[b] = ((recall - prev) * precision).sum()
    return out


def row_f1(mask, labels):
    """utils/metrices.py:26-38 as imagenet_seg_eval.py:268 calls it: F1 of every image ROW -> float64 [B, H]."""
    p, t = mask.long() == 1, labels.long() == 1
    tp = (p & t).sum(-1).double()
    fp = (p & ~t).sum(-1).double()
    fn = (~p & t).sum(-1).double()
    den = 2 * tp + fp + fn
    return torch.where(den > 0, 2 * tp / den.clamp(min=1), torch.zeros_like(den))


class SegmentationEvaluator:
    """Running totals of imagenet_seg_eval.py:274-311.  ``explain(images) -> patch maps [B, g*g]`` is any generator of
    this package (e.g. ``lambda x: lrp.generate_LRP(x, start_layer=1)``)."""

    def __init__(self, explain, scale=16):
        self.explain, self.scale = explain, scale
        self.total_correct = self.total_label = 0
        self.total_inter = np.zeros(2, dtype=np.int64)
        self.total_union = np.zeros(2, dtype=np.int64)
        self.total_ap, self.total_f1 = [], []

    def update(self, image, labels):
        maps = self.explain(image).detach()
        heat, mask = foreground_split(maps.reshape(maps.shape[0], -1), self.scale)
        return self.update_from_heat(heat, mask, labels)

    def update_from_heat(self, heat, mask, labels):
        correct, labeled = pixel_accuracy(mask, labels)
        inter, union = intersection_union(mask, labels)
        ap, f1 = average_precision(heat, labels), row_f1(mask, labels)
        self.total_correct += int(correct.sum())
        self.total_label += int(labeled.sum())
        self.total_inter += inter.sum(0).cpu().numpy()
        self.total_union += union.sum(0).cpu().numpy()
        self.total_ap += [float(v) for v in ap.cpu()]
        self.total_f1 += [r for r in f1.cpu().numpy()]
        return correct, labeled, inter, union, ap, f1

    def summary(self):
        eps = np.spacing(1, dtype=np.float64)                                                   # :306-307
        iou = np.float64(1.0) * self.total_inter / (eps + self.total_union)
        return {"pixAcc": float(np.float64(1.0) * self.total_correct / (eps + self.total_label)),
                "mIoU": float(iou.mean()), "mAP": float(np.mean(self.total_ap)), "mF1": float(np.mean(self.total_f1))}
