"""Perturbation test (positive / negative), SURVEY.md section 8(f) row 4: mirror of the evaluation loop of
baselines/ViT/pertubation_eval_from_hdf5.py:25-144 without its dataset / hdf5 / directory plumbing.

Per batch the reference runs, for each of 9 perturbation steps, torch.topk over 50,176 relevance values per image, an
index repeat, a clone, a scatter and a normalisation, then one model forward and five small reductions with a
device-to-host copy each.  Here the 9 masked + normalised copies (and the normalised original, as step k = 0) come out
of ONE te_perturb_f32 call, the forwards run as one batch of 10 B images (chunked by ``max_forward_batch``), and the
metrics stay on the device until ``arrays()`` / ``save()``.

Result arrays have the reference's names and shapes (``save`` writes the same six .npy files, :120-125).
Differences: ties among equal relevance values are removed in ascending pixel order (torch.topk leaves that
unspecified); with ``wrong=True`` the per-step differences use the SELECTED samples' original probabilities / logits
(the reference subtracts the unselected batch there, which only works when every sample of the batch is wrong).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops

PER_STEPS = (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9)          # pertubation_eval_from_hdf5.py:34
ABS_STEPS = (5, 10, 15, 20, 25, 30, 35, 40, 45)                    # :37


class PerturbationEvaluator:
    def __init__(self, model, num_samples, scale="per", neg=True, wrong=False, mean=(0.5, 0.5, 0.5),
                 std=(0.5, 0.5, 0.5), image_pixels=224 * 224, max_forward_batch=512):
        if scale == "per":
            self.base_size, self.steps = image_pixels, PER_STEPS                     # :32-34
        elif scale == "100":
            self.base_size, self.steps = 100, ABS_STEPS                              # :35-37
        else:
            raise Exception("scale not valid")                                       # :39
        self.model = model.eval()
        self.neg, self.wrong = bool(neg), bool(wrong)
        self.mean, self.std = tuple(mean), tuple(std)
        self.ks = [int(self.base_size * s) for s in self.steps]                      # :91
        self.max_forward_batch = int(max_forward_batch)
        self.num_samples = int(num_samples)
        self._model_hits, self._model_dis = [], []
        self._hits, self._dis, self._logit, self._prob = [], [], [], []

    # ------------------------------------------------------------------------------------------
    def _forward(self, images):
        outs = [self.model(images[i:i + self.max_forward_batch])
                for i in range(0, images.shape[0], self.max_forward_batch)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    @staticmethod
    def _dissimilarity(logits, target):
        probs = torch.softmax(logits, dim=-1)
        target_probs = torch.gather(probs, -1, target.unsqueeze(-1)).squeeze(-1)
        second_probs = probs.topk(2, dim=-1).values[..., 1]
        return torch.log(target_probs / second_probs)

    @torch.no_grad()
    def update(self, data, vis, target):
        """data [B,3,H,W] in [0,1]; vis [B,1,H,W] (or [B,H*W]) relevance; target [B] labels -- one loader batch
        (pertubation_eval_from_hdf5.py:47-118)."""
        B = data.shape[0]
        vis = vis.reshape(B, -1).float()
        if self.neg:
            vis = -vis                                                               # :85-86
        x = ops.perturb(vis, data.float(), [0] + self.ks, self.mean, self.std)       # [10,B,C,H,W]; step 0 = original
        S = len(self.ks)
        logits = self._forward(x.reshape((S + 1) * B, *data.shape[1:])).reshape(S + 1, B, -1).float()
        pred = logits[0]                                                             # :56-66
        pred_org_logit = pred.max(dim=1).values
        pred_org_prob = torch.softmax(pred, dim=1).max(dim=1).values
        tgt_pred = (target == pred.argmax(dim=1))
        self._model_hits.append(tgt_pred.to(torch.float64))
        self._model_dis.append(self._dissimilarity(pred, target).to(torch.float64))
        out = logits[1:]                                                             # [S,B,K]  :100-116
        sel = None
        if self.wrong:                                                               # :72-79
            sel = (~tgt_pred).nonzero().flatten()
            if sel.numel() == 0:
                return
            out, target = out[:, sel], target[sel]
            pred_org_logit, pred_org_prob = pred_org_logit[sel], pred_org_prob[sel]
        self._prob.append((torch.softmax(out, dim=-1).max(dim=-1).values - pred_org_prob).to(torch.float64))
        self._logit.append((out.max(dim=-1).values - pred_org_logit).to(torch.float64))
        self._hits.append((out.argmax(dim=-1) == target).to(torch.float64))
        self._dis.append(self._dissimilarity(out, target.unsqueeze(0).expand(S, -1)).to(torch.float64))

    # ------------------------------------------------------------------------------------------
    def arrays(self):
        """The six arrays the script saves (:120-125), keyed by file name."""
        def cat(parts, dim, empty_shape):
            return torch.cat(parts, dim).cpu().numpy() if parts else np.zeros(empty_shape)

        def padded(parts):                       # the model arrays are allocated for the whole dataset (:27-28)
            a = np.zeros((self.num_samples,))
            v = cat(parts, 0, (0,))
            a[:len(v)] = v
            return a
        S = len(self.ks)
        return {"model_hits.npy": padded(self._model_hits),
                "model_dissimilarities.npy": padded(self._model_dis),
                "perturbations_hits.npy": cat(self._hits, 1, (S, 0)),
                "perturbations_dissimilarities.npy": cat(self._dis, 1, (S, 0)),
                "perturbations_logit_diff.npy": cat(self._logit, 1, (S, 0)),
                "perturbations_prob_diff.npy": cat(self._prob, 1, (S, 0))}

    def save(self, experiment_dir):
        os.makedirs(experiment_dir, exist_ok=True)
        arrs = self.arrays()
        for name, a in arrs.items():
            np.save(os.path.join(experiment_dir, name), a)
        return arrs
