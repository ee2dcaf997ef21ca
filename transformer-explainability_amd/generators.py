"""Explanation generators: mirrors of baselines/ViT/ViT_explanation_generator.py (LRP) and
BERT_explainability/modules/BERT/ExplanationGenerator.py (Generator) of the reference.

Same method names and arguments; differences (results identical at batch 1):
  * a batch of B inputs is explained in one pass (B independent samples) -> [B, N-1] / [B, N]
  * the class index defaults to the per-sample argmax, computed on the device (no D2H round trip)
  * the attention gradients are obtained with torch.autograd.grad w.r.t. the attention tensors only,
    so no weight gradients are computed (the reference's loss.backward() computes and discards them)
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def _one_hot(output: torch.Tensor, index) -> torch.Tensor:
    B, K = output.shape
    if index is None:
        idx = output.detach().argmax(dim=-1)
    else:
        idx = torch.as_tensor(np.asarray(index) if not torch.is_tensor(index) else index, device=output.device)
        idx = idx.reshape(-1).long()
        if idx.numel() == 1 and B > 1:
            idx = idx.expand(B)
    one_hot = torch.zeros((B, K), dtype=output.dtype, device=output.device)
    one_hot.scatter_(1, idx.view(B, 1), 1.0)
    return one_hot


def _attention_gradients(loss, attn_modules):
    attns = [m.get_attn() for m in attn_modules]
    grads = torch.autograd.grad(loss, attns, retain_graph=False, allow_unused=False)
    for m, g in zip(attn_modules, grads):
        m.save_attn_gradients(g)


class LRP:
    """baselines/ViT/ViT_explanation_generator.py:20-41."""

    def __init__(self, model):
        self.model = model
        self.model.eval()

    def generate_LRP(self, input, index=None, method="transformer_attribution", is_ablation=False, start_layer=0):
        output = self.model(input)
        kwargs = {"alpha": 1}
        one_hot = _one_hot(output, index)
        loss = torch.sum(one_hot * output)
        _attention_gradients(loss, [blk.attn for blk in self.model.blocks])
        return self.model.relprop(one_hot, method=method, is_ablation=is_ablation, start_layer=start_layer, **kwargs)


class Generator:
    """BERT_explainability/modules/BERT/ExplanationGenerator.py:20-59 (generate_LRP)."""

    def __init__(self, model):
        self.model = model
        self.model.eval()

    def forward(self, input_ids, attention_mask):
        return self.model(input_ids, attention_mask)

    def _explain(self, input_ids, attention_mask, index):
        output = self.model(input_ids=input_ids, attention_mask=attention_mask)[0]
        one_hot = _one_hot(output, index)
        loss = torch.sum(one_hot * output)
        layers = self.model.bert.encoder.layer
        _attention_gradients(loss, [lay.attention.self for lay in layers])
        self.model.relprop(one_hot, alpha=1)
        return layers

    def generate_LRP(self, input_ids, attention_mask, index=None, start_layer=11):
        self._explain(input_ids, attention_mask, index)
        return self.attribution_tail(start_layer)

    def attribution_tail(self, start_layer=11):
        """ExplanationGenerator.py:47-59 on the attn_cam / attention gradients cached by relprop + backward."""
        layers = self.model.bert.encoder.layer
        first = layers[0].attention.self.get_attn_cam()
        B, _, N, _ = first.shape
        stack = torch.empty((len(layers), B, N, N), dtype=first.dtype, device=first.device)
        for i, lay in enumerate(layers):
            sa = lay.attention.self
            ops.gradcam_headmean(sa.get_attn_gradients(), sa.get_attn_cam(), out=stack[i])
        # ExplanationGenerator.py:7-18 (row-normalised rollout) + :58 (CLS fix-up) -> row 0
        joint = ops.rollout(stack, start_layer=start_layer, normalise=True, cls_fixup=True)
        return joint[:, 0]

    def generate_LRP_last_layer(self, input_ids, attention_mask, index=None):
        """ExplanationGenerator.py:62-84: head-mean of the last layer's attn_cam, CLS row, CLS slot zeroed."""
        layers = self._explain(input_ids, attention_mask, index)
        cam = layers[-1].attention.self.get_attn_cam().clamp(min=0).mean(dim=1)[:, 0].clone()
        cam[:, 0] = 0
        return cam
