"""Relprop rule classes -- host-side mirror of the reference's rule library, backed by HIP kernels.

Same class names, constructor signatures, ``forward`` behaviour, ``self.X`` / ``self.Y`` state
convention and ``relprop(R, alpha)`` signature as modules/layers_ours.py (variant "ours") and
modules/layers_lrp.py (variant "lrp") of the reference, and their BERT copies
(BERT_explainability/modules/layers_ours.py, layers_lrp.py: + MatMul, Mul, Tanh).  The reference
evaluates each rule with ``torch.autograd.grad`` on a re-built micro-graph (layers_ours.py:41-43);
here ``relprop`` calls one fused closed-form kernel through the C ABI (include/te_relprop.h).

Batch semantics: the reference is batch-1 only; a batch of B samples is B independent batch-1
problems (Add's "whole tensor" sums are per sample).  With B = 1 the results match the reference.

Off the accelerated hot path ("API surface"): BatchNorm2d / pools / Cat / AddEye /
Mul keep their forward so models build and run, but their relprop raises NotImplementedError here.
Conv2d.relprop (method="full": the patch embedding's z^B rule, layers_ours.py:256-286) is implemented
(csrc/te_conv.hip).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops

__all__ = ['forward_hook', 'Clone', 'Add', 'Cat', 'ReLU', 'GELU', 'Dropout', 'BatchNorm2d', 'Linear', 'MaxPool2d',
           'AdaptiveAvgPool2d', 'AvgPool2d', 'Conv2d', 'Sequential', 'safe_divide', 'einsum', 'Softmax',
           'IndexSelect', 'LayerNorm', 'AddEye', 'Tanh', 'MatMul', 'Mul']


def safe_divide(a, b):
    """modules/layers_ours.py:10-13 (plain torch; host-side helper, not used by the kernels)."""
    den = b.clamp(min=1e-9) + b.clamp(max=1e-9)
    den = den + den.eq(0).type(den.type()) * 1e-9
    return a / den * b.ne(0).type(b.type())


def forward_hook(self, input, output):
    """Capture the detached input(s) as self.X and the output as self.Y (layers_ours.py:16-27).
    No copy is made: detach() aliases the activation the forward pass produced."""
    if type(input[0]) in (list, tuple):
        self.X = [i.detach() for i in input[0]]
    else:
        self.X = input[0].detach()
    self.Y = output
    # the rules that take Z from the cached forward output (Linear, einsum, MatMul) only do so while that tensor
    # -- and the weight that produced it -- are exactly what the forward pass wrote (see _cached_y)
    self._y_version = output._version if torch.is_tensor(output) else None
    w = getattr(self, "weight", None)
    b = getattr(self, "bias", None)
    self._w_version = (w._version if torch.is_tensor(w) else None, b._version if torch.is_tensor(b) else None)


class StopRelprop(Exception):
    """Raised by an attention module whose ``_stop_after_attn_cam`` flag is set, right after it stored its attn_cam:
    the model-level relprop loop catches it (extension: ``prune_below_start_layer``, see vit.VisionTransformer)."""


def _cached_y(module):
    """The forward output forward_hook stored (the product whose rule is being evaluated), or None.

    self.Y is the LIVE output tensor (an alias, like the reference's), so an in-place op after the layer
    (``ReLU(inplace=True)``, ``x += ...``) or an optimizer step on the weight would silently change what Z is derived
    from, whereas the reference recomputes Z from X and W.  forward_hook records the tensors' version counters; a
    mismatch means "modified since the forward pass" and the rule falls back to recomputing Z itself."""
    y = getattr(module, "Y", None)
    if not torch.is_tensor(y) or getattr(module, "_y_version", None) != y._version:
        return None
    w = getattr(module, "weight", None)
    b = getattr(module, "bias", None)
    now = (w._version if torch.is_tensor(w) else None, b._version if torch.is_tensor(b) else None)
    if getattr(module, "_w_version", now) != now:
        return None
    return y


class RelProp(nn.Module):
    variant = "ours"

    def __init__(self):
        super(RelProp, self).__init__()
        self.register_forward_hook(forward_hook)

    def relprop(self, R, alpha):
        return R


class _Identity(RelProp):
    """Rules that pass relevance through unchanged (layers_ours.py:67-80,86-87) need no cached input."""


def _no_cache_hook(self, input, output):
    self.Y = output


class ReLU(nn.ReLU, RelProp):
    pass


class GELU(nn.GELU, RelProp):
    def feeds(self, linear):
        """Model code may name the Linear layer this activation's output goes to (vit.Mlp, bert.BertLayer): the producer
        then emits that layer's operand planes itself (producers._Gelu).  A hint only -- the consumer checks that the planes
        belong to the tensor it received.  Kept in a tuple: not a registered submodule, and copy / pickle follow it."""
        self.__dict__["_te_feeds"] = (linear,)
        return self

    def forward(self, x):
        from . import producers                      # 8f.1: csrc/te_norm_act.hip behind ops.USE_FUSED_PRODUCERS
        if getattr(self, "approximate", "none") == "none" and not self.training and producers.gelu_usable(x):
            nxt = self.__dict__.get("_te_feeds", (None,))[0]
            return producers.gelu(x, nxt, x6_cache(nxt) if isinstance(nxt, Linear) else None)
        return super().forward(x)


class Softmax(nn.Softmax, RelProp):
    pass


class LayerNorm(nn.LayerNorm, RelProp):
    def forward(self, x):
        from . import producers
        if producers.norm_usable(x, self):
            return producers.layer_norm(x, self)
        return super().forward(x)


class Dropout(nn.Dropout, RelProp):
    pass


class Tanh(nn.Tanh, RelProp):
    pass


# ------------------------------------------------------------------------------------------ hot path
def x6_cache(module) -> dict:
    """Per-layer store of derived operands that outlive one relprop call (the bf16 planes of the weight,
    ops.x6_weight_planes); not part of the state dict."""
    c = module.__dict__.get("_te_cache")
    if c is None:
        c = module.__dict__["_te_cache"] = {}
    return c


class Linear(nn.Linear, RelProp):
    """layers_ours.py:207-230 / layers_lrp.py:188-211 -> te_linear_relprop_f32."""

    # The derived operand planes (x6_cache) are device scratch, not state: they are rebuilt on demand, so they are kept out of
    # pickling / deepcopy / torch.save(model), and dropped whenever the parameters are replaced wholesale.
    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_te_cache", None)
        return state

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        ops.x6_invalidate(self)

    def _apply(self, fn, *args, **kwargs):
        ops.x6_invalidate(self)
        return super()._apply(fn, *args, **kwargs)

    def forward(self, x):
        from . import producers                      # 8f.1: forward and / or input gradient on te_gemm_x6_f32
        plan = producers.linear_plan(x, self)
        if not plan[0]:
            x6_cache(self).pop("x_abs_planes", None)      # no x6 forward product of THIS input: nothing for the rule to reuse
            x6_cache(self).pop("x_planes_from_producer", None)
        if plan[0] or plan[1]:
            return producers.linear(x, self, x6_cache(self), plan)
        return super().forward(x)

    def relprop(self, R, alpha):
        # self.Y (the forward output, cached by forward_hook like the reference does) lets the kernel derive
        # Z = X+ W+^T + X- W-^T from one product instead of two
        Y = _cached_y(self)
        if Y is not None and Y.shape[:-1] != self.X.shape[:-1]:
            Y = None
        return ops.linear_relprop(R, self.X, self.weight.detach(), alpha=alpha, variant=self.variant, Y=Y,
                                  bias=self.bias, cache=x6_cache(self))


class Add(RelProp):
    """layers_ours.py:97-120 (ours) / layers_lrp.py:98-100 (lrp) -> te_add_relprop_f32 /
    te_add_bcast_relprop_f32 (broadcast mask operand of BERT self-attention)."""

    def forward(self, inputs):
        return torch.add(*inputs)

    def relprop(self, R, alpha, deferred=False):
        # deferred=True (model-internal call sites only): the per-sample rescale of layers_ours.py:117-118 travels with
        # the two outputs as an ``ops.Deferred`` and is applied inside the consuming Clone / Linear kernels -- the
        # rule then streams its operands once instead of twice; ``.materialise()`` is bitwise the plain result
        x0, x1 = self.X
        bcast_mask = x0.dim() == 4 and x1.dim() == 4 and x1.shape[1] == 1 and x1.shape[2] == 1       # BERT.py:386-388
        a, b = ops.add_relprop(R, x0, x1, variant=self.variant,
                               deferred=deferred and ops.USE_DEFERRED_ADD and (x0.shape == x1.shape or bcast_mask))
        return [a, b]


class einsum(RelProp):
    """layers_ours.py:122-127 (RelPropSimple) for the two attention products."""

    def __init__(self, equation):
        super().__init__()
        self.equation = equation

    def forward(self, *operands):
        ops_ = operands[0] if (len(operands) == 1 and isinstance(operands[0], (list, tuple))) else operands
        eq = self.equation.replace(" ", "")
        # the two attention products as plain batched matmuls: torch.einsum materialises k^T first (a [B,H,D,N] copy
        # per block); matmul hands the transposed strides to the BLAS
        if eq == 'bhid,bhjd->bhij' and len(ops_) == 2:
            return torch.matmul(ops_[0], ops_[1].transpose(-1, -2))
        if eq == 'bhij,bhjd->bhid' and len(ops_) == 2:
            return torch.matmul(ops_[0], ops_[1])
        return torch.einsum(self.equation, *operands)

    def relprop(self, R, alpha):
        eq = self.equation.replace(" ", "")
        if eq == 'bhij,bhjd->bhid':
            cam_attn, cam_v = ops.matmul_relprop_av(R, self.X[0], self.X[1], variant=self.variant, z=_cached_y(self))
            return [cam_attn, cam_v]
        if eq == 'bhid,bhjd->bhij':
            cam_q, cam_k = ops.matmul_relprop_qk(R, self.X[0], self.X[1], variant=self.variant, z=_cached_y(self))
            return [cam_q, cam_k]
        raise NotImplementedError(f"einsum.relprop: equation {self.equation!r} is not on the accelerated path")


class MatMul(RelProp):
    """BERT_explainability/modules/layers_ours.py:89-91: [probs, V] and [Q, K^T] products."""

    def forward(self, inputs):
        return torch.matmul(*inputs)

    def relprop(self, R, alpha):
        x0, x1 = self.X
        if x0.dim() != 4 or x1.dim() != 4:
            raise NotImplementedError("MatMul.relprop: only [B,H,.,.] attention products are accelerated")
        if x1.stride(-1) != 1 and x1.stride(-2) == 1:          # [Q, K^T] with K^T a transposed view
            k = x1.transpose(-1, -2)
            cam_q, cam_k = ops.matmul_relprop_qk(R, x0, k, variant=self.variant, z=_cached_y(self))
            return [cam_q, cam_k.transpose(-1, -2)]
        if x0.shape[-1] == x0.shape[-2] == x1.shape[-2]:       # [probs, V]
            cam_attn, cam_v = ops.matmul_relprop_av(R, x0, x1, variant=self.variant, z=_cached_y(self))
            return [cam_attn, cam_v]
        if x0.shape[-1] == x1.shape[-2]:                        # [Q, K^T] materialised contiguously
            k = x1.transpose(-1, -2).contiguous()
            cam_q, cam_k = ops.matmul_relprop_qk(R, x0, k, variant=self.variant, z=_cached_y(self))
            return [cam_q, cam_k.transpose(-1, -2)]
        raise NotImplementedError("MatMul.relprop: operand shapes are not an attention product")


class Clone(RelProp):
    """layers_ours.py:151-169 -> te_clone_relprop_f32."""

    def forward(self, input, num):
        self.__setattr__('num', num)
        return [input for _ in range(num)]

    def relprop(self, R, alpha):
        return ops.clone_relprop(list(R), self.X)


class IndexSelect(RelProp):
    """layers_ours.py:129-147 -> te_index_select_relprop_f32 (dim 1, one index)."""

    def forward(self, inputs, dim, indices):
        self.__setattr__('dim', dim)
        self.__setattr__('indices', indices)
        # a host copy of a single index, taken where it is free: python ints and CPU tensors here, a device tensor
        # once per tensor object (models pass the same buffer every call) -- relprop then never synchronises and is
        # capturable in a HIP graph
        if not torch.is_tensor(indices):
            self._index_host = int(indices)
        elif indices.numel() == 1 and (not indices.is_cuda or getattr(self, "_index_src", None) is not indices
                                       or getattr(self, "_index_version", None) != indices._version):
            # (identity AND version counter: an in-place edit of the index buffer, ``idx.fill_(k)``, refreshes the copy)
            if not (indices.is_cuda and torch.cuda.is_current_stream_capturing()):
                self._index_host = int(indices)
                self._index_src = indices
                self._index_version = indices._version
        return torch.index_select(inputs, dim, indices.reshape(-1) if torch.is_tensor(indices) else indices)

    def relprop(self, R, alpha):
        idx = self.indices
        if self.dim != 1 or self.X.dim() != 3 or (torch.is_tensor(idx) and idx.numel() != 1):
            raise NotImplementedError("IndexSelect.relprop: only dim=1 with a single index is accelerated")
        host = getattr(self, "_index_host", None)
        if host is None or (torch.is_tensor(idx) and (getattr(self, "_index_src", idx) is not idx
                                                      or getattr(self, "_index_version", idx._version) != idx._version)):
            host = int(idx)                                        # (device sync; not capturable)
        return ops.index_select_relprop(R, self.X, host)


class Sequential(nn.Sequential):
    def relprop(self, R, alpha):
        for m in reversed(self._modules.values()):
            R = m.relprop(R, alpha)
        return R


# ------------------------------------------------------------------------- API surface, off the hot path
class _OffPath(RelProp):
    def relprop(self, R, alpha):
        raise NotImplementedError(f"{type(self).__name__}.relprop is off the accelerated transformer_attribution path")


class Mul(_OffPath):
    def forward(self, inputs):
        return torch.mul(*inputs)


class AddEye(_OffPath):
    def forward(self, input):
        return input + torch.eye(input.shape[2], device=input.device).expand_as(input)


class Cat(_OffPath):
    def forward(self, inputs, dim):
        self.__setattr__('dim', dim)
        return torch.cat(inputs, dim)


class MaxPool2d(nn.MaxPool2d, _OffPath):
    pass


class AdaptiveAvgPool2d(nn.AdaptiveAvgPool2d, _OffPath):
    pass


class AvgPool2d(nn.AvgPool2d, _OffPath):
    pass


class BatchNorm2d(nn.BatchNorm2d, _OffPath):
    pass


class Conv2d(nn.Conv2d, RelProp):
    """layers_ours.py:232-279.  Accelerated: the z^B rule of an image-input (3-channel) convolution whose stride
    equals its kernel with no padding -- the ViT patch embedding, reached by method="full" (ViT_LRP.py:337-343) ->
    te_conv2d_zb_relprop_f32.  Other geometries are off the transformer path."""

    def relprop(self, R, alpha):
        k = self.kernel_size
        patch = (self.X.shape[1] == 3 and k[0] == k[1] and tuple(self.stride) == tuple(k)
                 and tuple(self.padding) == (0, 0) and tuple(self.dilation) == (1, 1) and self.groups == 1)
        if not patch:
            raise NotImplementedError("Conv2d.relprop: only the z^B rule of a patch-embedding convolution "
                                      "(3 input channels, stride == kernel, no padding) is accelerated")
        return ops.conv2d_zb_relprop(R, self.X, self.weight, self.Y, self.bias)
