"""Host side of the producer kernels of SURVEY.md 8f.1 that are plain layers (LayerNorm, GELU): autograd Functions
over ``ops.layernorm_* / ops.gelu_*`` (csrc/te_norm_act.hip), used by ``rules.LayerNorm`` / ``rules.GELU`` -- the
reference's ``modules/layers_ours.py:70-77`` classes -- and by the residual blocks when ``ops.USE_FUSED_PRODUCERS`` is on.

Only the INPUT gradient is produced (the explanation differentiates the logit w.r.t. activations, never w.r.t.
parameters): the LayerNorm producer is used in eval mode only and leaves ``weight.grad`` / ``bias.grad`` untouched.
"""
from __future__ import annotations

import torch

from . import ops


def usable(x: torch.Tensor) -> bool:
    """fp32 device tensor on which the producer kernels apply (rows of <= 2048 elements, a multiple of 4)."""
    return ops.USE_FUSED_PRODUCERS and torch.is_tensor(x) and ops.layernorm_supported(x)


def gelu_usable(x: torch.Tensor) -> bool:
    return (ops.USE_FUSED_PRODUCERS and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32
            and x.numel() > 0 and x.numel() % 4 == 0)


def norm_usable(x: torch.Tensor, norm) -> bool:
    return (usable(x) and not norm.training and getattr(norm, "weight", None) is not None
            and tuple(norm.normalized_shape) == (x.shape[-1],))


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, mean, rstd = ops.layernorm_forward(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        return ops.layernorm_backward(dy, x, weight, mean, rstd), None, None, None


class _ResidualLayerNorm(torch.autograd.Function):
    """``x1, x2 = clone(x); n = norm(x2)`` of a pre-norm residual block (ViT_LRP.py:203-205) as one node: forward returns
    (x, LayerNorm(x)); backward adds the gradient that arrives on the bypass to the LayerNorm's input gradient INSIDE
    the backward kernel (autograd would run a separate [T,C] addition for the two uses of x)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, mean, rstd = ops.layernorm_forward(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, d_bypass, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        if dy is None:
            return d_bypass, None, None, None
        return ops.layernorm_backward(dy, x, weight, mean, rstd, add=d_bypass), None, None, None


class _Gelu(torch.autograd.Function):
    """nn.GELU.  Round 5: between two Linear layers that run on the x6 kernels it emits their operand
    planes itself (csrc/te_linear_x6.hip: gelu_split_kernel) --
      forward   feeds = the cache dict of the Linear that consumes the output (rules.GELU.feeds): the output is written as
                fp32 and as the two plane sets that layer's forward product and rule would otherwise split from it; they wait
                in that dict under "x_planes_from_producer", keyed on the output tensor (ops.gemm_x6 checks the key);
      backward  source = the cache dict of the Linear whose output this node received, if that layer forms its input gradient
                on gemm_x6 (producers.linear marks its output): the gradient leaves this node as planes in that dict
                ("dy_planes_from_consumer") and as a zero-stride NaN placeholder in autograd's hands -- _Linear.backward
                recognises the placeholder by its address and never reads it; anything else that consumed it would read
                NaN, not stale memory.  ONLY inside ``ops.gelu_backward_plane_handoff()`` (round 6): the
                generators of this package, which drive the backward pass themselves towards attention tensors only, opt
                in for their own forward pass; under a plain ``model(x)`` the node returns the real fp32 gradient, so
                ``autograd.grad(out, h)``, ``h.retain_grad()``, tensor hooks and second consumers see standard autograd."""

    @staticmethod
    def forward(ctx, x, feeds, source):
        ctx.save_for_backward(x)
        ctx.source = source
        if feeds is not None:
            y, xs, xa = ops.gelu_forward_planes(x)
            K = y.shape[-1]
            feeds["x_planes_from_producer"] = (ops._x_abs_key(y, y.numel() // K, K), xs, xa, y)
            return y
        return ops.gelu_forward(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if ctx.source is not None:
            planes = ops.gelu_backward_planes(dy, x)
            base = torch.full((1,), float("nan"), dtype=x.dtype, device=x.device)
            ctx.source["dy_planes_from_consumer"] = (base, tuple(x.shape), planes)
            return base.expand(x.shape), None, None
        return ops.gelu_backward(dy, x), None, None


class _Linear(torch.autograd.Function):
    """nn.Linear (modules/layers_ours.py:207: ``class Linear(nn.Linear, RelProp)``) with y = x W^T + b and / or, backward,
    d_x = d_y W on the split-operand bf16 kernels of csrc/te_linear_x6.hip -- each fp32 operand as three bf16 planes, six
    partial products, fp32 accumulation (fp32-class accuracy: tests/test_gpu_producers.py); the direction that is not
    worth an operand split (ops.gemm_x6_wanted) stays on the stock fp32 GEMM.  No weight / bias gradient (the explanation
    differentiates w.r.t. activations only; eval mode only, like the LayerNorm producer)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, fwd_x6, bwd_x6):
        ctx.weight, ctx.cache, ctx.bwd_x6 = weight, cache, bwd_x6
        if fwd_x6:
            return ops.gemm_x6(x, ops.x6_matrix_planes(weight, False, cache), bias, weight.shape[0], "linear_forward_x6",
                               keep_abs=cache)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        w = ctx.weight
        if ctx.bwd_x6:
            # the GELU node that consumed this layer's output may have left the gradient as operand planes (_Gelu.backward);
            # dy is then its zero-stride placeholder, recognised by address and shape, and is never read
            hit = ctx.cache.pop("dy_planes_from_consumer", None)
            planes = None
            if (hit is not None and dy.data_ptr() == hit[0].data_ptr() and tuple(dy.shape) == hit[1]
                    and not any(dy.stride())):
                planes = hit[2]
            dx = ops.gemm_x6(dy, ops.x6_matrix_planes(w, True, ctx.cache), None, w.shape[1], "linear_backward_x6",
                             x_planes=planes)
        else:
            dx = torch.matmul(dy, w)
        return dx, None, None, None, None, None


def linear_plan(x: torch.Tensor, lin):
    """(forward on x6?, input gradient on x6?) for this call of an nn.Linear; (False, False) = leave the layer alone."""
    if not (ops.USE_FUSED_PRODUCERS and ops.X6_GEMM != "off" and torch.is_tensor(x) and x.is_cuda
            and x.dtype == torch.float32 and not lin.training and x.dim() >= 2):
        return False, False
    out_f, in_f = lin.weight.shape
    T = x.numel() // in_f
    return ops.gemm_x6_wanted(T, in_f, out_f), ops.gemm_x6_wanted(T, out_f, in_f)


# The parameters enter the producer nodes DETACHED: these nodes produce the input gradient only, and autograd is told so
# (a caller who differentiates w.r.t. LayerNorm / Linear parameters under ops.USE_FUSED_PRODUCERS gets "no gradient" for
# them from autograd itself instead of a silent None from a node that claimed to be differentiable).
def _d(t):
    return None if t is None else t.detach()


def linear(x, lin, cache, plan):
    out = _Linear.apply(x, _d(lin.weight), _d(lin.bias), cache, plan[0], plan[1])
    if plan[1]:
        out._te_bwd_x6_cache = cache      # (for a GELU that follows: this layer's input gradient takes operand planes)
    return out


def layer_norm(x, norm):
    return _LayerNorm.apply(x, _d(norm.weight), _d(norm.bias), norm.eps)


def residual_layer_norm(x, norm):
    return _ResidualLayerNorm.apply(x, _d(norm.weight), _d(norm.bias), norm.eps)


def gelu(x, consumer=None, consumer_cache=None):
    """consumer: the Linear layer the output goes to, if the caller knows it (rules.GELU.feeds), and its cache dict."""
    feeds = source = None
    if ops.gelu_planes_supported(x):
        if consumer is not None and consumer.weight.shape[1] == x.shape[-1] and linear_plan(x, consumer)[0]:
            out_f, in_f = consumer.weight.shape
            if (ops.USE_LINEAR_X6 and ops.X6_KEEP_ABS
                    and ops.linear_relprop_x6_supported(x.numel() // in_f, in_f, out_f)):
                feeds = consumer_cache
        if ops.gelu_backward_handoff_active():     # opt-in of the caller that owns the backward pass (ops.py)
            source = getattr(x, "_te_bwd_x6_cache", None)
    if feeds is None and consumer_cache is not None:
        consumer_cache.pop("x_planes_from_producer", None)      # nothing of an earlier call may wait there for this one's output
    return _Gelu.apply(x, feeds, source)
