"""TEST-ONLY stand-in for transformer_explainability_amd.ops backed by the CPU oracle.

Lets the `-m "not gpu"` suite exercise the host logic (rule classes, model relprop composition,
stride/view plumbing, generators, sharding) on a machine without a GPU.  The product never uses this.
"""
import contextlib

import torch

from oracle import relprop_oracle as O


def _plain(r):
    """A Deferred relevance operand (unscaled tensor + per-sample factor) as the plain tensor the rule consumes."""
    from transformer_explainability_amd import ops
    return r.materialise() if isinstance(r, ops.Deferred) else r


def linear_relprop(R, X, W, alpha=1.0, variant="ours", Y=None, bias=None, cache=None):
    return O.linear_relprop(_plain(R), X, W, alpha=alpha, variant=variant)     # Y / bias: a device-side shortcut only


def matmul_relprop_av(R, attn, v, out_scale=1.0, cam_v_out=None, variant="ours", z=None):
    c_attn, c_v = O.einsum_av_relprop(R, attn, v)
    c_attn = c_attn * out_scale
    c_v = c_v * out_scale
    if cam_v_out is not None:
        cam_v_out.copy_(c_v)
        c_v = cam_v_out
    return c_attn.contiguous(), c_v


def matmul_relprop_qk(R, q, k, out_scale=1.0, cam_q_out=None, cam_k_out=None, variant="ours", z=None):
    c_q, c_k = O.einsum_qk_relprop(_plain(R), q, k)
    c_q = c_q * out_scale
    c_k = c_k * out_scale
    if cam_q_out is not None:
        cam_q_out.copy_(c_q)
        c_q = cam_q_out
    if cam_k_out is not None:
        cam_k_out.copy_(c_k)
        c_k = cam_k_out
    return c_q, c_k


def add_relprop(R, X0, X1, variant="ours", deferred=False):
    a, b = O.add_relprop(R, X0, X1, variant)
    if deferred and variant == "ours" and a.shape == b.shape:
        # same host-side contract as the device op: (tensor, per-sample factor) pairs.  The oracle has already applied
        # the rescale, so the factor is an exact 1 (x * 1.0f is the identity in fp32).
        from transformer_explainability_amd import ops
        one = torch.ones((X0.shape[0], 2), dtype=a.dtype)
        return ops.Deferred(a, one[:, 0]), ops.Deferred(b, one[:, 1])
    return a, b


def clone_relprop(Rs, X):
    return O.clone_relprop([_plain(r) for r in Rs], X)


def index_select_relprop(R, X, index):
    return O.index_select_relprop(R.reshape(X.shape[0], 1, X.shape[2]), X, 1, index)


def gradcam_headmean(grad, cam, out=None):
    r = O.gradcam_headmean(grad, cam)
    if out is not None:
        out.copy_(r)
        return out
    return r


def rollout(cams, start_layer=0, normalise=False, cls_fixup=False, row0_only=False):
    joint = O.rollout(list(cams), start_layer, normalise=normalise).clone()
    if cls_fixup:
        joint[:, 0, 0] = joint[:, 0].min(dim=-1).values
    return joint[:, 0] if row0_only else joint


def conv2d_zb_relprop(R, X, W, Y, bias=None):
    return O.conv2d_zb_relprop(R.contiguous(), X, W.detach(), W.shape[-1])   # Y / bias: a device-side shortcut only


def perturb(vis, data, ks, mean=None, std=None):
    return O.perturb(vis.reshape(data.shape[0], -1), data, ks, mean, std)


def heatmap(maps, scale=16, normalise=True, with_mask=False):
    B = maps.shape[0]
    g = int(round((maps.numel() // B) ** 0.5))
    heat, mask = O.heatmap(maps.reshape(B, g * g), scale=scale, normalise=normalise)
    return (heat, mask) if with_mask else heat


_NAMES = ["heatmap", "perturb", "conv2d_zb_relprop", "linear_relprop", "matmul_relprop_av", "matmul_relprop_qk", "add_relprop", "clone_relprop",
          "index_select_relprop", "gradcam_headmean", "rollout"]


@contextlib.contextmanager
def oracle_ops():
    """Temporarily route transformer_explainability_amd.ops.* to the oracle (CPU tensors)."""
    from transformer_explainability_amd import ops
    saved = {n: getattr(ops, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(ops, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
