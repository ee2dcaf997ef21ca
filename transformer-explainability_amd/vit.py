"""ViT / DeiT with relevance propagation on MI355X kernels.

Host-side mirror of baselines/ViT/ViT_LRP.py (rule variant "ours", default method
"transformer_attribution") and baselines/ViT/ViT_orig_LRP.py (variant "lrp", method "grad") of the
reference: same constructor arguments, parameter names (timm / reference checkpoints load
unchanged), accessors (``blk.attn.get_attn()/get_attn_cam()/get_attn_gradients()/get_v()/
get_v_cam()``) and ``model.relprop(cam, method=..., is_ablation=..., start_layer=..., alpha=1)``.

What differs from the reference (by design, results identical at batch 1):
  * forward + backward stay on stock PyTorch-ROCm; every ``relprop`` below is HIP kernels via the C ABI
  * batch B >= 1 = B independent samples: relprop(one_hot[B,K]) returns [B, N-1]
  * q/k/v are consumed in place from the fused qkv activation and cam_q/cam_k/cam_v are written in
    place into the 'b n (qkv h d)' relevance buffer -- no rearrange copies (ViT_LRP.py:135,157,175)
  * the /2 of ViT_LRP.py:161-162,172-173 is folded into the kernels' store (exact: power of two)
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops, producers
from . import rules as R_ours

__all__ = ["VisionTransformer", "vit_base_patch16_224", "vit_large_patch16_224", "deit_base_patch16_224",
           "compute_rollout_attention", "make_vit_module"]


def compute_rollout_attention(all_layer_matrices, start_layer=0):
    """ViT_LRP.py:38-49 (identity added, NO row normalisation).  list of L [B,N,N] -> [B,N,N]."""
    return ops.rollout(torch.stack(list(all_layer_matrices), 0), start_layer=start_layer, normalise=False)


def _trunc_normal_(t, std=.02):
    with torch.no_grad():
        return nn.init.trunc_normal_(t, mean=0., std=std, a=-2., b=2.)


class _SplitQKV(torch.autograd.Function):
    """'b n (qkv h d) -> qkv b h n d' as three contiguous [B,H,N,D] tensors.  Same values as the permuted views the
    reference slices (ViT_LRP.py:135-136); the batched products need contiguous operands anyway, and the backward
    writes dq / dk / dv straight into one [B,N,3C] gradient (three strided copies) where autograd's select_backward
    would zero-fill a [3,B,H,N,D] buffer per operand and sum the three (1.6 ms per ViT-B/16 batch-64 step)."""

    @staticmethod
    def forward(ctx, qkv, num_heads):
        B, N, C3 = qkv.shape
        parts = qkv.view(B, N, 3, num_heads, C3 // (3 * num_heads)).permute(2, 0, 3, 1, 4)
        return parts[0].contiguous(), parts[1].contiguous(), parts[2].contiguous()

    @staticmethod
    def backward(ctx, dq, dk, dv):
        ref = next(g for g in (dq, dk, dv) if g is not None)
        B, H, N, D = ref.shape
        grad = torch.empty((B, N, 3, H, D), dtype=ref.dtype, device=ref.device)
        slots = grad.permute(2, 0, 3, 1, 4)
        for slot, g in zip(slots, (dq, dk, dv)):
            if g is None:
                slot.zero_()
            else:
                slot.copy_(g)
        return grad.view(B, N, 3 * H * D), None


def split_qkv(qkv, num_heads):
    return _SplitQKV.apply(qkv, num_heads)


class _FusedAttention(torch.autograd.Function):
    """the attention block of ViT_LRP.py:132-152 on the hand-written producer kernels
    (te_attention_forward_f32 / te_attention_backward_f32) instead of ~10 stock launches per direction.

    forward : qkv [B,N,3C] -> out [B,N,C]; the probabilities and the unscaled scores (the tensors the relprop rules and
              the accessors read) come back as non-differentiable by-products.
    backward: computes d_attn -- the attention gradient the explanation needs -- and hands it to the module
              (save_attn_gradients, what the reference's register_hook does, ViT_LRP.py:144-145), then d_qkv for the
              layers below; a module whose ``_fused_stop_backward`` is set is the lowest block whose gradient is
              wanted: nothing consumes d_qkv there, so only d_attn is formed."""

    @staticmethod
    def forward(ctx, qkv, num_heads, scale, module, feeds=None):
        # feeds (round 6): the cache dict of the projection layer, if that layer will run its forward product and its rule on
        # the x6 kernels -- the producer then writes the operand planes of `out` itself and the layer's split pass disappears
        # (the GELU pattern: producers._Gelu; the consumer checks that the planes belong to the tensor it receives)
        if feeds is not None:
            out, attn, zqk, xs, xa = ops.attention_forward(qkv, num_heads, scale, planes=True)
            feeds["x_planes_from_producer"] = (ops._x_abs_key(out, out.numel() // out.shape[-1], out.shape[-1]), xs, xa, out)
        else:
            out, attn, zqk = ops.attention_forward(qkv, num_heads, scale)
        ctx.save_for_backward(qkv, attn, out)      # (out: the projection's input, alive anyway; the backward's row sums come from it)
        ctx.num_heads, ctx.scale, ctx.module = num_heads, scale, module
        ctx.mark_non_differentiable(attn, zqk)
        # (round 6) no zero gradients for the two by-products: autograd otherwise fills a [B,H,N,N] zero tensor for each of them
        # in front of every backward call -- 24 x 119 MB of stores per ViT-B/16 batch-64 step that nobody reads
        ctx.set_materialize_grads(False)
        return out, attn, zqk

    @staticmethod
    def backward(ctx, d_out, _d_attn_unused, _d_zqk_unused):
        qkv, attn, out = ctx.saved_tensors
        if d_out is None:          # (nothing downstream of `out` reached the loss: with unmaterialised gradients that is a None)
            return None, None, None, None, None
        stop = bool(getattr(ctx.module, "_fused_stop_backward", False))
        d_attn, d_qkv = ops.attention_backward(d_out, qkv, attn, ctx.num_heads, ctx.scale, need_qk=not stop, out=out)
        ctx.module.save_attn_gradients(d_attn)
        return (None if stop else d_qkv), None, None, None, None


# the checkpoints the reference's factories fetch (ViT_LRP.py:24-36, 428-435 -- the same URLs in ViT_new.py / ViT_orig_LRP.py)
PRETRAINED_URLS = {
    "vit_base_patch16_224": "https://github.com/rwightman/pytorch-image-models/releases/download/v0.1-vitjx/"
                            "jx_vit_base_p16_224-80ecf9dd.pth",
    "vit_large_patch16_224": "https://github.com/rwightman/pytorch-image-models/releases/download/v0.1-vitjx/"
                             "jx_vit_large_p16_224-4ee7a4dc.pth",
    "deit_base_patch16_224": "https://dl.fbaipublicfiles.com/deit/deit_base_patch16_224-b5f2ef4d.pth",
}


def load_pretrained_weights(model, url, patch_size=16, key=None):
    """``pretrained=True`` of the reference's factories (helpers.py:87-150 ``load_pretrained`` -> ``model_zoo.load_url``;
    DeiT: ``torch.hub.load_state_dict_from_url``, ViT_LRP.py:432): the checkpoint comes from the torch hub cache
    (``$TORCH_HOME/hub/checkpoints/<file name of the URL>``) and is downloaded only if it is not there -- on a host without
    a network that means: put the file there once.  Same post-processing as the reference: a linear patch-embedding
    weight is reshaped to the convolution's [C, 3, p, p] (``_conv_filter``, ViT_LRP.py:399-406); a classifier of another
    width than the model's is dropped and the rest loaded non-strictly (helpers.py:143-147)."""
    import os
    name = os.path.basename(url)
    try:
        state = torch.hub.load_state_dict_from_url(url, map_location="cpu", progress=False)
    except Exception as exc:       # urllib's errors on a host without a network
        where = os.path.join(torch.hub.get_dir(), "checkpoints", name)
        raise RuntimeError(f"pretrained=True: {name} is not in the torch hub cache and could not be downloaded "
                           f"({type(exc).__name__}: {exc}); place the checkpoint at {where}") from exc
    if key is not None:
        state = state[key]
    state = dict(state)
    w = state.get("patch_embed.proj.weight")
    if w is not None and w.dim() == 2:
        state["patch_embed.proj.weight"] = w.reshape(w.shape[0], 3, patch_size, patch_size)
    strict = True
    if "head.weight" in state and state["head.weight"].shape[0] != model.head.weight.shape[0]:
        del state["head.weight"], state["head.bias"]
        strict = False
    model.load_state_dict(state, strict=strict)
    return model


def make_vit_module(L):
    """Build the model classes over a rule namespace ``L`` (rules for 'ours', rules_lrp for 'lrp')."""

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = L.Linear(in_features, hidden_features)
            self.act = L.GELU()
            self.fc2 = L.Linear(hidden_features, out_features)
            self.drop = L.Dropout(drop)
            if hasattr(self.act, "feeds"):
                self.act.feeds(self.fc2)      # (a hint for the producers: the activation may emit fc2's operand planes)

        def forward(self, x):
            return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

        def relprop(self, cam, **kwargs):
            # ViT_LRP.py:69-74 -- dropout / GELU rules are the identity
            cam = self.fc2.relprop(cam, **kwargs)
            return self.fc1.relprop(cam, **kwargs)

    class Attention(nn.Module):
        def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
            super().__init__()
            self.num_heads = num_heads
            self.scale = (dim // num_heads) ** -0.5
            self.matmul1 = L.einsum('bhid,bhjd->bhij')      # A = Q K^T
            self.matmul2 = L.einsum('bhij,bhjd->bhid')      # out = A V
            self.qkv = L.Linear(dim, dim * 3, bias=qkv_bias)
            self.attn_drop = L.Dropout(attn_drop)
            self.proj = L.Linear(dim, dim)
            self.proj_drop = L.Dropout(proj_drop)
            self.softmax = L.Softmax(dim=-1)
            self.attn_cam = self.attn = self.v = self.v_cam = self.attn_gradients = None

        # accessors of ViT_LRP.py:102-130
        def get_attn(self): return self.attn
        def save_attn(self, attn): self.attn = attn
        # accessor names of the plain (non-LRP) model, baselines/ViT/ViT_new.py:78-88
        def get_attention_map(self): return self.attn
        def save_attention_map(self, attn): self.attn = attn
        def save_attn_cam(self, cam): self.attn_cam = cam
        def get_attn_cam(self): return self.attn_cam
        def get_v(self): return self.v
        def save_v(self, v): self.v = v
        def save_v_cam(self, cam): self.v_cam = cam
        def get_v_cam(self): return self.v_cam
        def save_attn_gradients(self, g): self.attn_gradients = g
        def get_attn_gradients(self): return self.attn_gradients

        def forward(self, x):
            B, N, C = x.shape
            H = self.num_heads
            self._fused_anchor = None
            if (ops.USE_FUSED_PRODUCERS and x.is_cuda and x.dtype == torch.float32 and not self.training
                    and ops.attention_forward_supported(N, C // H)):
                return self._forward_fused(x)
            q, k, v = split_qkv(self.qkv(x), H)                                 # 'b n (qkv h d) -> qkv b h n d'
            self.save_v(v)
            attn = self.attn_drop(self.softmax(self.matmul1([q, k]) * self.scale))
            self.save_attn(attn)
            if attn.requires_grad:
                attn.register_hook(self.save_attn_gradients)
            out = self.matmul2([attn, v]).permute(0, 2, 1, 3).reshape(B, N, C)  # 'b h n d -> b n (h d)'
            return self.proj_drop(self.proj(out))

        def _forward_fused(self, x):
            """ViT_LRP.py:132-152 on the producer kernels.  q / k / v are never copied out of the fused activation: the
            rule modules' caches (matmul1.X / .Y, matmul2.X / .Y) are strided views of it and of the 'b n (h d)'
            output, which the relprop kernels read in place."""
            B, N, C = x.shape
            H, D = self.num_heads, C // self.num_heads
            qkv = self.qkv(x)
            feeds = None
            if isinstance(self.proj, L.Linear) and ops.attention_forward_planes_supported(qkv, H):
                from .rules import x6_cache
                out_f, in_f = self.proj.weight.shape
                # (the layer's own plan for an input of out's shape: producers.linear_plan / producers.gelu)
                if (in_f == C and ops.USE_FUSED_PRODUCERS and ops.X6_GEMM != "off" and not self.proj.training
                        and ops.gemm_x6_wanted(B * N, in_f, out_f) and ops.USE_LINEAR_X6 and ops.X6_KEEP_ABS
                        and ops.linear_relprop_x6_supported(B * N, in_f, out_f)):
                    feeds = x6_cache(self.proj)
            out, attn, zqk = _FusedAttention.apply(qkv, H, self.scale, self, feeds)
            self._fused_anchor = qkv if qkv.requires_grad else None
            q, k, v = qkv.detach().view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
            self.save_v(v)
            self.save_attn(attn)
            zav = out.detach().view(B, N, H, D).permute(0, 2, 1, 3)
            for mod, X, Y in ((self.matmul1, [q, k], zqk), (self.matmul2, [attn, v], zav)):
                mod.X, mod.Y = X, Y
                mod._y_version, mod._w_version = Y._version, (None, None)
            return self.proj_drop(self.proj(out))

        def relprop(self, cam, **kwargs):
            """ViT_LRP.py:154-177."""
            return self.relprop_after_proj(self.proj.relprop(cam, **kwargs), **kwargs)

        def relprop_after_proj(self, cam, **kwargs):
            """ViT_LRP.py:157-177: everything below proj.relprop (the two attention rules and qkv.relprop)."""
            B, N, C = cam.shape
            H = self.num_heads
            D = C // H
            r_heads = cam.view(B, N, H, D).permute(0, 2, 1, 3)                  # strided view, no copy
            attn, v = self.matmul2.X
            q, k = self.matmul1.X
            cam_qkv = torch.empty((B, N, 3 * C), dtype=cam.dtype, device=cam.device)
            slots = cam_qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)          # [3][B,H,N,D] views
            var = self.matmul2.variant
            cam1, cam_v = ops.matmul_relprop_av(r_heads, attn, v, out_scale=0.5, cam_v_out=slots[2], variant=var,
                                                z=R_ours._cached_y(self.matmul2))
            self.save_v_cam(cam_v)
            self.save_attn_cam(cam1)
            if getattr(self, "_stop_after_attn_cam", False):
                raise L.StopRelprop()
            ops.matmul_relprop_qk(cam1, q, k, out_scale=0.5, cam_q_out=slots[0], cam_k_out=slots[1], variant=var,
                                  z=R_ours._cached_y(self.matmul1))
            return self.qkv.relprop(cam_qkv, **kwargs)

    class Block(nn.Module):
        def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., norm_eps=1e-6):
            super().__init__()
            self.norm1 = L.LayerNorm(dim, eps=norm_eps)
            self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
            self.norm2 = L.LayerNorm(dim, eps=norm_eps)
            self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)
            self.add1 = L.Add()
            self.add2 = L.Add()
            self.clone1 = L.Clone()
            self.clone2 = L.Clone()

        def forward(self, x):
            x1, n = self._clone_norm(self.clone1, self.norm1, x)
            x = self.add1([x1, self.attn(n)])
            x1, n = self._clone_norm(self.clone2, self.norm2, x)
            return self.add2([x1, self.mlp(n)])

        @staticmethod
        def _clone_norm(clone, norm, x):
            """``x1, x2 = clone(x, 2); n = norm(x2)`` (ViT_LRP.py:203-205).  With the producer kernels on, the pair is one
            autograd node whose backward kernel also adds the bypass gradient (producers._ResidualLayerNorm); the two
            modules' caches are filled as their forward hooks would."""
            if producers.norm_usable(x, norm):
                x1, n = producers.residual_layer_norm(x, norm)
                clone.X, clone.num = x.detach(), 2
                norm.X, norm.Y = clone.X, n
                return x1, n
            x1, x2 = clone(x, 2)
            return x1, norm(x2)

        def relprop(self, cam, **kwargs):
            """ViT_LRP.py:203-213 (LayerNorm rules are the identity)."""
            cam1, cam2 = self.add2.relprop(cam, deferred=True, **kwargs)
            cam2 = self.mlp.relprop(cam2, **kwargs)
            cam = self.clone2.relprop((cam1, cam2), **kwargs)
            cam1, cam2 = self.add1.relprop(cam, deferred=True, **kwargs)
            cam2 = self.attn.relprop(cam2, **kwargs)
            return self.clone1.relprop((cam1, cam2), **kwargs)

        def relprop_cls_only(self, cam_cls, **kwargs):
            """Block.relprop for relevance that lives on token 0 only (cam_cls [B,1,C]) -- the state right after
            pool.relprop (ViT_LRP.py:329), i.e. the LAST block.  Every rule from add2 down to proj maps a zero
            relevance row to an exact zero row (S = safe_divide(0, Z) = 0), and Add's per-sample sums gain only
            zeros from those rows, so the six rules are evaluated on the [B,1,C] slice of their cached inputs and
            the result is scattered into a zero [B,N,C] tensor before the attention rules, which spread relevance
            to all tokens.  Bitwise equal to the dense evaluation at 1/N of its Linear work."""
            alpha = kwargs.get("alpha", 1)
            var = self.add2.variant
            cls = lambda t: t[:, :1]                                             # noqa: E731
            dfr = ops.USE_DEFERRED_ADD
            c1, c2 = ops.add_relprop(cam_cls, cls(self.add2.X[0]), cls(self.add2.X[1]), variant=var, deferred=dfr)
            def lin(r, m):       # the staleness guard of the cached forward output applies here as in Linear.relprop
                y = R_ours._cached_y(m)
                return ops.linear_relprop(r, cls(m.X), m.weight.detach(), alpha=alpha, variant=var,
                                          Y=None if y is None else cls(y), bias=m.bias, cache=R_ours.x6_cache(m))
            c2 = lin(lin(c2, self.mlp.fc2), self.mlp.fc1)
            cam = ops.clone_relprop((c1, c2), cls(self.clone2.X))
            c1, c2 = ops.add_relprop(cam, cls(self.add1.X[0]), cls(self.add1.X[1]), variant=var, deferred=dfr)
            c2 = lin(c2, self.attn.proj)
            if isinstance(c1, ops.Deferred):
                c1 = c1.materialise()
            B, N, C = self.clone1.X.shape
            dense = torch.zeros((2, B, N, C), dtype=c2.dtype, device=c2.device)
            dense[0, :, 0] = c1[:, 0]
            dense[1, :, 0] = c2[:, 0]
            cam2 = self.attn.relprop_after_proj(dense[1], **kwargs)
            return self.clone1.relprop((dense[0], cam2), **kwargs)

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
            patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
            self.img_size, self.patch_size = img_size, patch_size
            self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
            self.proj = L.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

        def forward(self, x):
            B, C, Hh, Ww = x.shape
            assert Hh == self.img_size[0] and Ww == self.img_size[1], \
                f"Input image size ({Hh}*{Ww}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
            return self.proj(x).flatten(2).transpose(1, 2)

        def relprop(self, cam, **kwargs):
            # ViT_LRP.py:238-242: [B,P,E] -> [B,E,Hp,Wp]; built as a VIEW so the kernel reads the token-major memory
            # in place (the reference's transpose + reshape copies)
            cam = cam.unflatten(1, (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]))
            return self.proj.relprop(cam.permute(0, 3, 1, 2), **kwargs)

    class VisionTransformer(nn.Module):
        default_method = "transformer_attribution" if L.RelProp.variant == "ours" else "grad"

        def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                     num_heads=12, mlp_ratio=4., qkv_bias=False, mlp_head=False, drop_rate=0., attn_drop_rate=0.,
                     block_norm_eps=1e-6, final_norm_eps=1e-5):
            # (the LayerNorm epsilons are ViT_LRP.py:184,187,266; ViT_new.py uses other values -- see the drop-in)
            super().__init__()
            self.num_classes = num_classes
            self.num_features = self.embed_dim = embed_dim
            self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                          embed_dim=embed_dim)
            num_patches = self.patch_embed.num_patches
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
            self.blocks = nn.ModuleList([
                Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                      attn_drop=attn_drop_rate, norm_eps=block_norm_eps) for _ in range(depth)])
            self.norm = L.LayerNorm(embed_dim, eps=final_norm_eps)
            self.head = Mlp(embed_dim, int(embed_dim * mlp_ratio), num_classes) if mlp_head \
                else L.Linear(embed_dim, num_classes)
            _trunc_normal_(self.pos_embed, std=.02)
            _trunc_normal_(self.cls_token, std=.02)
            self.apply(self._init_weights)
            self.pool = L.IndexSelect()
            self.add = L.Add()
            self.inp_grad = None
            self.exploit_cls_sparsity = True     # exact; set False to evaluate the last block densely
            # (extension, off by default) "transformer_attribution" / "grad" read attn_cam of blocks >= start_layer only
            # (ViT_LRP.py:357-369), yet the reference propagates relevance through every block.  With this flag the
            # loop stops right after block start_layer's AV rule stored its attn_cam: same map bit for bit, 1/L of the
            # relprop work less per skipped block -- but get_attn_cam() of blocks < start_layer is no longer refreshed.
            self.prune_below_start_layer = False
            self.register_buffer("_cls_index", torch.zeros((), dtype=torch.long), persistent=False)

        def save_inp_grad(self, grad): self.inp_grad = grad
        def get_inp_grad(self): return self.inp_grad

        def _init_weights(self, m):
            if isinstance(m, nn.Linear):
                _trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

        @property
        def no_weight_decay(self):
            return {'pos_embed', 'cls_token'}

        def forward(self, x, register_hook=False):
            # register_hook (ViT_new.py:195): the attention-gradient hook is always registered when a graph is built
            B = x.shape[0]
            x = self.patch_embed(x)
            x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
            x = self.add([x, self.pos_embed])
            if x.requires_grad:
                x.register_hook(self.save_inp_grad)
            for blk in self.blocks:
                x = blk(x)
            x = self.norm(x)
            x = self.pool(x, dim=1, indices=self._cls_index).squeeze(1)   # (a device tensor made once: graph-capturable)
            return self.head(x)

        # ------------------------------------------------------------------------------------------
        def relprop(self, cam=None, method=None, is_ablation=False, start_layer=0, **kwargs):
            """ViT_LRP.py:324-398.  cam: one-hot [B, num_classes]; returns per-sample maps."""
            if method is None:
                method = self.default_method
            prune = self.prune_below_start_layer and method in ("transformer_attribution", "grad")
            stop_at = self.blocks[start_layer].attn if prune else None
            if stop_at is not None:
                stop_at._stop_after_attn_cam = True
            try:
                cam = self.head.relprop(cam, **kwargs)
                if self.exploit_cls_sparsity and isinstance(self.head, nn.Linear):
                    # pool.relprop puts relevance on the class token only: keep it as a [B,1,C] row through the last
                    # block's dense rules instead of a [B,N,C] tensor that is zero everywhere else
                    cam = ops.index_select_relprop(cam.unsqueeze(1), self.pool.X[:, :1], 0)
                    cam = self.blocks[-1].relprop_cls_only(cam, **kwargs)
                    rest = list(self.blocks)[:-1]
                else:
                    cam = self.pool.relprop(cam.unsqueeze(1), **kwargs)
                    rest = list(self.blocks)
                for blk in reversed(rest):
                    cam = blk.relprop(cam, **kwargs)
            except L.StopRelprop:
                cam = None
            finally:
                if stop_at is not None:
                    stop_at._stop_after_attn_cam = False

            hook = getattr(self, "_before_tail", None)    # LRP(overlap_backward=True): join with the backward pass here
            if hook is not None:
                hook()

            if method == "full":
                # ViT_LRP.py:337-343: position-embedding Add, drop the class token, z^B rule of the patch
                # embedding, sum over the colour channels -> [B, H, W]
                cam, _ = self.add.relprop(cam, **kwargs)
                cam = self.patch_embed.relprop(cam[:, 1:], **kwargs)
                return cam.sum(dim=1)

            if method == "rollout":
                mats = [blk.attn.get_attn_cam().clamp(min=0).mean(dim=1) for blk in self.blocks]
                return compute_rollout_attention(mats, start_layer=start_layer)[:, 0, 1:]

            if method in ("transformer_attribution", "grad"):
                # ViT_LRP.py:357-369: per block mean_h max(grad * attn_cam, 0), then rollout, row 0
                first = self.blocks[-1].attn.get_attn_cam()
                Bn, _, N, _ = first.shape
                stack = torch.empty((len(self.blocks), Bn, N, N), dtype=first.dtype, device=first.device)
                for i, blk in enumerate(self.blocks):
                    if i >= start_layer or not prune:       # (the rollout reads layers >= start_layer only)
                        ops.gradcam_headmean(blk.attn.get_attn_gradients(), blk.attn.get_attn_cam(), out=stack[i])
                return ops.rollout(stack, start_layer=start_layer, normalise=False, row0_only=True)[:, 1:]

            if method in ("last_layer", "second_layer"):
                blk = self.blocks[-1] if method == "last_layer" else self.blocks[1]
                c = blk.attn.get_attn_cam()
                if is_ablation:
                    c = blk.attn.get_attn_gradients() * c
                return c.clamp(min=0).mean(dim=1)[:, 0, 1:]

            if method == "last_layer_attn":
                return self.blocks[-1].attn.get_attn().clamp(min=0).mean(dim=1)[:, 0, 1:]
            return None   # unknown method: the reference falls through silently

    def vit_base_patch16_224(pretrained=False, **kwargs):
        model = VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                                  **kwargs)
        if pretrained:
            load_pretrained_weights(model, PRETRAINED_URLS["vit_base_patch16_224"], patch_size=16)
        return model

    def vit_large_patch16_224(pretrained=False, **kwargs):
        model = VisionTransformer(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                                  **kwargs)
        if pretrained:
            load_pretrained_weights(model, PRETRAINED_URLS["vit_large_patch16_224"], patch_size=16)
        return model

    def deit_base_patch16_224(pretrained=False, **kwargs):
        model = VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                                  **kwargs)
        if pretrained:
            load_pretrained_weights(model, PRETRAINED_URLS["deit_base_patch16_224"], patch_size=16, key="model")
        return model

    ns = dict(Mlp=Mlp, Attention=Attention, Block=Block, PatchEmbed=PatchEmbed, VisionTransformer=VisionTransformer,
              vit_base_patch16_224=vit_base_patch16_224, vit_large_patch16_224=vit_large_patch16_224,
              deit_base_patch16_224=deit_base_patch16_224)
    return ns


_ns = make_vit_module(R_ours)
Mlp, Attention, Block, PatchEmbed = _ns["Mlp"], _ns["Attention"], _ns["Block"], _ns["PatchEmbed"]
VisionTransformer = _ns["VisionTransformer"]
vit_base_patch16_224 = _ns["vit_base_patch16_224"]
vit_large_patch16_224 = _ns["vit_large_patch16_224"]
deit_base_patch16_224 = _ns["deit_base_patch16_224"]
