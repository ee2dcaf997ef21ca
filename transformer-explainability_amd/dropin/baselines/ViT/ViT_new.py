"""Drop-in for baselines/ViT/ViT_new.py of the reference: the plain ViT used by the attention baselines and by the
perturbation evaluation.  The LRP-instrumented model is a superset (same parameters / state-dict keys, plus
``get_attention_map()`` / ``forward(x, register_hook=...)``), so it is re-used under the plain names with ViT_new's
LayerNorm epsilons: nn.LayerNorm's default 1e-5 everywhere when the class is constructed directly (ViT_new.py:113,154),
1e-6 everywhere from the factory functions (ViT_new.py:224,234)."""
import os as _os
import sys as _sys

_root = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "..", "..", ".."))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from transformer_explainability_amd import vit as _vit  # noqa: E402
from transformer_explainability_amd.vit import Attention, Block, Mlp, PatchEmbed  # noqa: E402,F401


def _eps_of(norm_layer):
    return 1e-5 if norm_layer is None else float(norm_layer(1).eps)


class VisionTransformer(_vit.VisionTransformer):
    def __init__(self, *args, norm_layer=None, **kwargs):
        eps = _eps_of(norm_layer)
        kwargs.setdefault("block_norm_eps", eps)
        kwargs.setdefault("final_norm_eps", eps)
        super().__init__(*args, **kwargs)


def vit_base_patch16_224(pretrained=False, **kwargs):
    model = VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                              block_norm_eps=1e-6, final_norm_eps=1e-6, **kwargs)
    if pretrained:       # ViT_new.py:222-230: the same checkpoint as the LRP-instrumented model, from the torch hub cache
        _vit.load_pretrained_weights(model, _vit.PRETRAINED_URLS["vit_base_patch16_224"], patch_size=16)
    return model


def vit_large_patch16_224(pretrained=False, **kwargs):
    model = VisionTransformer(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                              block_norm_eps=1e-6, final_norm_eps=1e-6, **kwargs)
    if pretrained:
        _vit.load_pretrained_weights(model, _vit.PRETRAINED_URLS["vit_large_patch16_224"], patch_size=16)
    return model
