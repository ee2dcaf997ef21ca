"""Drop-in for baselines/ViT/ViT_LRP.py of the reference."""
import os as _os
import sys as _sys

_root = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "..", "..", ".."))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from transformer_explainability_amd.vit import (  # noqa: E402,F401
    Attention, Block, Mlp, PatchEmbed, VisionTransformer, compute_rollout_attention,
    deit_base_patch16_224, vit_base_patch16_224, vit_large_patch16_224)
