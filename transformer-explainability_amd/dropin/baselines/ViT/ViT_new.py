"""Drop-in for baselines/ViT/ViT_new.py of the reference: the plain ViT used by the attention baselines and by the
perturbation evaluation.  The LRP-instrumented model is a superset (same parameters / state-dict keys, same forward,
plus ``get_attention_map()`` / ``forward(x, register_hook=...)``), so it is re-exported under the plain names."""
import os as _os
import sys as _sys

_root = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "..", "..", ".."))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from transformer_explainability_amd.vit import (  # noqa: E402,F401
    Attention, Block, Mlp, PatchEmbed, VisionTransformer, vit_base_patch16_224, vit_large_patch16_224)
