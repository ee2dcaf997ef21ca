"""Drop-in for baselines/ViT/ViT_orig_LRP.py of the reference (layers_lrp rules, method "grad")."""
import os as _os
import sys as _sys

_root = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "..", "..", ".."))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from transformer_explainability_amd import rules_lrp as _rules  # noqa: E402
from transformer_explainability_amd.vit import compute_rollout_attention, make_vit_module  # noqa: E402,F401

_ns = make_vit_module(_rules)
Mlp, Attention, Block, PatchEmbed = _ns['Mlp'], _ns['Attention'], _ns['Block'], _ns['PatchEmbed']
VisionTransformer = _ns['VisionTransformer']
vit_base_patch16_224 = _ns['vit_base_patch16_224']
vit_large_patch16_224 = _ns['vit_large_patch16_224']
