"""Drop-in for baselines/ViT/ViT_explanation_generator.py of the reference (classes LRP and Baselines)."""
import os as _os
import sys as _sys

_root = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "..", "..", ".."))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from transformer_explainability_amd import ops as _ops  # noqa: E402
from transformer_explainability_amd.generators import LRP, Baselines  # noqa: E402,F401


def compute_rollout_attention(all_layer_matrices, start_layer=0):
    """ViT_explanation_generator.py:7-18: rollout WITH row normalisation."""
    import torch
    return _ops.rollout(torch.stack(list(all_layer_matrices), 0), start_layer=start_layer, normalise=True)
