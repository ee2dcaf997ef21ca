"""Drop-in for modules/layers_ours.py of the reference: same names, HIP-backed relprop."""
import os as _os
import sys as _sys

_root = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "..", ".."))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from transformer_explainability_amd.rules import *  # noqa: E402,F401,F403
from transformer_explainability_amd.rules import RelProp, __all__  # noqa: E402,F401
