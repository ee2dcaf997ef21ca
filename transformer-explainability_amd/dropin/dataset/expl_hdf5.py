"""Drop-in for dataset/expl_hdf5.py of the reference (class ImagenetResults): reads the reference's results.hdf5 (needs
h5py) or the sharded .npy store the sweep of this package writes."""
import os as _os
import sys as _sys

_root = _os.path.abspath(_os.path.join(_os.path.dirname(__file__), "..", "..", ".."))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from transformer_explainability_amd.sweep import ImagenetResults  # noqa: E402,F401
