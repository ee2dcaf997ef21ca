"""transformer-explainability_amd: MI355X-native relevance propagation ("transformer_attribution")
behind the API of hila-chefer/Transformer-Explainability.

Importable as ``transformer_explainability_amd`` (see the alias module at the repo root; the
directory name carries a hyphen).  Layout:

  csrc/ + ../include/te_relprop.h   hand-written HIP kernels (gfx950) and the C ABI
  _lib.py, ops.py                   ctypes binding, tensor-level wrappers (no CPU fallback)
  rules.py, rules_lrp.py            rule classes = modules/layers_ours.py / layers_lrp.py of the reference
  vit.py, bert.py                   LRP-instrumented models = baselines/ViT/ViT_LRP.py, BERT.py ... of the reference
  generators.py                     LRP.generate_LRP / Generator.generate_LRP
  dropin/                           the reference's import paths (modules.layers_ours, baselines.ViT.ViT_LRP, ...)
  parallel.py                       one-process-per-GPU sharding + RCCL gather of the finished maps
  sweep.py, perturbation.py, segmentation.py   the evaluation protocols around the maps
  tuning/                           PyTorch TunableOp selection of the stock fp32 GEMMs of forward + backward
"""
from . import _lib, ops, rules, rules_lrp  # noqa: F401
from ._lib import TeError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"


def library_path() -> str:
    return LIB_PATH


def install_dropin() -> str:
    """Put the reference-compatible import paths (modules.*, baselines.*, BERT_explainability.*) on
    sys.path so the reference's evaluation scripts import this implementation unchanged."""
    import os
    import sys
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")
    # the evaluation scripts run from baselines/ViT and import ``ViT_LRP`` / ``ViT_new`` / ``ViT_explanation_generator``
    # by their bare names (imagenet_seg_eval.py:19-22), the BERT pipeline uses package paths (bert_pipeline.py:17)
    for p in (os.path.join(d, "baselines", "ViT"), d):
        if p not in sys.path:
            sys.path.insert(0, p)
    return d


def enable_tuned_gemms(path=None, tune=False) -> bool:
    """Producers: the ViT / BERT forward and attention-gradient backward stay on stock PyTorch-ROCm,
    whose default heuristic picks fp32 GEMM kernels that run at 97-125 TF on the ViT-B/16 batch-64 shapes; PyTorch's
    own TunableOp, given one tuning pass over rocBLAS / hipBLASLt's solutions, finds ones at 117-148 TF (+7 % on the
    whole generate_LRP step).  This loads the committed selection for gfx950 (tuning/tunableop_gfx950.csv, keyed by
    GEMM shape and validated against the PyTorch / rocBLAS / hipBLASLt versions; unknown shapes keep the default
    kernel).  ``tune=True`` additionally tunes shapes the file does not hold (seconds per shape, then written back to
    ``path``).  Returns False when TunableOp is unavailable or the file does not match this PyTorch / rocBLAS /
    hipBLASLt build (PyTorch's default kernels are then left in place)."""
    import os
    import torch
    try:
        import torch.cuda.tunable as tunable
    except ImportError:
        return False
    if not torch.cuda.is_available():
        return False
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")
    tunable.enable(True)
    tunable.tuning_enable(bool(tune))
    tunable.set_filename(path, insert_device_ordinal=False)
    loaded = bool(os.path.exists(path) and tunable.read_file(path))   # False: validators (library versions) differ
    if not loaded and not tune:
        tunable.enable(False)                                           # nothing to apply: leave PyTorch's default
    return loaded or bool(tune)
