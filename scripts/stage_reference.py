#!/usr/bin/env python
"""Stage the reference's own hot-path files into the git-ignored ``oracle/_ref/``.

    python scripts/stage_reference.py [--src /root/reference] [--check]

``/root/reference`` exists only in the build container.  The GPU box receives a snapshot of this repository (minus
``.git`` and ``.gpurunignore`` paths), so a *staged, git-ignored* copy of the files the harness imports is what lets
``bench.py``'s ``cpu_baseline`` leg time THE REFERENCE ITSELF (``kind: "reference"``) on the host cores of the box the
kernels are measured on, and lets the CPU tests run the live reference against the oracle wherever the stage exists.

Rules of the stage:
  * outputs go to ``oracle/_ref/`` only; that directory is listed in ``.gitignore`` (never in history) and NOT in
    ``.gpurunignore`` (it travels with the snapshot like the built ``.so`` files);
  * files are copied byte for byte (a MANIFEST with sha256 sums is written next to them) -- nothing is edited; the
    compat shims stay in ``oracle/ref_harness.py``;
  * nothing under ``transformer-explainability_amd/`` may import from the stage (tests/test_no_oracle_in_product.py);
    users: ``oracle/ref_harness.py`` (fallback root), ``tests/``, ``bench.py``'s cpu_baseline leg, ``smoke()``.

``__graft_entry__.build()`` calls ``stage()`` when ``/root/reference`` is present (building the checker is not using
it); on a host without the checkout the existing stage, if any, is left untouched.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEST = os.path.join(ROOT, "oracle", "_ref")
DEFAULT_SRC = os.environ.get("TE_REFERENCE_SRC", "/root/reference")

# The import closure of oracle/ref_harness.load_reference_{vit,bert,perturbation_eval}: the rule modules, the
# LRP-instrumented models, the generators, and the few helper modules those import at module level.
FILES = [
    "modules/__init__.py",
    "modules/layers_ours.py",
    "modules/layers_lrp.py",
    "baselines/ViT/ViT_LRP.py",
    "baselines/ViT/ViT_orig_LRP.py",
    "baselines/ViT/ViT_new.py",
    "baselines/ViT/ViT_explanation_generator.py",
    "baselines/ViT/helpers.py",
    "baselines/ViT/weight_init.py",
    "baselines/ViT/layer_helpers.py",
    "baselines/ViT/pertubation_eval_from_hdf5.py",
    # the evaluation scripts that tests/test_reference_scripts.py executes UNMODIFIED over the drop-in,
    # and the support modules they import next to the drop-in's (off the hot path: metrics, dataset readers, savers)
    "baselines/ViT/imagenet_seg_eval.py",
    "baselines/ViT/generate_visualizations.py",
    "baselines/ViT/misc_functions.py",
    "dataset/__init__.py",
    "dataset/expl_hdf5.py",
    "data/__init__.py",
    "data/Imagenet.py",
    "utils/render.py",
    "utils/saver.py",
    "utils/iou.py",
    "utils/metric.py",
    "utils/confusionmatrix.py",
    "BERT_explainability/modules/__init__.py",
    "BERT_explainability/modules/layers_ours.py",
    "BERT_explainability/modules/layers_lrp.py",
    "BERT_explainability/modules/BERT/BERT.py",
    "BERT_explainability/modules/BERT/BERT_orig_lrp.py",
    "BERT_explainability/modules/BERT/BERT_cls_lrp.py",
    "BERT_explainability/modules/BERT/BertForSequenceClassification.py",
    "BERT_explainability/modules/BERT/ExplanationGenerator.py",
    "BERT_rationale_benchmark/__init__.py",
    "BERT_rationale_benchmark/models/model_utils.py",
    "utils/__init__.py",
    "utils/metrices.py",
    "LICENSE",
]


def _sha(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def stage(src: str = DEFAULT_SRC, dest: str = DEST, quiet: bool = False) -> bool:
    """Copy FILES from `src` to `dest`.  Returns False (and touches nothing) when `src` is absent."""
    if not os.path.isdir(os.path.join(src, "modules")):
        return False
    manifest = {}
    for rel in FILES:
        s = os.path.join(src, rel)
        if not os.path.exists(s):
            continue
        d = os.path.join(dest, rel)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        if not os.path.exists(d) or _sha(d) != _sha(s):
            shutil.copyfile(s, d)
        manifest[rel] = _sha(d)
    with open(os.path.join(dest, "MANIFEST.json"), "w") as f:
        json.dump({"source": src, "note": "byte-for-byte copies; git-ignored test infrastructure (see "
                                          "scripts/stage_reference.py)", "sha256": manifest}, f, indent=1, sort_keys=True)
    if not quiet:
        print(f"staged {len(manifest)} reference files into {os.path.relpath(dest, ROOT)}/")
    return True


def check(dest: str = DEST) -> bool:
    """True iff a stage exists and every file matches its recorded checksum."""
    mpath = os.path.join(dest, "MANIFEST.json")
    if not os.path.exists(mpath):
        return False
    with open(mpath) as f:
        man = json.load(f)["sha256"]
    return all(os.path.exists(os.path.join(dest, rel)) and _sha(os.path.join(dest, rel)) == h for rel, h in man.items())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default=DEFAULT_SRC)
    ap.add_argument("--check", action="store_true", help="verify an existing stage instead of copying")
    a = ap.parse_args()
    if a.check:
        ok = check()
        print("stage ok" if ok else "no valid stage")
        sys.exit(0 if ok else 1)
    if not stage(a.src):
        print(f"{a.src} not found: nothing staged", file=sys.stderr)
        sys.exit(1)
