"""CPU check of the harness behind tests/test_reference_scripts.py: the same byte-identical reference scripts, the same
stubs, synthetic data and checkpoint -- run with plain ``python script.py`` (the script's own directory first, i.e. over the
REFERENCE'S OWN ViT_LRP / ViT_new / ViT_explanation_generator) and ``.cuda()`` shimmed to the identity.  Proves on a host
without a GPU that the stand-ins implement what the scripts call and that the synthetic datasets have the layout the
reference's readers expect; the `-m gpu` tests then swap the reference's modules for the drop-in and nothing else."""
import os
import subprocess
import sys

import numpy as np

from test_reference_scripts import N_IMAGES, ROOT, arena  # noqa: F401  (the fixture is shared)


def _run_plain(arena, script, *args):   # noqa: F811
    env = dict(arena["env"])
    env["PYTHONPATH"] = os.path.join(ROOT, "tests", "refscripts", "cpu_shim") + os.pathsep + env["PYTHONPATH"]
    env["HIP_VISIBLE_DEVICES"] = ""      # (a GPU box must run this leg on the host too)
    r = subprocess.run([sys.executable, os.path.join(arena["ref"], script), *args], cwd=arena["top"], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, f"{script} failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return r


def test_reference_scripts_run_on_their_own_modules(arena):   # noqa: F811
    _run_plain(arena, "baselines/ViT/generate_visualizations.py", "--method", "transformer_attribution",
               "--imagenet-validation-path", arena["imagenet"])
    out = os.path.join(arena["vit_dir"], "visualizations", "transformer_attribution", "top", "not_ablation", "results.hdf5")
    with np.load(out) as z:
        vis = z["vis"]
        assert vis.shape == (N_IMAGES, 1, 224, 224) and z["image"].shape == (N_IMAGES, 3, 224, 224)
    assert np.isfinite(vis).all() and vis.min() == 0.0 and vis.max() == 1.0
    _run_plain(arena, "baselines/ViT/pertubation_eval_from_hdf5.py", "--method", "transformer_attribution", "--batch-size", "4")
    exp = os.path.join(arena["vit_dir"], "experiments", "perturbations", "transformer_attribution_neg", "top", "not_ablation",
                       "experiment_0")
    assert np.load(os.path.join(exp, "perturbations_logit_diff.npy")).shape == (9, N_IMAGES)
    r = _run_plain(arena, "baselines/ViT/imagenet_seg_eval.py", "--method", "transformer_attribution",
                   "--imagenet-seg-path", arena["seg"])
    assert "Mean IoU over 2 classes" in r.stdout
    os.remove(out)      # the GPU tests of the same session start from an empty arena
