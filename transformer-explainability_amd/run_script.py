"""Run one of the reference's evaluation scripts, unmodified, over this implementation:

    python -m transformer_explainability_amd.run_script <reference>/baselines/ViT/imagenet_seg_eval.py --method ...

``python script.py`` puts the script's own directory first on ``sys.path``; for the scripts under ``baselines/ViT``
that directory holds the reference's ``ViT_LRP.py`` / ``ViT_new.py`` / ``ViT_explanation_generator.py``, which would
shadow the drop-in modules of the same names.  This runner does what ``python script.py`` does -- ``__main__`` =
the script, ``sys.argv[0]`` = its path, its directory importable -- with the drop-in import paths AHEAD of it
(``install_dropin()``), so every ``from ViT_LRP import ...`` / ``from modules.layers_ours import ...`` /
``from dataset.expl_hdf5 import ...`` of the script resolves to the MI355X path, while the script's other imports
(``utils.metrices``, ``data.Imagenet``, ``misc_functions`` ...) still resolve to the reference checkout on
``PYTHONPATH``, as its README runs them (``PYTHONPATH=./:$PYTHONPATH python3 baselines/ViT/imagenet_seg_eval.py``).
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        sys.exit(__doc__)
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        sys.exit(f"run_script: {script} is not a file")
    import transformer_explainability_amd as te
    script_dir = os.path.dirname(script)
    if script_dir not in sys.path:
        sys.path.insert(0, script_dir)        # what `python script.py` would have done ...
    te.install_dropin()                       # ... and the drop-in paths ahead of it
    sys.argv = [script, *argv[1:]]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
