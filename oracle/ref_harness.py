"""Harness that imports and runs the UNMODIFIED reference from ``/root/reference`` on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/relprop_oracle.py header).  ``/root/reference`` exists only
in the build container, never on the GPU box; there the harness falls back to ``oracle/_ref/``, the
git-ignored byte-for-byte stage of the hot-path files made by ``scripts/stage_reference.py``
(called from ``__graft_entry__.build()`` in the build container).  This module is used (a) by
``tests/golden/make_golden.py`` to generate the committed fixtures, (b) by CPU tests that are
skipped when the checkout is absent, (c) by ``bench.py``'s optional ``cpu_baseline`` leg of kind
"reference" when the checkout is present.

Shims -- none touches relprop arithmetic:
  * ``torch.Tensor.cuda -> identity`` on a CPU-only host (generate_LRP hard-codes ``.cuda()``:
    baselines/ViT/ViT_explanation_generator.py:35; ExplanationGenerator.py:40)
  * BERT: stub ``gensim`` (BERT_rationale_benchmark/models/model_utils.py:5), transformers 3.5.1
    -> 5.x API drift (init_weights, get_extended_attention_mask, get_head_mask, return_dict).
"""
from __future__ import annotations

import hashlib
import os
import sys
import types
from contextlib import contextmanager

import torch

_STAGE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def _pick_root() -> str:
    """The reference checkout: $TE_REFERENCE_ROOT, else /root/reference (build container), else the git-ignored stage
    ``oracle/_ref/`` that scripts/stage_reference.py copies from it (what travels to the GPU box)."""
    env = os.environ.get("TE_REFERENCE_ROOT")
    if env:
        return env
    for cand in ("/root/reference", _STAGE):
        if os.path.isdir(os.path.join(cand, "modules")):
            return cand
    return "/root/reference"


REFERENCE_ROOT = _pick_root()


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modules"))


def reference_origin() -> str:
    """'checkout' (the read-only reference tree), 'stage' (oracle/_ref) or 'absent'."""
    if not reference_available():
        return "absent"
    return "stage" if os.path.abspath(REFERENCE_ROOT) == os.path.abspath(_STAGE) else "checkout"


@contextmanager
def reference_on_path():
    """Temporarily put the reference checkout first on sys.path, hiding same-named drop-in
    modules (``modules``, ``baselines``, ``BERT_explainability``) that may already be imported."""
    clash = ("modules", "baselines", "BERT_explainability", "BERT_rationale_benchmark")
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in clash}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        yield
    finally:
        sys.path.remove(REFERENCE_ROOT)
        ref_mods = {k: v for k, v in sys.modules.items() if k.split(".")[0] in clash}
        for k in ref_mods:
            del sys.modules[k]
        sys.modules.update(saved)
        _REF_CACHE.update(ref_mods)


_REF_CACHE: dict = {}


def _cuda_identity_shim():
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self


@contextmanager
def reference_on_cpu():
    """Run the reference on the HOST even where a GPU is visible (bench.py's cpu_baseline leg on the GPU box): its
    generators hard-code ``.cuda()`` (ViT_explanation_generator.py:35, ExplanationGenerator.py:40), which would move
    the one-hot to the device while the model sits on the CPU.  ``Tensor.cuda`` is the identity inside the block."""
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = saved


def load_reference_vit():
    """Returns the reference modules (ViT_LRP, ViT_orig_LRP, generator, layers_ours, layers_lrp)."""
    _cuda_identity_shim()
    with reference_on_path():
        import importlib
        mods = {}
        mods["layers_ours"] = importlib.import_module("modules.layers_ours")
        mods["layers_lrp"] = importlib.import_module("modules.layers_lrp")
        mods["ViT_LRP"] = importlib.import_module("baselines.ViT.ViT_LRP")
        mods["ViT_orig_LRP"] = importlib.import_module("baselines.ViT.ViT_orig_LRP")
        mods["gen"] = importlib.import_module("baselines.ViT.ViT_explanation_generator")
    return mods


def load_reference_perturbation_eval():
    """Import baselines/ViT/pertubation_eval_from_hdf5.py as a module (its ``eval(args)`` reads the loader, dataset,
    model and device from module globals, which the caller sets).  Stubs: ``dataset.expl_hdf5`` (needs h5py, absent
    here; only the class name is imported at module level)."""
    _cuda_identity_shim()
    import importlib.util
    stub_pkg = types.ModuleType("dataset")
    stub_mod = types.ModuleType("dataset.expl_hdf5")
    stub_mod.ImagenetResults = object
    stub_pkg.expl_hdf5 = stub_mod
    vit_dir = os.path.join(REFERENCE_ROOT, "baselines", "ViT")
    with reference_on_path():
        saved = {k: sys.modules.get(k) for k in ("dataset", "dataset.expl_hdf5", "ViT_explanation_generator", "ViT_new")}
        sys.modules["dataset"], sys.modules["dataset.expl_hdf5"] = stub_pkg, stub_mod
        for k in ("ViT_explanation_generator", "ViT_new"):
            sys.modules.pop(k, None)
        sys.path.insert(0, vit_dir)
        try:
            spec = importlib.util.spec_from_file_location("ref_pertubation_eval",
                                                          os.path.join(vit_dir, "pertubation_eval_from_hdf5.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            vnew = sys.modules["ViT_new"]
        finally:
            sys.path.remove(vit_dir)
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
    return mod, vnew


def load_reference_bert():
    """Returns reference BERT modules with the Appendix-C compat shims applied."""
    _cuda_identity_shim()
    stub = types.ModuleType("gensim")
    stub_models = types.ModuleType("gensim.models")
    stub_models.KeyedVectors = object
    stub.models = stub_models
    sys.modules.setdefault("gensim", stub)
    sys.modules.setdefault("gensim.models", stub_models)
    from transformers import BertPreTrainedModel

    if not getattr(BertPreTrainedModel, "_te_shimmed", False):
        def init_weights(self):
            if getattr(self, "_te_in_init", False):
                return
            self._te_in_init = True
            try:
                self.post_init()
            finally:
                self._te_in_init = False

        BertPreTrainedModel.init_weights = init_weights
        BertPreTrainedModel.get_extended_attention_mask = (
            lambda self, m, shape=None, *a, **k: (1.0 - m[:, None, None, :].float()) * -10000.0)
        BertPreTrainedModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
        BertPreTrainedModel._te_shimmed = True
    with reference_on_path():
        import importlib
        mods = {}
        mods["layers_ours"] = importlib.import_module("BERT_explainability.modules.layers_ours")
        mods["layers_lrp"] = importlib.import_module("BERT_explainability.modules.layers_lrp")
        mods["BERT"] = importlib.import_module("BERT_explainability.modules.BERT.BERT")
        mods["cls"] = importlib.import_module(
            "BERT_explainability.modules.BERT.BertForSequenceClassification")
        mods["gen"] = importlib.import_module("BERT_explainability.modules.BERT.ExplanationGenerator")
    return mods


# --------------------------------------------------------------------------------------------
# deterministic synthetic parameters, independent of module construction order
# --------------------------------------------------------------------------------------------
def synthetic_init(model: torch.nn.Module, seed: int = 0) -> None:
    """Fill every parameter / buffer by NAME from its own seeded generator, so that the reference
    model and the drop-in model (same state_dict keys) get bit-identical weights on any host.

      *.weight of LayerNorm (1-D, name contains 'norm'/'LayerNorm') : 1 + 0.1 N(0,1)
      other 1-D tensors (biases)                                     : 0.02 N(0,1)
      >=2-D tensors (Linear/Conv/Embedding weights, pos_embed, cls)  : 0.02 N(0,1) clipped at 2 sigma
    integer buffers (position_ids) are left untouched.
    """
    sd = model.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if not t.dtype.is_floating_point:
                continue
            h = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:8], "little")
            g = torch.Generator().manual_seed(h % (2 ** 63))
            r = torch.randn(t.shape, generator=g, dtype=torch.float32)
            lname = name.lower()
            if t.dim() == 1 and "norm" in lname and name.endswith("weight"):
                v = 1.0 + 0.1 * r
            elif t.dim() <= 1:
                v = 0.02 * r
            else:
                v = 0.02 * r.clamp(-2, 2)
            t.copy_(v.to(t.dtype))


def seeded_randn(shape, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def state_checksum(model: torch.nn.Module) -> float:
    """Order-independent fp64 checksum of all floating parameters (detects RNG drift)."""
    tot = 0.0
    for name, t in sorted(model.state_dict().items()):
        if t.dtype.is_floating_point:
            tot += float(t.double().abs().sum())
    return tot


# --------------------------------------------------------------------------------------------
# cache extraction from a reference model after forward + backward
# --------------------------------------------------------------------------------------------
def vit_cache_from_reference(model) -> dict:
    """Collect the tensors relprop reads (names as in relprop_oracle.vit_block_relprop)."""
    def d(t):
        return t.detach().clone()

    blocks = []
    for blk in model.blocks:
        blocks.append({
            "add2_x0": d(blk.add2.X[0]), "add2_x1": d(blk.add2.X[1]),
            "fc2_x": d(blk.mlp.fc2.X), "fc2_w": d(blk.mlp.fc2.weight),
            "fc1_x": d(blk.mlp.fc1.X), "fc1_w": d(blk.mlp.fc1.weight),
            "clone2_x": d(blk.clone2.X),
            "add1_x0": d(blk.add1.X[0]), "add1_x1": d(blk.add1.X[1]),
            "proj_x": d(blk.attn.proj.X), "proj_w": d(blk.attn.proj.weight),
            "attn": d(blk.attn.get_attn()), "attn_grad": d(blk.attn.get_attn_gradients()),
            "qkv_out": d(blk.attn.qkv.Y), "qkv_x": d(blk.attn.qkv.X), "qkv_w": d(blk.attn.qkv.weight),
            "clone1_x": d(blk.clone1.X),
        })
    return {"head_x": d(model.head.X), "head_w": d(model.head.weight), "pool_x": d(model.pool.X),
            "blocks": blocks}


def bert_cache_from_reference(model) -> dict:
    def d(t):
        return None if t is None else t.detach().clone()

    layers = []
    for lay in model.bert.encoder.layer:
        sa = lay.attention.self
        masked = sa.attention_mask is not None
        layers.append({
            "out_add_x0": d(lay.output.add.X[0]), "out_add_x1": d(lay.output.add.X[1]),
            "out_dense_x": d(lay.output.dense.X), "out_dense_w": d(lay.output.dense.weight),
            "inter_x": d(lay.intermediate.dense.X), "inter_w": d(lay.intermediate.dense.weight),
            "clone_x": d(lay.clone.X),
            "att_add_x0": d(lay.attention.output.add.X[0]), "att_add_x1": d(lay.attention.output.add.X[1]),
            "att_dense_x": d(lay.attention.output.dense.X), "att_dense_w": d(lay.attention.output.dense.weight),
            "probs": d(sa.get_attn()), "attn_grad": d(sa.get_attn_gradients()),
            "q": d(sa.query.Y), "k": d(sa.key.Y), "v": d(sa.value.Y),
            "mask_add_x0": d(sa.add.X[0]) if masked else None,
            "ext_mask": d(sa.add.X[1]) if masked else None,
            "q_x": d(sa.query.X), "q_w": d(sa.query.weight),
            "k_x": d(sa.key.X), "k_w": d(sa.key.weight),
            "v_x": d(sa.value.X), "v_w": d(sa.value.weight),
            "self_clone_x": d(sa.clone.X), "att_clone_x": d(lay.attention.clone.X),
        })
    return {"cls_x": d(model.classifier.X), "cls_w": d(model.classifier.weight),
            "pool_dense_x": d(model.bert.pooler.dense.X), "pool_dense_w": d(model.bert.pooler.dense.weight),
            "pool_x": d(model.bert.pooler.pool.X), "layers": layers}
