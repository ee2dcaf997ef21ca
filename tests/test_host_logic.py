"""Host-logic tests on CPU: the model / rule / generator plumbing of the package, with the device ops
swapped for the oracle (tests/oracle_backend.py).  Checks against the golden fixtures produced by the
reference, so they also pin our forward pass and state_dict layout."""
import pytest
import torch

from oracle_backend import oracle_ops


def _state(g, prefix="state."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def _rel(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    nan = torch.isnan(b)                      # 0/0 of an all-clamped map: the reference yields NaN there, so must we
    assert bool((torch.isnan(a) == nan).all())
    a, b = torch.where(nan, torch.zeros_like(a), a), torch.where(nan, torch.zeros_like(b), b)
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


@pytest.fixture(scope="module")
def tiny_vit(golden_vit_tiny):
    from transformer_explainability_amd import vit
    m = vit.VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=4, num_classes=10,
                              qkv_bias=True).eval()
    missing = m.load_state_dict(_state(golden_vit_tiny), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


def test_vit_forward_matches_reference(tiny_vit, golden_vit_tiny):
    g = golden_vit_tiny
    assert _rel(tiny_vit(g["x"]).detach(), g["ours.logits"]) < 1e-5


@pytest.mark.parametrize("start_layer", [0, 1])
def test_vit_generate_lrp_batched(tiny_vit, golden_vit_tiny, start_layer):
    from transformer_explainability_amd.generators import LRP
    g = golden_vit_tiny
    with oracle_ops():
        out = LRP(tiny_vit).generate_LRP(g["x"], method="transformer_attribution", start_layer=start_layer)
    assert out.shape == (2, 16)
    assert _rel(out.detach(), g[f"ours.map_sl{start_layer}"]) < 1e-4
    for i, blk in enumerate(tiny_vit.blocks):
        assert _rel(blk.attn.get_attn_cam()[:1], g[f"ours.attn_cam.{i}"]) < 1e-4
        assert _rel(blk.attn.get_v_cam()[:1], g[f"ours.v_cam.{i}"]) < 1e-4


def test_vit_explicit_index_and_other_methods(tiny_vit, golden_vit_tiny):
    from transformer_explainability_amd.generators import LRP
    g = golden_vit_tiny
    with oracle_ops():
        lrp = LRP(tiny_vit)
        out = lrp.generate_LRP(g["x"], index=3, start_layer=0)
        assert _rel(out.detach(), g["ours.map_sl0_idx3"]) < 1e-4
        out = lrp.generate_LRP(g["x"], method="rollout", start_layer=0)
        assert _rel(out.detach(), g["ours.rollout_sl0"]) < 1e-4
        out = lrp.generate_LRP(g["x"][:1], method="last_layer")
        assert _rel(out.detach(), g["ours.last_layer"]) < 1e-4
        assert lrp.generate_LRP(g["x"][:1], method="no_such_method") is None


def test_vit_full_and_attention_methods(tiny_vit, golden_vit_tiny, golden_methods):
    """method="full" (position-embedding Add + Conv2d z^B rule) and the attention-only branches."""
    from transformer_explainability_amd.generators import LRP
    g, gm = golden_vit_tiny, golden_methods
    with oracle_ops():
        lrp = LRP(tiny_vit)
        full = lrp.generate_LRP(g["x"], method="full")
        assert full.shape == (2, 32, 32)
        assert _rel(full.detach(), gm["ours.full"]) < 1e-4
        for method, key in (("second_layer", "ours.second_layer"), ("last_layer_attn", "ours.last_layer_attn")):
            out = lrp.generate_LRP(g["x"], method=method)
            assert _rel(out.detach().reshape(2, -1), gm[key]) < 1e-4, method
        out = lrp.generate_LRP(g["x"], method="last_layer", is_ablation=True)
        assert _rel(out.detach().reshape(2, -1), gm["ours.last_layer_ablation"]) < 1e-4
        out = lrp.generate_LRP(g["x"], method="rollout", start_layer=1)
        assert _rel(out.detach(), gm["ours.rollout_sl1"]) < 1e-4


def test_baselines_against_reference(golden_methods):
    from transformer_explainability_amd import vit
    from transformer_explainability_amd.generators import Baselines
    from oracle.ref_harness import seeded_randn
    gm = golden_methods
    m = vit.VisionTransformer(img_size=224, patch_size=16, embed_dim=64, depth=2, num_heads=4, num_classes=10,
                              qkv_bias=True, block_norm_eps=1e-5, final_norm_eps=1e-5).eval()   # ViT_new.py:113,154
    m.load_state_dict(_state(gm, "baselines.state."), strict=True)
    x = seeded_randn((2, 3, 224, 224), 2)
    assert _rel(m(x).detach(), gm["baselines.logits"]) < 1e-5
    with oracle_ops():
        b = Baselines(m)
        assert _rel(b.generate_cam_attn(x).detach(), gm["baselines.cam_attn"]) < 1e-4
        assert _rel(b.generate_cam_attn(x[:1], index=3).detach(), gm["baselines.cam_attn_idx3"]) < 1e-4
        for sl in (0, 1):
            assert _rel(b.generate_rollout(x, start_layer=sl).detach(), gm[f"baselines.rollout_sl{sl}"]) < 1e-5


@pytest.mark.parametrize("start_layer", [0, 1, 2])
def test_vit_pruned_relprop_is_bitwise_the_full_one(tiny_vit, golden_vit_tiny, start_layer):
    """LRP(prune=True): stop after block start_layer's attn_cam, attention gradients of blocks >= start_layer only."""
    from transformer_explainability_amd.generators import LRP
    g = golden_vit_tiny
    with oracle_ops():
        full = LRP(tiny_vit).generate_LRP(g["x"], start_layer=start_layer).clone()
        for blk in tiny_vit.blocks:
            blk.attn.attn_cam = None
        pruned = LRP(tiny_vit, prune=True).generate_LRP(g["x"], start_layer=start_layer)
        assert torch.equal(full, pruned)
        assert all((blk.attn.get_attn_cam() is None) == (i < start_layer) for i, blk in enumerate(tiny_vit.blocks))
        assert not any(getattr(blk.attn, "_stop_after_attn_cam", False) for blk in tiny_vit.blocks)
        # other methods are served in full whatever the flag says
        out = LRP(tiny_vit, prune=True).generate_LRP(g["x"], method="rollout", start_layer=1)
        assert all(blk.attn.get_attn_cam() is not None for blk in tiny_vit.blocks) and out.shape == (2, 16)
    tiny_vit.prune_below_start_layer = False


def test_vit_lrp_variant(golden_vit_tiny):
    from transformer_explainability_amd import rules_lrp, vit
    from transformer_explainability_amd.generators import LRP
    g = golden_vit_tiny
    ns = vit.make_vit_module(rules_lrp)
    m = ns["VisionTransformer"](img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=4, num_classes=10,
                                qkv_bias=True).eval()
    m.load_state_dict(_state(g))
    with oracle_ops():
        out = LRP(m).generate_LRP(g["x"], method="grad", start_layer=0)
    assert _rel(out.detach(), g["lrp.map_sl0"]) < 1e-4


@pytest.fixture(scope="module")
def tiny_bert(golden_bert_tiny):
    from transformer_explainability_amd import bert
    cfg = bert.BertConfigLite(vocab_size=100, hidden_size=64, num_hidden_layers=3, num_attention_heads=4,
                              intermediate_size=128, max_position_embeddings=40, num_labels=2)
    m = bert.BertForSequenceClassification(cfg).eval()
    m.load_state_dict(_state(golden_bert_tiny), strict=True)
    return m


@pytest.mark.parametrize("start_layer", [0, 2])
def test_bert_generate_lrp_batched(tiny_bert, golden_bert_tiny, start_layer):
    from transformer_explainability_amd.generators import Generator
    g = golden_bert_tiny
    ids, mask = g["input_ids"].long(), g["attention_mask"]
    assert _rel(tiny_bert(input_ids=ids, attention_mask=mask)[0].detach(), g["logits"]) < 1e-5
    with oracle_ops():
        out = Generator(tiny_bert).generate_LRP(input_ids=ids, attention_mask=mask, start_layer=start_layer)
    assert out.shape == (2, 24)
    assert _rel(out.detach(), g[f"map_sl{start_layer}"]) < 1e-4
    for i, lay in enumerate(tiny_bert.bert.encoder.layer):
        assert _rel(lay.attention.self.get_attn_cam()[:1], g[f"attn_cam.{i}"]) < 1e-4


def test_bert_other_generator_methods(tiny_bert, golden_bert_tiny, golden_methods):
    """ExplanationGenerator.py:62-155, batched (the reference explains sample 0 only)."""
    from transformer_explainability_amd.generators import Generator
    g, gm = golden_bert_tiny, golden_methods
    ids, mask = g["input_ids"].long(), g["attention_mask"]
    with oracle_ops():
        gen = Generator(tiny_bert)
        assert _rel(gen.generate_LRP_last_layer(ids, mask).detach(), gm["bert.last_layer"]) < 1e-4
        assert _rel(gen.generate_full_lrp(ids, mask).detach(), gm["bert.full_lrp"]) < 1e-4
        assert _rel(gen.generate_attn_last_layer(ids, mask).detach(), gm["bert.attn_last_layer"]) < 1e-5
        assert _rel(gen.generate_rollout(ids, mask, start_layer=0).detach(), gm["bert.rollout_sl0"]) < 1e-5
        assert _rel(gen.generate_rollout(ids, mask, start_layer=1).detach(), gm["bert.rollout_sl1"]) < 1e-5
        assert _rel(gen.generate_attn_gradcam(ids, mask).detach(), gm["bert.attn_gradcam"]) < 1e-4


@pytest.mark.parametrize("start_layer", [0, 1, 2])
def test_bert_pruned_relprop_is_bitwise_the_full_one(tiny_bert, golden_bert_tiny, start_layer):
    from transformer_explainability_amd.generators import Generator
    g = golden_bert_tiny
    ids, mask = g["input_ids"].long(), g["attention_mask"]
    with oracle_ops():
        full = Generator(tiny_bert).generate_LRP(ids, mask, start_layer=start_layer).clone()
        last = Generator(tiny_bert).generate_LRP_last_layer(ids, mask).clone()
        for lay in tiny_bert.bert.encoder.layer:
            lay.attention.self.attn_cam = None
        gen = Generator(tiny_bert, prune=True)
        assert torch.equal(gen.generate_LRP(ids, mask, start_layer=start_layer), full)
        cams = [lay.attention.self.get_attn_cam() for lay in tiny_bert.bert.encoder.layer]
        assert all((c is None) == (i < start_layer) for i, c in enumerate(cams))
        assert torch.equal(gen.generate_LRP_last_layer(ids, mask), last)
        assert torch.equal(gen.generate_full_lrp(ids, mask), Generator(tiny_bert).generate_full_lrp(ids, mask))


def test_bert_full_relprop_conservation(tiny_bert, golden_bert_tiny):
    g = golden_bert_tiny
    ids, mask = g["input_ids"][:1].long(), g["attention_mask"][:1]
    logits = tiny_bert(input_ids=ids, attention_mask=mask)[0]
    oh = torch.zeros_like(logits)
    oh[0, logits.argmax(-1)] = 1
    with oracle_ops():
        cam = tiny_bert.relprop(oh, alpha=1)
    assert _rel(cam.detach(), g["cam_tokens"]) < 1e-4
    assert abs(float(cam.double().sum()) - 1.0) < 1e-4


def test_dropin_import_paths(golden_vit_tiny):
    """The reference's import paths resolve to this implementation."""
    import importlib
    import sys
    import transformer_explainability_amd as te
    d = te.install_dropin()
    try:
        for name in ("modules.layers_ours", "modules.layers_lrp", "baselines.ViT.ViT_LRP",
                     "baselines.ViT.ViT_orig_LRP", "baselines.ViT.ViT_explanation_generator",
                     "BERT_explainability.modules.layers_ours", "BERT_explainability.modules.layers_lrp",
                     "BERT_explainability.modules.BERT.BERT", "BERT_explainability.modules.BERT.BERT_orig_lrp",
                     "BERT_explainability.modules.BERT.BertForSequenceClassification",
                     "BERT_explainability.modules.BERT.BERT_cls_lrp",
                     "BERT_explainability.modules.BERT.ExplanationGenerator"):
            mod = importlib.import_module(name)
            assert mod.__file__.startswith(d), (name, mod.__file__)
        lo = importlib.import_module("modules.layers_ours")
        for cls in ("Linear", "Add", "Clone", "einsum", "IndexSelect", "LayerNorm", "GELU", "Softmax", "Dropout",
                    "Conv2d", "safe_divide", "forward_hook"):
            assert hasattr(lo, cls)
        V = importlib.import_module("baselines.ViT.ViT_LRP")
        m = V.vit_base_patch16_224  # factory exists with the reference's name
        G = importlib.import_module("baselines.ViT.ViT_explanation_generator")
        assert hasattr(G.LRP, "generate_LRP")
        assert callable(m)
    finally:
        sys.path.remove(d)
        for k in [k for k in sys.modules if k.split(".")[0] in ("modules", "baselines", "BERT_explainability")]:
            del sys.modules[k]


def test_split_qkv_matches_the_permuted_views_and_their_gradient():
    """vit.split_qkv (one autograd node, no zero-fill / accumulate in backward) == the reference's permuted views
    (ViT_LRP.py:135-136), values and gradients, also when only some of q / k / v receive a gradient."""
    from transformer_explainability_amd.vit import split_qkv
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 5, 3 * 4 * 8), generator=g, requires_grad=True)
    w = torch.randn((2, 4, 5, 8), generator=g)
    q, k, v = split_qkv(x, 4)
    ref = x.view(2, 5, 3, 4, 8).permute(2, 0, 3, 1, 4)
    assert all(t.is_contiguous() for t in (q, k, v))
    assert torch.equal(q, ref[0]) and torch.equal(k, ref[1]) and torch.equal(v, ref[2])
    (ga,) = torch.autograd.grad((q * w).sum() + (k * 2).sum() + (v * w * 3).sum(), x)
    (gb,) = torch.autograd.grad((ref[0] * w).sum() + (ref[1] * 2).sum() + (ref[2] * w * 3).sum(), x)
    assert torch.equal(ga, gb)
    (gk,) = torch.autograd.grad((split_qkv(x, 4)[1] * w).sum(), x)
    (gr,) = torch.autograd.grad((ref[1] * w).sum(), x)
    assert torch.equal(gk, gr)


def test_cpu_only_hosts_get_no_silent_fallback():
    """No GPU here: the tuned-GEMM switch reports False, and the device ops refuse CPU tensors loudly."""
    import transformer_explainability_amd as te
    from transformer_explainability_amd import ops
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    assert te.enable_tuned_gemms() is False
    with pytest.raises(te.TeError):
        ops.clone_relprop((torch.zeros(1, 2, 4), torch.zeros(1, 2, 4)), torch.ones(1, 2, 4))


def test_producer_switch_is_inert_off_the_gpu(tiny_vit, golden_vit_tiny):
    """ops.USE_FUSED_PRODUCERS only selects kernels for fp32 device tensors: on host tensors the LayerNorm / GELU /
    attention modules must stay on their stock forward (same logits, bit for bit) instead of reaching for the
    library, and the producer autograd nodes must not appear in the graph."""
    from transformer_explainability_amd import ops, producers
    g = golden_vit_tiny
    stock = tiny_vit(g["x"]).detach().clone()
    ops.USE_FUSED_PRODUCERS = True
    try:
        x = g["x"].clone().requires_grad_(True)
        out = tiny_vit(x)
        assert torch.equal(out.detach(), stock)
        assert not producers.usable(x) and not producers.gelu_usable(x)
        assert not producers.norm_usable(torch.zeros(2, 5, 64), tiny_vit.blocks[0].norm1)
        assert all(getattr(b.attn, "_fused_anchor", None) is None for b in tiny_vit.blocks)
        (gx,) = torch.autograd.grad(out.sum(), x)             # the stock graph differentiates as usual
        assert torch.isfinite(gx).all()
    finally:
        ops.USE_FUSED_PRODUCERS = False


def test_x6_gemm_direction_policy():
    """ops.gemm_x6_wanted / producers.linear_plan: which products of a Linear layer go to te_gemm_x6_f32 (host logic only:
    the C ABI answers the shape question without a GPU).  all = every supported product of >= 256 rows; auto = narrow
    operand, wide output; off = none; host tensors and training-mode layers are never planned."""
    from transformer_explainability_amd import ops, producers, rules
    was = (ops.USE_FUSED_PRODUCERS, ops.X6_GEMM)
    try:
        ops.X6_GEMM = "all"
        assert ops.gemm_x6_wanted(12608, 768, 2304) and ops.gemm_x6_wanted(12608, 3072, 768)
        assert not ops.gemm_x6_wanted(197, 768, 2304)            # below 256 rows: the stock GEMM
        assert not ops.gemm_x6_wanted(12608, 768, 1000)          # the head: 1000 classes are not a multiple of 128
        ops.X6_GEMM = "auto"
        assert ops.gemm_x6_wanted(12608, 768, 2304) and ops.gemm_x6_wanted(12608, 768, 3072)
        assert not ops.gemm_x6_wanted(12608, 768, 768) and not ops.gemm_x6_wanted(12608, 3072, 768)
        ops.X6_GEMM = "off"
        assert not ops.gemm_x6_wanted(12608, 768, 2304)
        ops.USE_FUSED_PRODUCERS, ops.X6_GEMM = True, "all"
        lin = rules.Linear(768, 2304).eval()
        assert producers.linear_plan(torch.zeros(2, 197, 768), lin) == (False, False)        # host tensor
        y = lin(torch.zeros(2, 197, 768))                                                   # ... and the stock forward runs
        assert y.shape == (2, 197, 2304) and "x_abs_planes" not in rules.x6_cache(lin)
    finally:
        ops.USE_FUSED_PRODUCERS, ops.X6_GEMM = was


def test_bench_refuses_fewer_gpus_than_ranks():
    """`python bench.py --gpus 2` on a host with fewer than 2 GPUs must fail loudly -- never print an n_gpus: 1 line."""
    import os
    import subprocess
    import sys
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("needs a host with < 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TE_DEVICE_OVERRIDE",
                                                            "TE_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing to run 2 ranks" in r.stderr and '"n_gpus"' not in r.stdout
    # under a launcher whose world size disagrees with --gpus: also an error, not a silent 1-rank run
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr and '"n_gpus"' not in r.stdout


def test_reference_stage_matches_checkout():
    """oracle/_ref (what travels to the GPU box for the cpu_baseline leg) is a byte-for-byte copy of the reference's
    hot-path files: MANIFEST checksums hold, and against /root/reference when it is present."""
    import hashlib
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_stage", os.path.join(root, "scripts", "stage_reference.py"))
    st = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(st)
    if not os.path.exists(os.path.join(st.DEST, "MANIFEST.json")):
        if not st.stage(quiet=True):
            pytest.skip("no reference checkout and no stage on this host")
    assert st.check()
    if os.path.isdir("/root/reference/modules"):
        with open(os.path.join(st.DEST, "MANIFEST.json")) as f:
            man = json.load(f)["sha256"]
        for rel, h in man.items():
            with open(os.path.join("/root/reference", rel), "rb") as f:
                assert hashlib.sha256(f.read()).hexdigest() == h, rel


def test_index_select_host_copy_follows_in_place_edits():
    """the host copy of a device / tensor index is keyed on identity AND version counter, so
    ``idx.fill_(k)`` between forwards is seen (the cached value used to go stale silently)."""
    import torch
    from transformer_explainability_amd import rules
    m = rules.IndexSelect()
    x = torch.randn(2, 5, 4)
    idx = torch.tensor([0])
    m(x, 1, idx)
    assert m._index_host == 0
    idx.fill_(3)
    y = m(x, 1, idx)
    assert m._index_host == 3 and torch.equal(y, x[:, 3:4])


def test_x6_plane_cache_is_scratch_not_state():
    """the bf16 operand planes a Linear layer caches (rules.x6_cache) outlive a forward pass.  They are keyed on
    the weight as the caller holds it (address, strides, version counter): an in-place edit autograd records changes the
    key; one it cannot see (``.data``) needs ops.x6_invalidate; load_state_dict and .to() / .float() drop them; they are
    never pickled or deep-copied (torch.save(model) must not serialise device scratch)."""
    import copy
    import pickle
    from transformer_explainability_amd import ops, rules
    lin = rules.Linear(8, 4)
    k0 = ops._weight_key(lin.weight.detach())
    assert ops._weight_key(lin.weight.detach()) == k0                       # detach() shares storage AND version counter
    with torch.no_grad():
        lin.weight.mul_(2.0)
    k1 = ops._weight_key(lin.weight.detach())
    assert k1 != k0                                                         # recorded in-place edit -> new key
    lin.weight.data.mul_(0.5)                                               # NOT recorded: same key (documented) ...
    assert ops._weight_key(lin.weight.detach()) == k1
    rules.x6_cache(lin)["x6_planes"] = (k1, torch.zeros(3))
    assert ops.x6_invalidate(lin) == 1 and not rules.x6_cache(lin)          # ... hence the explicit invalidation
    # a non-contiguous weight is keyed as the caller holds it (its contiguous copy would have version 0 and a recycled address)
    wt = torch.randn(8, 4).t()
    assert ops._weight_key(wt) != ops._weight_key(wt.contiguous())
    # whole-model invalidation, load_state_dict, dtype / device moves
    model = torch.nn.Sequential(rules.Linear(8, 8), rules.Linear(8, 4))
    for m in model:
        rules.x6_cache(m)["x6_planes"] = ("k", torch.zeros(1))
    assert ops.x6_invalidate(model) == 2
    for m in model:
        rules.x6_cache(m)["x6_planes"] = ("k", torch.zeros(1))
    model.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    assert all(not rules.x6_cache(m) for m in model)
    rules.x6_cache(model[0])["x6_planes"] = ("k", torch.zeros(1))
    model.double()
    assert not rules.x6_cache(model[0])
    # pickling / deepcopy leave the scratch behind
    rules.x6_cache(lin)["x6_planes"] = (k1, torch.zeros(1000))
    lin2 = copy.deepcopy(lin)
    assert "_te_cache" not in lin2.__dict__ and torch.equal(lin2.weight, lin.weight)
    lin3 = pickle.loads(pickle.dumps(lin))
    assert "_te_cache" not in lin3.__dict__
    assert "x6_planes" in rules.x6_cache(lin)                               # the original keeps its planes


def test_gelu_consumer_hint_is_not_state():
    """Round 5: vit.Mlp / bert.BertLayer tell their GELU which Linear layer its output feeds (rules.GELU.feeds), so that on
    the GPU the activation can emit that layer's operand planes itself (producers._Gelu).  The hint is not a registered
    submodule (state_dict and parameter lists are untouched), deepcopy follows it to the COPY's layer, a bare GELU has none,
    and on the CPU path (no HIP producers) the module is plain nn.GELU."""
    import copy
    from transformer_explainability_amd import bert, rules, vit
    mlp = vit.Mlp(16, 64)
    assert mlp.act.__dict__["_te_feeds"][0] is mlp.fc2
    assert list(mlp.state_dict().keys()) == ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    assert [n for n, _ in mlp.act.named_modules()] == [""] and not list(mlp.act.parameters())
    twin = copy.deepcopy(mlp)
    assert twin.act.__dict__["_te_feeds"][0] is twin.fc2 and twin.fc2 is not mlp.fc2
    x = torch.randn(2, 5, 16)
    assert torch.equal(mlp.eval()(x), mlp.fc2(torch.nn.functional.gelu(mlp.fc1(x))))
    assert "_te_feeds" not in rules.GELU().__dict__
    model = bert.BertForSequenceClassification(bert.BertConfigLite(hidden_size=32, num_attention_heads=2, intermediate_size=64,
                                                                   num_hidden_layers=1, vocab_size=50, num_labels=2))
    layer = model.bert.encoder.layer[0]
    assert layer.intermediate.intermediate_act_fn.__dict__["_te_feeds"][0] is layer.output.dense
    assert not any("act_fn" in k for k in model.state_dict())
