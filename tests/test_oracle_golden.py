"""Pin the CPU oracle (oracle/relprop_oracle.py) against outputs of the reference itself.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py running the
unmodified reference on CPU (the reference has no tests of its own).  The
closed forms reproduce the reference bit-for-bit on the host that generated the fixtures; on
another host the CPU GEMM blocking may differ, so GEMM-carrying rules get a 1e-6 relative guard
band while purely element-wise rules must match exactly.
"""
import pytest
import torch

from oracle import relprop_oracle as O
from conftest import unflatten_cache


def _close(a, b, rel=1e-6):
    scale = max(float(b.abs().max()), 1e-30)
    assert float((a - b).abs().max()) <= rel * scale, (float((a - b).abs().max()), scale)


@pytest.mark.parametrize("variant", ["ours", "lrp"])
@pytest.mark.parametrize("alpha", [1, 2])
def test_linear(golden_rules, variant, alpha):
    g = golden_rules
    out = O.linear_relprop(g[f"linear_{variant}.R"], g[f"linear_{variant}.X"], g[f"linear_{variant}.W"],
                           alpha=alpha, variant=variant)
    _close(out, g[f"linear_{variant}.out_a{alpha}"])


def test_einsum_av_qk(golden_rules):
    g = golden_rules
    o0, o1 = O.einsum_av_relprop(g["av.R"], g["av.attn"], g["av.v"])
    _close(o0, g["av.out0"]); _close(o1, g["av.out1"])
    o0, o1 = O.einsum_qk_relprop(g["qk.R"], g["qk.q"], g["qk.k"])
    _close(o0, g["qk.out0"]); _close(o1, g["qk.out1"])
    o0, o1 = O.matmul_relprop(g["qk.R"], g["qk.q"], g["qk.k"].transpose(-1, -2))
    _close(o0, g["matmul_qkT.out0"]); _close(o1, g["matmul_qkT.out1"])


@pytest.mark.parametrize("variant", ["ours", "lrp"])
def test_add(golden_rules, variant):
    g = golden_rules
    a, b = O.add_relprop(g[f"add_{variant}.R"], g[f"add_{variant}.X0"], g[f"add_{variant}.X1"], variant)
    assert torch.equal(a, g[f"add_{variant}.out0"]) and torch.equal(b, g[f"add_{variant}.out1"])


def test_add_mask(golden_rules):
    g = golden_rules
    a, b = O.add_relprop(g["add_mask.R"], g["add_mask.X0"], g["add_mask.X1"], "ours")
    _close(a, g["add_mask.out0"], 1e-6); _close(b, g["add_mask.out1"], 1e-6)


@pytest.mark.parametrize("num", [2, 3])
def test_clone(golden_rules, num):
    g = golden_rules
    out = O.clone_relprop([g[f"clone{num}.R{i}"] for i in range(num)], g[f"clone{num}.X"])
    assert torch.equal(out, g[f"clone{num}.out"])


def test_index_select(golden_rules):
    g = golden_rules
    out = O.index_select_relprop(g["index_select.R"], g["index_select.X"], 1, 0)
    assert torch.equal(out, g["index_select.out"])


@pytest.mark.parametrize("variant", ["ours", "lrp"])
def test_vit_tiny_end_to_end(golden_vit_tiny, variant):
    g = golden_vit_tiny
    cache = unflatten_cache(g, f"{variant}.cache.")
    logits = g[f"{variant}.logits"][:1]
    oh = torch.zeros_like(logits)
    oh[0, logits.argmax(-1)] = 1
    res = O.vit_relprop(oh, cache, num_heads=4, start_layer=0, variant=variant)
    for i in range(3):
        _close(res["attn_cams"][i], g[f"{variant}.attn_cam.{i}"], 1e-5)
    _close(res["cam"], g[f"{variant}.cam_tokens"], 1e-5)
    _close(res["map"], g[f"{variant}.map_sl0"][:1], 1e-5)
    res1 = O.vit_relprop(oh, cache, num_heads=4, start_layer=1, variant=variant)
    _close(res1["map"], g[f"{variant}.map_sl1"][:1], 1e-5)


def test_bert_tiny_end_to_end(golden_bert_tiny):
    g = golden_bert_tiny
    cache = unflatten_cache(g, "cache.")
    logits = g["logits"][:1]
    oh = torch.zeros_like(logits)
    oh[0, logits.argmax(-1)] = 1
    res = O.bert_relprop(oh, cache, num_heads=4, start_layer=0)
    for i in range(3):
        _close(res["attn_cams"][i], g[f"attn_cam.{i}"], 1e-5)
    _close(res["cam"], g["cam_tokens"], 1e-5)
    _close(res["map"], g["map_sl0"][:1], 1e-5)
    assert abs(float(res["cam"].double().sum()) - 1.0) < 1e-5       # LRP conservation
    res2 = O.bert_relprop(oh, cache, num_heads=4, start_layer=2)
    _close(res2["map"], g["map_sl2"][:1], 1e-5)


def test_safe_divide_branches():
    a = torch.tensor([1.0, 1.0, 1.0, 1.0])
    b = torch.tensor([0.0, -1e-9, 2.0, -2.0])
    out = O.safe_divide(a, b)
    assert out[0] == 0.0                       # b == 0 -> 0
    assert out[1] == 1.0 / 1e-9 or torch.isfinite(out[1])   # den == 0 -> 1e-9 replacement
    assert out[2] == 1.0 / (2.0 + 1e-9) or abs(float(out[2]) - 0.5) < 1e-6
    assert abs(float(out[3]) + 0.5) < 1e-6


# ---- Conv2d z^B rule and method="full"
@pytest.mark.parametrize("variant", ["ours", "lrp"])
def test_conv2d_zb(golden_methods, variant):
    g = golden_methods
    out = O.conv2d_zb_relprop(g[f"conv_{variant}.R"], g[f"conv_{variant}.X"], g[f"conv_{variant}.W"], 4)
    _close(out, g[f"conv_{variant}.out"])


def test_vit_tiny_full_tail(golden_vit_tiny, golden_methods):
    """oracle block stack (pinned above) -> position-embedding Add -> z^B rule -> channel sum == the reference's
    method="full" map of sample 0."""
    g = golden_vit_tiny
    cache = unflatten_cache(g, "ours.cache.")
    state = {k[len("state."):]: v for k, v in g.items() if k.startswith("state.")}
    x = g["x"][:1]
    p = state["patch_embed.proj.weight"].shape[-1]
    tokens = torch.nn.functional.conv2d(x, state["patch_embed.proj.weight"], state["patch_embed.proj.bias"], stride=p)
    tokens = torch.cat([state["cls_token"], tokens.flatten(2).transpose(1, 2)], dim=1)
    cache.update(pos_add_x0=tokens, pos_embed=state["pos_embed"], patch_x=x, patch_w=state["patch_embed.proj.weight"])
    full = O.vit_full_tail(g["ours.cam_tokens"], cache)
    _close(full, golden_methods["ours.full"][:1], rel=1e-5)


# ---- live reference (the checkout, or its stage oracle/_ref) vs the oracle on FRESH seeds --------------------------
def _live_reference():
    from oracle import ref_harness as rh
    if not rh.reference_available():
        pytest.skip("no reference checkout / stage on this host")
    return rh


@pytest.mark.parametrize("seed", [101, 202])
@pytest.mark.parametrize("variant", ["ours", "lrp"])
def test_live_reference_vit_vs_oracle(seed, variant):
    """Runs the unmodified reference NOW (not a replayed fixture) on a freshly seeded small ViT and input, extracts the
    tensors its relprop reads, and requires the oracle to reproduce every block's attn_cam and the final map."""
    rh = _live_reference()
    vit = rh.load_reference_vit()
    modname, method = (("ViT_LRP", "transformer_attribution") if variant == "ours" else ("ViT_orig_LRP", "grad"))
    model = vit[modname].VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=4, num_classes=10,
                                           qkv_bias=True).eval()
    rh.synthetic_init(model, seed)
    x = rh.seeded_randn((1, 3, 32, 32), seed + 1)
    gen = vit["gen"].LRP(model)
    for sl in (0, 1):
        ref_map = gen.generate_LRP(x, method=method, start_layer=sl).detach().clone()
        cache = rh.vit_cache_from_reference(model)
        logits = model(x)
        oh = torch.zeros_like(logits)
        oh[0, logits.argmax(-1)] = 1
        res = O.vit_relprop(oh, cache, num_heads=4, start_layer=sl, variant=variant)
        _close(res["map"], ref_map, 1e-5)
        for i, blk in enumerate(model.blocks):
            _close(res["attn_cams"][i], blk.attn.get_attn_cam().detach(), 1e-5)


@pytest.mark.parametrize("seed", [303])
def test_live_reference_bert_vs_oracle(seed):
    rh = _live_reference()
    bert = rh.load_reference_bert()
    from transformers import BertConfig
    cfg = BertConfig(vocab_size=100, hidden_size=64, num_hidden_layers=3, num_attention_heads=4, intermediate_size=128,
                     max_position_embeddings=40, num_labels=2)
    cfg.return_dict = False
    model = bert["cls"].BertForSequenceClassification(cfg).eval()
    rh.synthetic_init(model, seed)
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, 100, (1, 24), generator=g)
    mask = torch.ones(1, 24)
    mask[:, 18:] = 0
    gen = bert["gen"].Generator(model)
    for sl in (0, 2):
        ref_map = gen.generate_LRP(input_ids=ids, attention_mask=mask, start_layer=sl).detach().clone()
        cache = rh.bert_cache_from_reference(model)
        logits = model(input_ids=ids, attention_mask=mask)[0]
        oh = torch.zeros_like(logits)
        oh[0, logits.argmax(-1)] = 1
        res = O.bert_relprop(oh, cache, num_heads=4, start_layer=sl)
        _close(res["map"], ref_map, 1e-5)
