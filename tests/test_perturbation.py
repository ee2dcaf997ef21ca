"""the perturbation test.  The golden arrays were produced by the reference's own eval(args)
(pertubation_eval_from_hdf5.py:25-144, tests/golden/make_golden.py make_perturbation) on seeded inputs; CPU tests run our
evaluator with the device op swapped for the oracle, `-m gpu` tests run the HIP kernels."""
import numpy as np
import pytest
import torch

from oracle import relprop_oracle as O
from oracle.ref_harness import state_checksum, synthetic_init

CFG = dict(img_size=224, patch_size=16, embed_dim=64, depth=2, num_heads=4, num_classes=10, qkv_bias=True,
           block_norm_eps=1e-5, final_norm_eps=1e-5)        # ViT_new.VisionTransformer constructed directly
MODES = [("per_neg", "per", True), ("per_pos", "per", False), ("abs_neg", "100", True)]


def inputs():
    g = torch.Generator().manual_seed(7)                    # = make_golden.perturbation_inputs
    data = torch.rand((4, 3, 224, 224), generator=g)
    # a random permutation of 50,176 DISTINCT values: torch.topk leaves the order among ties unspecified, and 50k
    # randn draws do contain equal pairs
    vis = torch.stack([torch.randperm(224 * 224, generator=g) for _ in range(4)]).float().reshape(4, 1, 224, 224)
    vis = vis / (224 * 224) - 0.5
    return data, vis, torch.tensor([1, 4, 7, 2])


def model_for(golden):
    from transformer_explainability_amd import vit
    m = vit.VisionTransformer(**CFG).eval()
    synthetic_init(m, 0)
    assert abs(state_checksum(m) - golden["state_checksum"]) < 1e-6 * abs(golden["state_checksum"])
    return m


def run_and_compare(model, golden, device, tag, scale, neg):
    from transformer_explainability_amd.perturbation import PerturbationEvaluator
    data, vis, target = (t.to(device) for t in inputs())
    ev = PerturbationEvaluator(model, num_samples=4, scale=scale, neg=neg)
    for lo in (0, 2):                                       # two loader batches of 2, as the fixture
        ev.update(data[lo:lo + 2], vis[lo:lo + 2], target[lo:lo + 2])
    arrs = ev.arrays()
    assert sorted(arrs) == sorted(k.split(".", 1)[1] for k in golden if k.startswith(tag + "."))
    for name, got in arrs.items():
        ref = golden[f"{tag}.{name}"].numpy()
        assert got.shape == ref.shape and got.dtype == np.float64, name
        if "hits" in name:
            # a hit can only flip where the top-2 margin is within rounding
            dis = golden[f"{tag}.{name.replace('hits', 'dissimilarities')}"].numpy()
            assert ((got == ref) | (np.abs(dis) < 1e-4)).all(), name
        else:
            assert np.abs(got - ref).max() < 2e-5, (name, np.abs(got - ref).max())


@pytest.mark.parametrize("tag,scale,neg", MODES)
def test_evaluator_matches_reference_cpu(golden_perturbation, tag, scale, neg):
    from oracle_backend import oracle_ops
    with oracle_ops():
        run_and_compare(model_for(golden_perturbation), golden_perturbation, torch.device("cpu"), tag, scale, neg)


def test_evaluator_wrong_mode_and_save(golden_perturbation, tmp_path):
    from oracle_backend import oracle_ops
    from transformer_explainability_amd.perturbation import PerturbationEvaluator
    model = model_for(golden_perturbation)
    data, vis, target = inputs()
    with oracle_ops():
        ev = PerturbationEvaluator(model, num_samples=4, wrong=True)
        ev.update(data, vis, target)
    arrs = ev.save(str(tmp_path))
    wrong = np.flatnonzero(golden_perturbation["per_neg.model_hits.npy"].numpy() == 0)
    assert arrs["perturbations_hits.npy"].shape == (9, len(wrong))
    ref = golden_perturbation["per_neg.perturbations_logit_diff.npy"].numpy()[:, wrong]
    assert np.abs(arrs["perturbations_logit_diff.npy"] - ref).max() < 2e-5
    assert sorted(p.name for p in tmp_path.iterdir()) == sorted(arrs)
    with pytest.raises(Exception):
        PerturbationEvaluator(model, 4, scale="nope")


def test_oracle_perturb_is_the_scripts_topk_scatter():
    """tie-free input: the stable-sort restatement removes exactly the pixels torch.topk + scatter_ removes (:91-95)."""
    g = torch.Generator().manual_seed(3)
    vis, data = torch.randn((2, 64), generator=g), torch.rand((2, 3, 8, 8), generator=g)
    out = O.perturb(vis, data, [0, 5, 64])
    _, idx = torch.topk(vis, 5, dim=-1)
    ref = data.clone().reshape(2, 3, -1).scatter_(-1, idx.unsqueeze(1).repeat(1, 3, 1), 0).reshape(2, 3, 8, 8)
    assert torch.equal(out[0], data) and torch.equal(out[1], ref) and float(out[2].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------ device
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 3, 224, 224), (2, 3, 7, 9), (3, 1, 32, 32), (1, 4, 16, 12)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("ties", ["none", "upsampled", "constant"])
def test_perturb_kernel(shape, ties):
    """te_perturb_f32 vs the oracle, bit for bit: tie-free maps, bilinearly up-sampled maps (replicated border rows =
    real ties), a constant map (everything tied); k = 0, k = HW, k > HW and the 9 fractional steps."""
    from transformer_explainability_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(11)
    data = torch.rand(shape, generator=g)
    if ties == "none":
        vis = torch.randn((B, H * W), generator=g)
    elif ties == "upsampled":
        small = torch.rand((B, 1, max(H // 4, 1), max(W // 4, 1)), generator=g)
        vis = torch.nn.functional.interpolate(small, size=(H, W), mode="bilinear").reshape(B, -1)
        vis[:, : W // 2] = vis[:, :1]                       # plus an explicit run of equal values
    else:
        vis = torch.full((B, H * W), 0.25)
        vis[0, 0] = -0.0
        vis[0, 1] = 0.0
    ks = [0, 1, H * W, H * W + 5] + [int(H * W * f) for f in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9)]
    mean, std = [0.5] * C, [0.25] * C
    d = torch.device("cuda:0")
    got = ops.perturb(vis.to(d), data.to(d), ks, mean, std).cpu()
    ref = O.perturb(vis, data, ks, mean, std)
    assert got.shape == ref.shape == (len(ks), B, C, H, W)
    assert torch.equal(got, ref)
    removed = (got[:, :, 0] == (0.0 - mean[0]) / std[0]).reshape(len(ks), B, -1).sum(-1)     # data > 0 a.s.
    assert torch.equal(removed, torch.tensor([min(k, H * W) for k in ks]).unsqueeze(1).expand(-1, B))


@pytest.mark.gpu
def test_perturb_nan_and_negated():
    from transformer_explainability_amd import ops
    g = torch.Generator().manual_seed(5)
    vis = torch.randn((2, 400), generator=g)
    vis[0, 17] = float("nan")                               # torch.topk ranks NaN first
    data = torch.rand((2, 3, 20, 20), generator=g)
    d = torch.device("cuda:0")
    got = ops.perturb(vis.to(d), data.to(d), [1, 40]).cpu()
    assert float(got[0, 0, :, 0, 17].abs().max()) == 0.0
    _, idx = torch.topk(vis, 40, dim=-1)
    ref = data.clone().reshape(2, 3, -1).scatter_(-1, idx.unsqueeze(1).repeat(1, 3, 1), 0).reshape(data.shape)
    assert torch.equal(got[1], ref)
    got_neg = ops.perturb((-vis[1:]).to(d), data[1:].to(d), [40]).cpu()
    _, idx = torch.topk(-vis[1:], 40, dim=-1)
    ref = data[1:].clone().reshape(1, 3, -1).scatter_(-1, idx.unsqueeze(1).repeat(1, 3, 1), 0).reshape(1, 3, 20, 20)
    assert torch.equal(got_neg[0], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,scale,neg", MODES)
def test_evaluator_matches_reference_gpu(golden_perturbation, tag, scale, neg):
    d = torch.device("cuda:0")
    run_and_compare(model_for(golden_perturbation).to(d), golden_perturbation, d, tag, scale, neg)
