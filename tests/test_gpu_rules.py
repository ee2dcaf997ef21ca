"""`-m gpu`: every relprop rule, HIP kernels through the C ABI vs the CPU oracle on the same seeded
inputs, vs the golden fixtures produced by the reference, for the tiled (MFMA) AND the simple device
kernels.  Tolerances: element-wise rules are evaluated op for op like the reference (exact or 1 ulp-
level); GEMM-carrying rules differ only by fp32 summation order (relative 2e-5 of the tensor max)."""
import pytest
import torch

from gpu_util import check, check_conditioned, dev, record, rnd
from oracle import relprop_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _impl(request):
    from transformer_explainability_amd import ops
    simple = getattr(request, "param", None)
    yield
    ops.FORCE_SIMPLE = False


def set_impl(simple):
    from transformer_explainability_amd import ops
    ops.FORCE_SIMPLE = bool(simple)


def test_single_hip_runtime():
    from transformer_explainability_amd import _lib
    _lib.require_device()
    rts = _lib.hip_runtimes_loaded()
    record("hip_runtimes", runtimes=rts)
    assert len(rts) == 1, rts


# ------------------------------------------------------------------------------------------ clone
@pytest.mark.parametrize("num", [2, 3])
@pytest.mark.parametrize("shape", [(2, 5, 12), (3, 197, 768), (1, 7, 3)])
def test_clone(num, shape):
    from transformer_explainability_amd import ops
    X = rnd(shape, 1)
    X.view(-1)[0] = 0.0
    Rs = [rnd(shape, 10 + i, 0.01) for i in range(num)]
    got = ops.clone_relprop([r.to(dev()) for r in Rs], X.to(dev()))
    check(f"clone{num}{shape}", got, O.clone_relprop(Rs, X), 1e-6)


# ------------------------------------------------------------------------------------------ add
@pytest.mark.parametrize("variant", ["ours", "lrp"])
@pytest.mark.parametrize("shape", [(1, 9, 16), (3, 197, 768), (2, 5, 3), (64, 33, 64)])
def test_add(variant, shape):
    from transformer_explainability_amd import ops
    X0, X1, R = rnd(shape, 2), rnd(shape, 3), rnd(shape, 4, 0.01)
    X0.view(-1)[0] = 0.0
    X1.view(-1)[0] = 0.0
    a, b = ops.add_relprop(R.to(dev()), X0.to(dev()), X1.to(dev()), variant=variant)
    ra, rb = O.add_relprop(R, X0, X1, variant)
    # the per-sample sums are accumulated in fp64 here and pairwise in fp32 by torch: ~1e-6 relative
    check(f"add_{variant}{shape}.a", a, ra, 2e-5)
    check(f"add_{variant}{shape}.b", b, rb, 2e-5)


def test_add_batch_independent_and_shared_x1():
    from transformer_explainability_amd import ops
    shape = (4, 50, 64)
    X0, X1, R = rnd(shape, 5).to(dev()), rnd(shape, 6).to(dev()), rnd(shape, 7, 0.01).to(dev())
    a, b = ops.add_relprop(R, X0, X1)
    for i in range(4):
        ai, bi = ops.add_relprop(R[i:i + 1], X0[i:i + 1], X1[i:i + 1])
        assert torch.equal(ai, a[i:i + 1]) and torch.equal(bi, b[i:i + 1])
    # shared (batch-less) second operand, e.g. pos_embed
    a2, b2 = ops.add_relprop(R, X0, X1[:1])
    ra, rb = O.add_relprop(R.cpu(), X0.cpu(), X1[:1].cpu())
    check("add_shared_x1.a", a2, ra, 2e-5)
    check("add_shared_x1.b", b2, rb, 2e-5)


@pytest.mark.parametrize("B,H,N", [(1, 3, 7), (2, 12, 128), (3, 4, 300)])
def test_add_bcast_mask(B, H, N):
    from transformer_explainability_amd import ops
    X0 = rnd((B, H, N, N), 8)
    R = rnd((B, H, N, N), 9, 0.01)
    mask = torch.zeros(B, 1, 1, N)
    mask[..., N - max(1, N // 5):] = -10000.0
    R[..., N - max(1, N // 5):] = 0.0          # relevance of masked keys is zero in practice (probs = 0)
    a, b = ops.add_relprop(R.to(dev()), X0.to(dev()), mask.to(dev()))
    ra, rb = O.add_relprop(R, X0, mask)
    check(f"add_bcast({B},{H},{N}).a", a, ra, 2e-5)
    assert b.shape == (B, 1, 1, N)


def test_add_bcast_nonzero_mask_relevance():
    from transformer_explainability_amd import ops
    B, H, N = 2, 3, 9
    X0, R = rnd((B, H, N, N), 18), rnd((B, H, N, N), 19, 0.01)
    mask = rnd((B, 1, 1, N), 20)
    a, b = ops.add_relprop(R.to(dev()), X0.to(dev()), mask.to(dev()))
    ra, rb = O.add_relprop(R, X0, mask)
    check("add_bcast_dense.a", a, ra, 5e-5)
    check("add_bcast_dense.b", b, rb, 5e-5)


# ------------------------------------------------------------------------------------------ index select
def test_index_select():
    from transformer_explainability_amd import ops
    X, R = rnd((3, 197, 768), 11), rnd((3, 1, 768), 12, 0.01)
    got = ops.index_select_relprop(R.to(dev()), X.to(dev()), 0)
    check("index_select", got, O.index_select_relprop(R, X, 1, 0), 1e-6)


# ------------------------------------------------------------------------------------------ head mean
@pytest.mark.parametrize("B,H,N", [(2, 12, 197), (1, 4, 17), (2, 3, 64)])
def test_headmean(B, H, N):
    from transformer_explainability_amd import ops
    g, c = rnd((B, H, N, N), 13), rnd((B, H, N, N), 14, 0.01)
    got = ops.gradcam_headmean(g.to(dev()), c.to(dev()))
    check(f"headmean({B},{H},{N})", got, O.gradcam_headmean(g, c), 1e-6)


# ------------------------------------------------------------------------------------------ linear
LINEAR_SHAPES = [(10, 24, 40), (300, 768, 2304), (197, 3072, 768), (64, 768, 1000), (7, 10, 2), (130, 64, 192),
                 (257, 1024, 4096)]


@pytest.mark.parametrize("simple", [False, True], ids=["tiled", "simple"])
@pytest.mark.parametrize("variant", ["ours", "lrp"])
@pytest.mark.parametrize("T,in_f,out_f", LINEAR_SHAPES)
def test_linear(simple, variant, T, in_f, out_f):
    from transformer_explainability_amd import ops
    set_impl(simple)
    X, W, R = rnd((T, in_f), 21), rnd((out_f, in_f), 22, 0.05), rnd((T, out_f), 23, 0.01)
    X[0, :3] = 0.0
    got = ops.linear_relprop(R.to(dev()), X.to(dev()), W.to(dev()), alpha=1.0, variant=variant)
    ref = O.linear_relprop(R, X, W, 1.0, variant)
    check(f"linear_{variant}_{'simple' if simple else 'tiled'}({T},{in_f},{out_f})", got, ref, 2e-5)


@pytest.mark.parametrize("T,in_f,out_f", LINEAR_SHAPES + [(788, 768, 768), (5, 3072, 768)])
def test_linear_from_forward_output(T, in_f, out_f):
    """Z-pass fed by the forward output Y = X W^T + b (te_linear_relprop_fwd_f32): Z = ((Y - b) + |X||W|^T) / 2 must
    reproduce the oracle's X+W+^T + X-W-^T to GEMM-rounding accuracy; Y comes from rocBLAS (F.linear on the GPU)."""
    from transformer_explainability_amd import ops
    if in_f % 4 or out_f % 4:
        pytest.skip("tiled kernels need in_f, out_f multiples of 4")
    X, W, R = rnd((T, in_f), 21), rnd((out_f, in_f), 22, 0.05), rnd((T, out_f), 23, 0.01)
    bias = rnd((out_f,), 24, 0.3)
    X[0, :3] = 0.0
    Xd, Wd, bd = X.to(dev()), W.to(dev()), bias.to(dev())
    Y = torch.nn.functional.linear(Xd, Wd, bd)
    got = ops.linear_relprop(R.to(dev()), Xd, Wd, alpha=1.0, variant="ours", Y=Y, bias=bd)
    ref = O.linear_relprop(R, X, W, 1.0, "ours")
    check(f"linear_fwd({T},{in_f},{out_f})", got, ref, 2e-5)
    two = ops.linear_relprop(R.to(dev()), Xd, Wd, alpha=1.0, variant="ours")
    check(f"linear_fwd_vs_two_gemm({T},{in_f},{out_f})", got, two, 2e-5)
    got_nb = ops.linear_relprop(R.to(dev()), Xd, Wd, alpha=1.0, variant="ours", Y=torch.nn.functional.linear(Xd, Wd))
    check(f"linear_fwd_nobias({T},{in_f},{out_f})", got_nb, ref, 2e-5)


# every Linear shape of the three bench configurations (ViT-B/16: 768 -> 2304 / 768 / 3072, 3072 -> 768; ViT-L/16:
# 1024 -> 3072 / 1024 / 4096, 4096 -> 1024; BERT-base: 768 -> 768 / 3072, 3072 -> 768) + two small odd ones
@pytest.mark.parametrize("T,in_f,out_f", [(394, 768, 3072), (140, 256, 384), (130, 3072, 768), (257, 128, 128),
                                          (1576, 768, 768), (600, 1024, 1024), (394, 768, 2304), (300, 1024, 3072),
                                          (300, 1024, 4096), (300, 4096, 1024)])
def test_linear_x6_split_operand_path(T, in_f, out_f):
    """DEFAULT path since round 3 (csrc/te_linear_x6.hip): the rule's three products on bf16 MFMAs with every fp32
    operand split into three bf16 parts (six partial products kept), persistent sequential stream-K schedule.  Must agree
    with the oracle as closely as the fp32-MFMA path does (same 2e-5 bar), with that path itself, be AT LEAST as
    accurate against fp64 ("error-vs-fp64 <= the fp32-MFMA path on every Linear shape"), honour the deferred
    per-sample factor, leave no row dependent on its tile neighbours or on where the stream-K cuts fall (bitwise; both
    tile geometries), and fall back to the exact positive-part sum where (Y - b) and |X||W|^T cancel."""
    from transformer_explainability_amd import ops
    X, W, R = rnd((T, in_f), 121), rnd((out_f, in_f), 122, 0.05), rnd((T, out_f), 123, 0.01)
    bias = rnd((out_f,), 124, 0.3)
    X[0, :3] = 0.0
    X[1] = 0.0                                                   # an all-zero row: Z = 0 -> S = 0
    X[2] = X[2].abs() + 0.01
    W[:5] = -W[:5].abs() - 0.001                                 # row 2 against these: every product negative
    Xd, Wd, bd, Rd = X.to(dev()), W.to(dev()), bias.to(dev()), R.to(dev())
    Y = torch.nn.functional.linear(Xd, Wd, bd)
    was = ops.USE_LINEAR_X6
    ops.USE_LINEAR_X6 = False
    fp32 = ops.linear_relprop(Rd, Xd, Wd, alpha=1.0, variant="ours", Y=Y, bias=bd)
    ops.USE_LINEAR_X6 = True
    ops.X6_CHECK = True
    try:
        assert bool(ops._lib.load().te_linear_relprop_x6_supported(T, in_f, out_f))
        cache = {}
        got = ops.linear_relprop(Rd, Xd, Wd, alpha=1.0, variant="ours", Y=Y, bias=bd, cache=cache)
        assert "x6_planes" in cache                              # the x6 kernels ran, and the weight planes are kept
        ref = O.linear_relprop(R, X, W, 1.0, "ours")
        assert torch.isfinite(got).all()
        check(f"linear_x6({T},{in_f},{out_f})", got, ref, 2e-5)
        check(f"linear_x6_vs_fp32_mfma({T},{in_f},{out_f})", got, fp32, 2e-5)
        # fp64: no less accurate than the fp32-MFMA path
        ref64 = O.linear_relprop(R.double(), X.double(), W.double(), 1.0, "ours")
        d6, d32 = got.cpu().double() - ref64, fp32.cpu().double() - ref64
        e6, e32 = float(d6.abs().max()), float(d32.abs().max())
        r6, r32 = float(d6.pow(2).mean().sqrt()), float(d32.pow(2).mean().sqrt())
        record(f"linear_x6_fp64({T},{in_f},{out_f})", x6_max=e6, fp32_mfma_max=e32, x6_rms=r6, fp32_mfma_rms=r32)
        # the rms error is the stable statistic (the maximum of ~1e6 errors of either path fluctuates by +-50 %)
        assert r6 <= 1.1 * r32 + 1e-10 * float(ref64.abs().max()), (r6, r32)
        assert e6 <= 2.0 * e32 + 1e-9 * float(ref64.abs().max()), (e6, e32)
        # second call: planes from the cache, same bits; the other tile geometry: same bits
        assert torch.equal(ops.linear_relprop(Rd, Xd, Wd, alpha=1.0, variant="ours", Y=Y, bias=bd, cache=cache), got)
        ops.X6_TILE = 1
        assert torch.equal(ops.linear_relprop(Rd, Xd, Wd, alpha=1.0, variant="ours", Y=Y, bias=bd, cache=cache), got)
        ops.X6_TILE = 0
        # a row's result does not depend on its tile neighbours; a sample-wise factor on R == the scaled R
        half = ops.linear_relprop(Rd[:T // 2].contiguous(), Xd[:T // 2].contiguous(), Wd, alpha=1.0, variant="ours",
                                  Y=Y[:T // 2].contiguous(), bias=bd, cache=cache)
        assert torch.equal(half, got[:T // 2])
        mid = ops.linear_relprop(Rd[33:97].contiguous(), Xd[33:97].contiguous(), Wd, alpha=1.0, variant="ours",
                                 Y=Y[33:97].contiguous(), bias=bd, cache=cache)
        assert torch.equal(mid, got[33:97])
        if T % 2 == 0:
            fac = torch.tensor([0.75, 1.5], device=dev())
            scaled = ops.linear_relprop(ops.Deferred(Rd.view(2, T // 2, out_f), fac), Xd.view(2, T // 2, in_f), Wd,
                                        alpha=1.0, variant="ours", Y=Y.view(2, T // 2, out_f), bias=bd, cache=cache)
            plain = ops.linear_relprop((Rd.view(2, T // 2, out_f) * fac[:, None, None]).contiguous(),
                                       Xd.view(2, T // 2, in_f), Wd, alpha=1.0, variant="ours",
                                       Y=Y.view(2, T // 2, out_f), bias=bd, cache=cache)
            assert torch.equal(scaled, plain)
        # an in-place edit of the weight invalidates the cached planes
        Wd.mul_(1.5)
        Y2 = torch.nn.functional.linear(Xd, Wd, bd)
        got2 = ops.linear_relprop(Rd, Xd, Wd, alpha=1.0, variant="ours", Y=Y2, bias=bd, cache=cache)
        check(f"linear_x6_reweighted({T},{in_f},{out_f})", got2, O.linear_relprop(R, X, 1.5 * W, 1.0, "ours"), 2e-5)
    finally:
        ops.USE_LINEAR_X6 = was
        ops.X6_CHECK = False
        ops.X6_TILE = 0


def _x6_operands(T, in_f, out_f, seed):
    X, W, R = rnd((T, in_f), seed), rnd((out_f, in_f), seed + 1, 0.05), rnd((T, out_f), seed + 2, 0.01)
    bias = rnd((out_f,), seed + 3, 0.3)
    Xd, Wd, bd, Rd = X.to(dev()), W.to(dev()), bias.to(dev()), R.to(dev())
    return Xd, Wd, bd, Rd, torch.nn.functional.linear(Xd, Wd, bd)


@pytest.mark.parametrize("T,in_f,out_f", [(1100, 768, 2304), (900, 3072, 768), (700, 256, 512), (257, 128, 256)])
def test_linear_x6_schedules_are_bitwise_equal(T, in_f, out_f):
    """One k-ordered chain per output whatever the schedule: the three tile geometries (per pass), two and three LDS stages
    (round 4 study: prefetch distance 2 in the 256-row geometry), and a 16-workgroup grid that cuts nearly every tile of these
    small shapes in two (the stream-K hand-over, which the default grid only uses at batch-64 sizes) give the same bits --
    for the rule and for the plain product of the same kernel."""
    from transformer_explainability_amd import ops
    Xd, Wd, bd, Rd, Y = _x6_operands(T, in_f, out_f, 300)
    was = ops.USE_LINEAR_X6
    ops.USE_LINEAR_X6, ops.X6_CHECK = True, True
    try:
        cache = {}
        ops.x6_raise_if_failed()
        base = ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache)
        wp = ops.x6_matrix_planes(Wd, False, cache)
        gbase = ops.gemm_x6(Xd, wp, bd, out_f)
        check(f"gemm_x6_small({T},{in_f},{out_f})", gbase, Y, 1e-5)
        # the study schedules (three LDS stages, the K split) exist in -DTE_X6_STUDY builds only
        from transformer_explainability_amd import _lib
        study = ops.x6_study_build()
        if not study:
            for fl in (ops.TE_X6_STAGES_3, ops.TE_X6_KSPLIT):
                ops.X6_TILE, ops.X6_FLAGS = 0, fl
                with pytest.raises(_lib.TeError):
                    ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache)
                with pytest.raises(_lib.TeError):
                    ops.gemm_x6(Xd, wp, bd, out_f)
        for tile in (1, 2, 3):                   # 128 x 256, 256 x 256, 128 x 128 tiles
            for st in ((0, ops.TE_X6_STAGES_3, ops.TE_X6_WHOLE_TILES) if study else (0, ops.TE_X6_WHOLE_TILES)):      # (+ ranges cut at tile boundaries only)
                for grid in (0, ops.TE_X6_TEST_SMALL_GRID):
                    ops.X6_TILE, ops.X6_FLAGS = tile, st | grid
                    got = ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache)
                    assert torch.equal(got, base), (tile, st, grid)
                    assert torch.equal(ops.gemm_x6(Xd, wp, bd, out_f), gbase), (tile, st, grid)
        # the K-split study (TE_X6_KSPLIT: two k-ordered chains per output for K >= 1536 into <= 768 weight rows; off by
        # default, it changes the bits): within the setting every geometry and schedule agrees bit for bit as well
        if study and in_f >= 1536 and out_f <= 768:
            ops.X6_TILE, ops.X6_FLAGS = 0, ops.TE_X6_KSPLIT
            kbase = ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache)
            gk = ops.gemm_x6(Xd, wp, bd, out_f)
            check(f"linear_x6_ksplit_vs_single_chain({T},{in_f},{out_f})", kbase, base, 1e-5)
            for tile in (1, 2, 3):
                for grid in (0, ops.TE_X6_TEST_SMALL_GRID):
                    ops.X6_TILE, ops.X6_FLAGS = tile, ops.TE_X6_KSPLIT | grid
                    assert torch.equal(ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache), kbase), (tile, grid)
                    assert torch.equal(ops.gemm_x6(Xd, wp, bd, out_f), gk), (tile, grid)
        # per-pass pins: Z on 128-row tiles, C on 256-row tiles and the other way round
        ops.X6_TILE = 0
        for fl in ((1 << ops.TE_X6_TILE_Z_SHIFT) | (2 << ops.TE_X6_TILE_C_SHIFT),
                   (2 << ops.TE_X6_TILE_Z_SHIFT) | (3 << ops.TE_X6_TILE_C_SHIFT),
                   (3 << ops.TE_X6_TILE_Z_SHIFT) | (1 << ops.TE_X6_TILE_C_SHIFT)):
            ops.X6_FLAGS = fl | ops.TE_X6_TEST_SMALL_GRID
            assert torch.equal(ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache), base), fl
        ops.x6_raise_if_failed()
    finally:
        ops.USE_LINEAR_X6, ops.X6_CHECK, ops.X6_TILE, ops.X6_FLAGS = was, False, 0, 0


@pytest.mark.parametrize("variant,alpha", [("lrp", 1.0), ("lrp", 2.0), ("ours", 2.0), ("ours", 1.5)])
@pytest.mark.parametrize("T,in_f,out_f", [(394, 768, 2304), (300, 3072, 768), (257, 128, 256), (600, 1024, 1024)])
def test_linear_x6_variant_lrp_and_alpha(variant, alpha, T, in_f, out_f):
    """variant lrp (modules/layers_lrp.py:188-211: S1 = sd(R, X+ W+^T), S2 = sd(R, X- W-^T)) and the
    inhibitor half alpha * f(pw, nw, px, nx) - beta * f(nw, pw, px, nx) (layers_ours.py:225-228) on the x6 kernels
    (te_linear_relprop_x6_general_f32): against the oracle, against the fp32-MFMA kernels of te_linear.hip, no less accurate
    against fp64 than those, and bitwise independent of tile geometry, stream-K cuts and batch composition."""
    from transformer_explainability_amd import ops
    X, W, R = rnd((T, in_f), 141), rnd((out_f, in_f), 142, 0.05), rnd((T, out_f), 143, 0.01)
    bias = rnd((out_f,), 144, 0.3)
    X[1] = 0.0
    X[2] = X[2].abs() + 0.01
    W[:5] = -W[:5].abs() - 0.001                                 # row 2 against these: every product negative
    Xd, Wd, bd, Rd = X.to(dev()), W.to(dev()), bias.to(dev()), R.to(dev())
    Y = torch.nn.functional.linear(Xd, Wd, bd)
    was = ops.USE_LINEAR_X6
    try:
        ops.USE_LINEAR_X6 = False
        fp32 = ops.linear_relprop(Rd, Xd, Wd, alpha=alpha, variant=variant, Y=Y, bias=bd)
        ops.USE_LINEAR_X6, ops.X6_CHECK = True, True
        assert bool(ops._lib.load().te_linear_relprop_x6_general_supported(T, in_f, out_f, 1 if variant == "lrp" else 0))
        cache = {}
        ops.x6_raise_if_failed()
        got = ops.linear_relprop(Rd, Xd, Wd, alpha=alpha, variant=variant, Y=Y, bias=bd, cache=cache)
        assert ("x6_planes_lrp" if variant == "lrp" else "x6_planes") in cache, "the x6 kernels must have run"
        ref = O.linear_relprop(R, X, W, alpha, variant)
        tag = f"({variant},{alpha},{T},{in_f},{out_f})"
        assert torch.isfinite(got).all()
        check("linear_x6_general" + tag, got, ref, 4e-5)
        check("linear_x6_general_vs_fp32_mfma" + tag, got, fp32, 4e-5)
        ref64 = O.linear_relprop(R.double(), X.double(), W.double(), alpha, variant)
        d6, d32 = got.cpu().double() - ref64, fp32.cpu().double() - ref64
        r6, r32 = float(d6.pow(2).mean().sqrt()), float(d32.pow(2).mean().sqrt())
        record("linear_x6_general_fp64" + tag, x6_rms=r6, fp32_mfma_rms=r32, x6_max=float(d6.abs().max()),
               fp32_mfma_max=float(d32.abs().max()))
        assert r6 <= 1.25 * r32 + 1e-10 * float(ref64.abs().max()), (r6, r32)
        # geometry / schedule / batch composition: the same bits
        for tile in (1, 2, 3):
            for grid in (0, ops.TE_X6_TEST_SMALL_GRID):
                ops.X6_TILE, ops.X6_FLAGS = tile, grid
                assert torch.equal(ops.linear_relprop(Rd, Xd, Wd, alpha=alpha, variant=variant, Y=Y, bias=bd, cache=cache),
                                   got), (tile, grid)
        ops.X6_TILE, ops.X6_FLAGS = 0, 0
        h = T // 2
        half = ops.linear_relprop(Rd[:h].contiguous(), Xd[:h].contiguous(), Wd, alpha=alpha, variant=variant,
                                  Y=Y[:h].contiguous(), bias=bd, cache=cache)
        assert torch.equal(half, got[:h])
        ops.x6_raise_if_failed()
    finally:
        ops.USE_LINEAR_X6, ops.X6_CHECK, ops.X6_TILE, ops.X6_FLAGS = was, False, 0, 0


@pytest.mark.parametrize("tile", [1, 2, 3], ids=["128x256", "256x256", "128x128"])
def test_linear_x6_lost_handover_is_loud(tile):
    """VERDICT r3 item 2 / ADVICE r3: a stream-K hand-over that never arrives must not yield a plausible result.  The test
    hook keeps every publisher's flag down: the waiting workgroup gives up after its bounded wait, ORs the sticky status
    word (ops.x6_raise_if_failed raises), poisons what it computed from the missing accumulators (NaN in the outputs of
    the C-pass and of the plain product) -- and every later wait gives up at once, so the call takes ~0.25 s, not one
    timeout per workgroup.  Afterwards the same call without the hook is clean and correct."""
    import time
    from transformer_explainability_amd import _lib, ops
    T, in_f, out_f = 1100, 768, 2304
    Xd, Wd, bd, Rd, Y = _x6_operands(T, in_f, out_f, 310)
    was = ops.USE_LINEAR_X6
    ops.USE_LINEAR_X6 = True
    try:
        cache = {}
        ops.x6_raise_if_failed()
        ops.X6_TILE = tile
        good = ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache)
        wp = ops.x6_matrix_planes(Wd, False, cache)
        ggood = ops.gemm_x6(Xd, wp, bd, out_f)
        torch.cuda.synchronize()
        assert not ops.x6_failed()
        ops.X6_FLAGS = ops.TE_X6_TEST_SMALL_GRID | ops.TE_X6_TEST_DROP_HANDOVER
        t0 = time.perf_counter()
        bad = ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert ops.x6_failed(), "the sticky status word must be set"
        assert not torch.isfinite(bad).all(), "outputs computed from missing accumulators must be NaN"
        assert dt < 5.0, f"later waits must give up at once (took {dt:.1f} s)"
        with pytest.raises(_lib.TeError, match="waited in vain"):
            ops.x6_raise_if_failed()
        assert not ops.x6_failed()                 # raising resets the word
        gbad = ops.gemm_x6(Xd, wp, bd, out_f)
        torch.cuda.synchronize()
        assert ops.x6_failed() and not torch.isfinite(gbad).all()
        with pytest.raises(_lib.TeError):
            ops.x6_raise_if_failed()
        record(f"x6_lost_handover(tile={tile})", seconds_for_failing_rule=dt, nan_fraction=float(torch.isnan(bad).float().mean()))
        # the per-call check of the C ABI sees it too
        ops.X6_CHECK = True
        with pytest.raises(_lib.TeError):
            ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache)
        ops.X6_CHECK = False
        for t in ops._x6_status.values():
            t.zero_()
        # the hook gone: clean and bit-identical again (on the small grid too)
        ops.X6_FLAGS = ops.TE_X6_TEST_SMALL_GRID
        assert torch.equal(ops.linear_relprop(Rd, Xd, Wd, Y=Y, bias=bd, cache=cache), good)
        assert torch.equal(ops.gemm_x6(Xd, wp, bd, out_f), ggood)
        ops.x6_raise_if_failed()
    finally:
        ops.USE_LINEAR_X6, ops.X6_CHECK, ops.X6_TILE, ops.X6_FLAGS = was, False, 0, 0
        for t in ops._x6_status.values():
            t.zero_()


def test_linear_x6_two_launches_on_two_streams_vit_b16_b64():
    """two persistent x6 rules side by side (two HIP streams) at the headline's own size -- T = 64 x 197
    rows, the qkv and fc2 shapes, whose C- / Z-passes are stream-K launches that want every CU.  A workgroup only ever
    waits for a lower-numbered workgroup of its OWN launch, whose publishing fragment is the first thing that one runs, so
    two launches cannot deadlock each other; whatever the dispatcher does, the call must end (bounded waits) and either
    give the serial bits or raise."""
    from transformer_explainability_amd import _lib, ops
    T = 64 * 197
    a = _x6_operands(T, 768, 2304, 320)
    b = _x6_operands(T, 3072, 768, 330)
    was = ops.USE_LINEAR_X6
    ops.USE_LINEAR_X6 = True
    try:
        ca, cb = {}, {}
        ops.x6_raise_if_failed()
        run = lambda o, c: ops.linear_relprop(o[3], o[0], o[1], Y=o[4], bias=o[2], cache=c)      # noqa: E731
        ra, rb = run(a, ca), run(b, cb)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for _ in range(4):
            s1.wait_stream(torch.cuda.current_stream())
            s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s1):
                xa = run(a, ca)
            with torch.cuda.stream(s2):
                xb = run(b, cb)
            torch.cuda.current_stream().wait_stream(s1)
            torch.cuda.current_stream().wait_stream(s2)
            outs.append((xa, xb))
        torch.cuda.synchronize()
        try:
            ops.x6_raise_if_failed()
        except _lib.TeError:
            record("x6_two_streams", outcome="raised")
            assert any(not torch.isfinite(x).all() for pair in outs for x in pair)
            return
        record("x6_two_streams", outcome="bitwise equal to the serial results")
        for xa, xb in outs:
            assert torch.equal(xa, ra) and torch.equal(xb, rb)
    finally:
        ops.USE_LINEAR_X6 = was
        for t in ops._x6_status.values():
            t.zero_()


@pytest.mark.parametrize("T,in_f,out_f", [(140, 256, 192), (140, 256, 256), (300, 768, 768)],
                         ids=["fp32-mfma", "x6-128", "x6-256"])
def test_linear_from_forward_output_cancellation(T, in_f, out_f):
    """Where (Y - b) and |X||W|^T cancel the kernel must fall back to the plain positive-part sum:
      * rows whose products are ALL negative (X > 0 against W rows < 0): reference Z = 0 exactly -> S = 0;
      * rows with a single positive product of relative size 1e-6: Z tiny but exact;
      * all-zero rows of X: Z = 0."""
    from transformer_explainability_amd import ops
    X, W, R = rnd((T, in_f), 51), rnd((out_f, in_f), 52, 0.05), rnd((T, out_f), 53, 0.01)
    X[:40] = X[:40].abs() + 0.01           # positive inputs ...
    W[:50] = -W[:50].abs() - 0.001         # ... against negative weight rows: every product negative
    X[40:45] = 0.0                         # zero rows
    X[45:50] = X[45:50].abs() + 0.01
    W[50:60] = -W[50:60].abs() - 0.001
    W[50:60, 7] = 1e-6                     # one tiny positive product among negatives
    bias = rnd((out_f,), 54, 0.3)
    Xd, Wd, bd = X.to(dev()), W.to(dev()), bias.to(dev())
    Y = torch.nn.functional.linear(Xd, Wd, bd)
    got = ops.linear_relprop(R.to(dev()), Xd, Wd, alpha=1.0, variant="ours", Y=Y, bias=bd)
    ref = O.linear_relprop(R, X, W, 1.0, "ours")
    assert torch.isfinite(got).all()
    check("linear_fwd_cancellation", got, ref, 2e-5)


@pytest.mark.parametrize("simple", [False, True], ids=["tiled", "simple"])
@pytest.mark.parametrize("variant", ["ours", "lrp"])
def test_linear_alpha2(simple, variant):
    from transformer_explainability_amd import ops
    set_impl(simple)
    T, in_f, out_f = 70, 96, 160
    X, W, R = rnd((2, 35, in_f), 24), rnd((out_f, in_f), 25, 0.05), rnd((2, 35, out_f), 26, 0.01)
    got = ops.linear_relprop(R.to(dev()), X.to(dev()), W.to(dev()), alpha=2.0, variant=variant)
    check(f"linear_alpha2_{variant}_{simple}", got, O.linear_relprop(R, X, W, 2.0, variant), 5e-5)
    assert got.shape == (2, 35, in_f)


def test_linear_rows_independent_and_homogeneous():
    """size-independent properties at the full ViT-B width: a row's result does not depend on which
    other rows share its tile (bitwise), and relprop(2R) == 2 relprop(R) bitwise."""
    from transformer_explainability_amd import ops
    T, in_f, out_f = 394, 768, 3072
    X, W, R = rnd((T, in_f), 27).to(dev()), rnd((out_f, in_f), 28, 0.05).to(dev()), rnd((T, out_f), 29, 0.01).to(dev())
    full = ops.linear_relprop(R, X, W)
    part = ops.linear_relprop(R[100:231].contiguous(), X[100:231].contiguous(), W)
    assert torch.equal(full[100:231], part)
    twice = ops.linear_relprop(2 * R, X, W)
    assert torch.equal(twice, 2 * full)
    # conservation: sum_i out[t,i] == sum_j R[t,j] (all Z != 0 here)
    cons = (full.double().sum(-1) - R.double().sum(-1)).abs().max() / R.double().abs().sum(-1).max()
    record("linear_conservation", rel=float(cons))
    assert float(cons) < 1e-4


# ------------------------------------------------------------------------------------------ attention rules
@pytest.mark.parametrize("simple", [False, True], ids=["tiled", "simple"])
@pytest.mark.parametrize("with_z", [False, True], ids=["computeZ", "forwardZ"])
@pytest.mark.parametrize("signed", [False, True], ids=["positive", "mixedsign"])
@pytest.mark.parametrize("B,H,N,D", [(2, 3, 7, 8), (2, 12, 197, 64), (1, 4, 130, 64), (1, 2, 577, 64), (4, 12, 197, 64),
                                     (1, 2, 64, 64), (1, 1, 65, 64), (2, 2, 512, 64), (1, 3, 33, 32), (1, 2, 1, 64),
                                     (1, 1, 63, 64), (1, 1, 129, 64)])
def test_attention_rules_fused_qkv_layout(simple, with_z, signed, B, H, N, D):
    """q/k/v read in place from the fused qkv activation [B,N,3HD]; outputs written in place into the
    'b n (qkv h d)' relevance buffer (strided views).

    positive : q, k, v > 0 => every Z = sum of positive products is well conditioned, so the kernels must
               agree with the fp32 oracle to summation-order accuracy (3e-5 of the tensor max).
    mixedsign: Z = attn v and Z = q k^T change sign, sd(R, Z) is ill conditioned wherever Z ~ 0 and the
               fp32 oracle itself is only as accurate as its distance to the fp64 oracle; the kernels are
               held to that band (gpu_util.check_conditioned)."""
    from transformer_explainability_amd import ops
    set_impl(simple)
    C = H * D
    qkv = rnd((B, N, 3 * C), 31)
    if not signed:
        qkv = qkv.abs() + 0.05
    v5 = qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    q, k, v = v5[0], v5[1], v5[2]
    attn = torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1).contiguous()
    Rav = rnd((B, N, C), 32, 0.01)
    r_heads = Rav.view(B, N, H, D).permute(0, 2, 1, 3)

    d = dev()
    qkv_d = qkv.to(d)
    v5d = qkv_d.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    cam_qkv = torch.full((B, N, 3 * C), float("nan"), device=d)
    slots = cam_qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    # with_z: Z handed over as the forward pass produced it on the device (rocBLAS products), as the model path does
    attn_d = attn.to(d)
    z_av = (attn_d @ v5d[2]) if with_z else None
    z_qk = (v5d[0] @ v5d[1].transpose(-1, -2)) if with_z else None
    cam1, cam_v = ops.matmul_relprop_av(Rav.to(d).view(B, N, H, D).permute(0, 2, 1, 3), attn_d, v5d[2],
                                        out_scale=0.5, cam_v_out=slots[2], z=z_av)
    ops.matmul_relprop_qk(cam1, v5d[0], v5d[1], out_scale=0.5, cam_q_out=slots[0], cam_k_out=slots[1], z=z_qk)
    assert not torch.isnan(cam_qkv).any()       # every slot of the fused buffer was written

    # the QK rule is checked on the relevance the device AV rule produced (its own input), so that each
    # rule is compared on identical inputs
    cam1_c = cam1.cpu()
    tag = f"({B},{H},{N},{D}){'simple' if simple else 'tiled'}{'+-' if signed else '+'}{'z' if with_z else ''}"
    # with_z: the forward products are inputs of the rule (oracle/relprop_oracle.py: matmul_relprop), so the oracle
    # gets the same Z and even the mixed-sign case is well conditioned (S is then identical up to one division)
    zc_av = z_av.cpu() if with_z else None
    zc_qk = z_qk.cpu() if with_z else None
    ref_attn, ref_v = O.einsum_av_relprop(r_heads, attn, v, zc_av)
    ref_q, ref_k = O.einsum_qk_relprop(cam1_c, q, k, zc_qk)
    if not signed or with_z:
        check("av.cam_attn" + tag, cam1, ref_attn * 0.5, 3e-5)
        check("av.cam_v" + tag, slots[2], ref_v * 0.5, 3e-5)
        check("qk.cam_q" + tag, slots[0], ref_q * 0.5, 3e-5)
        check("qk.cam_k" + tag, slots[1], ref_k * 0.5, 3e-5)
        if D == 64 and not simple and N >= 64:      # (N >= 64: enough elements for an rms ratio to mean something)
            # The AV rule runs on bf16 MFMAs with three-way split operands (csrc/te_attn_kb.hip: av6_kb_kernel; VERDICT r4
            # item 1): it must be as accurate against fp64 as the fp32 evaluation it replaces -- rms error vs the fp64
            # rule <= 2.5 x the fp32 oracle's, maximum <= 4 x.  Measured 0.6-1.4 up to N = 197 (the six kept partial
            # products are exact in the fp32 accumulator, like a plain fp32 product in another summation order) and 1.9
            # at N = 577 with all-positive operands, where the CPU oracle's blocked fp32 summation beats ANY kernel that
            # accumulates 577 terms in sequence (both errors are 1e-9 of the tensor's maximum there)
            a64, v64 = O.einsum_av_relprop(r_heads.double(), attn.double(), v.double(), None if zc_av is None else zc_av.double())
            for nm, got_, r32, r64 in (("cam_attn", cam1, ref_attn * 0.5, a64 * 0.5), ("cam_v", slots[2], ref_v * 0.5, v64 * 0.5)):
                e_k = (got_.detach().cpu().double() - r64)
                e_o = (r32.double() - r64)
                rms_k, rms_o = float(e_k.pow(2).mean().sqrt()), float(e_o.pow(2).mean().sqrt())
                mx_k, mx_o = float(e_k.abs().max()), float(e_o.abs().max())
                from gpu_util import record
                record("av_x6_vs_fp64." + nm + tag, rms_kernel=rms_k, rms_oracle32=rms_o, max_kernel=mx_k, max_oracle32=mx_o,
                       rms_ratio=rms_k / max(rms_o, 1e-300))
                assert rms_k <= 2.5 * rms_o + 1e-12 * float(r64.abs().max()), (nm, tag, rms_k, rms_o)
                assert mx_k <= 4.0 * mx_o + 1e-10 * float(r64.abs().max()), (nm, tag, mx_k, mx_o)
    else:
        a64, v64 = O.einsum_av_relprop(r_heads.double(), attn.double(), v.double())
        q64, k64 = O.einsum_qk_relprop(cam1_c.double(), q.double(), k.double())   # (computeZ: Z is part of the rule)
        check_conditioned("av.cam_attn" + tag, cam1, ref_attn * 0.5, a64 * 0.5, 3e-5)
        check_conditioned("av.cam_v" + tag, slots[2], ref_v * 0.5, v64 * 0.5, 3e-5)
        check_conditioned("qk.cam_q" + tag, slots[0], ref_q * 0.5, q64 * 0.5, 3e-5)
        check_conditioned("qk.cam_k" + tag, slots[1], ref_k * 0.5, k64 * 0.5, 3e-5)


def test_attention_rules_batch_independent():
    """(b,h) problems are independent: a batch run equals per-sample runs bitwise (full ViT-B geometry)."""
    from transformer_explainability_amd import ops
    B, H, N, D = 3, 12, 197, 64
    C = H * D
    d = dev()
    qkv = rnd((B, N, 3 * C), 33).to(d)
    v5 = qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    attn = torch.softmax(v5[0] @ v5[1].transpose(-1, -2) * D ** -0.5, -1).contiguous()
    R = rnd((B, H, N, D), 34, 0.01).to(d)
    cam1, cam_v = ops.matmul_relprop_av(R, attn, v5[2], out_scale=0.5)
    cam_q, cam_k = ops.matmul_relprop_qk(cam1, v5[0], v5[1], out_scale=0.5)
    for i in range(B):
        c1, cv = ops.matmul_relprop_av(R[i:i + 1], attn[i:i + 1], v5[2][i:i + 1], out_scale=0.5)
        cq, ck = ops.matmul_relprop_qk(c1, v5[0][i:i + 1], v5[1][i:i + 1], out_scale=0.5)
        assert torch.equal(c1, cam1[i:i + 1]) and torch.equal(cv, cam_v[i:i + 1])
        assert torch.equal(cq, cam_q[i:i + 1]) and torch.equal(ck, cam_k[i:i + 1])


# ------------------------------------------------------------------------------------------ rollout
@pytest.mark.parametrize("simple", [False, True], ids=["mfma", "simple"])
@pytest.mark.parametrize("L,B,N,start", [(4, 2, 50, 0), (12, 2, 197, 1), (3, 1, 197, 2), (5, 3, 64, 4), (4, 2, 577, 1),
                                         (3, 2, 512, 0), (3, 1, 1, 0)])
@pytest.mark.parametrize("normalise", [False, True])
def test_rollout(L, B, N, start, normalise, simple):
    from transformer_explainability_amd import ops
    set_impl(simple)
    cams = rnd((L, B, N, N), 41).abs() * 0.01
    got = ops.rollout(cams.to(dev()), start_layer=start, normalise=normalise)
    ref = O.rollout(list(cams), start, normalise=normalise)
    check(f"rollout({L},{B},{N},{start},{normalise})", got, ref, 1e-5)
    got2 = ops.rollout(cams.to(dev()), start_layer=start, normalise=normalise, cls_fixup=True)
    ref2 = ref.clone()
    ref2[:, 0, 0] = ref[:, 0].min(dim=-1).values
    check(f"rollout_fixup({L},{B},{N},{start},{normalise})", got2, ref2, 1e-5)


@pytest.mark.parametrize("L,B,N,start", [(4, 2, 50, 0), (12, 2, 197, 1), (3, 1, 197, 2), (5, 3, 64, 4), (4, 2, 577, 1),
                                         (3, 2, 512, 0), (3, 1, 1, 0), (12, 64, 197, 1), (2, 1, 1024, 0)])
@pytest.mark.parametrize("normalise", [False, True])
def test_rollout_row0_chain(L, B, N, start, normalise):
    """TE_ROLLOUT_ROW0: row 0 of the chain product as a vector x matrix chain (what both generators consume) vs the
    oracle's full product chain, with and without the CLS fix-up; a batch equals its samples run alone, bitwise."""
    from transformer_explainability_amd import ops
    cams = rnd((L, B, N, N), 43).abs() * 0.01
    ref = O.rollout(list(cams), start, normalise=normalise)
    got = ops.rollout(cams.to(dev()), start_layer=start, normalise=normalise, row0_only=True)
    assert got.shape == (B, N)
    check(f"rollout_row0({L},{B},{N},{start},{normalise})", got, ref[:, 0], 1e-5)
    got2 = ops.rollout(cams.to(dev()), start_layer=start, normalise=normalise, cls_fixup=True, row0_only=True)
    ref2 = ref[:, 0].clone()
    ref2[:, 0] = ref[:, 0].min(dim=-1).values
    check(f"rollout_row0_fixup({L},{B},{N},{start},{normalise})", got2, ref2, 1e-5)
    # the full-matrix path sliced afterwards (flag off) agrees to rounding
    ops.USE_ROW0_CHAIN = False
    try:
        full = ops.rollout(cams.to(dev()), start_layer=start, normalise=normalise, row0_only=True)
    finally:
        ops.USE_ROW0_CHAIN = True
    check(f"rollout_row0_vs_matrix({L},{B},{N},{start},{normalise})", got, full, 1e-5)
    if B > 1:
        for i in (0, B - 1):
            one = ops.rollout(cams[:, i:i + 1].contiguous().to(dev()), start_layer=start, normalise=normalise,
                              row0_only=True)
            assert torch.equal(one[0], got[i]), (i, float((one[0] - got[i]).abs().max()))


# ------------------------------------------------------------------------------------------ deferred Add factor
@pytest.mark.parametrize("shape", [(1, 9, 16), (3, 197, 768), (2, 5, 3), (64, 33, 64), (5, 1, 768)])
def test_add_deferred_equals_two_pass(shape):
    """te_add_relprop_deferred_f32 (one streaming pass, per-sample factor handed to the consumers) == the two-pass rule,
    BITWISE, once the factor is applied; and the consumers applying it in-kernel (Clone, Linear Z-pass epilogue) ==
    the same consumers fed the materialised tensors, bitwise."""
    from transformer_explainability_amd import ops
    X0, X1, R = rnd(shape, 2), rnd(shape, 3), rnd(shape, 4, 0.01)
    X0.view(-1)[0] = 0.0
    X1.view(-1)[0] = 0.0
    d = dev()
    a, b = ops.add_relprop(R.to(d), X0.to(d), X1.to(d), variant="ours")
    da, db = ops.add_relprop(R.to(d), X0.to(d), X1.to(d), variant="ours", deferred=True)
    assert isinstance(da, ops.Deferred) and isinstance(db, ops.Deferred)
    assert torch.equal(da.materialise(), a), float((da.materialise() - a).abs().max())
    assert torch.equal(db.materialise(), b), float((db.materialise() - b).abs().max())
    ra, rb = O.add_relprop(R, X0, X1, "ours")
    check(f"add_deferred{shape}.a", da.materialise(), ra, 2e-5)
    check(f"add_deferred{shape}.b", db.materialise(), rb, 2e-5)
    # Clone consuming (deferred, plain) / (deferred, deferred, plain)
    Xc = rnd(shape, 5).to(d)
    other = rnd(shape, 6, 0.01).to(d)
    assert torch.equal(ops.clone_relprop((da, other), Xc), ops.clone_relprop((a, other), Xc))
    assert torch.equal(ops.clone_relprop((other, db, da), Xc), ops.clone_relprop((other, b, a), Xc))
    # lrp variant has no rescale: the flag is ignored
    la, lb = ops.add_relprop(R.to(d), X0.to(d), X1.to(d), variant="lrp", deferred=True)
    assert torch.is_tensor(la) and torch.is_tensor(lb)


@pytest.mark.parametrize("B,H,N", [(1, 3, 7), (2, 12, 128), (3, 4, 300), (4, 12, 512)])
def test_add_bcast_mask_deferred(B, H, N):
    """Deferred broadcast-mask Add (one pass, factor handed to the QK rule) == the two-pass rule once the factor is
    applied, bitwise; the QK rule taking the factor in its S tile == the QK rule on the materialised operand, bitwise."""
    from transformer_explainability_amd import ops
    d = dev()
    X0 = rnd((B, H, N, N), 81).to(d)
    mask = torch.zeros(B, 1, 1, N)
    mask[::2, ..., N - max(1, N // 4):] = -10000.0
    mask = mask.to(d)
    R = rnd((B, H, N, N), 82, 0.01).to(d)
    a, b1 = ops.add_relprop(R, X0, mask, variant="ours")
    da, db1 = ops.add_relprop(R, X0, mask, variant="ours", deferred=True)
    assert isinstance(da, ops.Deferred)
    assert torch.equal(da.materialise(), a), float((da.materialise() - a).abs().max())
    assert torch.equal(db1, b1)
    D = 64
    q, k = rnd((B, H, N, D), 83).to(d), rnd((B, H, N, D), 84).to(d)
    z = q @ k.transpose(-1, -2)
    cq, ck = ops.matmul_relprop_qk(da, q, k, out_scale=0.5, z=z)
    rq, rk = ops.matmul_relprop_qk(a, q, k, out_scale=0.5, z=z)
    assert torch.equal(cq, rq) and torch.equal(ck, rk)


@pytest.mark.parametrize("B,N,in_f,out_f", [(3, 197, 768, 768), (2, 197, 3072, 768), (64, 1, 768, 768), (2, 50, 64, 192),
                                            (5, 33, 24, 40)])
def test_linear_with_deferred_relevance(B, N, in_f, out_f):
    """Linear.relprop whose relevance operand carries a deferred per-sample factor (row t scaled by s[t // N] inside
    the Z-pass epilogue) == the rule on the materialised operand, bitwise -- interior and edge tiles, rows of one
    32-row accumulator block spanning two samples (N = 197, 50, 33) and N = 1 (class-token path)."""
    from transformer_explainability_amd import ops
    d = dev()
    X, W, bias = rnd((B, N, in_f), 61).to(d), (rnd((out_f, in_f), 62) * 0.05).to(d), (rnd((out_f,), 63) * 0.1).to(d)
    Y = torch.nn.functional.linear(X, W, bias)
    R = rnd((B, N, out_f), 64, 0.01).to(d)
    fac = (rnd((B, 2), 65).abs() + 0.5).to(d)
    Rd = ops.Deferred(R, fac[:, 1])
    got = ops.linear_relprop(Rd, X, W, Y=Y, bias=bias)
    ref = ops.linear_relprop(Rd.materialise(), X, W, Y=Y, bias=bias)
    assert torch.equal(got, ref), float((got - ref).abs().max())
    # no cached forward output: the operand is materialised on the host side
    got2 = ops.linear_relprop(Rd, X, W)
    ref2 = ops.linear_relprop(Rd.materialise(), X, W)
    assert torch.equal(got2, ref2)
    check(f"linear_deferred({B},{N},{in_f},{out_f})", got, O.linear_relprop(Rd.materialise().cpu(), X.cpu(), W.cpu()), 3e-5)


@pytest.mark.parametrize("B,H,N", [(3, 16, 50), (1, 12, 577), (2, 20, 33), (64, 12, 197)])
def test_headmean_more_shapes(B, H, N):
    """Head-mean kernel (flat form: every head's loads in flight, H <= 16; grid-stride form beyond) on 16 / 20 heads,
    N = 577 and the bench shape; heads are added in index order, then divided by H, like torch's mean."""
    from transformer_explainability_amd import ops
    g, c = rnd((B, H, N, N), 13), rnd((B, H, N, N), 14, 0.01)
    got = ops.gradcam_headmean(g.to(dev()), c.to(dev()))
    check(f"headmean_more({B},{H},{N})", got, O.gradcam_headmean(g, c), 1e-6)


# ------------------------------------------------------------------------------------------ golden (reference outputs)
def test_golden_rules(golden_rules):
    from transformer_explainability_amd import ops
    g = golden_rules
    d = dev()
    for variant in ("ours", "lrp"):
        for alpha in (1, 2):
            got = ops.linear_relprop(g[f"linear_{variant}.R"].to(d), g[f"linear_{variant}.X"].to(d),
                                     g[f"linear_{variant}.W"].to(d), alpha=alpha, variant=variant)
            check(f"golden.linear_{variant}_a{alpha}", got, g[f"linear_{variant}.out_a{alpha}"], 2e-5)
        a, b = ops.add_relprop(g[f"add_{variant}.R"].to(d), g[f"add_{variant}.X0"].to(d), g[f"add_{variant}.X1"].to(d),
                               variant=variant)
        check(f"golden.add_{variant}.a", a, g[f"add_{variant}.out0"], 2e-5)
        check(f"golden.add_{variant}.b", b, g[f"add_{variant}.out1"], 2e-5)
    # the golden attention inputs are mixed-sign (Z ~ 0 somewhere): reference outputs vs their fp64 restatement
    # give the accuracy band of a plain fp32 evaluation (gpu_util.check_conditioned)
    o0, o1 = ops.matmul_relprop_av(g["av.R"].to(d), g["av.attn"].to(d), g["av.v"].to(d))
    r0, r1 = O.einsum_av_relprop(g["av.R"].double(), g["av.attn"].double(), g["av.v"].double())
    check_conditioned("golden.av.out0", o0, g["av.out0"], r0, 2e-5)
    check_conditioned("golden.av.out1", o1, g["av.out1"], r1, 2e-5)
    o0, o1 = ops.matmul_relprop_qk(g["qk.R"].to(d), g["qk.q"].to(d), g["qk.k"].to(d))
    r0, r1 = O.einsum_qk_relprop(g["qk.R"].double(), g["qk.q"].double(), g["qk.k"].double())
    check_conditioned("golden.qk.out0", o0, g["qk.out0"], r0, 2e-5)
    check_conditioned("golden.qk.out1", o1, g["qk.out1"], r1, 2e-5)
    a, b = ops.add_relprop(g["add_mask.R"].to(d), g["add_mask.X0"].to(d), g["add_mask.X1"].to(d))
    check("golden.add_mask.a", a, g["add_mask.out0"], 2e-5)
    for num in (2, 3):
        got = ops.clone_relprop([g[f"clone{num}.R{i}"].to(d) for i in range(num)], g[f"clone{num}.X"].to(d))
        check(f"golden.clone{num}", got, g[f"clone{num}.out"], 1e-6)
    got = ops.index_select_relprop(g["index_select.R"].to(d), g["index_select.X"].to(d), 0)
    check("golden.index_select", got, g["index_select.out"], 1e-6)


def test_rule_modules_match_reference_api(golden_rules):
    """The rule CLASSES (forward hook state + relprop(R, alpha)) reproduce the reference's outputs."""
    from transformer_explainability_amd import rules, rules_lrp
    g = golden_rules
    d = dev()
    for variant, mod in (("ours", rules), ("lrp", rules_lrp)):
        lin = mod.Linear(24, 40).to(d)
        with torch.no_grad():
            lin.weight.copy_(g[f"linear_{variant}.W"])
        lin(g[f"linear_{variant}.X"].to(d))
        check(f"module.linear_{variant}", lin.relprop(g[f"linear_{variant}.R"].to(d), 1),
              g[f"linear_{variant}.out_a1"], 2e-5)
        add = mod.Add()
        add([g[f"add_{variant}.X0"].to(d), g[f"add_{variant}.X1"].to(d)])
        a, b = add.relprop(g[f"add_{variant}.R"].to(d), 1)
        check(f"module.add_{variant}.a", a, g[f"add_{variant}.out0"], 2e-5)
    e2 = rules.einsum('bhij,bhjd->bhid')
    e2([g["av.attn"].to(d), g["av.v"].to(d)])
    o0, o1 = e2.relprop(g["av.R"].to(d), 1)
    r0, r1 = O.einsum_av_relprop(g["av.R"].double(), g["av.attn"].double(), g["av.v"].double())
    check_conditioned("module.einsum_av.0", o0, g["av.out0"], r0, 2e-5)
    check_conditioned("module.einsum_av.1", o1, g["av.out1"], r1, 2e-5)
    mm = rules.MatMul()
    mm([g["qk.q"].to(d), g["qk.k"].to(d).transpose(-1, -2)])
    o0, o1 = mm.relprop(g["qk.R"].to(d), 1)
    r0, r1 = O.matmul_relprop(g["qk.R"].double(), g["qk.q"].double(), g["qk.k"].double().transpose(-1, -2))
    check_conditioned("module.matmul_qkT.0", o0, g["matmul_qkT.out0"], r0, 2e-5)
    check_conditioned("module.matmul_qkT.1", o1, g["matmul_qkT.out1"], r1, 2e-5)
    cl = rules.Clone()
    cl(g["clone3.X"].to(d), 3)
    check("module.clone3", cl.relprop([g[f"clone3.R{i}"].to(d) for i in range(3)], 1), g["clone3.out"], 1e-6)
    isel = rules.IndexSelect()
    isel(g["index_select.X"].to(d), 1, torch.tensor(0, device=d))
    check("module.index_select", isel.relprop(g["index_select.R"].to(d), 1), g["index_select.out"], 1e-6)


# ------------------------------------------------------------------------------------------ consumer (8f.2)
@pytest.mark.parametrize("B,g,scale", [(3, 14, 16), (1, 24, 16), (2, 7, 4), (1, 1, 16)])
@pytest.mark.parametrize("normalise", [True, False])
def test_heatmap_consumer(B, g, scale, normalise):
    """bilinear x16 + per-map min-max + mean threshold vs the reference's own torch calls on the CPU."""
    from transformer_explainability_amd import ops
    if g == 1 and normalise:
        pytest.skip("a constant map has max == min (0/0 in the reference too)")
    maps = rnd((B, g * g), 61, 1e-4).abs()
    heat, mask = ops.heatmap(maps.to(dev()), scale=scale, normalise=normalise, with_mask=True)
    ref_h, ref_m = O.heatmap(maps, scale=scale, normalise=normalise)
    check(f"heatmap({B},{g},{scale},{normalise})", heat, ref_h, 2e-6)
    # the mask may differ only where a pixel sits within rounding of the mean
    disagree = (mask.cpu() != ref_m)
    near = (ref_h - ref_h.reshape(B, -1).mean(1).reshape(B, 1, 1, 1)).abs() <= 1e-6 * ref_h.abs().max()
    assert not (disagree & ~near).any()


# ------------------------------------------------------------------------------------------ Conv2d z^B (8f.3)
@pytest.mark.parametrize("simple", [False, True], ids=["tiled", "simple"])
@pytest.mark.parametrize("geom", [(2, 8, 12, 4, 12), (3, 32, 32, 8, 64), (2, 224, 224, 16, 768), (1, 48, 32, 16, 20)],
                         ids=lambda g: "B%d_%dx%d_p%d_E%d" % g)
def test_conv2d_zb(geom, simple):
    """z^B rule of a patch convolution vs the oracle's three convolutions + three transposed convolutions, R handed
    over as the token-major VIEW cam[:, 1:] that PatchEmbed.relprop builds (batch stride (P+1) E)."""
    from transformer_explainability_amd import ops
    set_impl(simple)
    B, H, W, p, E = geom
    X = rnd((B, 3, H, W), 71)
    Wt, bias = rnd((E, 3, p, p), 72, 0.05), rnd((E,), 73, 0.1)
    Hp, Wp = H // p, W // p
    cam = rnd((B, Hp * Wp + 1, E), 74, 0.01)                       # [B, N, E] with the class token in front
    R_view = cam[:, 1:].unflatten(1, (Hp, Wp)).permute(0, 3, 1, 2)  # [B,E,Hp,Wp], token-major memory
    Y = torch.nn.functional.conv2d(X, Wt, bias, stride=p)
    ref = O.conv2d_zb_relprop(R_view.contiguous(), X, Wt, p)
    ref64 = O.conv2d_zb_relprop(R_view.contiguous().double(), X.double(), Wt.double(), p)
    d = dev()
    got = ops.conv2d_zb_relprop(cam.to(d)[:, 1:].unflatten(1, (Hp, Wp)).permute(0, 3, 1, 2), X.to(d), Wt.to(d), Y.to(d),
                                bias.to(d))
    check_conditioned(f"conv2d_zb{geom}{'simple' if simple else ''}", got, ref, ref64, 2e-6)
    # an NCHW-contiguous R (what the reference's transpose + reshape produces) takes the copy path: same result
    got2 = ops.conv2d_zb_relprop(R_view.contiguous().to(d), X.to(d), Wt.to(d), Y.to(d), bias.to(d))
    assert torch.equal(got, got2)
    # conservation of the z^B rule: sum(out) = sum(R * (Za - 1e-9) / Za) ~ sum(R)
    assert abs(float(got.double().sum()) - float(R_view.double().sum())) <= 1e-4 * float(R_view.abs().double().sum())


@pytest.mark.parametrize("variant", ["ours", "lrp"])
def test_conv2d_zb_golden(golden_methods, variant):
    from transformer_explainability_amd import ops
    g, d = golden_methods, dev()
    X, Wt, b, R = (g[f"conv_{variant}.{k}"] for k in ("X", "W", "b", "R"))
    Y = torch.nn.functional.conv2d(X, Wt, b, stride=4)
    got = ops.conv2d_zb_relprop(R.to(d), X.to(d), Wt.to(d), Y.to(d), b.to(d))
    check(f"conv2d_zb.golden.{variant}", got, g[f"conv_{variant}.out"], 1e-5)
