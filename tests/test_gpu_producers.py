"""`-m gpu`: the producer kernels of SURVEY.md 8f.1 -- attention forward (scores, softmax, attn v in one pass over the
fused qkv activation) and the attention-gradient backward -- through the C ABI, against stock PyTorch on the same
device (tolerance: fp32 summation order, 1e-6-class) and, at model level, against the oracle on the tensors they cached.
Reference bodies: baselines/ViT/ViT_LRP.py:132-152 (forward), :144-145 (the gradient hook)."""
import pytest
import torch

from gpu_util import check, dev, map_stats, record, rnd, vit_cache_from_model
from oracle import relprop_oracle as O
from oracle.ref_harness import seeded_randn, synthetic_init

pytestmark = pytest.mark.gpu

SHAPES = [(2, 12, 197), (1, 3, 50), (2, 4, 224), (1, 2, 33), (3, 2, 1), (1, 1, 32), (2, 2, 64),
          # beyond 224 tokens: the row-tile kernels of csrc/te_attn_long.hip (ViT-L/16-384: 577; BERT: 512)
          (1, 2, 577), (2, 3, 512), (1, 2, 225), (1, 1, 640), (1, 2, 300)]
# round 6: the chunked long-sequence producers (csrc/te_attn_fwd6l.hip, te_attn_bwd6l.hip; 64 < N <= 640 through the strided entry
# points) at their edges: the shortest sequence they take, one key past a chunk / a block boundary, both workgroup cuts (4 and 8 waves)
L6_SHAPES = [(2, 2, 65), (1, 3, 96), (2, 1, 97), (1, 2, 129), (1, 1, 257), (1, 2, 639), (1, 1, 578), (2, 1, 131)]      # (N % 4 = 0 .. 3: the row tails)
SHAPES = SHAPES + [(1, 2, 639), (1, 1, 578), (2, 1, 131), (1, 2, 65)]


def _stock(qkv, H, scale):
    B, N, C3 = qkv.shape
    D = C3 // 3 // H
    q, k, v = qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    zqk = q @ k.transpose(-1, -2)
    attn = torch.softmax(zqk * scale, dim=-1)
    out = (attn @ v).permute(0, 2, 1, 3).reshape(B, N, H * D)
    return out, attn, zqk


@pytest.mark.parametrize("B,H,N", SHAPES)
def test_attention_forward_producer(B, H, N):
    from transformer_explainability_amd import ops
    D = 64
    assert ops.attention_forward_supported(N, D) and not ops.attention_forward_supported(641, D)
    qkv = rnd((B, N, 3 * H * D), 71).to(dev())
    scale = D ** -0.5
    out, attn, zqk = ops.attention_forward(qkv, H, scale)
    r_out, r_attn, r_zqk = _stock(qkv, H, scale)
    check(f"producer.fwd.zqk({B},{H},{N})", zqk, r_zqk, 2e-6)
    check(f"producer.fwd.attn({B},{H},{N})", attn, r_attn, 3e-6)
    check(f"producer.fwd.out({B},{H},{N})", out, r_out, 3e-6)
    assert float((attn.sum(-1) - 1).abs().max()) < 1e-5
    # fp64 on the host: both implementations are fp32-accurate
    q64 = qkv.double().cpu()
    e_out, e_attn, _ = _stock(q64, H, scale)
    assert float((out.cpu().double() - e_out).abs().max()) <= 2 * float((r_out.cpu().double() - e_out).abs().max()) + 1e-6
    # peaked rows (one score far above the rest) and a batch of one (b,h) equal to the batched run, bitwise
    spiky = qkv.clone()
    spiky[:, 0, :H * D] *= 30.0
    s_out, s_attn, _ = ops.attention_forward(spiky, H, scale)
    rs_out, rs_attn, _ = _stock(spiky, H, scale)
    # (scores up to a few hundred: one fp32 ulp of the scaled score is ~3e-5, and exp() turns it into a relative error)
    check(f"producer.fwd.spiky.attn({B},{H},{N})", s_attn, rs_attn, 3e-5)
    check(f"producer.fwd.spiky.out({B},{H},{N})", s_out, rs_out, 3e-5)
    one = ops.attention_forward(qkv[:1].contiguous(), H, scale)
    assert torch.equal(one[0], out[:1]) and torch.equal(one[1], attn[:1]) and torch.equal(one[2], zqk[:1])


@pytest.mark.parametrize("need_qk", [True, False])
@pytest.mark.parametrize("B,H,N", SHAPES)
def test_attention_backward_producer(B, H, N, need_qk):
    from transformer_explainability_amd import ops
    D = 64
    d = dev()
    qkv = rnd((B, N, 3 * H * D), 72).to(d).requires_grad_(True)
    scale = D ** -0.5
    g_out = rnd((B, N, H * D), 73).to(d)
    # stock autograd: gradient w.r.t. the probabilities (what register_hook receives) and w.r.t. qkv
    Bq, Nq, C3 = qkv.shape
    q, k, v = qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    attn = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
    attn.retain_grad()
    out = (attn @ v).permute(0, 2, 1, 3).reshape(B, N, H * D)
    out.backward(g_out)
    _, attn_k, _ = ops.attention_forward(qkv.detach(), H, scale)
    d_attn, d_qkv = ops.attention_backward(g_out, qkv.detach(), attn_k, H, scale, need_qk=need_qk)
    check(f"producer.bwd.d_attn({B},{H},{N})", d_attn, attn.grad, 3e-6)
    C = H * D
    check(f"producer.bwd.d_v({B},{H},{N})", d_qkv[..., 2 * C:], qkv.grad[..., 2 * C:], 3e-6)
    if need_qk:
        check(f"producer.bwd.d_q({B},{H},{N})", d_qkv[..., :C], qkv.grad[..., :C], 1e-5)
        check(f"producer.bwd.d_k({B},{H},{N})", d_qkv[..., C:2 * C], qkv.grad[..., C:2 * C], 1e-5)
        # round 6: with the forward output at hand the row sums of the softmax backward come from d_out . out
        # (te_attention_backward_out_f32; N <= 224: the rc kernel for every such N) -- same bars, same d_attn / d_v bits,
        # and a batch equals its samples bit for bit
        out_k, attn_k2, _ = ops.attention_forward(qkv.detach(), H, scale)
        d_attn2, d_qkv2 = ops.attention_backward(g_out, qkv.detach(), attn_k2, H, scale, need_qk=True, out=out_k)
        # (beyond 224 tokens the row side with `out` is te_attn_bwd6l.hip -- bf16 MFMAs with split operands -- and without it the
        #  fp32-MFMA kernel of round 3: d_attn agrees to rounding there, not bit for bit; d_v comes from the same column kernel)
        if N <= 224:
            assert torch.equal(d_attn2, d_attn)
        else:
            check(f"producer.bwd_out.d_attn({B},{H},{N})", d_attn2, attn.grad, 3e-6)
        assert torch.equal(d_qkv2[..., 2 * C:], d_qkv[..., 2 * C:])
        # The bar: 1e-5 of the gradient's maximum or -- where the softmax backward cancels, d_s = attn (d_attn - rowsum) ~ 0, at
        # N = 1 exactly 0 -- of the magnitude of the terms that cancel (scale |d_attn| |k|): the row sum is the same quantity
        # summed another way, so what is left of a cancellation is rounding of that size, not of the result's.
        nat = float(scale * attn.grad.abs().max() * qkv.detach().abs().max())
        for nm, sl in (("d_q", slice(0, C)), ("d_k", slice(C, 2 * C))):
            err = float((d_qkv2[..., sl] - qkv.grad[..., sl]).abs().max())
            bar = 1e-5 * max(float(qkv.grad[..., sl].abs().max()), nat)
            record(f"producer.bwd_out.{nm}({B},{H},{N})", max_abs=err, bar=bar)
            assert torch.isfinite(d_qkv2[..., sl]).all() and err <= bar, (nm, B, H, N, err, bar)
        one = ops.attention_backward(g_out[:1].contiguous(), qkv.detach()[:1].contiguous(), attn_k2[:1].contiguous(), H, scale,
                                     need_qk=True, out=out_k[:1].contiguous())
        assert torch.equal(one[1], d_qkv2[:1])
    # (need_qk=False: the q / k thirds of d_qkv are scratch -- the block has no consumer for them)


@pytest.mark.parametrize("B,H,N,masked", [(2, 12, 512, True), (2, 2, 128, True), (1, 3, 577, False), (3, 2, 40, True),
                                            (1, 12, 512, False)] + [(b, h, n, bool(i % 2)) for i, (b, h, n) in enumerate(L6_SHAPES)])
def test_attention_producer_bert_layout(B, H, N, masked):
    """The BERT form (BERT.py:336-352): three separate 'b n (h d)' activations, scores / sqrt(D), additive padding mask,
    softmax, probs v -- forward by-products (unscaled scores, masked scaled scores, probabilities) and all gradients
    against stock PyTorch on the device; a batch equals its samples bit for bit."""
    import math
    from transformer_explainability_amd import ops
    D = 64
    C = H * D
    d = dev()
    q, k, v = (rnd((B, N, C), 81 + i).to(d).requires_grad_(True) for i in range(3))
    mask = None
    if masked:
        m = torch.ones(B, N)
        m[::2, N - max(1, N // 8):] = 0
        mask = ((1.0 - m) * -10000.0).view(B, 1, 1, N).to(d)
    scale = 1.0 / math.sqrt(D)
    heads = lambda t: t.view(B, N, H, D).permute(0, 2, 1, 3)      # noqa: E731
    z = heads(q) @ heads(k).transpose(-1, -2)
    x = z / math.sqrt(D)
    if masked:
        x = x + mask
    probs = torch.softmax(x, dim=-1)
    probs.retain_grad()
    ctx = (probs @ heads(v)).permute(0, 2, 1, 3).reshape(B, N, C)
    g_out = rnd((B, N, C), 85).to(d)
    ctx.backward(g_out)
    out, attn, zqk, xsc = ops.attention_forward_qkv(q.detach(), k.detach(), v.detach(), H, scale, mask=mask, want_z=True,
                                                    want_x=masked)
    tag = f"({B},{H},{N},{masked})"
    check("producer.bert.zqk" + tag, zqk, z, 2e-6)
    if masked:
        # the kernel's x_scaled = the scaled scores BEFORE the mask (the Add module's first operand, BERT.py:339-342)
        check("producer.bert.x_scaled" + tag, xsc, (z / math.sqrt(D)).detach(), 2e-6)
    check("producer.bert.attn" + tag, attn, probs, 3e-6)
    check("producer.bert.out" + tag, out, ctx, 3e-6)
    for need_qk in (True, False):
        d_q, d_k, d_v = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        d_attn = ops.attention_backward_qkv(g_out, q.detach(), k.detach(), v.detach(), attn, H, scale,
                                            d_q if need_qk else None, d_k if need_qk else None, d_v, need_qk=need_qk)
        check("producer.bert.d_attn" + tag, d_attn, probs.grad, 3e-6)
        check("producer.bert.d_v" + tag, d_v, v.grad, 3e-6)
        if need_qk:
            check("producer.bert.d_q" + tag, d_q, q.grad, 1e-5)
            check("producer.bert.d_k" + tag, d_k, k.grad, 1e-5)
    # with the forward output at hand (te_attention_backward_strided_out_f32: row sums from d_out . out; beyond 64 tokens the row
    # side on te_attn_bwd6l.hip): the same bars, and sample 0 alone == sample 0 of the batch, bitwise
    d_q, d_k, d_v = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    d_attn = ops.attention_backward_qkv(g_out, q.detach(), k.detach(), v.detach(), attn, H, scale, d_q, d_k, d_v, need_qk=True, out=out)
    check("producer.bert_out.d_attn" + tag, d_attn, probs.grad, 3e-6)
    check("producer.bert_out.d_v" + tag, d_v, v.grad, 3e-6)
    check("producer.bert_out.d_q" + tag, d_q, q.grad, 1e-5)
    check("producer.bert_out.d_k" + tag, d_k, k.grad, 1e-5)
    d_q1, d_k1, d_v1 = torch.empty_like(q[:1]), torch.empty_like(k[:1]), torch.empty_like(v[:1])
    d_attn1 = ops.attention_backward_qkv(g_out[:1].contiguous(), q.detach()[:1], k.detach()[:1], v.detach()[:1], attn[:1].contiguous(), H,
                                         scale, d_q1, d_k1, d_v1, need_qk=True, out=out[:1].contiguous())
    assert torch.equal(d_attn1, d_attn[:1]) and torch.equal(d_q1, d_q[:1]) and torch.equal(d_k1, d_k[:1]) and torch.equal(d_v1, d_v[:1])
    # sample 0 alone == sample 0 of the batch, bitwise
    one = ops.attention_forward_qkv(q.detach()[:1], k.detach()[:1], v.detach()[:1], H, scale,
                                    mask=None if mask is None else mask[:1], want_z=True, want_x=masked)
    assert torch.equal(one[0], out[:1]) and torch.equal(one[1], attn[:1]) and torch.equal(one[2], zqk[:1])


@pytest.mark.parametrize("T,K,M", [(788, 768, 2304), (1576, 3072, 768), (600, 1024, 4096), (300, 128, 384), (12608, 768, 768)])
def test_linear_producer_x6_gemm(T, K, M):
    """SURVEY.md 8f.1, the Linear layers themselves: y = x W^T + b and d_x = d_y W on the split-operand bf16 kernels
    (te_gemm_x6_f32) against stock PyTorch on the device and against fp64: fp32-class accuracy (no worse than rocBLAS's
    fp32 GEMM), rows independent of the batch (bitwise), weight planes cached per weight version."""
    from transformer_explainability_amd import ops
    d = dev()
    x, W, b = rnd((T, K), 91).to(d), rnd((M, K), 92, 0.05).to(d), rnd((M,), 93, 0.3).to(d)
    dy = rnd((T, M), 94).to(d)
    assert ops.gemm_x6_supported(T, K, M)
    cache = {}
    y = ops.gemm_x6(x, ops.x6_matrix_planes(W, False, cache), b, M)
    ys = torch.nn.functional.linear(x, W, b)
    check(f"producer.linear.y({T},{K},{M})", y, ys, 4e-6)      # (two fp32-class roundings; 2.9e-6 at K = 3072)
    y64 = torch.nn.functional.linear(x.double().cpu(), W.double().cpu(), b.double().cpu())
    e6, es = float((y.cpu().double() - y64).pow(2).mean().sqrt()), float((ys.cpu().double() - y64).pow(2).mean().sqrt())
    record(f"producer.linear.fp64({T},{K},{M})", x6_rms=e6, stock_rms=es)
    assert e6 <= 1.1 * es + 1e-9, (e6, es)
    if ops.gemm_x6_supported(T, M, K):
        dx = ops.gemm_x6(dy, ops.x6_matrix_planes(W, True, cache), None, K)
        check(f"producer.linear.dx({T},{K},{M})", dx, dy @ W, 4e-6)      # (two fp32-class roundings, K up to 4096)
        assert "x6_gemm_planes_T" in cache
    assert "x6_gemm_planes" in cache
    h = T // 2
    assert torch.equal(ops.gemm_x6(x[:h].contiguous(), ops.x6_matrix_planes(W, False, cache), b, M), y[:h])
    assert torch.equal(ops.gemm_x6(x[7:300].contiguous(), ops.x6_matrix_planes(W, False, cache), b, M), y[7:300])


def _reference_planes(x):
    """The P3 plane layout of csrc/te_linear_x6.hip written with torch: p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1)
    (round to nearest even, exact residuals), stored [rows / 32][K / 16][plane 3][k half 2][row 32][8 bf16]."""
    T, K = x.shape
    Tp = (T + 31) // 32 * 32
    xp = torch.zeros(Tp, K, dtype=torch.float32, device=x.device)
    xp[:T] = x
    p0 = xp.bfloat16()
    r1 = xp - p0.float()
    p1 = r1.bfloat16()
    p2 = (r1 - p1.float()).bfloat16()
    pl = torch.stack([p0, p1, p2], 0).view(3, Tp // 32, 32, K // 16, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
    return pl.view(torch.uint8).flatten()


@pytest.mark.parametrize("T,K", [(788, 768), (300, 3072), (70, 128), (33, 144)])
def test_x6_split_dual_planes(T, K):
    """te_linear_x6_split_dual_f32: one pass, the signed planes of X and the planes of |X| -- bit for bit what the two
    single-purpose split kernels write (zeros, negative zeros and tiny values included), and bit for bit the layout and
    rounding of a torch restatement of the split (K = 144: a last group of K16 steps that is not full)."""
    from transformer_explainability_amd import _lib
    d = dev()
    lib = _lib.load()
    x = rnd((T, K), 97).to(d)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1.0, -1.0, 3.0e38, -3.0e38], device=d)
    x[1] = -x[1].abs()
    nb = lib.te_linear_x6_planes_bytes(T, K)
    s = torch.cuda.current_stream().cuda_stream
    bufs = [torch.zeros(nb, dtype=torch.uint8, device=d) for _ in range(4)]
    _lib.check(lib.te_linear_x6_split_dual_f32(x.data_ptr(), T, K, bufs[0].data_ptr(), bufs[1].data_ptr(), nb, s), "dual")
    _lib.check(lib.te_linear_x6_split_matrix_f32(x.data_ptr(), T, K, 0, bufs[2].data_ptr(), nb, s), "signed")
    _lib.check(lib.te_linear_x6_split_abs_f32(x.data_ptr(), T, K, bufs[3].data_ptr(), nb, s), "abs")
    torch.cuda.synchronize()
    assert torch.equal(bufs[0], bufs[2])
    assert torch.equal(bufs[1], bufs[3])
    assert torch.equal(bufs[2], _reference_planes(x))
    assert torch.equal(bufs[3], _reference_planes(x.abs()))


def test_linear_rule_reuses_forward_planes():
    """A Linear layer whose forward product ran on te_gemm_x6_f32 leaves the planes of |X| for its relprop rule
    (ops.gemm_x6 keep_abs -> ops.linear_relprop): the rule's result is bitwise the one with its own split pass, and a
    changed or different input makes the rule split again."""
    from transformer_explainability_amd import ops, rules
    d = dev()
    lin = rules.Linear(768, 2304).to(d).eval()
    x = rnd((2, 197, 768), 98).to(d)
    R = rnd((2, 197, 2304), 99, 1e-3).to(d)
    was = (ops.USE_FUSED_PRODUCERS, ops.X6_GEMM)
    ops.USE_FUSED_PRODUCERS, ops.X6_GEMM = True, "all"
    try:
        lin(x)
        cache = rules.x6_cache(lin)
        assert "x_abs_planes" in cache and cache["x_abs_planes"][0][0] == x.data_ptr()
        kept = cache["x_abs_planes"]
        with_planes = lin.relprop(R, 1.0)
        # consumed once (6 B per input element must not stay parked in the layer after its rule has run)
        assert "x_abs_planes" not in cache
        own_split = lin.relprop(R, 1.0)
        assert torch.equal(with_planes, own_split)
        cache["x_abs_planes"] = kept
        x.mul_(1.0)                                  # version bump: the planes no longer describe self.X
        assert ops._x_abs_key(lin.X, 394, 768) != kept[0]
        assert torch.equal(lin.relprop(R, 1.0), own_split)
        ops.X6_GEMM = "off"
        lin(x)                                       # stock forward: the stale planes are dropped
        assert "x_abs_planes" not in cache
    finally:
        ops.USE_FUSED_PRODUCERS, ops.X6_GEMM = was


def test_linear_layer_on_producers():
    """rules.Linear.forward / backward through autograd with ops.USE_FUSED_PRODUCERS: same values and input gradient as the
    stock layer to fp32 rounding; the stock path below 256 rows and for shapes the kernels do not tile."""
    from transformer_explainability_amd import ops, rules
    d = dev()
    lin = rules.Linear(768, 3072).to(d).eval()
    x = rnd((4, 197, 768), 95).to(d).requires_grad_(True)
    g = rnd((4, 197, 3072), 96).to(d)
    y0 = lin(x)
    (dx0,) = torch.autograd.grad(y0, x, g)
    was = ops.X6_GEMM
    ops.USE_FUSED_PRODUCERS, ops.X6_GEMM = True, "all"
    try:
        y1 = lin(x)
        assert "x6_gemm_planes" in rules.x6_cache(lin)
        (dx1,) = torch.autograd.grad(y1, x, g)
        check("producer.linear.layer.y", y1, y0, 2e-6)
        check("producer.linear.layer.dx", dx1, dx0, 5e-6)      # (K = 3072: two fp32-class roundings)
        assert lin.Y is y1 and lin.X.shape == x.shape
        small = lin(x[:1])                       # 197 rows: stock kernels
        assert torch.equal(small, torch.nn.functional.linear(x[:1], lin.weight, lin.bias))
        # per-direction policy (ops.gemm_x6_wanted): 768 -> 3072 has a narrow operand forward (x6) and a wide one backward (stock)
        from transformer_explainability_amd import producers
        ops.X6_GEMM = "auto"
        assert producers.linear_plan(x, lin) == (True, False)
        y2 = lin(x)
        (dx2,) = torch.autograd.grad(y2, x, g)
        assert torch.equal(y2, y1)
        check("producer.linear.layer.dx_auto", dx2, dx0, 2e-6)
        ops.X6_GEMM = "off"
        assert producers.linear_plan(x, lin) == (False, False)
    finally:
        ops.USE_FUSED_PRODUCERS = False
        ops.X6_GEMM = was


@pytest.fixture()
def fused():
    from transformer_explainability_amd import ops
    ops.USE_FUSED_PRODUCERS = True
    yield
    ops.USE_FUSED_PRODUCERS = False


def test_vit_b16_with_fused_producers(fused, golden_bands):
    """ViT-B/16 with the attention blocks on the producer kernels: logits and attention gradients agree with the stock
    forward / backward to fp32 rounding, the HIP relprop on the tensors the producers cached agrees with the oracle on
    the same tensors (tight), the map stays within the reference's own noise band of the sample, a batch equals its
    samples bitwise, and the whole pass is capturable in a HIP graph."""
    from transformer_explainability_amd import ops, vit
    from transformer_explainability_amd.generators import LRP, GraphedLRP
    from gpu_util import sliced_relprop_state
    model = vit.vit_base_patch16_224().eval()
    synthetic_init(model, 0)
    model.to(dev())
    x = seeded_randn((2, 3, 224, 224), 1).to(dev())
    lrp = LRP(model)
    ops.USE_FUSED_PRODUCERS = False
    stock_map = lrp.generate_LRP(x, start_layer=1).clone()
    stock_logits = model.head.Y.detach().clone()
    stock_grads = [b.attn.get_attn_gradients().clone() for b in model.blocks]
    ops.USE_FUSED_PRODUCERS = True
    fused_map = lrp.generate_LRP(x, start_layer=1).clone()
    assert all(b.attn._fused_anchor is not None for b in model.blocks)
    check("producer.vit_b16.logits", model.head.Y.detach(), stock_logits, 1e-5)
    for i in (11, 6, 0):
        check(f"producer.vit_b16.attn_grad.{i}", model.blocks[i].attn.get_attn_gradients(), stock_grads[i], 1e-4)
    # kernels in isolation: oracle on the very tensors the producers cached
    cache = vit_cache_from_model(model)
    oh = torch.zeros_like(stock_logits.cpu())
    oh.scatter_(1, model.head.Y.detach().cpu().argmax(-1, keepdim=True), 1.0)
    ref = O.vit_relprop(oh, cache, num_heads=12, start_layer=1)
    s = map_stats(fused_map, ref["map"])
    record("producer.vit_b16.oracle_same_cache", **s)
    assert s["normalised_max_abs"] <= 2e-4 and s["rel_linf"] <= 3e-4, s
    # against the reference's own map: within the sample's noise band (like the stock-producer path)
    for i in range(2):
        key = f"vit_b16.seed1.img{i}.sl1"
        st = map_stats(fused_map[i:i + 1], golden_bands[key + ".map"])
        record(f"producer.vit_b16.golden[{i}]", **st, band_norm=golden_bands[key + ".band_norm"])
        assert st["raw_max_abs"] <= 1e-4
        assert st["normalised_max_abs"] <= 5 * golden_bands[key + ".band_norm"] + 2e-5, (key, st)
    # batch == samples on the same cache, bitwise; batch == separate forward passes, bitwise too (the producer kernels
    # work per (b, h): unlike rocBLAS their summation order does not depend on the batch size -- the Linear layers
    # around them still do, so this is asserted on the attention block alone)
    ohd = oh.to(dev())
    for i in range(2):
        with sliced_relprop_state(model, i, 2):
            one = model.relprop(ohd[i:i + 1], method="transformer_attribution", start_layer=1, alpha=1)
        assert torch.equal(one, fused_map[i:i + 1]), i
    glrp = GraphedLRP(lrp, x, method="transformer_attribution", start_layer=1)
    assert torch.equal(glrp(x), fused_map)
    x2 = seeded_randn((2, 3, 224, 224), 8).to(dev())
    assert torch.equal(glrp(x2).clone(), lrp.generate_LRP(x2, start_layer=1))
    # pruned path: the lowest block whose gradient is wanted moves up to start_layer
    assert torch.equal(LRP(model, prune=True).generate_LRP(x, start_layer=1), fused_map)


# ------------------------------------------------------------------------------------------ LayerNorm / GELU producers
LN_SHAPES = [(2, 197, 768), (3, 50, 1024), (1, 7, 64), (2, 33, 2048), (5, 1, 4), (2, 512, 768)]


@pytest.mark.parametrize("B,N,C", LN_SHAPES)
@pytest.mark.parametrize("with_bias", [True, False])
def test_layernorm_producer(B, N, C, with_bias):
    """te_layernorm_forward_f32 / _backward_f32 against torch.nn.functional.layer_norm and its autograd on the same
    device (fp32 rounding), against fp64 on the host, with and without the bypass gradient, batch = samples bitwise."""
    from transformer_explainability_amd import ops
    import torch.nn.functional as F
    d = dev()
    x = (rnd((B, N, C), 81) * 3.0 + 0.5).to(d)
    w = (rnd((C,), 82) * 0.5 + 1.0).to(d)
    b = rnd((C,), 83).to(d) if with_bias else None
    dy = rnd((B, N, C), 84).to(d)
    byp = rnd((B, N, C), 85).to(d)
    assert ops.layernorm_supported(x)
    y, mean, rstd = ops.layernorm_forward(x, w, b, 1e-6)
    xr = x.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), w, b, 1e-6)
    (dxr,) = torch.autograd.grad(yr, xr, dy)
    check(f"producer.ln.fwd({B},{N},{C})", y, yr.detach(), 3e-6)
    dx = ops.layernorm_backward(dy, x, w, mean, rstd)
    check(f"producer.ln.bwd({B},{N},{C})", dx, dxr, 5e-6)
    dx2 = ops.layernorm_backward(dy, x, w, mean, rstd, add=byp)
    assert torch.equal(dx2, byp + dx)                                  # the fused addition is the plain fp32 sum
    # fp64 on the host: as accurate as the stock kernel
    x64 = x.double().cpu().requires_grad_(True)
    y64 = F.layer_norm(x64, (C,), w.double().cpu(), None if b is None else b.double().cpu(), 1e-6)
    (dx64,) = torch.autograd.grad(y64, x64, dy.double().cpu())
    e_mine, e_stock = float((y.cpu().double() - y64.detach()).abs().max()), float((yr.detach().cpu().double() - y64.detach()).abs().max())
    assert e_mine <= 2 * e_stock + 1e-6, (e_mine, e_stock)
    g_mine, g_stock = float((dx.cpu().double() - dx64).abs().max()), float((dxr.cpu().double() - dx64).abs().max())
    assert g_mine <= 2 * g_stock + 1e-6, (g_mine, g_stock)
    one = ops.layernorm_forward(x[:1].contiguous(), w, b, 1e-6)
    assert torch.equal(one[0], y[:1]) and torch.equal(one[1], mean[:N]) and torch.equal(one[2], rstd[:N])
    assert torch.equal(ops.layernorm_backward(dy[:1].contiguous(), x[:1].contiguous(), w, one[1], one[2]), dx[:1])


@pytest.mark.parametrize("shape", [(2, 197, 3072), (1, 50, 4096), (3, 5, 8), (1, 1, 4)])
def test_gelu_producer(shape):
    from transformer_explainability_amd import ops
    import torch.nn.functional as F
    d = dev()
    x = (rnd(shape, 91) * 3.0).to(d)
    x.view(-1)[:4] = torch.tensor([0.0, -0.0, 30.0, -30.0], device=d)      # zero and the saturated tails
    dy = rnd(shape, 92).to(d)
    y = ops.gelu_forward(x)
    xr = x.clone().requires_grad_(True)
    yr = F.gelu(xr)
    (dxr,) = torch.autograd.grad(yr, xr, dy)
    check(f"producer.gelu.fwd{shape}", y, yr.detach(), 1e-6)
    dx = ops.gelu_backward(dy, x)
    check(f"producer.gelu.bwd{shape}", dx, dxr, 2e-6)
    assert torch.equal(ops.gelu_forward(x[:1].contiguous()), y[:1])


@pytest.mark.parametrize("shape", [(2, 197, 3072), (1, 50, 4096), (3, 11, 144), (1, 33, 16)])
def test_gelu_producers_emit_operand_planes(shape):
    """Round 5: the GELU producers that write operand planes themselves.  te_gelu_forward_x6_planes_f32 =
    te_gelu_forward_f32's bits plus the two plane sets te_linear_x6_split_dual_f32 builds from them;
    te_gelu_backward_x6_planes_f32 = the planes te_linear_x6_split_matrix_f32 builds from te_gelu_backward_f32's output --
    bit for bit (ragged last row block and a last group of K16 steps that is not full included)."""
    from transformer_explainability_amd import _lib, ops
    d = dev()
    lib = _lib.load()
    x = (rnd(shape, 91) * 3.0).to(d)
    x.view(-1)[:4] = torch.tensor([0.0, -0.0, 30.0, -30.0], device=d)
    dy = rnd(shape, 92).to(d)
    K = shape[-1]
    T = x.numel() // K
    nb = lib.te_linear_x6_planes_bytes(T, K)
    s = torch.cuda.current_stream().cuda_stream
    y0, dx0 = ops.gelu_forward(x), ops.gelu_backward(dy, x)
    ref = [torch.zeros(nb, dtype=torch.uint8, device=d) for _ in range(3)]
    _lib.check(lib.te_linear_x6_split_dual_f32(y0.data_ptr(), T, K, ref[0].data_ptr(), ref[1].data_ptr(), nb, s), "dual")
    _lib.check(lib.te_linear_x6_split_matrix_f32(dx0.data_ptr(), T, K, 0, ref[2].data_ptr(), nb, s), "signed")
    y1, xs, xa = ops.gelu_forward_planes(x)
    dxp = ops.gelu_backward_planes(dy, x)
    torch.cuda.synchronize()
    assert torch.equal(y1, y0)
    assert torch.equal(xs[:nb], ref[0]) and torch.equal(xa[:nb], ref[1])
    assert torch.equal(dxp[:nb], ref[2])


def test_mlp_block_gelu_hands_planes_to_its_neighbours():
    """vit.Mlp (ViT_LRP.py:57-74: fc1 -> GELU -> fc2) with the Linear layers on the x6 kernels: the activation emits fc2's
    operand planes in its forward pass and fc1's input-gradient operand in its backward pass (producers._Gelu) -- output,
    input gradient and the relprop result are bitwise those of the separate split passes (ops.X6_FUSE_GELU = False), the
    fused kernels are the ones that ran.  The backward hand-off is an OPT-IN of the caller that owns the backward pass
    (ops.gelu_backward_plane_handoff): under a plain forward the hidden gradient is the real fp32 tensor for
    every consumer (autograd.grad / retain_grad / hooks); inside the context it is a NaN placeholder for anyone but fc1."""
    from transformer_explainability_amd import ops, producers, rules, vit
    d = dev()
    torch.manual_seed(3)
    mlp = vit.Mlp(768, 3072).to(d).eval()
    x = rnd((4, 197, 768), 95).to(d).requires_grad_(True)
    g = rnd((4, 197, 768), 96).to(d)
    R = rnd((4, 197, 768), 97, 1e-3).to(d)
    was = (ops.USE_FUSED_PRODUCERS, ops.X6_GEMM, ops.X6_FUSE_GELU)
    ops.USE_FUSED_PRODUCERS, ops.X6_GEMM = True, "all"
    calls = {"fwd": 0, "bwd": 0}
    f0, b0 = ops.gelu_forward_planes, ops.gelu_backward_planes

    def fwd(*a):
        calls["fwd"] += 1
        return f0(*a)

    def bwd(*a):
        calls["bwd"] += 1
        return b0(*a)

    ops.gelu_forward_planes, ops.gelu_backward_planes = fwd, bwd
    try:
        outs = []
        for fuse in (False, True):
            ops.X6_FUSE_GELU = fuse
            with ops.gelu_backward_plane_handoff():      # what LRP.generate_LRP / Generator.generate_LRP do for their forward
                y = mlp(x)
            (dx,) = torch.autograd.grad(y, x, g)
            cam = mlp.relprop(R, alpha=1.0)
            outs.append((y.detach().clone(), dx.clone(), cam.clone()))
            assert "dy_planes_from_consumer" not in rules.x6_cache(mlp.fc1)      # consumed by fc1's backward
            assert "x_planes_from_producer" not in rules.x6_cache(mlp.fc2)       # consumed by fc2's forward
        assert calls == {"fwd": 1, "bwd": 1}
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        assert torch.isfinite(outs[1][1]).all()
        ops.X6_FUSE_GELU = True
        gh = rnd((4, 197, 3072), 98).to(d)
        # plain forward (no opt-in): standard autograd semantics for the hidden activation -- the real gradient, for
        # autograd.grad, retain_grad and a tensor hook alike; the whole-block gradient is still bitwise the fused one
        ops.X6_FUSE_GELU = False
        h0 = mlp.fc1(x)
        (dh_ref,) = torch.autograd.grad(mlp.act(h0), h0, gh)
        ops.X6_FUSE_GELU = True
        h = mlp.fc1(x)
        h.retain_grad()
        seen = []
        h.register_hook(lambda t: seen.append(t.detach().clone()))
        a = mlp.act(h)
        (dh,) = torch.autograd.grad(a, h, gh, retain_graph=True)
        assert torch.isfinite(dh).all() and torch.equal(dh, dh_ref)
        y2 = mlp.fc2(a)
        (dx2,) = torch.autograd.grad(y2, x, g)
        assert torch.equal(dx2, outs[1][1]) and len(seen) == 2 and all(torch.isfinite(t).all() for t in seen)
        assert "dy_planes_from_consumer" not in rules.x6_cache(mlp.fc1)
        assert calls["bwd"] == 1                         # the plane-emitting backward did not run outside the context
        # inside the context the placeholder between the two nodes is NaN for anyone but fc1's input-gradient product
        with ops.gelu_backward_plane_handoff():
            h = mlp.fc1(x)
            a = mlp.act(h)
        (dh,) = torch.autograd.grad(a, h, gh)
        assert dh.shape == h.shape and not any(dh.stride()) and torch.isnan(dh).all()
        rules.x6_cache(mlp.fc1).pop("dy_planes_from_consumer", None)
        rules.x6_cache(mlp.fc2).pop("x_planes_from_producer", None)
        # the forward planes are held together with the tensor they were split from: a consumer that receives another
        # tensor falls back to its own split pass (and the entry is dropped)
        a = mlp.act(mlp.fc1(x))
        assert "x_planes_from_producer" in rules.x6_cache(mlp.fc2)
        assert rules.x6_cache(mlp.fc2)["x_planes_from_producer"][3] is a or \
            rules.x6_cache(mlp.fc2)["x_planes_from_producer"][3].data_ptr() == a.data_ptr()
        other = a.detach().clone()
        y3 = mlp.fc2(other)
        assert "x_planes_from_producer" not in rules.x6_cache(mlp.fc2) and torch.equal(y3, mlp.fc2(a.detach()))
    finally:
        ops.gelu_forward_planes, ops.gelu_backward_planes = f0, b0
        ops.USE_FUSED_PRODUCERS, ops.X6_GEMM, ops.X6_FUSE_GELU = was


def test_residual_layernorm_node(fused):
    """producers._ResidualLayerNorm (clone -> norm of a pre-norm block as one autograd node) gives the same values and
    the same input gradient as the two stock nodes."""
    from transformer_explainability_amd import producers, rules
    d = dev()
    norm = rules.LayerNorm(768, eps=1e-6).to(d).eval()
    with torch.no_grad():
        norm.weight.copy_(rnd((768,), 95) * 0.5 + 1.0)
        norm.bias.copy_(rnd((768,), 96))
    x = (rnd((2, 197, 768), 97) * 2.0).to(d)
    g1, g2 = rnd((2, 197, 768), 98).to(d), rnd((2, 197, 768), 99).to(d)
    xa = x.clone().requires_grad_(True)
    assert producers.norm_usable(xa, norm)
    x1, n = producers.residual_layer_norm(xa, norm)
    (dxa,) = torch.autograd.grad([x1, n], xa, [g1, g2])
    xb = x.clone().requires_grad_(True)
    nb = torch.nn.functional.layer_norm(xb, (768,), norm.weight, norm.bias, 1e-6)
    (dxb,) = torch.autograd.grad([xb * 1.0, nb], xb, [g1, g2])
    assert torch.equal(x1, x)
    check("producer.residual_ln.fwd", n, nb.detach(), 3e-6)
    check("producer.residual_ln.bwd", dxa, dxb, 5e-6)


@pytest.mark.parametrize("B,H,N", [(2, 12, 197), (3, 2, 50), (1, 4, 224), (5, 2, 33), (2, 1, 1)])
def test_attention_forward_emits_the_projection_planes(B, H, N):
    """Round 6: te_attention_forward_planes_f32 = te_attention_forward_f32's bits plus the two plane sets
    te_linear_x6_split_dual_f32 builds from `out` -- bit for bit, ragged last row block (B N not a multiple of 32) included."""
    from transformer_explainability_amd import _lib, ops
    d = dev()
    lib = _lib.load()
    D = 64
    qkv = (rnd((B, N, 3 * H * D), 101) * 2.0).to(d)
    scale = D ** -0.5
    out0, attn0, z0 = ops.attention_forward(qkv, H, scale)
    assert ops.attention_forward_planes_supported(qkv, H)
    out1, attn1, z1, xs, xa = ops.attention_forward(qkv, H, scale, planes=True)
    T, K = B * N, H * D
    nb = lib.te_linear_x6_planes_bytes(T, K)
    ref = [torch.zeros(nb, dtype=torch.uint8, device=d) for _ in range(2)]
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.te_linear_x6_split_dual_f32(out0.data_ptr(), T, K, ref[0].data_ptr(), ref[1].data_ptr(), nb, s), "dual")
    torch.cuda.synchronize()
    assert torch.equal(out1, out0) and torch.equal(attn1, attn0) and torch.equal(z1, z0)
    assert torch.equal(xs[:nb], ref[0]) and torch.equal(xa[:nb], ref[1])


def test_attention_block_hands_planes_to_the_projection():
    """vit.Attention on the producer kernels with the projection on the x6 kernels: the forward producer leaves proj's operand
    planes in the layer's cache, the layer's split pass does not run, and output, attention gradient and the relprop result
    are bitwise those of the separate passes (ops.X6_KEEP_ABS = False turns the hand-over off)."""
    from transformer_explainability_amd import ops, rules, vit
    d = dev()
    torch.manual_seed(5)
    res = {}
    x = rnd((4, 197, 768), 111).to(d)
    R = rnd((4, 197, 768), 112, 1e-3).to(d)
    saved = (ops.USE_FUSED_PRODUCERS, ops.X6_GEMM, ops.USE_LINEAR_X6, ops.X6_KEEP_ABS)
    try:
        ops.USE_FUSED_PRODUCERS, ops.X6_GEMM, ops.USE_LINEAR_X6 = True, "all", True
        blk = vit.Attention(768, num_heads=12, qkv_bias=True).to(d).eval()
        for handover in (True, False):
            ops.X6_KEEP_ABS = handover
            calls = {"planes": 0}
            orig = ops.attention_forward

            def counting(*a, **k):
                calls["planes"] += int(bool(k.get("planes")))
                return orig(*a, **k)

            ops.attention_forward = counting
            try:
                y = blk(x)
            finally:
                ops.attention_forward = orig
            assert calls["planes"] == (1 if handover else 0)
            cam = blk.proj.relprop(R, alpha=1)
            res[handover] = (y.detach().clone(), cam.detach().clone())
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    finally:
        ops.USE_FUSED_PRODUCERS, ops.X6_GEMM, ops.USE_LINEAR_X6, ops.X6_KEEP_ABS = saved
