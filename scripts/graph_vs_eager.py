#!/usr/bin/env python
"""Debug aid: ViT-B/16 batch 64, serial eager vs relprop-beside-backward (eager and graph replay): bitwise?  Where not, which
block's attn_cam / attention gradient differs first."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import transformer_explainability_amd as te  # noqa: E402
from transformer_explainability_amd import ops, vit  # noqa: E402
from transformer_explainability_amd.generators import LRP, GraphedCall  # noqa: E402
d = torch.device("cuda:0")
torch.manual_seed(0)
model = vit.vit_base_patch16_224().eval().to(d)
ops.USE_FUSED_PRODUCERS, ops.USE_LINEAR_X6 = True, True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g0 = torch.Generator().manual_seed(1)
x = torch.randn((B, 3, 224, 224), generator=g0).to(d)
def snap():
    return ([blk.attn.get_attn_cam().clone() for blk in model.blocks], [blk.attn.get_attn_gradients().clone() for blk in model.blocks])
# record the outputs of every relprop op, in call order
TRACE = None
def _flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x_ in o for t in _flat(x_)]
    if hasattr(o, "tensor"):      # ops.Deferred
        return _flat(o.tensor) + _flat(getattr(o, "fac", None))
    return []
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        out = fn(*a, **k)
        if TRACE is not None:
            TRACE.append((name, [t.detach().clone() for t in _flat(out)]))
        return out
    setattr(ops, name, w)
for n_ in ("add_relprop", "clone_relprop", "index_select_relprop", "linear_relprop", "matmul_relprop_av", "matmul_relprop_qk",
           "gradcam_headmean", "add_relprop_deferred", "clone_relprop_scaled"):
    if hasattr(ops, n_):
        wrap(n_)
lrp = LRP(model)
TRACE = []
maps = lrp.generate_LRP(x, method="transformer_attribution", start_layer=1).clone()
torch.cuda.synchronize()
TRACE0, TRACE = TRACE, None
print("ops traced:", len(TRACE0), flush=True)
cams0, grads0 = snap()
def compare(tag, out):
    if torch.equal(out, maps):
        print(f"{tag}: bitwise equal", flush=True)
        return
    cams, grads = snap()
    dm = float((out - maps).abs().max() / maps.abs().max())
    bad_c = [i for i in range(12) if not torch.equal(cams[i], cams0[i])]
    bad_g = [i for i in range(12) if not torch.equal(grads[i], grads0[i])]
    print(f"{tag}: DIFFERENT rel max {dm:.3e}; attn_cam differs in blocks {bad_c}; attention gradients differ in blocks {bad_g}", flush=True)
    for i in bad_c[-1:]:
        e = (cams[i] - cams0[i]).abs()
        idx = (e > 0).nonzero()
        print(f"   block {i} attn_cam: {idx.shape[0]} elements differ, max {float(e.max()):.3e} (tensor max {float(cams0[i].abs().max()):.3e}); b,h of first: {idx[:4].tolist()}", flush=True)
for r in range(3):
    compare(f"serial eager again {r}", lrp.generate_LRP(x, method="transformer_attribution", start_layer=1))
lrp_ov = LRP(model, overlap_backward=True)
for r in range(2):
    TRACE = []
    out = lrp_ov.generate_LRP(x, method="transformer_attribution", start_layer=1)
    torch.cuda.synchronize()
    tr, TRACE = TRACE, None
    compare(f"overlapped eager {r}", out)
    for i, ((n0, t0), (n1, t1)) in enumerate(zip(TRACE0, tr)):
        if n0 != n1 or len(t0) != len(t1) or not all(torch.equal(u, v_) for u, v_ in zip(t0, t1)):
            dd = [float((u - v_).abs().max()) for u, v_ in zip(t0, t1)] if n0 == n1 and len(t0) == len(t1) else "shape/name mismatch"
            prev = TRACE0[i - 1][0] if i else None
            print(f"   first differing op: #{i} {n0} (previous op: {prev}); max abs diff per output {dd}; shapes {[tuple(u.shape) for u in t0]}", flush=True)
            break
g = GraphedCall(lambda t: lrp_ov.generate_LRP(t, method="transformer_attribution", start_layer=1), (x,))
NREP = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for r in range(NREP):
    compare(f"graph replay {r}", g(x))
