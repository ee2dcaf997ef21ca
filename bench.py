#!/usr/bin/env python
"""bench.py -- relevance maps/sec for ViT-B/16 224^2 at batch 64 per GPU (BASELINE.json metric; configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config vit_b16_224 | vit_l16_384 | bert_base_512]

``--gpus N`` with N > 1 launches N ranks by itself (one process per GPU under ``torch.distributed.run`` on
127.0.0.1) when it is not already running under a launcher; under one (RANK in the environment, as the driver starts
it) it asserts WORLD_SIZE == N.  It refuses to run with fewer visible GPUs than ranks.

A "step" = one pass of the hot path over one batch of synthetic inputs already resident in HBM: PyTorch-ROCm forward +
attention-gradient backward (stock GEMMs / LayerNorm / GELU; the attention blocks themselves on the producer kernels of
SURVEY.md 8f.1 where the shape qualifies, --producers stock for PyTorch everywhere), then the HIP relprop rules, the gradient x relevance head-mean and the rollout
chain (LRP.generate_LRP, method "transformer_attribution", start_layer 1 as baselines/ViT/imagenet_seg_eval.py:196 of
the reference calls it), fp32 end to end.  All blocks are propagated; the only shortcuts are exact or rounding-level
(DESIGN.md section 3).  Every rank replays its step from a HIP graph captured BEFORE the process group exists (so
RCCL's threads never see an open capture); ONE step of the timed region runs eagerly with a HIP-event pair around
every C-ABI call of the relprop path -- the source of the roofline block.

Weak scaling: every rank runs the same batch size on its own inputs; the only communication is one all_gather (RCCL)
of the finished maps of the last step, inside the timed region.

The JSON line carries
  roofline      dominant kernel (Linear.relprop C-pass, fp32 MFMA) + ``kernels``: one entry per relprop kernel group
                with its ALGORITHMIC FLOPs / bytes per launch, the average launch duration from
                HIP events on the launch stream, and the fraction of the roofline that bounds it
  cpu_baseline  THE REFERENCE ITSELF (imported from /root/reference, or from its staged copy oracle/_ref -- see
                scripts/stage_reference.py) timed on the host cores of this box: >= 5 maps after one warm-up with all
                usable cores, plus the 1-thread figure; kind "port" (our CPU forward + the oracle) only if neither
                exists.  Rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import contextlib
import faulthandler
import fcntl
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (measured 2495)
X6_DTYPE = "f32 (3xbf16 split operands, 6 products, fp32 accumulate)"
HBM_PEAK_TBS = 8.0               # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
_T0 = time.perf_counter()

CONFIGS = {
    # name: (BASELINE.json configs index, default batch per GPU, description)
    "vit_b16_224": (1, 64, "ViT-B/16 224^2"),
    "vit_l16_384": (2, 32, "ViT-L/16 384^2"),
    "bert_base_512": (3, 32, "BERT-base 512 tokens"),
    # configs[4]: the ImageNet-seg style sweep, GLOBAL batch 256 sharded over the ranks (32 per GPU on 8), every image keyed by
    # its global index, explained + up-sampled (SaliencySweep), ONE gather of all maps at the end.  --steps = global batches
    # (default: the whole 50 000-image sweep = 196 steps); a 1-GPU run processes 256 per step itself.
    "sweep50k": (4, 0, "ViT-B/16 224^2 sweep"),
}
SWEEP_GLOBAL_BATCH = 256
SWEEP_IMAGES = 50_000


def log(msg):
    """Progress to stderr (the JSON line is the only thing on stdout)."""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def _resolve(args):
    if args.overlap_backward == "auto":
        args.overlap_backward = "on"
    if args.steps is None:
        args.steps = -(-SWEEP_IMAGES // SWEEP_GLOBAL_BATCH) if args.config == "sweep50k" else 5
    args.inflight_auto = args.inflight is None       # resolved in main() once the per-GPU batch is known
    if args.inflight is None:
        args.inflight = 1
    return args


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 5; --config sweep50k: 196 = the whole 50 000-image sweep in global batches of 256)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="vit_b16_224",
                    help="vit_b16_224 = BASELINE.json's headline (configs[1]); vit_l16_384 / bert_base_512 = configs[2] / [3]; "
                         "sweep50k = configs[4]: global batch 256 sharded over --gpus ranks, images keyed by global index, "
                         "SaliencySweep (explain + x16 up-sampling), one gather of all maps after the last step")
    ap.add_argument("--batch", type=int, default=0, help="inputs per GPU per step (0 = the configuration's own)")
    ap.add_argument("--start-layer", type=int, default=None,
                    help="default: 1 for ViT (imagenet_seg_eval.py:196), 0 for BERT (every layer reaches the map)")
    ap.add_argument("--cpu-baseline", choices=["auto", "port", "off"], default="auto")
    ap.add_argument("--cpu-maps", type=int, default=5, help="timed maps of the all-cores cpu_baseline leg")
    ap.add_argument("--parity", choices=["on", "off"], default="on",
                    help="N = 1: after the timed region, compare the step's maps with the cpu_baseline leg's maps of the same "
                         "inputs and with the CPU oracle on the step's own cached tensors (the line's `parity` object)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay each step from a HIP graph (one eager step inside the timed region carries the "
                         "per-kernel HIP events of the roofline); auto = on for the headline configuration, off for "
                         "ViT-L/384 and BERT-512 (a captured step pins a second copy of the 40-120 GB of activations)")
    ap.add_argument("--tuned-gemms", choices=["on", "off", "tune"], default="on",
                    help="stock fp32 GEMMs of forward/backward selected by PyTorch TunableOp from the committed results "
                         "file (on), PyTorch's default heuristic (off), or tune now and write gpurun_out/ (tune)")
    ap.add_argument("--prune", choices=["on", "off"], default="off",
                    help="skip the relprop rules and attention gradients of the blocks below --start-layer (their "
                         "attn_cam never reaches the map); off = every block, as the reference does")
    ap.add_argument("--overlap-backward", choices=["auto", "on", "off"], default="auto",
                    help="run the relprop rules on a side stream beside the attention-gradient backward pass: bitwise-equal "
                         "maps, graph-capturable.  auto (default since round 3) = on (ViT-B/16: 851 vs 803 maps/s under rocprofv3, one "
                         "trip; ViT-L/16-384: 81.7 vs 79.1; BERT-512: profiles/r03_overlap_backward_ab.log); the eager probe "
                         "step that feeds the roofline block always runs serially")
    ap.add_argument("--inflight", type=int, default=None,
                    help="consecutive steps (batches) in flight, each replayed / launched on its own HIP stream.  Default: 2 "
                         "where the step is a replayed HIP graph (ViT-B/16, the sweep and, since round 6, BERT-512: 925 vs 896 maps/s, same box, A B A B, "
                         "profiles/r04_inflight2.log), 1 for the eager configurations (a second ViT-L/16-384 step in flight would "
                         "be another 117 GB of activations)")
    ap.add_argument("--linear", choices=["x6", "fp32"], default="x6",
                    help="Linear.relprop kernels: x6 (default) = bf16 MFMAs on three-way split fp32 operands, six partial "
                         "products, fp32 accumulation (csrc/te_linear_x6.hip; fp32-class accuracy, asserted by the parity "
                         "tests); fp32 = the fp32-MFMA kernels of csrc/te_linear.hip.  With x6 the line also carries the "
                         "fp32-MFMA throughput of the same workload from a second timed run (config.fp32_mfma_*)")
    ap.add_argument("--x6-gemm", choices=["auto", "all", "off"], default="all",
                    help="with --producers fused: which forward / input-gradient products of the Linear layers run on the "
                         "split-operand bf16 kernel (te_gemm_x6_f32) instead of the stock fp32 GEMM: all (default; 73.3 vs "
                         "78.9 ms per ViT-B step), auto = only where the operand is narrow and the output wide "
                         "(ops.gemm_x6_wanted), off; the fp32-MFMA comparison run of the line always uses the stock GEMMs")
    ap.add_argument("--rules", choices=["ours", "lrp"], default="ours",
                    help="ViT configurations: the rule library the model is built over -- ours = modules/layers_ours.py "
                         "(baselines/ViT/ViT_LRP.py, method transformer_attribution: the headline), lrp = modules/layers_lrp.py "
                         "(baselines/ViT/ViT_orig_LRP.py, method grad: separate denominators per sign, 4 instead of 3 products per "
                         "Linear rule -- te_linear_relprop_x6_general_f32)")
    ap.add_argument("--x6-tile", choices=["auto", "lib", "128", "256"], default="auto",
                    help="tile geometry of the x6 Linear kernels (the maps do not depend on it, bit for bit): auto = 256 x 256 "
                         "tiles wherever the feature counts allow while the step runs concurrent streams (least CU-time per "
                         "launch), else the library's per-launch policy; lib = that policy always; 128 / 256 = pinned")
    ap.add_argument("--producers", choices=["stock", "fused"], default="fused",
                    help="fused (default): the attention blocks' forward and attention-gradient backward run on the "
                         "hand-written producer kernels (SURVEY.md 8f.1: head dim 64; N <= 224 one workgroup per head, N <= 640 "
                         "row tiles -- ViT-B/16 224^2, ViT-L/16 384^2 and BERT-512 alike); stock: PyTorch-ROCm everywhere")
    return _resolve(ap.parse_args(argv))


# --------------------------------------------------------------------------------------------------------- launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def build_once():
    """__graft_entry__.build() under an exclusive file lock: every rank may call it, one compiles, the rest wait and
    find the library up to date (no process group needed, so it can run before init_process_group)."""
    import __graft_entry__
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            __graft_entry__.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) and relay their exit code."""
    import torch
    rig = "TE_DEVICE_OVERRIDE" in os.environ or os.environ.get("TE_DIST_BACKEND") == "gloo"
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < args.gpus and not rig:
        sys.exit(f"bench.py --gpus {args.gpus}: only {visible} GPU(s) visible -- refusing to run {args.gpus} ranks on "
                 f"fewer devices (set TE_DIST_BACKEND=gloo + TE_DEVICE_OVERRIDE=0 for the one-GPU test rig)")
    build_once()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


# ----------------------------------------------------------------------------------------------------- kernel timer
class KernelTimer:
    """Brackets single C-ABI calls with HIP events recorded on torch's current stream (the stream the C ABI launches
    on).  Events are resolved after the timed region's final synchronise."""

    def __init__(self):
        self.records = []          # (name, flops, bytes, start_event, end_event)
        self.enabled = False

    @contextlib.contextmanager
    def __call__(self, name, flops, nbytes):
        if not self.enabled:
            yield
            return
        import torch
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        yield
        e.record()
        self.records.append((name, flops, nbytes, s, e))

    def names(self):
        return sorted({r[0] for r in self.records})

    def summary(self, name, min_share=0.05):
        """Launches of `name` carrying at least `min_share` of the largest launch's work: the class-token-only launches
        of the last block move ~1/N of a full launch's work and are launch-latency, not roofline, business.
        min_share = 0 keeps every launch (what rocprofv3's per-kernel average covers)."""
        rows = [(f, b, s.elapsed_time(e) * 1e-3) for n, f, b, s, e in self.records if n == name]
        if not rows:
            return None
        work = [max(f / (mfma_peak(name) * 1e12), b / (HBM_PEAK_TBS * 1e12)) for f, b, _ in rows]
        keep = [r for r, w in zip(rows, work) if w >= min_share * max(work)]
        n = len(keep)
        flops, nbytes, secs = (sum(r[i] for r in keep) for i in range(3))
        return {"launches": n, "launches_dropped_as_small": len(rows) - n, "avg_us": secs / n * 1e6,
                "flops_per_launch": flops / n, "bytes_per_launch": nbytes / n,
                "tflops": flops / secs / 1e12, "tbs": nbytes / secs / 1e12}


# configurations whose step is replayed as a HIP graph, two batches in flight, by default (ViT-L/16-384: a second step in flight would be
# another 117 GB of activations, and its graph alone measures -0.7 %: eager, one in flight; BERT-512: 354 vs 335 sequences/s, same box,
# round 6)
GRAPH_CONFIGS = ("vit_b16_224", "sweep50k", "bert_base_512")


def mfma_peak(name):
    """The MFMA roof a kernel group is priced against: the x6 kernels execute bf16 MFMAs (their flops are the EXECUTED
    bf16 work, six partial products per fp32 product), everything else fp32 MFMAs."""
    return MFMA_BF16_PEAK_TFLOPS if "x6" in name else MFMA_F32_PEAK_TFLOPS


KERNEL_GROUPS = {
    # timer name: (device kernels, reference rule)
    "linear_x6_cpass": ("x6_kernel<WM, MODE_C, 0, 2, 1> (rocprofv3 prints the numbers: <2, 1, 0, 2, 1>)", "Linear.relprop C-pass on bf16 MFMAs (P+ and P- side by side: 12 bf16 "
                                                "products of 2*T*in*out flops), layers_ours.py:220-225"),
    "linear_x6_zpass": ("x6_kernel<WM, MODE_Z, 0, 2, 1> (<2, 0, 0, 2, 1>; proj: <0, 0, 0, 2, 1>)", "Linear.relprop Z-pass from the forward output on bf16 MFMAs (6 products), "
                                                "S written as bf16 planes, layers_ours.py:216-219"),
    "linear_x6_split": ("zero_words_kernel + split_kernel<OP_ABS>", "|X| -> three bf16 planes in MFMA-fragment order"),
    "linear_x6_general": ("split_kernel<OP_POS / OP_NEG> + x6_kernel<., MODE_Z1> x 2 + x6_kernel<., MODE_X> x 2 (variant lrp) | "
                          "x6_kernel<., MODE_Z / MODE_C / MODE_ZI / MODE_CI> (ours, alpha != 1)",
                          "Linear.relprop for variant lrp / alpha != 1 on bf16 MFMAs: one-sided products, 24 (lrp) or 18 (ours) "
                          "bf16 product units of 2*T*in*out per half, layers_lrp.py:188-211, layers_ours.py:225-228"),
    "linear_forward_x6": ("split_kernel<OP_ID> (fc2: none, GELU emitted the planes) + x6_kernel<WM, MODE_G, 0, 2, 1>", "producer: y = x W^T + b (nn.Linear, "
                          "layers_ours.py:207) on bf16 MFMAs, 6 products of 2*T*in*out flops"),
    "linear_backward_x6": ("split_kernel<OP_ID> (fc1: none, GELU's backward emitted the planes) + x6_kernel<WM, MODE_G, 0, 2, 1>",
                           "producer: d_x = d_y W on bf16 MFMAs"),
    "linear_cpass": ("linear_k2_kernel<0,false,false>", "Linear.relprop C-pass, layers_ours.py:220-225"),
    "linear_zpass_fwd": ("linear_k1_kernel<ZM_FWD>", "Linear.relprop Z-pass from the forward output, layers_ours.py:216-219"),
    "linear_zpass": ("linear_k1_kernel<ZM_OURS>", "Linear.relprop Z-pass (two products), layers_ours.py:216-219"),
    "attention_av_rule": ("te_attn_kb::av6_kb_kernel<RULE> (round 5: wave-owned key blocks, bf16 MFMAs with split operands)",
                          "einsum 'bhij,bhjd->bhid' / MatMul rule, layers_ours.py:48-60"),
    "attention_qk_rule": ("qk_rule_kernel", "einsum 'bhid,bhjd->bhij' / MatMul rule, layers_ours.py:48-60"),
    "attention_fused_rules": ("attn_rules_kernel", "both attention rules of a ViT block in one pass, ViT_LRP.py:157-173"),
    "attention_forward": ("te_attn_fwd6::fwd6_kernel (N <= 224) / te_attn_fwd6l::fwd6l_kernel (64 < N <= 640, separate q / k / v, mask): row-block "
                          "owners, bf16 MFMAs with split operands (the fp32-MFMA roof `frac` is priced against when the "
                          "algorithmic fp32 flops outweigh the bytes is NOT the pipe these kernels run on: read `hbm_frac`)",
                          "producer: scores + softmax + attn v, ViT_LRP.py:132-152, BERT.py:336-352"),
    "attention_backward": ("N <= 224: te_attn_kb::av6_kb_kernel<BWD> (d_attn, d_v) + te_attn_rc::qk_rc_kernel<BWD>; 64 < N <= 640 (separate q / k / v): "
                           "te_attn_bwd6l::bwd6l_rows_kernel (d_attn, d_q) + bwd6l_cols_kernel (d_v, d_k) -- bf16 MFMAs with split operands: "
                           "read `hbm_frac`", "producer: attention-gradient backward, ViT_LRP.py:144-145, BERT.py:349-350"),
    "layernorm_forward": ("ln_fwd_kernel", "producer: LayerNorm forward, layers_ours.py:76 (ViT_LRP.py:184,187,266)"),
    "layernorm_backward": ("ln_bwd_kernel", "producer: LayerNorm input gradient (+ the bypass gradient of the residual "
                                            "block), ViT_LRP.py:203-205"),
    "gelu_forward": ("gelu_split_lds_kernel<SRC_GELU_FWD> (fp32 output + the operand planes of fc2's forward product and rule; "
                     "gelu_fwd_kernel where no x6 Linear follows)", "producer: GELU forward, layers_ours.py:70 (ViT_LRP.py:57)"),
    "gelu_backward": ("gelu_split_lds_kernel<SRC_GELU_BWD> (the gradient leaves as the operand planes of fc1's input-gradient "
                      "product, no fp32 tensor; gelu_bwd_kernel otherwise)", "producer: GELU input gradient"),
    "add_deferred": ("add_deferred_kernel + add_factors_kernel", "Add.relprop (one pass; rescale applied by the "
                                                                 "consumers), layers_ours.py:97-120"),
    "add": ("add_sums_kernel + add_apply_kernel", "Add.relprop, layers_ours.py:97-120"),
    "add_bcast_mask_deferred": ("addb_sums_kernel<store> + addb_finalize_kernel", "Add.relprop with the BERT mask operand, "
                                "one pass; rescale applied in the QK rule, BERT.py:386-388"),
    "add_bcast_mask": ("addb_sums + addb_finalize + addb_apply", "Add.relprop with the BERT mask operand, BERT.py:386-388"),
    "clone": ("clone_kernel / clone_scaled_kernel", "Clone.relprop, layers_ours.py:151-169"),
    "headmean": ("headmean_flat_kernel", "mean_h max(grad * attn_cam, 0), ViT_LRP.py:359-366"),
    "rollout_row0_chain": ("rollout_row_step_kernel x (L - start) + finish", "compute_rollout_attention row 0, "
                           "ViT_LRP.py:38-49,369 -- NOT an MFMA kernel on the hot path: the generators read row 0 of the "
                           "rollout only, which is a chain of L - start vector x matrix steps (3.4 MB per ViT-B map instead of "
                           "0.15 GFLOP of (N x N)(N x N) products); 12 launches of ~5 us, launch-bound, so the HBM fraction "
                           "below says nothing about the kernel.  The MFMA full-matrix chain (rollout_bmm_mfma_kernel) serves "
                           "method='rollout' / compute_rollout_attention callers and is timed as rollout_matrix_chain"),
    "rollout_matrix_chain": ("rollout_prep + rollout_bmm_mfma_kernel x (L-1-start)", "compute_rollout_attention"),
}


def kernel_table(timer):
    out = []
    for name in timer.names():
        s = timer.summary(name)
        if not s:
            continue
        t = s["avg_us"] * 1e-6
        t_f = s["flops_per_launch"] / (mfma_peak(name) * 1e12)
        t_b = s["bytes_per_launch"] / (HBM_PEAK_TBS * 1e12)
        bound = "mfma" if t_f >= t_b else "hbm"
        kern, rule = KERNEL_GROUPS.get(name, (name, ""))
        row = {"name": name, "kernels": kern, "rule": rule, "bound": bound, "launches": s["launches"],
               "avg_us": round(s["avg_us"], 2),
               "algorithmic_flops_per_launch": s["flops_per_launch"], "algorithmic_bytes_per_launch": s["bytes_per_launch"],
               "achieved": round(s["tflops"] if bound == "mfma" else s["tbs"], 4),
               "peak": mfma_peak(name) if bound == "mfma" else HBM_PEAK_TBS,
               "mfma_dtype": ("bf16 (executed flops)" if "x6" in name else "f32") if bound == "mfma" else None,
               "unit": "TFLOP/s" if bound == "mfma" else "TB/s", "frac": round(max(t_f, t_b) / t, 4),
               "hbm_frac": round(t_b / t, 4)}
        out.append(row)
    out.sort(key=lambda r: -r["avg_us"] * r["launches"])
    return out


# ------------------------------------------------------------------------------------------------------- workloads
def synthetic_image(global_index, shape=(3, 224, 224), seed=1):
    import torch
    g = torch.Generator().manual_seed(seed * 1_000_003 + global_index)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def synthetic_tokens(global_index, n_tokens=512, seed=1):
    import torch
    g = torch.Generator().manual_seed(seed * 1_000_003 + global_index)
    return torch.randint(1000, 20000, (n_tokens,), generator=g)


class Workload:
    """Model + resident inputs + the step function of one BASELINE.json configuration."""

    def __init__(self, args, rank, dev):
        import torch
        from transformer_explainability_amd import bert, vit
        from transformer_explainability_amd.generators import LRP, Generator
        self.args, self.name = args, args.config
        _, self.B, self.title = CONFIGS[args.config]
        if args.batch:
            self.B = args.batch
        B = self.B
        torch.manual_seed(0)
        self.sweep = args.config == "sweep50k"
        if self.sweep:
            world = int(os.environ.get("WORLD_SIZE", "1"))
            if SWEEP_GLOBAL_BATCH % world:
                sys.exit(f"--config sweep50k: the global batch of {SWEEP_GLOBAL_BATCH} does not divide over {world} ranks")
            self.B = B = args.batch or SWEEP_GLOBAL_BATCH // world
        self.rules = getattr(args, "rules", "ours")
        if args.config.startswith("vit") or self.sweep:
            if self.rules == "lrp":          # baselines/ViT/ViT_orig_LRP.py: the same architecture over the lrp rule library
                from transformer_explainability_amd import rules_lrp
                ns = vit.make_vit_module(rules_lrp)
                mk_b, mk_l = ns["vit_base_patch16_224"], ns["vit_large_patch16_224"]
            else:
                mk_b, mk_l = vit.vit_base_patch16_224, vit.vit_large_patch16_224
            if args.config in ("vit_b16_224", "sweep50k"):
                model, side = mk_b().eval(), 224
            else:
                model, side = mk_l(img_size=384).eval(), 384
            with torch.no_grad():          # non-trivial biases / LayerNorm scales so every path is exercised
                for _, p in model.named_parameters():
                    if p.dim() == 1:
                        p.add_(0.02 * torch.randn_like(p))
            self.cpu_state = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
            self.start_layer = 1 if args.start_layer is None else args.start_layer
            if self.sweep:
                # The layout the CPU tests validate (parallel.sweep_layout): rank r owns the contiguous block r of the
                # 50 000 GLOBAL indices and walks it in batches of 256 / world -- 195 whole batches and one short one (80
                # images on one rank, 10 per rank on eight); --steps k < 196 runs the first k batches of every rank's block.
                # --batch pins a per-rank batch instead (a rank's share measured on fewer GPUs): whole batches only.
                # All inputs resident in HBM before the timed region (600 KB per image).
                from transformer_explainability_amd import parallel as par
                world = int(os.environ.get("WORLD_SIZE", "1"))
                if args.batch:
                    lo = rank * args.steps * B
                    batches = [(lo + k * B, lo + (k + 1) * B) for k in range(args.steps)]
                else:
                    lo, _, bl = par.sweep_layout(SWEEP_IMAGES, world, SWEEP_GLOBAL_BATCH)[rank]
                    if args.steps > len(bl):
                        sys.exit(f"--config sweep50k: {args.steps} steps, but the sweep is {len(bl)} batches per rank")
                    batches = bl[:args.steps]
                self.sweep_batches = batches
                self.sweep_inputs = []
                for b_lo, b_hi in batches:
                    x = torch.empty((b_hi - b_lo, 3, side, side), dtype=torch.float32, device=dev)
                    for j, gi in enumerate(range(b_lo, b_hi)):
                        x[j] = par.synthetic_image_on(gi, dev, (3, side, side))
                    self.sweep_inputs.append(x)
                self.inputs = (self.sweep_inputs[0],)             # (warm-up steps and graph capture: the first batch)
                self.sweep_lo = lo
                self.sweep_local = sum(hi_ - lo_ for lo_, hi_ in batches)
            else:
                self.inputs = (torch.stack([synthetic_image(rank * B + i, (3, side, side)) for i in range(B)]).to(dev),)
            self.model = model.to(dev)
            self.gen = LRP(self.model, overlap_backward=(args.overlap_backward == "on"),
                           prune=(args.prune == "on"))
            if self.sweep:
                from transformer_explainability_amd.sweep import SaliencySweep
                self.saliency = SaliencySweep("transformer_attribution", lrp=self.gen, device=dev)
            self.tokens = (side // 16) ** 2 + 1
            self.out_cols = self.tokens - 1
            self.blocks = len(model.blocks)
            self.unit, self.noun = "maps/s", "maps"
            self.side = side
        else:
            model = bert.BertForSequenceClassification(bert.BertConfigLite(num_labels=2)).eval()
            with torch.no_grad():
                for _, p in model.named_parameters():
                    if p.dim() == 1:
                        p.add_(0.02 * torch.randn_like(p))
            self.cpu_state = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
            self.start_layer = 0 if args.start_layer is None else args.start_layer
            ids = torch.stack([synthetic_tokens(rank * B + i) for i in range(B)]).to(dev)
            mask = torch.ones(B, 512)
            mask[::2, 512 - 64:] = 0       # half of the batch padded: the broadcast-mask Add rule is exercised
            self.inputs = (ids, mask.to(dev))
            self.model = model.to(dev)
            self.gen = Generator(self.model, prune=(args.prune == "on"), overlap_backward=(args.overlap_backward == "on"))
            self.tokens, self.out_cols, self.blocks = 512, 512, 12
            self.unit, self.noun = "sequences/s", "sequences"
            self.side = None

    def eager(self, *inputs):
        if self.sweep:
            # generate_visualizations.py:60-98: normalise, explain (method "grad" = transformer_attribution, start_layer 1),
            # bilinear x16 + min-max; the [B,196] maps are what the final gather carries
            heat, maps = self.saliency.explain(inputs[0], return_maps=True)
            self.last_heat = heat
            return maps
        if self.name.startswith("vit"):
            return self.gen.generate_LRP(inputs[0], method="transformer_attribution" if self.rules == "ours" else "grad",
                                         start_layer=self.start_layer)
        return self.gen.generate_LRP(input_ids=inputs[0], attention_mask=inputs[1], start_layer=self.start_layer)

    def eager_serial(self, *inputs):
        """The probe step: kernel durations must be a kernel's own, so no relprop-beside-backward overlap."""
        ov = getattr(self.gen, "overlap_backward", False)
        if ov:
            self.gen.overlap_backward = False
        try:
            return self.eager(*inputs)
        finally:
            if ov:
                self.gen.overlap_backward = True


# ----------------------------------------------------------------------------------------------------- cpu baseline
def usable_cores():
    """Cores this process may actually run on (affinity and cgroup-quota aware), uncapped."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:   # cgroup v2 CPU quota
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, wl):
    """Time the CPU path on this box's host cores, one sample at a time (the reference is batch-1): all usable cores
    (1 warm-up + args.cpu_maps maps, median), a 32-thread leg when the host has more cores than that (torch's CPU GEMMs
    on M = 197 rows stop scaling long before a 100+ core host is filled; the best leg is the reported value), and the
    1-thread figure (1 warm-up + 2 maps)."""
    import torch
    from oracle import ref_harness as rh
    cores = usable_cores()
    kind = "reference" if (rh.reference_available() and args.cpu_baseline != "port") else "port"
    run64 = None
    log(f"cpu_baseline: kind {kind} ({rh.reference_origin()}), {cores} usable cores (os.cpu_count() = {os.cpu_count()}), "
        f"{cpu_model_string()}")
    n_in = max(args.cpu_maps, 2) + 1
    is_vit = wl.name.startswith("vit") or wl.sweep
    if is_vit:
        xs = torch.stack([synthetic_image(i, (3, wl.side, wl.side)) for i in range(n_in)])
    else:
        xs = torch.stack([synthetic_tokens(i) for i in range(n_in)])
        ms = torch.ones(n_in, 512)
        ms[::2, 512 - 64:] = 0

    if kind == "reference":
        if is_vit:
            mods = rh.load_reference_vit()
            ref_mod = mods["ViT_LRP"] if wl.rules == "ours" else mods["ViT_orig_LRP"]
            def fresh():
                if wl.name in ("vit_b16_224", "sweep50k"):
                    m_ = ref_mod.vit_base_patch16_224(pretrained=False).eval()
                else:
                    m_ = ref_mod.vit_large_patch16_224(pretrained=False, img_size=384).eval()
                m_.load_state_dict(wl.cpu_state)
                return m_
            model = fresh()
            gen = mods["gen"].LRP(model)
            meth = "transformer_attribution" if wl.rules == "ours" else "grad"
            run = lambda i: gen.generate_LRP(xs[i:i + 1], method=meth, start_layer=wl.start_layer)    # noqa: E731

            def run64(i, _cache=[]):      # the same reference code and weights evaluated in fp64 (parity block's yardstick)
                if not _cache:
                    _cache.append(mods["gen"].LRP(fresh().double()))
                return _cache[0].generate_LRP(xs[i:i + 1].double(), method=meth, start_layer=wl.start_layer)
            what = (f"reference LRP.generate_LRP over {'ViT_LRP' if wl.rules == 'ours' else 'ViT_orig_LRP'} "
                    f"(baselines/ViT/ViT_explanation_generator.py:25-41, method {meth})")
        else:
            mods = rh.load_reference_bert()
            from transformers import BertConfig
            cfg = BertConfig(num_labels=2)
            cfg.return_dict = False
            def fresh():
                m_ = mods["cls"].BertForSequenceClassification(cfg).eval()
                m_.load_state_dict(wl.cpu_state, strict=False)
                return m_
            model = fresh()
            gen = mods["gen"].Generator(model)
            run = lambda i: gen.generate_LRP(input_ids=xs[i:i + 1], attention_mask=ms[i:i + 1],   # noqa: E731
                                             start_layer=wl.start_layer)

            def run64(i, _cache=[]):
                if not _cache:
                    _cache.append(mods["gen"].Generator(fresh().double()))
                return _cache[0].generate_LRP(input_ids=xs[i:i + 1], attention_mask=ms[i:i + 1].double(),
                                              start_layer=wl.start_layer)
            what = "reference Generator.generate_LRP (BERT_explainability/modules/BERT/ExplanationGenerator.py:28-59)"
    else:
        from oracle import relprop_oracle as O
        from oracle.model_cache import bert_cache_from_model, vit_cache_from_model
        from transformer_explainability_amd import bert, vit
        if is_vit:
            model = (vit.vit_base_patch16_224() if wl.name in ("vit_b16_224", "sweep50k")
                     else vit.vit_large_patch16_224(img_size=384)).eval()
            model.load_state_dict(wl.cpu_state)
            heads = model.blocks[0].attn.num_heads

            def run(i):
                out = model(xs[i:i + 1])
                oh = torch.zeros_like(out)
                oh.scatter_(1, out.argmax(-1, keepdim=True), 1.0)
                grads = torch.autograd.grad((oh * out).sum(), [b.attn.get_attn() for b in model.blocks])
                for b, g in zip(model.blocks, grads):
                    b.attn.save_attn_gradients(g)
                return O.vit_relprop(oh, vit_cache_from_model(model), num_heads=heads, start_layer=wl.start_layer)["map"]
        else:
            model = bert.BertForSequenceClassification(bert.BertConfigLite(num_labels=2)).eval()
            model.load_state_dict(wl.cpu_state)

            def run(i):
                out = model(input_ids=xs[i:i + 1], attention_mask=ms[i:i + 1])[0]
                oh = torch.zeros_like(out)
                oh.scatter_(1, out.argmax(-1, keepdim=True), 1.0)
                layers = model.bert.encoder.layer
                grads = torch.autograd.grad((oh * out).sum(), [l.attention.self.get_attn() for l in layers])
                for l, g in zip(layers, grads):
                    l.attention.self.save_attn_gradients(g)
                return O.bert_relprop(oh, bert_cache_from_model(model), num_heads=12, start_layer=wl.start_layer)["map"]
        what = "our CPU forward/backward + the CPU oracle's relprop (no reference checkout or stage on this host)"

    ref_maps = {}      # input index -> the map this leg computed (BASELINE.json's metric: "max |delta| vs CPU ref")
    maps_by_threads = {}   # threads -> {input index -> map}: the same inputs under another thread count = another GEMM
    #                        summation order on the host: how far the reference moves from ITSELF (the parity block's floor)

    def leg(threads, n_maps):
        torch.set_num_threads(threads)
        times = []
        for i in range(n_maps + 1):
            t0 = time.perf_counter()
            with rh.reference_on_cpu():         # the reference hard-codes .cuda(); this leg runs on the host cores
                m = run(i % n_in)
            times.append(time.perf_counter() - t0)
            ref_maps.setdefault(i % n_in, m.detach().float().reshape(1, -1).clone())
            maps_by_threads.setdefault(threads, {}).setdefault(i % n_in, m.detach().float().reshape(1, -1).clone())
            log(f"cpu_baseline({kind}, {threads} threads) {wl.noun[:-1]} {i}: {times[-1]:.2f} s")
        times = sorted(times[1:])
        return times[len(times) // 2]

    legs = {cores: leg(cores, args.cpu_maps)}
    if cores > 48:
        legs[32] = leg(32, args.cpu_maps)
    best = min(legs, key=legs.get)
    # the 1-thread figure for the headline configuration only: a ViT-L / BERT-512 map takes
    # minutes on one thread
    one = None
    if wl.name in ("vit_b16_224", "sweep50k"):
        one = leg(1, 2) if cores > 1 else legs[cores]
    torch.set_num_threads(cores)
    first = next(iter(maps_by_threads))             # the all-cores leg: what ref_maps holds
    self_rows = [{"sample": i, "threads": f"{first} vs {t}", **_map_stats(m, maps_by_threads[first][i])}
                 for t, d in maps_by_threads.items() if t != first for i, m in sorted(d.items()) if i in maps_by_threads[first]]
    ref_maps["self"] = self_rows
    # the reference in fp64 on the first inputs (outside every timing): how far its own fp32 map is from the exact one
    ref64 = {}
    if run64 is not None and args.parity != "off":
        n64 = 3 if wl.name in ("vit_b16_224", "sweep50k") else 2
        for i in range(min(n64, n_in)):
            t0 = time.perf_counter()
            try:
                with rh.reference_on_cpu():
                    ref64[i] = run64(i).detach().double().reshape(1, -1).clone()
            except Exception as exc:      # (a reference module that does not survive .double(): the block says so)
                ref64 = {"error": f"{type(exc).__name__}: {exc}"}
                break
            log(f"cpu_baseline: reference in fp64, {wl.noun[:-1]} {i}: {time.perf_counter() - t0:.2f} s")
    ref_maps["ref64"] = ref64
    return ref_maps, {"value": 1.0 / legs[best], "unit": wl.unit, "cores": best, "kind": kind,
            "cpu_model": cpu_model_string(), "usable_cores": cores,
            "seconds_per_unit_by_threads": {str(k): round(v, 4) for k, v in sorted(legs.items())},
            "one_thread": None if one is None else {"value": 1.0 / one, "seconds_per_unit": round(one, 4),
                                                    "sample": "2 after 1 warm-up"},
            "sample": f"{what}: {args.cpu_maps} {wl.title} {wl.noun}, batch 1, after 1 warm-up; median "
                      f"{legs[best]:.3f} s each with {best} torch threads, fp32"}


def _map_stats(got, ref):
    """SURVEY.md 8d's three statistics of one map against its reference: raw max |delta|, max |delta| after per-map min-max
    normalisation (what imagenet_seg_eval.py:217 consumes), and raw max |delta| over max |ref|."""
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    mm = lambda m: (m - m.min()) / (m.max() - m.min())      # noqa: E731
    raw = float((got - ref).abs().max())
    return {"raw_max_abs": raw, "normalised_max_abs": float((mm(got) - mm(ref)).abs().max()),
            "rel_linf": raw / max(float(ref.abs().max()), 1e-300)}


def _worst(rows, vs, note):
    keys = ("raw_max_abs", "normalised_max_abs", "rel_linf")
    out = {k: max(r[k] for r in rows) for k in keys}
    med = sorted(r["normalised_max_abs"] for r in rows)
    out.update({"median_normalised_max_abs": med[len(med) // 2], "samples": len(rows), "vs": vs, "note": note,
                "per_sample": [{"sample": r["sample"], **{k: float(f"{r[k]:.3e}") for k in keys}} for r in rows]})
    return out


def cpu_baseline_parity(args, wl, gpu_maps, ref_maps):
    """Second half of the cpu_baseline leg: the accuracy half of BASELINE.json's metric ("max |delta| vs CPU ref"), OUTSIDE
    the timed region, after the line's throughput is final:

      vs_reference_cpu   the maps of the timed region's last step against the maps the cpu_baseline leg computed for the
                         same inputs (images / sequences 0 .. cpu_maps of rank 0's batch, the same weights): the whole
                         pipeline against the reference's, producers included.  With random-init weights and start_layer 1
                         LRP amplifies rounding-level differences of the forward / backward pass (the reference does not
                         reproduce ITSELF to 1e-4 there: DESIGN.md section 4), so the normalised statistic of this row is a
                         noise-floor measurement, not a kernel property; the raw bar (1e-4) always holds.
      vs_oracle_same_cache   the HIP relprop / head-mean / rollout kernels against the CPU oracle (bit-exact restatement of
                         the reference's rules, tests/test_oracle_golden.py) on the tensors one more eager step of this
                         process cached on the GPU, four samples: the kernels alone, north-star bar 1e-4.
    The oracle is used here as the checker only (as in tests/ and smoke())."""
    import torch
    from oracle import relprop_oracle as O
    from oracle.model_cache import bert_cache_from_model, sliced_relprop_state, vit_cache_from_model
    out = {"bar": 1e-4, "statistics": "raw_max_abs = max |ours - ref|; normalised_max_abs = the same after per-map min-max "
                                      "normalisation; rel_linf = raw / max |ref|; each the worst over the samples listed"}
    B = wl.B
    got = gpu_maps.detach().float().cpu()
    self_rows = ref_maps.pop("self", []) if ref_maps else []
    ref64 = ref_maps.pop("ref64", {}) if ref_maps else {}
    if ref_maps:
        rows = [{"sample": i, **_map_stats(got[i], m)} for i, m in sorted(ref_maps.items()) if i < B]
        out["vs_reference_cpu"] = _worst(rows, "reference CPU", "whole pipeline vs the reference's generate_LRP on the host "
                                         "cores (the cpu_baseline leg's own maps): different producers, see DESIGN.md section 4")
        if ref64 and "error" not in ref64:
            # the yardstick for vs_reference_cpu: the SAME reference code and weights evaluated in fp64 on the same inputs.
            # LRP divides by mixed-sign sums near zero, so with random-init weights at start_layer 1 the reference's fp32 map
            # is itself far from its fp64 map; "ours" is held against the same fp64 map beside it
            r32 = [{"sample": i, **_map_stats(ref_maps[i], m)} for i, m in sorted(ref64.items()) if i in ref_maps]
            o64 = [{"sample": i, **_map_stats(got[i], m)} for i, m in sorted(ref64.items()) if i < B]
            out["reference_fp32_vs_reference_fp64"] = _worst(r32, "reference CPU fp64", "the reference's fp32 map against its "
                                                             "own fp64 evaluation (same code, weights, inputs)")
            out["ours_vs_reference_fp64"] = _worst(o64, "reference CPU fp64", "this pipeline's maps against the reference's "
                                                   "fp64 evaluation of the same inputs")
        elif ref64:
            out["reference_fp64_error"] = ref64["error"]
        if self_rows:
            keys = ("raw_max_abs", "normalised_max_abs", "rel_linf")
            out["reference_cpu_vs_itself"] = {
                **{k: max(r[k] for r in self_rows) for k in keys}, "samples": len(self_rows),
                "note": "the reference's own maps of the same inputs under another torch thread count on the same host (oneDNN / "
                        "MKL keep the k-order of a GEMM when they re-partition rows over threads, so this only shows the host "
                        "path is deterministic; the yardstick for vs_reference_cpu is reference_fp32_vs_reference_fp64)",
                "per_sample": [{"sample": r["sample"], "threads": r["threads"], **{k: float(f"{r[k]:.3e}") for k in keys}}
                               for r in self_rows]}
    is_vit = wl.name.startswith("vit")
    # one more eager, serial step: its module caches are what the oracle reads; its maps must equal the replayed step's
    eager = wl.eager_serial(*wl.inputs).detach().clone()
    torch.cuda.synchronize()
    out["replayed_step_equals_eager_step_bitwise"] = bool(torch.equal(eager.cpu(), got))
    model = wl.model
    logits = (model.head.Y if is_vit else model.classifier.Y).detach().float().cpu()
    oh = torch.zeros_like(logits)
    oh.scatter_(1, logits.argmax(-1, keepdim=True), 1.0)
    heads = model.blocks[0].attn.num_heads if is_vit else 12
    rows = []
    for i in sorted({0, B // 3, (2 * B) // 3, B - 1}):
        with sliced_relprop_state(model, i, B):
            cache = vit_cache_from_model(model) if is_vit else bert_cache_from_model(model)
        if is_vit:     # (--rules lrp: ViT_orig_LRP's method "grad" is the same walk over modules/layers_lrp.py)
            ref = O.vit_relprop(oh[i:i + 1], cache, num_heads=heads, start_layer=wl.start_layer, variant=wl.rules)["map"]
        else:
            ref = O.bert_relprop(oh[i:i + 1], cache, num_heads=heads, start_layer=wl.start_layer)["map"]
        rows.append({"sample": i, **_map_stats(eager[i], ref)})
    out["vs_oracle_same_cache"] = _worst(rows, "oracle, same cache", "HIP kernels vs oracle/relprop_oracle.py on the tensors "
                                         "an eager step of this process cached (kernels only; bar 1e-4 on every statistic)")
    out["within_bar"] = bool(out["vs_oracle_same_cache"]["normalised_max_abs"] <= 1e-4
                             and out["vs_oracle_same_cache"]["raw_max_abs"] <= 1e-4
                             and out.get("vs_reference_cpu", {"raw_max_abs": 0.0})["raw_max_abs"] <= 1e-4)
    return out


# -------------------------------------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)                      # does not return
    faulthandler.enable()
    faulthandler.dump_traceback_later(900, repeat=True, file=sys.stderr)   # a stuck run leaves a stack trace

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", rank))
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} is running under a launcher with WORLD_SIZE={world}: they must agree")
    build_once()
    import transformer_explainability_amd as te
    from transformer_explainability_amd import ops, parallel
    from transformer_explainability_amd.generators import GraphedCall

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    rig = "TE_DEVICE_OVERRIDE" in os.environ or os.environ.get("TE_DIST_BACKEND") == "gloo"
    if torch.cuda.device_count() < world and not rig:
        sys.exit(f"rank {rank}: {torch.cuda.device_count()} GPU(s) visible for {world} ranks")
    te._lib.require_device()
    dev = torch.device("cuda", int(os.environ.get("TE_DEVICE_OVERRIDE", local)))   # (override: one-GPU test rig)
    torch.cuda.set_device(dev)
    cores = parallel.pin_rank_to_cores(local, world)      # N > 1: each rank enqueues from its own slice of the host cores
    if cores:
        log(f"rank {rank}: pinned to {cores} host cores (OMP_NUM_THREADS = {cores})")
    tuned = False
    if args.tuned_gemms == "on":
        tuned = te.enable_tuned_gemms()
    elif args.tuned_gemms == "tune":
        tuned = te.enable_tuned_gemms(os.path.join(ROOT, "gpurun_out", f"tunableop_gfx950_rank{rank}.csv"), tune=True)
    if args.producers == "fused":
        ops.USE_FUSED_PRODUCERS = True
    ops.X6_GEMM = args.x6_gemm
    ops.USE_LINEAR_X6 = args.linear == "x6"
    # x6 tile geometry.  The library's own policy (TE_X6_TILE_AUTO) optimises a launch running ALONE (few weight rows ->
    # smaller tiles so that every CU gets work).  A step, however, keeps several streams busy (relprop beside the backward
    # pass, two steps in flight): the CUs a narrow launch leaves idle are used by the other streams' kernels, and what counts
    # is CU-time per launch -- 150 workgroups of 256 x 256 tiles on 150 CUs cost 56 k CU-us where 128 x 128 tiles on all 256
    # cost 92 k (fc2's forward).  Measured, same box, A B A B, two steps in flight: 948 vs 928 maps/s
    # (profiles/r04_x6_geometry_step_ab.log).  So "auto" pins the large tiles whenever the step runs concurrent streams.
    # (resolved below, once --inflight auto is known: ADVICE r4)

    wl = Workload(args, rank, dev)
    B = wl.B
    if args.inflight_auto:
        # two replayed steps in flight where the step is a HIP graph AND a second step's activations are affordable: a ViT-B
        # step holds ~0.38 GB per sample (24 GB at batch 64; the one-GPU sweep's batch of 256 would be 2 x 96 GB + the eager
        # probe step's 96 GB: out of memory on a 288 GB part -- found the hard way, trip t16)
        graph = args.graph == "on" or (args.graph == "auto" and args.config in GRAPH_CONFIGS)
        args.inflight = 2 if (graph and B <= 64) else 1
    concurrent = args.overlap_backward == "on" or args.inflight > 1       # from the RESOLVED number of steps in flight
    ops.X6_TILE = {"auto": 2 if concurrent else 0, "lib": 0, "128": 1, "256": 2}[args.x6_tile]
    log(f"rank {rank}/{world}: {wl.title} model + {B} inputs resident on {dev}; x6 tile pin {ops.X6_TILE} "
        f"({'concurrent streams' if concurrent else 'single stream'}), {args.inflight} step(s) in flight")

    timer = KernelTimer()
    if not args.no_roofline:
        ops.KERNEL_TIMER = timer

    # --inflight N: round 3 refused it (two step graphs in flight "hung": every expired hand-over wait of the x6 kernels cost
    # seconds, and they expired one after the other).  Since round 4 the waits are bounded at 250 ms, fail fast once one has
    # expired and end in a TeError below (a workgroup only ever waits for a LOWER-numbered workgroup of its own launch, whose
    # publishing fragment is the first thing that one runs, so concurrent launches cannot starve each other).  Two replayed
    # steps in flight let the forward pass of step k + 1 run beside the relprop tail of step k: +3.2 % (923.7 / 926.7 vs
    # 896.0 / 896.5 maps/s, same box, A B A B); three: nothing more.
    lanes = [torch.cuda.Stream(device=dev) for _ in range(args.inflight)] if args.inflight > 1 else None
    counter = [0]

    # The graph is captured BEFORE the process group exists: RCCL's proxy / watchdog threads touch the HIP runtime on
    # their own and must never meet an open capture; replay afterwards is an ordinary launch.
    graphed = None
    lane_graphs = None           # --inflight N with graphs: one captured step (own static buffers) per lane
    use_graph = args.graph == "on" or (args.graph == "auto" and args.config in GRAPH_CONFIGS)
    if use_graph:
        try:
            if args.inflight == 1:
                graphed = GraphedCall(wl.eager, wl.inputs)
            else:
                lane_graphs = [GraphedCall(wl.eager, wl.inputs) for _ in range(args.inflight)]
            log(f"HIP graph of one step captured (x{max(1, args.inflight)})")
        except Exception as exc:      # capture is an optimisation of the host side only: fall back to eager launches
            graphed = lane_graphs = None
            torch.cuda.synchronize()
            log(f"HIP graph capture failed ({type(exc).__name__}: {exc}); running eagerly")

    if world > 1:
        r, w, _ = parallel.init_distributed()
        assert (r, w) == (rank, world), (r, w, rank, world)
        log(f"rank {rank}: process group up ({torch.distributed.get_backend()})")

    sweep_maps = (torch.empty((wl.sweep_local, wl.out_cols), dtype=torch.float32, device=dev) if wl.sweep else None)
    sweep_off = ([b_lo - wl.sweep_batches[0][0] for b_lo, _ in wl.sweep_batches] if wl.sweep else None)

    def step(eager=False, k=None):
        inputs = wl.inputs if not wl.sweep else (wl.sweep_inputs[k if k is not None else 0],)

        def keep(out):        # the sweep keeps every step's maps (a graph's static output is overwritten by its next replay)
            if wl.sweep and k is not None:
                sweep_maps[sweep_off[k]: sweep_off[k] + out.shape[0]].copy_(out)
            return out
        if eager:
            join()                       # the probe step runs alone: its kernel durations must be its own
            return keep(wl.eager_serial(*inputs))
        if wl.sweep and inputs[0].shape[0] != B:
            join()                       # the short last batch of the sweep: not the captured shape -- eager launches
            return keep(wl.eager(*inputs))
        if lanes is None:
            return keep(graphed(*inputs) if graphed is not None else wl.eager(*inputs))
        # several steps in flight: every tensor of a step is allocated, produced and consumed on that step's stream
        i = counter[0] % len(lanes)
        lane = lanes[i]
        counter[0] += 1
        lane.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(lane):
            return keep(lane_graphs[i](*inputs) if lane_graphs is not None else wl.eager(*inputs))

    def join():
        if lanes is not None:
            for lane in lanes:
                torch.cuda.current_stream(dev).wait_stream(lane)

    for w_ in range(args.warmup):
        maps = step()
        torch.cuda.synchronize()
        log(f"warmup step {w_} done")
    if (graphed is not None or lane_graphs is not None) and not args.no_roofline and args.warmup > 0:
        # The probe step of the timed region runs EAGERLY; capture emptied the caching allocator (GraphedCall), so its ~1000
        # allocations would each be a fresh hipMalloc inside the timed region (measured: 100-350 ms of host time for one
        # step, launches stalled for up to 17 ms between their two events).  One untimed eager step leaves the blocks in
        # the allocator's cache, exactly as the graph's own warm-up does for the capture.
        step(eager=True)
        torch.cuda.synchronize()
        log("warmup of the eager probe path done")
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # the probe step = the last step with a whole batch (the sweep's short last batch is not what a roofline row describes)
    probe_k = args.steps - 1
    if wl.sweep and args.steps > 1 and wl.sweep_inputs[probe_k].shape[0] != B:
        probe_k -= 1
    t0 = time.perf_counter()
    for k in range(args.steps):
        # with graph replay, ONE step of the timed region runs eagerly with a HIP-event pair around every C-ABI call of
        # the relprop path (events cannot be recorded inside a replayed graph); same kernels, same order, one stream
        # (eager configurations too: their other steps run as the library would -- relprop beside the backward pass, no events)
        probe = k == probe_k
        timer.enabled = probe and not args.no_roofline
        maps = step(eager=probe and not args.no_roofline, k=k)
    host_enqueue = time.perf_counter() - t0      # host time to enqueue all steps (GPU still running)
    join()
    t_compute = t_gather0 = None
    if world > 1:      # per-rank readiness figures: this rank's own compute time, then the gather's wall time
        torch.cuda.synchronize()
        t_gather0 = time.perf_counter()
        t_compute = t_gather0 - t0
    if wl.sweep:     # ONE collective for the whole sweep: every rank's [K * B, 196] block, in global order
        n_sweep = SWEEP_IMAGES if (not args.batch and args.steps == -(-SWEEP_IMAGES // SWEEP_GLOBAL_BATCH)) else world * wl.sweep_local
        gathered = parallel.gather_maps(sweep_maps, n_sweep)
    else:
        gathered = parallel.gather_maps(maps, world * B)
    torch.cuda.synchronize()
    t_gather = (time.perf_counter() - t_gather0) if t_gather0 is not None else None
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f"timed {args.steps} steps: {elapsed:.3f} s (host enqueue {host_enqueue:.3f} s)")
    timer.enabled = False
    ops.KERNEL_TIMER = None
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        mine = torch.tensor([t_compute, t_gather, float(wl.sweep_local if wl.sweep else B * args.steps)], dtype=torch.float64,
                            device=dev)
        if torch.distributed.get_backend() == "gloo":
            t, mine = t.cpu(), mine.cpu()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        every = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)        # (after the timed region: bookkeeping, not part of the data path)
        per_rank = [{"rank": r, "units": int(v[2]), "compute_s": round(float(v[0]), 4),
                     "units_per_s": round(float(v[2] / v[0]), 2), "gather_ms": round(float(v[1]) * 1e3, 3)}
                    for r, v in enumerate(every)]
    n_units = n_sweep if wl.sweep else world * B * args.steps          # what the timed region explained, all ranks
    gpu_maps = None if wl.sweep else maps.detach().clone()             # (a graph's static output: kept before anything re-runs)
    assert gathered.shape == ((n_sweep if wl.sweep else world * B), wl.out_cols) and torch.isfinite(gathered).all()
    ops.x6_raise_if_failed(dev)      # sticky device word of every x6 launch of the run (no synchronisation inside a step)

    # Rounds stay comparable: with the x6 Linear rules the same workload is timed once more on the fp32-MFMA kernels of
    # csrc/te_linear.hip (own graph capture, one warm-up, the same number of steps); N = 1 only, after the timed region.
    fp32_cmp = {}
    used_graph = graphed is not None or lane_graphs is not None
    if args.linear == "x6" and world == 1 and not wl.sweep:
        lane_graphs = None
        ops.USE_LINEAR_X6 = False
        ops.X6_GEMM = "off"          # the comparison run executes no bf16 MFMA at all: rules AND layer products on fp32 MFMAs
        try:
            g2 = None
            if used_graph:
                graphed = None      # release the first graphs' private memory pools before capturing the comparison's
                g2 = GraphedCall(wl.eager, wl.inputs)
            # ONE step in flight here whatever --inflight says: this path runs eight stock (hipBLASLt / rocBLAS, TunableOp-
            # selected) GEMMs per block, and two replayed graphs of it side by side stopped making progress once
            # (--overlap-backward off --inflight 2, trip t19: 300 s timeout; the x6 path of the same run had finished).  The
            # comparison therefore slightly favours the default path (two steps in flight: +3 %); the note says so.
            run2 = (lambda: g2(*wl.inputs)) if g2 is not None else (lambda: wl.eager(*wl.inputs))
            run2()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                m2 = run2()
            torch.cuda.synchronize()
            e2 = time.perf_counter() - t1
            assert torch.isfinite(m2).all()
            ops.x6_raise_if_failed(dev)
            fp32_cmp = {"fp32_mfma_maps_per_s" if wl.noun == "maps" else "fp32_mfma_sequences_per_s": B * args.steps / e2,
                        "fp32_mfma_ms_per_step": e2 / args.steps * 1e3,
                        "fp32_mfma_note": "the same step with the Linear rules on the fp32-MFMA kernels (te_linear.hip) and the "
                                          "layers' own products on the stock fp32 GEMMs (no bf16 MFMA anywhere), one graph "
                                          "replayed step after step (ONE step in flight, whatever the headline run used); "
                                          "second timed run of this process"}
            log(f"fp32-MFMA comparison run: {e2 / args.steps * 1e3:.2f} ms/step")
            del g2
        finally:
            ops.USE_LINEAR_X6 = True
            ops.X6_GEMM = args.x6_gemm

    if rank == 0 and os.environ.get("TE_BENCH_DUMP"):      # per-launch probe times by (group, algorithmic work) -> stderr
        import collections
        by = collections.defaultdict(list)
        for n_, f_, b_, s_, e_ in timer.records:
            by[(n_, round(f_ / 1e9, 1), round(b_ / 1e6, 1))].append(round(s_.elapsed_time(e_) * 1e3, 1))
        for k_, v_ in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            if sum(v_) > 300:
                log(f"probe {k_[0]} gflop {k_[1]} mb {k_[2]}: n {len(v_)} us {sorted(v_)[len(v_) // 2]} (min {min(v_)}, max {max(v_)})")
    if rank == 0:
        value = n_units / elapsed
        idx = CONFIGS[args.config][0]
        fused_on = args.producers == "fused" and ops.attention_forward_supported(wl.tokens, 64)
        fused_note = "attention blocks on the HIP producer kernels" if fused_on else "stock kernels throughout"
        line = {
            "metric": f"relevance {wl.noun}/sec ({wl.title}, batch {B} per GPU, generate_LRP "
                      f"{'transformer_attribution' if getattr(wl, 'rules', 'ours') == 'ours' else 'grad over the lrp rule library (ViT_orig_LRP)'})",
            "value": value, "unit": wl.unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": X6_DTYPE if args.linear == "x6" else "f32", "data": "synthetic",
            "build_id": te._lib.build_id(),
            **({"rig": True, "rig_note": "test rig: all ranks share ONE GPU over gloo -- not a multi-GPU measurement"}
               if rig else {}),
            "config": {"workload": f"{wl.title} batch {B} per GPU on "
                                   f"{(str(world) + ' ranks sharing one MI355X (rig)') if rig else (str(world) + 'xMI355X')}: "
                                   f"PyTorch-ROCm fwd + attn-grad bwd "
                                   f"({fused_note}) + fp32 relprop/head-mean/rollout HIP kernels"
                                   f"{' (Linear rules: fp32 operands as three bf16 planes on bf16 MFMAs)' if args.linear == 'x6' else ''} (BASELINE.json "
                                   f"configs[{idx}], sharded by sample)",
                       "batch_per_gpu": B, "global_batch": world * B, "world": world, "tokens": wl.tokens, "blocks": wl.blocks,
                       **({"per_rank": per_rank,
                           "per_rank_note": "compute_s = the rank's own wall time from the common start to its last kernel (synchronised), "
                                            "units_per_s = its units / compute_s, gather_ms = wall time of the single all_gather "
                                            "of the maps (includes waiting for the slowest rank); value = all units / the slowest "
                                            "rank's whole timed region",
                           "dist_backend": torch.distributed.get_backend()} if per_rank else {}),
                       **({"sweep_images": n_units, "sweep_last_batch_per_rank": wl.sweep_inputs[-1].shape[0],
                           "sweep_note": "parallel.sweep_layout: images keyed by GLOBAL index (rank r owns the "
                           "contiguous block r of the sweep, walked in batches of 256 / ranks; the whole sweep ends in one short "
                           "batch, run eagerly); each step = SaliencySweep.explain (generate_LRP + bilinear x16 + "
                           "min-max, generate_visualizations.py:60-98) of one batch; ONE all_gather of all [n,196] maps after "
                           "the last step, inside the timed region"} if wl.sweep else {}),
                       "start_layer": wl.start_layer, "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3,
                       "steps_in_flight": args.inflight,
                       "x6_tile": {"option": args.x6_tile, "pin": ops.X6_TILE,
                                   "meaning": {0: "library policy per launch (TE_X6_TILE_AUTO)", 1: "128 x 256 tiles",
                                               2: "256 x 256 tiles where the shape allows (the step runs concurrent streams)",
                                               3: "128 x 128 tiles"}[ops.X6_TILE],
                                   "extra_flags": hex(ops.X6_FLAGS)},
                       "gelu_emits_operand_planes": bool(ops.X6_FUSE_GELU and ops.USE_FUSED_PRODUCERS and ops.X6_GEMM != "off"),
                       "relprop_beside_backward": args.overlap_backward == "on",
                       "blocks_below_start_layer_pruned": args.prune == "on",
                       "producers": "fused attention forward/backward kernels" if fused_on else "stock",
                       "stock_gemm_selection": ("PyTorch TunableOp, committed results file" if tuned and
                                                args.tuned_gemms == "on" else
                                                "PyTorch TunableOp, tuned in this run" if tuned else "PyTorch default"),
                       "linear_relprop": ("x6: bf16 MFMAs on three-way split fp32 operands, six partial products, fp32 "
                                          "accumulation" if args.linear == "x6" else "fp32 MFMA"),
                       "rule_library": getattr(wl, "rules", "ours"),
                       "linear_forward_backward": (("stock fp32 GEMMs, except on the x6 kernel (te_gemm_x6_f32): "
                                                    + {"auto": "forward products and input gradients with K <= 1024 and "
                                                               "M >= 2 K (qkv / fc1 forward, fc2 input gradient)",
                                                       "all": "every supported product"}[args.x6_gemm])
                                                   if args.producers == "fused" and args.x6_gemm != "off"
                                                   else "stock fp32 GEMMs"),
                       **fp32_cmp,
                       "hip_graph": used_graph, "parallelism": f"dp{world} (independent samples, one "
                                                                      f"all_gather of the maps)"},
        }
        roof = None
        traffic = None
        traffic_src = None
        try:   # HBM-side bytes per C-pass launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE,
               # MI355X guide's gfx950 correction); only valid for the workload they were collected on
            if args.config == "vit_b16_224" and B == 64:
                cands = (("r06_linear_x6_traffic_pmc.json", "r05_linear_x6_traffic_pmc.json", "r04_linear_x6_traffic_pmc.json", "r03_linear_x6_traffic_pmc.json") if args.linear == "x6" else
                         ("r03_linear_traffic_pmc.json", "r02_linear_traffic_pmc.json", "r01_linear_traffic_pmc.json"))
                for cand in cands:
                    path = os.path.join(ROOT, "profiles", cand)
                    if os.path.exists(path):
                        with open(path) as f:
                            tr = json.load(f)
                        cps = [(v["traffic_bytes"], v.get("dispatches", 1)) for k, v in tr.items() if k.endswith(".cpass")]
                        traffic = sum(t * n for t, n in cps) / sum(n for _, n in cps)      # per launch, all C-pass shapes
                        traffic_src = cand
                        break
        except (OSError, ValueError, KeyError):
            traffic = None
        # the headline block averages EVERY launch of the kernel, like the rocprofv3 kernel-stats row it must agree with
        kernels_note = ("one eager step inside the timed region; per C-ABI call: HIP events on the launch stream, "
                        "ALGORITHMIC flops / bytes (SURVEY.md 8d, App. B; the x6 kernels: the bf16 MFMA flops they "
                        "EXECUTE, six partial products per fp32 product, against the bf16 peak), frac = max(flops / MFMA "
                        f"peak ({MFMA_F32_PEAK_TFLOPS} TF fp32, {MFMA_BF16_PEAK_TFLOPS:.0f} TF bf16), bytes / "
                        f"{HBM_PEAK_TBS} TB/s) / measured time; launches carrying < 5 % of the group's largest work "
                        "(class-token path of the last block) are excluded from the table's averages (the headline block "
                        "above averages every launch, like rocprofv3's kernel-stats row)")
        x6c = timer.summary("linear_x6_cpass", 0.0)
        x6z = timer.summary("linear_x6_zpass")      # (full-size launches; the table's row)
        cp = timer.summary("linear_cpass", 0.0)
        zp = timer.summary("linear_zpass_fwd", 0.0) or timer.summary("linear_zpass", 0.0)
        if x6c:
            # dominant kernel of the default path: the C-pass on bf16 MFMAs.  `achieved` = EXECUTED bf16 flops (12 bf16
            # products of 2*T*in*out per launch) / time against the 2.5 PF dense bf16 peak; the fp32-equivalent rate
            # (the 2 fp32 products the rule asks for) stands beside it.
            roof = {"bound": "mfma", "mfma_dtype": "bf16",
                    "kernel": "x6_kernel<2, MODE_C> (Linear.relprop C-pass, three-way split fp32 operands on bf16 MFMAs)",
                    "achieved": x6c["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": x6c["tflops"] / MFMA_BF16_PEAK_TFLOPS,
                    "fp32_equivalent_tflops": x6c["tflops"] / 6.0,
                    "fp32_equivalent_note": "algorithmic fp32 flops of the C-pass (2 products of 2*T*in*out) / time; the "
                                            f"fp32-MFMA peak this replaces is {MFMA_F32_PEAK_TFLOPS} TF",
                    "traffic": traffic,
                    "traffic_note": (f"bytes per launch, mean of the 4 C-pass shapes of a block; offline rocprofv3 --pmc "
                                     f"FETCH_SIZE / WRITE_SIZE passes (profiles/{traffic_src})") if traffic else None,
                    "launches_timed": x6c["launches"], "avg_launch_us": x6c["avg_us"],
                    "executed_bf16_flops_per_launch_avg": x6c["flops_per_launch"],
                    "zpass": {"kernel": "x6_kernel<2, MODE_Z> (Z-pass from the forward output, 6 bf16 products of "
                                        "2*T*in*out flops; S leaves as bf16 planes)",
                              "achieved": x6z["tflops"], "frac": x6z["tflops"] / MFMA_BF16_PEAK_TFLOPS,
                              "avg_launch_us": x6z["avg_us"]} if x6z else None,
                    "kernels": kernel_table(timer), "kernels_note": kernels_note}
        elif timer.summary("linear_x6_general", 0.0):
            x6g = timer.summary("linear_x6_general", 0.0)
            roof = {"bound": "mfma", "mfma_dtype": "bf16",
                    "kernel": "te_linear_relprop_x6_general_f32 (Linear.relprop of the lrp rule library: four one-sided x6 "
                              "products per rule, x6_kernel<., MODE_Z1> and x6_kernel<., MODE_X>, + the X+ / X- split)",
                    "achieved": x6g["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": x6g["tflops"] / MFMA_BF16_PEAK_TFLOPS, "fp32_equivalent_tflops": x6g["tflops"] / 6.0,
                    "traffic": None, "launches_timed": x6g["launches"], "avg_launch_us": x6g["avg_us"],
                    "executed_bf16_flops_per_launch_avg": x6g["flops_per_launch"],
                    "kernels": kernel_table(timer), "kernels_note": kernels_note}
        elif cp:
            roof = {"bound": "mfma", "mfma_dtype": "f32",
                    "kernel": "linear_k2_kernel<0,false,false> (Linear.relprop C-pass)",
                    "achieved": cp["tflops"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": cp["tflops"] / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_note": (f"bytes per launch, mean of the 4 C-pass shapes of a block; offline rocprofv3 --pmc "
                                     f"FETCH_SIZE / WRITE_SIZE passes (profiles/{traffic_src})") if traffic else None,
                    "launches_timed": cp["launches"], "avg_launch_us": cp["avg_us"],
                    "algorithmic_flops_per_launch_avg": cp["flops_per_launch"],
                    "zpass": {"kernel": "linear_k1_kernel<ZM_FWD> (Z-pass from the forward output, 2*T*in*out FLOP)",
                              "achieved": zp["tflops"], "avg_launch_us": zp["avg_us"]} if zp else None,
                    "kernels": kernel_table(timer), "kernels_note": kernels_note}
        line["roofline"] = roof
        base, ref_maps = None, {}
        if world == 1 and args.cpu_baseline != "off":
            ref_maps, base = cpu_baseline(args, wl)
        line["cpu_baseline"] = base
        if world == 1 and gpu_maps is not None and args.parity != "off":
            try:
                line["parity"] = cpu_baseline_parity(args, wl, gpu_maps, ref_maps)
            except Exception as exc:      # the line must still be printed: the failure is part of it
                line["parity"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(line), flush=True)
    faulthandler.cancel_dump_traceback_later()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
