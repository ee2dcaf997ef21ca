#!/usr/bin/env python
"""bench.py -- relevance maps/sec for ViT-B/16 224^2 at batch 64 per GPU (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic images already resident in HBM:
stock PyTorch-ROCm forward + attention-gradient backward, then the HIP relprop rules, the
gradient x relevance head-mean and the rollout chain (LRP.generate_LRP, method
"transformer_attribution", start_layer 1 as baselines/ViT/imagenet_seg_eval.py:196 of the reference calls
it), fp32 end to end.  All 12 blocks are propagated; the only shortcuts are exact or rounding-level:
the inhibitor half is dead at alpha = 1, the last block's dense rules run on the class-token row they are confined to,
and the Z-pass of Linear.relprop reuses the forward output X W^T + b instead of recomputing it.

Weak scaling: every rank runs the same batch size on its own images; the only communication is one
all_gather (RCCL) of the finished [B,196] maps of the last step, inside the timed region.

The JSON line also carries
  roofline      the dominant kernel (Linear.relprop C-pass, fp32 MFMA): algorithmic FLOPs per launch over
                the launch duration measured with HIP events on the launch stream, during the timed steps
  cpu_baseline  the same path on the host cores of this box: stock PyTorch CPU fwd/bwd + the CPU oracle's
                relprop (kind "port"), or the reference itself when /root/reference exists (kind
                "reference"); rank 0, N = 1 only, bounded to a few maps
"""
from __future__ import annotations

import argparse
import contextlib
import faulthandler
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz
_T0 = time.perf_counter()


def log(msg):
    """Progress to stderr (the JSON line is the only thing on stdout)."""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may actually run on (cgroup / affinity aware), capped: torch CPU GEMMs stop scaling
    long before a 100+ core host is filled and oversubscribing a container quota is catastrophic."""
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
    return max(1, min(n, 32))


class KernelTimer:
    """Brackets single kernel launches with HIP events recorded on torch's current stream (the stream
    the C ABI launches on).  Events are resolved after the timed region's final synchronise."""

    def __init__(self):
        self.records = []          # (name, flops, start_event, end_event)
        self.enabled = False

    @contextlib.contextmanager
    def __call__(self, name, flops):
        if not self.enabled:
            yield
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        yield
        e.record()
        self.records.append((name, flops, s, e))

    def summary(self, name):
        rows = [(f, s.elapsed_time(e) * 1e-3) for n, f, s, e in self.records if n == name]
        if not rows:
            return None
        flops = sum(f for f, _ in rows)
        secs = sum(t for _, t in rows)
        return {"launches": len(rows), "avg_us": secs / len(rows) * 1e6, "tflops": flops / secs / 1e12,
                "flops_per_launch_avg": flops / len(rows)}


def cpu_baseline(args, model_cpu_state, n_maps=3):
    """Time the CPU path on this box's host cores: one sample at a time (the reference is batch-1)."""
    from oracle import ref_harness as rh
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads (os.cpu_count() = {os.cpu_count()})")
    x = torch.stack([synthetic_image(i) for i in range(n_maps + 1)])
    if rh.reference_available() and args.cpu_baseline != "port":
        mods = rh.load_reference_vit()
        model = mods["ViT_LRP"].vit_base_patch16_224(pretrained=False).eval()
        model.load_state_dict(model_cpu_state)
        gen = mods["gen"].LRP(model)
        times = []
        for i in range(n_maps + 1):
            t0 = time.perf_counter()
            gen.generate_LRP(x[i:i + 1], method="transformer_attribution", start_layer=args.start_layer)
            times.append(time.perf_counter() - t0)
            log(f"cpu_baseline(reference) map {i}: {times[-1]:.2f} s")
        kind = "reference"
    else:
        from transformer_explainability_amd import vit
        from oracle import relprop_oracle as O
        from oracle.model_cache import vit_cache_from_model
        model = vit.vit_base_patch16_224().eval()
        model.load_state_dict(model_cpu_state)
        times = []
        for i in range(n_maps + 1):
            t0 = time.perf_counter()
            out = model(x[i:i + 1])
            oh = torch.zeros_like(out)
            oh.scatter_(1, out.argmax(-1, keepdim=True), 1.0)
            grads = torch.autograd.grad((oh * out).sum(), [b.attn.get_attn() for b in model.blocks])
            for b, g in zip(model.blocks, grads):
                b.attn.save_attn_gradients(g)
            O.vit_relprop(oh, vit_cache_from_model(model), num_heads=12, start_layer=args.start_layer)
            times.append(time.perf_counter() - t0)
            log(f"cpu_baseline(port) map {i}: {times[-1]:.2f} s")
        kind = "port"
    times = sorted(times[1:])
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "maps/s", "cores": cores, "kind": kind,
            "sample": f"{n_maps} ViT-B/16 224^2 maps, batch 1, after 1 warm-up; median {med:.3f} s/map, "
                      f"torch CPU fp32 with {cores} threads"}


def synthetic_image(global_index, shape=(3, 224, 224), seed=1):
    g = torch.Generator().manual_seed(seed * 1_000_003 + global_index)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--start-layer", type=int, default=1)
    ap.add_argument("--cpu-baseline", choices=["auto", "port", "off"], default="auto")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="micro-batches in flight on separate HIP streams")
    ap.add_argument("--graph", choices=["on", "off"], default="on",
                    help="replay each step from a HIP graph (one eager step inside the timed region carries the "
                         "per-kernel HIP events of the roofline)")
    ap.add_argument("--tuned-gemms", choices=["on", "off", "tune"], default="on",
                    help="stock fp32 GEMMs of forward/backward selected by PyTorch TunableOp from the committed results "
                         "file (on), PyTorch's default heuristic (off), or tune now and write gpurun_out/ (tune)")
    ap.add_argument("--prune", choices=["on", "off"], default="off",
                    help="skip the relprop rules and attention gradients of the blocks below --start-layer (their "
                         "attn_cam never reaches the map); off = every block, as the reference does")
    ap.add_argument("--overlap-backward", choices=["on", "off"], default="off",
                    help="run the relprop rules on a side stream beside the attention-gradient backward pass (they are "
                         "independent until the head-mean / rollout tail); the roofline probe step stays serial")
    ap.add_argument("--inflight", type=int, default=1,
                    help="consecutive steps (batches) in flight, each on its own HIP stream: the forward/backward of "
                         "step k+1 overlaps the relprop of step k")
    args = ap.parse_args()
    faulthandler.enable()
    faulthandler.dump_traceback_later(600, repeat=True, file=sys.stderr)   # a stuck run leaves a stack trace

    import __graft_entry__
    import transformer_explainability_amd as te
    from transformer_explainability_amd import ops, parallel, vit
    from transformer_explainability_amd.generators import LRP

    rank, world, local = parallel.init_distributed()
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        torch.distributed.barrier()
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    te._lib.require_device()
    dev = torch.device("cuda", int(os.environ.get("TE_DEVICE_OVERRIDE", local)))   # (override: test rigs with one GPU)
    torch.cuda.set_device(dev)
    tuned = False
    if args.tuned_gemms == "on":
        tuned = te.enable_tuned_gemms()
    elif args.tuned_gemms == "tune":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        tuned = te.enable_tuned_gemms(os.path.join(ROOT, "gpurun_out", f"tunableop_gfx950_rank{rank}.csv"), tune=True)

    torch.manual_seed(0)
    model = vit.vit_base_patch16_224().eval()
    with torch.no_grad():          # non-trivial biases / LayerNorm scales so every path is exercised
        for n_, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.02 * torch.randn_like(p))
    cpu_state = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model.to(dev)
    B = args.batch
    x = torch.stack([synthetic_image(rank * B + i) for i in range(B)]).to(dev)
    lrp = LRP(model, streams=args.streams, overlap_backward=(args.overlap_backward == "on"),
              prune=(args.prune == "on"))
    log(f"rank {rank}/{world}: model + {B} images resident on {dev}")

    timer = KernelTimer()
    if not args.no_roofline:
        ops.KERNEL_TIMER = timer

    lanes = [torch.cuda.Stream(device=dev) for _ in range(args.inflight)] if args.inflight > 1 else None
    counter = [0]

    graphed = None
    # (multi-rank runs stay eager: RCCL's watchdog thread may touch the HIP runtime while a capture is open, and the
    #  step is GPU-bound either way -- replay only frees the host)
    if args.graph == "on" and args.inflight == 1 and args.streams == 1 and world == 1:
        from transformer_explainability_amd.generators import GraphedLRP
        try:
            graphed = GraphedLRP(lrp, x, method="transformer_attribution", start_layer=args.start_layer)
            log("HIP graph of one step captured")
        except Exception as exc:      # capture is an optimisation of the host side only: fall back to eager launches
            graphed = None
            torch.cuda.synchronize()
            log(f"HIP graph capture failed ({type(exc).__name__}: {exc}); running eagerly")

    def step(eager=False):
        if graphed is not None and not eager:
            return graphed(x)
        if eager and lrp.overlap_backward:
            # the probe step times single launches with HIP events: run it serially so that a kernel's duration is
            # its own (with the overlap, kernels of the two streams share the CUs)
            lrp.overlap_backward = False
            try:
                return lrp.generate_LRP(x, method="transformer_attribution", start_layer=args.start_layer)
            finally:
                lrp.overlap_backward = True
        if lanes is None:
            return lrp.generate_LRP(x, method="transformer_attribution", start_layer=args.start_layer)
        # every tensor of a step is allocated, produced and consumed on that step's stream
        lane = lanes[counter[0] % len(lanes)]
        counter[0] += 1
        lane.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(lane):
            return lrp.generate_LRP(x, method="transformer_attribution", start_layer=args.start_layer)

    def join():
        if lanes is not None:
            for lane in lanes:
                torch.cuda.current_stream(dev).wait_stream(lane)

    for w in range(args.warmup):
        maps = step()
        torch.cuda.synchronize()
        log(f"warmup step {w} done")
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        # with graph replay, ONE step of the timed region runs eagerly with a HIP-event pair around every launch of
        # the Linear.relprop kernels (events cannot be recorded inside a replayed graph); the kernels and their
        # durations are the same in both modes (single stream, back to back)
        probe = (graphed is None) or (k == args.steps - 1)
        timer.enabled = probe and not args.no_roofline
        maps = step(eager=probe and graphed is not None and not args.no_roofline)
    host_enqueue = time.perf_counter() - t0      # host time to enqueue all steps (GPU still running)
    join()
    gathered = parallel.gather_maps(maps, world * B)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f"timed {args.steps} steps: {elapsed:.3f} s (host enqueue {host_enqueue:.3f} s)")
    timer.enabled = False
    ops.KERNEL_TIMER = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert gathered.shape == (world * B, 196) and torch.isfinite(gathered).all()

    if rank == 0:
        value = world * B * args.steps / elapsed
        line = {
            "metric": f"relevance maps/sec (ViT-B/16 224^2, batch {B} per GPU, generate_LRP transformer_attribution)",
            "value": value, "unit": "maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ViT-B/16 224^2 batch {B} per GPU on {world}xMI355X: stock fwd + attn-grad bwd + fp32 "
                                   "relprop/head-mean/rollout HIP kernels (BASELINE.json configs[1], sharded by sample)",
                       "batch_per_gpu": B, "global_batch": world * B, "tokens": 197, "blocks": 12,
                       "start_layer": args.start_layer, "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3,
                       "streams": args.streams, "steps_in_flight": args.inflight,
                       "relprop_beside_backward": args.overlap_backward == "on",
                       "blocks_below_start_layer_pruned": args.prune == "on",
                       "stock_gemm_selection": ("PyTorch TunableOp, committed results file" if tuned and
                                                args.tuned_gemms == "on" else
                                                "PyTorch TunableOp, tuned in this run" if tuned else "PyTorch default"),
                       "hip_graph": graphed is not None, "parallelism": f"dp{world} (independent samples, one "
                                                                      f"all_gather of the maps)"},
        }
        roof = None
        traffic = None
        try:   # HBM-side bytes per C-pass launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE,
               # MI355X guide's gfx950 correction); only valid for the B = 64 workload they were collected on
            if B == 64:
                with open(os.path.join(ROOT, "profiles", "r01_linear_traffic_pmc.json")) as f:
                    tr = json.load(f)
                cps = [v["traffic_bytes"] for k, v in tr.items() if k.endswith(".cpass")]
                traffic = sum(cps) / len(cps)
        except (OSError, ValueError, KeyError):
            traffic = None
        cp = timer.summary("linear_cpass")
        zp = timer.summary("linear_zpass_fwd") or timer.summary("linear_zpass")
        if cp:
            roof = {"bound": "mfma", "kernel": "linear_k2_kernel<0,false,false> (Linear.relprop C-pass)",
                    "achieved": cp["tflops"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": cp["tflops"] / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_note": "bytes per launch, mean of the 4 C-pass shapes of a block; offline rocprofv3 --pmc "
                                    "FETCH_SIZE / WRITE_SIZE passes (profiles/r01_linear_traffic_pmc.json)",
                    "launches_timed": cp["launches"], "avg_launch_us": cp["avg_us"],
                    "algorithmic_flops_per_launch_avg": cp["flops_per_launch_avg"],
                    "zpass": {"kernel": "linear_k1_kernel<ZM_FWD> (Z-pass from the forward output, 2*T*in*out FLOP)",
                              "achieved": zp["tflops"], "avg_launch_us": zp["avg_us"]} if zp else None}
        line["roofline"] = roof
        base = None
        if world == 1 and args.cpu_baseline != "off":
            base = cpu_baseline(args, cpu_state)
        line["cpu_baseline"] = base
        print(json.dumps(line), flush=True)
    faulthandler.cancel_dump_traceback_later()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
