"""Multi-GPU: one process per GPU, samples sharded with no data-path communication, one gather of the
finished maps (RCCL over xGMI on MI355X; gloo in the CPU tests).

Every image / sequence is an independent problem (per-sample semantics), so fwd, bwd and relprop never
communicate; weights are read-only replicas.  The only collective is the final all_gather of the
[n_local, N-1] fp32 maps (25 KB per rank per step at B = 256 over 8 GPUs -- latency-bound, so it is
issued once per sweep, not per step).  The reference has no equivalent (its utils/parallel.py is dead
code).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank).
    Single-process runs (no RANK in the environment) are a no-op returning (0, 1, 0)."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return 0, 1, 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        # "nccl" IS RCCL on ROCm; TE_DIST_BACKEND=gloo lets several ranks share ONE GPU (test rigs: RCCL refuses that)
        backend = os.environ.get("TE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("TE_DEVICE_OVERRIDE", local)))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition [lo, hi) of n_items over `world` ranks (first ranks get the remainder)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_indices(n_items: int, rank: int, world: int) -> range:
    lo, hi = shard_range(n_items, rank, world)
    return range(lo, hi)


def gather_maps(local_maps: torch.Tensor, n_items: int, force_collective: bool = False) -> torch.Tensor:
    """All-gather the per-rank [n_local, M] maps into the full [n_items, M] tensor in global order.
    Ranks may own different counts (block partition); shards are padded to the largest for the
    collective and trimmed afterwards.  force_collective (tests): issue the collective even in a
    one-rank group, so that the device branch (RCCL all_gather_into_tensor) can be executed on a
    single-GPU box."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_maps
    if dist.get_world_size() == 1 and not force_collective:
        return local_maps
    world = dist.get_world_size()
    if local_maps.is_cuda and dist.get_backend() == "gloo":      # gloo has no device all_gather: stage through the host
        return gather_maps(local_maps.cpu(), n_items, force_collective).to(local_maps.device)
    counts = [shard_range(n_items, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in counts)
    M = local_maps.shape[1:]
    pad = torch.zeros((max_n, *M), dtype=local_maps.dtype, device=local_maps.device)
    pad[: local_maps.shape[0]] = local_maps
    out = torch.empty((world, max_n, *M), dtype=local_maps.dtype, device=local_maps.device)
    dist.all_gather_into_tensor(out, pad.contiguous()) if hasattr(dist, "all_gather_into_tensor") and \
        local_maps.is_cuda else dist.all_gather(list(out.unbind(0)), pad.contiguous())
    return torch.cat([out[r, : hi - lo] for r, (lo, hi) in enumerate(counts)], 0)


def synthetic_image(global_index: int, shape=(3, 224, 224), seed: int = 1) -> torch.Tensor:
    """Deterministic synthetic sample keyed by its GLOBAL index, so any shard layout sees the same data."""
    g = torch.Generator().manual_seed(seed * 1_000_003 + global_index)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def pin_rank_to_cores(local_rank: int, ranks_on_node: int) -> int:
    """One process per GPU: give each rank its own slice of the host cores this process may run on (affinity / cgroup aware)
    and size torch's intra-op pool to it, so that eight ranks capturing / enqueueing at once do not fight for the same cores
    (the host side of a step is ~1000 launches; an oversubscribed host shows up as stream gaps on every GPU).  Returns the
    number of cores the rank was given (0 = left alone: single rank, or fewer cores than ranks)."""
    if ranks_on_node <= 1:
        return 0
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return 0
    per = len(cores) // ranks_on_node
    if per < 1:
        return 0
    mine = cores[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return 0
    os.environ["OMP_NUM_THREADS"] = str(per)
    torch.set_num_threads(per)
    return per


def sweep_layout(n_items: int, world: int, global_batch: int):
    """The 50 000-image sweep of BASELINE.json configs[4] (reference: baselines/ViT/generate_visualizations.py:27-100 over the
    ImageNet validation set) on `world` ranks: rank r owns the contiguous block shard_range(n_items, r, world) and walks it in
    batches of global_batch / world.  Returns [(lo, hi, [(b_lo, b_hi), ...]) per rank] in GLOBAL sample indices."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} does not divide over {world} ranks")
    per = global_batch // world
    out = []
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        out.append((lo, hi, [(s, min(s + per, hi)) for s in range(lo, hi, per)]))
    return out


def synthetic_image_on(global_index: int, device, shape=(3, 224, 224), seed: int = 1) -> torch.Tensor:
    """synthetic_image generated ON `device` (a 50 000-image sweep cannot afford 1 ms of host randn per image): keyed by the
    GLOBAL index alone, so every shard layout on the same kind of device sees the same image."""
    g = torch.Generator(device=device).manual_seed(seed * 1_000_003 + global_index)
    return torch.randn(shape, generator=g, dtype=torch.float32, device=device)
