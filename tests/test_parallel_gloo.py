"""Multi-process path on CPU: world_size-2 `gloo` run of the shard -> explain -> gather pipeline
(transformer_explainability_amd/parallel.py).  The per-rank "explain" step is the real host path (our
ViT + generators) with the device ops routed to the oracle, so the test checks that any shard layout
reproduces the single-process maps bit for bit and in global order."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _explain(indices):
    from oracle_backend import oracle_ops
    from transformer_explainability_amd import parallel, vit
    from transformer_explainability_amd.generators import LRP
    torch.manual_seed(0)
    model = vit.VisionTransformer(img_size=32, patch_size=8, embed_dim=32, depth=2, num_heads=2, num_classes=5,
                                  qkv_bias=True).eval()
    x = torch.stack([parallel.synthetic_image(i, (3, 32, 32)) for i in indices])
    with oracle_ops():
        # one sample at a time so that the forward GEMM shapes (and rounding) do not depend on the shard size
        return torch.cat([LRP(model).generate_LRP(x[i:i + 1], start_layer=0).detach() for i in range(len(indices))])


def _worker(rank, world, port, n_items, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from transformer_explainability_amd import parallel
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    idx = list(parallel.shard_indices(n_items, r, w))
    local = _explain(idx)
    full = parallel.gather_maps(local, n_items)
    torch.save({"full": full, "idx": idx}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_shard_ranges_cover_everything():
    from transformer_explainability_amd import parallel
    for n in (1, 5, 8, 50000):
        for world in (1, 2, 3, 8):
            ranges = [parallel.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
                assert a1 == b0 and a1 >= a0
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_world2_gloo_matches_single_process(tmp_path):
    n_items = 5      # odd on purpose: ranks own 3 and 2 samples (ragged shards)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    single = _explain(list(range(n_items)))
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_items, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        got = torch.load(os.path.join(str(tmp_path), f"rank{rank}.pt"))
        assert got["full"].shape == single.shape
        assert torch.equal(got["full"], single), f"rank {rank}: gathered maps differ from the single-process run"
    assert torch.load(os.path.join(str(tmp_path), "rank0.pt"))["idx"] == [0, 1, 2]
    assert torch.load(os.path.join(str(tmp_path), "rank1.pt"))["idx"] == [3, 4]


@pytest.mark.timeout(600)
def test_world8_gloo_matches_single_process(tmp_path):
    """The node-level layout the driver will run (one rank per GPU, eight of them), on eight CPU ranks over gloo: ragged
    block partition of 19 samples (3, 3, 3, 2, 2, 2, 2, 2), no communication while explaining, one all_gather -- every rank
    ends up with the single-process maps, bit for bit and in global order.  (No 8-GPU node was available in rounds 1-4:
    this is the only execution of the 8-rank path.)"""
    n_items, world = 19, 8
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    single = _explain(list(range(n_items)))
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_items, str(tmp_path)), nprocs=world, join=True)
    seen = []
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), f"rank{rank}.pt"))
        assert torch.equal(got["full"], single), f"rank {rank}: gathered maps differ from the single-process run"
        seen += got["idx"]
    assert seen == list(range(n_items))


# ------------------------------------------------------------------------------------------ the sweep layout
_COLLECTIVES = ("all_gather", "all_gather_into_tensor", "all_gather_object", "all_reduce", "broadcast", "reduce",
                "reduce_scatter", "reduce_scatter_tensor", "all_to_all", "all_to_all_single", "gather", "scatter", "send",
                "recv", "isend", "irecv", "broadcast_object_list")


def _sweep_worker(rank, world, port, n_items, out_dir):
    """BASELINE.json configs[4]'s layout on two CPU ranks: shard the dataset (block partition), explain + store the own
    shard (sweep.py), gather the maps -- with every torch.distributed collective counted."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import torch.distributed as dist
    from oracle_backend import oracle_ops
    from test_sweep import ToyImages, _generators
    from transformer_explainability_amd import parallel
    from transformer_explainability_amd.sweep import ResultsStore, SaliencySweep, shard_batches
    r, w, _ = parallel.init_distributed(backend="gloo")
    calls = {}
    for name in _COLLECTIVES:
        fn = getattr(dist, name, None)
        if fn is None:
            continue

        def counted(*a, _fn=fn, _name=name, **k):
            calls[_name] = calls.get(_name, 0) + 1
            return _fn(*a, **k)
        setattr(dist, name, counted)
    ds = ToyImages(n_items)
    dev = torch.device("cpu")
    with oracle_ops():
        lrp, orig, base = _generators(dev)
        sw = SaliencySweep("transformer_attribution", lrp=lrp, orig_lrp=orig, baselines=base, device=dev)
        batches, lo, hi = shard_batches(ds, 2, r, w)
        with ResultsStore(out_dir, len(ds), (3, 32, 32), (1, 32, 32), lo, hi, backend="npy") as store:
            sw.run(batches, store, r, w)
        local = torch.as_tensor(store_vis(out_dir, lo, hi))
    in_data_path = dict(calls)                      # collectives issued while explaining / storing: must be none
    full = parallel.gather_maps(local.reshape(hi - lo, -1), n_items)
    torch.save({"full": full, "lo": lo, "hi": hi, "data_path_calls": in_data_path, "calls": dict(calls)},
               os.path.join(out_dir, f"sweep_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def store_vis(out_dir, lo, hi):
    import numpy as np
    return np.load(os.path.join(out_dir, "results", f"vis.{lo:09d}-{hi:09d}.npy"))


@pytest.mark.timeout(300)
def test_world2_gloo_sweep_layout_and_single_collective(tmp_path):
    """The 50k-image sweep's layout (BASELINE.json configs[4]) at toy size on two gloo ranks: ragged block
    shards, per-rank result stores, ONE collective in the whole run (the all_gather of the finished maps) and none in the
    data path; the gathered maps are in global order and equal the stores read back through ImagenetResults."""
    n_items = 7
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    port = _free_port()
    mp.spawn(_sweep_worker, args=(2, port, n_items, str(tmp_path)), nprocs=2, join=True)
    from transformer_explainability_amd.sweep import ImagenetResults
    res = ImagenetResults(str(tmp_path))
    assert len(res) == n_items
    stored = torch.stack([res[i][1].reshape(-1) for i in range(n_items)])
    got = [torch.load(os.path.join(str(tmp_path), f"sweep_rank{r}.pt")) for r in range(2)]
    assert (got[0]["lo"], got[0]["hi"], got[1]["lo"], got[1]["hi"]) == (0, 4, 4, 7)
    for r in range(2):
        assert got[r]["data_path_calls"] == {}, got[r]["data_path_calls"]
        total = sum(got[r]["calls"].values())
        assert total == 1 and set(got[r]["calls"]) <= {"all_gather", "all_gather_into_tensor"}, got[r]["calls"]
        assert torch.equal(got[r]["full"], stored), f"rank {r}: gathered maps are not the stores in global order"


def test_sweep50k_layout_on_8_ranks():
    """BASELINE.json configs[4]: 50 000 images, global batch 256 over 8 ranks -- every rank owns one
    contiguous block of the global index space, walks it in batches of 32, the blocks tile the sweep exactly, and an image
    depends on its GLOBAL index only (any shard layout sees the same data)."""
    import bench
    from transformer_explainability_amd import parallel
    n, world, gb = 50_000, 8, 256
    lay = parallel.sweep_layout(n, world, gb)
    assert len(lay) == world and lay[0][0] == 0 and lay[-1][1] == n
    seen = 0
    for r, (lo, hi, batches) in enumerate(lay):
        assert (lo, hi) == parallel.shard_range(n, r, world) and lo == seen
        assert batches[0][0] == lo and batches[-1][1] == hi
        assert all(b1 - b0 == gb // world for b0, b1 in batches[:-1]) and 0 < batches[-1][1] - batches[-1][0] <= gb // world
        assert all(a[1] == b[0] for a, b in zip(batches, batches[1:]))
        seen = hi
    assert seen == n
    with pytest.raises(ValueError):
        parallel.sweep_layout(n, 3, gb)                       # 256 does not divide over 3 ranks
    # the same global index gives the same image whoever generates it, and different indices differ
    a = parallel.synthetic_image_on(31_337, "cpu", (3, 8, 8))
    assert torch.equal(a, parallel.synthetic_image_on(31_337, "cpu", (3, 8, 8)))
    assert not torch.equal(a, parallel.synthetic_image_on(31_338, "cpu", (3, 8, 8)))
    # bench.py: the sweep's default run is the whole sweep in global batches; per-rank batch = 256 / world
    args = bench.parse_args(["--config", "sweep50k"])
    assert args.steps == -(-bench.SWEEP_IMAGES // bench.SWEEP_GLOBAL_BATCH) == 196
    assert bench.parse_args([]).steps == 5
    # a single rank is left alone by the core pinning; N ranks get disjoint slices
    assert parallel.pin_rank_to_cores(0, 1) == 0
