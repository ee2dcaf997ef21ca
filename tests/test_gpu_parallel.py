"""The RCCL branch of parallel.gather_maps on the one GPU this box has: a one-rank `nccl` process
group (nccl IS RCCL on ROCm) and the collective forced (`force_collective=True` bypasses the world-1 early return), so
`all_gather_into_tensor` runs on DEVICE tensors -- the call the 8-GPU sweep issues once after its last step
(baselines/ViT/generate_visualizations.py:27-100 is the reference's 50k-image path; it has no collective of its own).
Runs in a child process: a process group must not outlive the test inside pytest's interpreter."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.environ["TE_ROOT"])
import torch.distributed as dist
from transformer_explainability_amd import parallel
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
calls = []
real = dist.all_gather_into_tensor
def counted(out, inp, *a, **k):
    assert out.is_cuda and inp.is_cuda, "the RCCL branch gathers device tensors"
    calls.append((tuple(out.shape), tuple(inp.shape)))
    return real(out, inp, *a, **k)
dist.all_gather_into_tensor = counted
g = torch.Generator().manual_seed(5)
maps = torch.randn((37, 196), generator=g).cuda()
assert parallel.gather_maps(maps, 37) is maps and not calls            # world 1: no collective by default
full = parallel.gather_maps(maps, 37, force_collective=True)
torch.cuda.synchronize()
assert calls == [((1, 37, 196), (37, 196))], calls
assert full.is_cuda and full.shape == maps.shape and torch.equal(full, maps)
dist.barrier()
dist.destroy_process_group()
print("RCCL_GATHER_OK")
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rccl_all_gather_branch_on_one_rank():
    env = dict(os.environ, TE_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=280)
    assert r.returncode == 0 and "RCCL_GATHER_OK" in r.stdout, r.stdout[-3000:]
