"""The N > 1 path of bench.py on its REAL backend, at the only world size a one-GPU box allows: `init_process_group("nccl")` -- RCCL on
ROCm -- with one rank on cuda:0, then the exact collectives of a multi-GPU bench run on GPU tensors (dctfhe/sharding.py: seed
broadcast, the all_reduce(MIN) that agrees the pass plan, the all_reduce(MAX) over elapsed times, the all_reduce(MIN) of the
bit-exactness flag, barrier, and the all_gather of logits in image order).  Round 2 had only ever run them over gloo; the first RCCL
initialisation must not be the driver's 8-GPU job.  Runs in a fresh child process (a process group per process; a failure in RCCL
must not take the test runner down) that exits non-zero on any failure."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "dct-cryptonets_amd"))
import torch
import torch.distributed as dist
from dctfhe.sharding import agree_min, all_true, barrier, broadcast_seed, gather_in_image_order, max_over_ranks, shard_indices
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)            # bench.py: dist.init_process_group("nccl", device_id=cdev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
seed = bytes(range(32))
assert broadcast_seed(seed, 1, dev) == seed
assert agree_min([5, 20], 1, dev) == [5, 20]
assert abs(max_over_ranks(12.5, 1, dev) - 12.5) < 1e-12
assert all_true(True, 1, dev) and not all_true(False, 1, dev)
barrier(1)
logits = torch.arange(30, dtype=torch.float32).reshape(3, 10).to(dev)
got = gather_in_image_order(logits, 1)
assert got.is_cuda and torch.equal(got.cpu(), torch.arange(30, dtype=torch.float32).reshape(3, 10))
assert shard_indices(8, 0, 1) == list(range(8))
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL one-rank smoke OK")
"""


def test_one_rank_rccl_runs_the_collectives_of_bench():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL one-rank smoke OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
