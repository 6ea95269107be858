"""N > 1 path on CPU: world_size-2 gloo run of the shard -> evaluate -> gather flow.
The per-image evaluation is replaced by a deterministic function of the image so the order is checkable."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_eval(images):
    return torch.from_numpy(np.stack([[im.sum() % 251, im.max(), im.min(), im[0, 0, 0]] for im in images]).astype(np.float32))


def _worker(rank, world, port, n_images, out_path):
    sys.path.insert(0, os.path.join(ROOT, "dct-cryptonets_amd"))
    from dctfhe.sharding import gather_in_image_order, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    imgs = np.random.default_rng(42).integers(0, 256, (n_images, 8, 8, 3))
    idx = shard_indices(n_images, rank, world)
    local = _fake_eval(imgs[idx])
    dist.barrier()
    full = gather_in_image_order(local, world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)           # the max-over-ranks timing reduction of bench.py
    if rank == 0:
        torch.save({"full": full, "tmax": t}, out_path)
    dist.destroy_process_group()


def test_shard_gather_matches_serial(tmp_path):
    world, n_images = 2, 6
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(world, _free_port(), n_images, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    imgs = np.random.default_rng(42).integers(0, 256, (n_images, 8, 8, 3))
    assert torch.equal(got["full"], _fake_eval(imgs))
    assert got["tmax"].item() == 2.0


def _worker_helpers(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "dct-cryptonets_amd"))
    from dctfhe.sharding import agree_min, all_true, barrier, broadcast_seed, max_over_ranks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    seed = broadcast_seed(bytes([rank + 1] * 32), world, dev)              # every rank ends with rank 0's bytes
    plan = agree_min([5 - rank, 20 + rank], world, dev)                    # the tightest plan wins
    tmax = max_over_ranks(10.0 + rank, world, dev)
    ok_all, ok_one = all_true(True, world, dev), all_true(rank == 0, world, dev)
    barrier(world)
    if rank == 1:
        torch.save({"seed": list(seed), "plan": plan, "tmax": tmax, "ok_all": ok_all, "ok_one": ok_one}, out_path)
    dist.destroy_process_group()


def test_bench_collectives_world_size_two(tmp_path):
    """the helpers bench.py calls for its seed broadcast, pass plan, max-over-ranks timing and exactness flag (dctfhe/sharding.py), seen
    from rank 1 of a two-rank gloo group; tests/test_gpu_rccl_smoke.py runs the same calls on RCCL"""
    out = str(tmp_path / "helpers.pt")
    mp.spawn(_worker_helpers, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    assert got["seed"] == [1] * 32 and got["plan"] == [4, 20] and got["tmax"] == 11.0 and got["ok_all"] is True and got["ok_one"] is False


def test_shard_indices_partition():
    sys.path.insert(0, os.path.join(ROOT, "dct-cryptonets_amd"))
    from dctfhe.sharding import shard_indices
    for world in (1, 2, 4, 8):
        allidx = sorted(i for r in range(world) for i in shard_indices(64, r, world))
        assert allidx == list(range(64))
        assert all(len(shard_indices(64, r, world)) == 64 // world for r in range(world))


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run starts its own two ranks as child processes;
    --launch-check keeps the ranks on CPU/gloo: shard -> all_gather in image order -> one JSON line from rank 0."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch-per-gpu", "3", "--launch-check"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res == {"launch_check": True, "n_gpus": 2, "gather_in_image_order": True}


def test_bench_budget_plan():
    sys.path.insert(0, ROOT)
    import bench
    # everything fits
    assert bench.plan_passes(5, 20, 1, 0, 10.0, 1000.0) == (5, 20)
    # the timed steps take priority over the remaining warm-up
    assert bench.plan_passes(5, 20, 1, 0, 15.0, 310.0) == (1, 20)
    assert bench.plan_passes(5, 20, 1, 0, 15.0, 340.0) == (3, 20)
    assert bench.plan_passes(5, 20, 1, 0, 15.0, 100.0) == (1, 6)
    # never fewer than one timed step, even with the budget already gone
    assert bench.plan_passes(5, 20, 1, 0, 15.0, -3.0) == (1, 1)
    # --budget-s 0: no cap
    assert bench.plan_passes(2, 7, 1, 0, 15.0, float("inf")) == (2, 7) and bench.plan_passes(0, 1, 0, 1, 60.0, float("inf")) == (0, 1)
    # --warmup 0: planned from the first timed step
    assert bench.plan_passes(0, 20, 0, 1, 15.0, 100.0) == (0, 7)


def test_bench_first_pass_rule():
    """the first, bracketed pass becomes the timed step exactly when another pass does not fit (config #4 with 8 images per
    GPU takes ~5 minutes per pass; a warm-up pass plus a timed pass blew the driver's 600 s)"""
    sys.path.insert(0, ROOT)
    import bench
    assert not bench.first_pass_is_the_step(12.3, 370.0, 5, 20)              # headline config: 25 more passes fit
    assert bench.first_pass_is_the_step(285.0, 65.0, 5, 20)                  # r18_3_32 x 8 images: nothing else fits
    assert bench.first_pass_is_the_step(100.0, 110.0, 1, 3)                  # one more pass would fit only without the 15 % margin
    assert not bench.first_pass_is_the_step(100.0, 116.0, 1, 3)
    assert bench.first_pass_is_the_step(1.0, 1e9, 0, 1)                      # --warmup 0 --steps 1: that pass is what was asked for
    assert bench.first_pass_is_the_step(1.0, 1e9, 5, 20, interrupted=True)
    assert not bench.first_pass_is_the_step(50.0, float("inf"), 0, 3)        # --budget-s 0
