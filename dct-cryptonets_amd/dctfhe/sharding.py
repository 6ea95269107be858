"""Image sharding for multi-GPU evaluation.

Each image is an independent circuit evaluation under the same keys (the reference loops samples one at a
time inside forward(), homomorphic_eval.py:70), so a batch shards by image: rank r of G takes images
r, r+G, r+2G, ...  Keys and circuit are regenerated per rank from the seed; the only exchange on the data
path is one all_gather of the decrypted-side logits (RCCL on GPUs, gloo in the CPU tests).

Every collective bench.py issues lives here, so that the CPU tests (gloo, world size 2) and the one-rank RCCL smoke test
(tests/test_gpu_rccl_smoke.py: `nccl` backend on cuda:0) run the very calls the 8-GPU job makes."""
import torch
import torch.distributed as dist


def shard_indices(n_images, rank, world):
    return list(range(rank, n_images, world))


def gather_in_image_order(local, world):
    """local: [B_local, F] tensor, same B_local on every rank -> [B_local*world, F] in global image order"""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous())
    return torch.stack(parts, dim=1).reshape(local.shape[0] * world, *local.shape[1:])


def _live(world):
    return world > 1 or (dist.is_available() and dist.is_initialized())


def broadcast_seed(seed32, world, dev):
    """rank 0's 32 key-seed bytes to every rank (keys are regenerated from them on each GPU: no key traffic)"""
    t = torch.tensor(list(seed32), dtype=torch.uint8)
    if _live(world):
        t = t.to(dev)
        dist.broadcast(t, 0)
        t = t.cpu()
    return bytes(t.tolist())


def agree_min(values, world, dev):
    """element-wise minimum over ranks of a list of ints (the pass plan every rank must share)"""
    if not _live(world):
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return [int(v) for v in t.tolist()]


def max_over_ranks(x, world, dev):
    """the slowest rank's elapsed time"""
    if not _live(world):
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_true(flag, world, dev):
    if not _live(world):
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def barrier(world):
    if _live(world):
        dist.barrier()
