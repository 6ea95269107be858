"""Image sharding for multi-GPU evaluation.

Each image is an independent circuit evaluation under the same keys (the reference loops samples one at a
time inside forward(), homomorphic_eval.py:70), so a batch shards by image: rank r of G takes images
r, r+G, r+2G, ...  Keys and circuit are regenerated per rank from the seed; the only exchange is one
all_gather of the decrypted-side logits (RCCL on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_indices(n_images, rank, world):
    return list(range(rank, n_images, world))


def gather_in_image_order(local, world):
    """local: [B_local, F] tensor, same B_local on every rank -> [B_local*world, F] in global image order"""
    if world == 1:
        return local
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local.contiguous())
    return torch.stack(parts, dim=1).reshape(local.shape[0] * world, *local.shape[1:])
