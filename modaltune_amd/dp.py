"""Data-parallel pieces of the hot path: slide sharding and the one gradient collective.

Host/torch.distributed logic only, so it is exercised on CPU with gloo (tests/test_dp_cpu.py) and runs unchanged
over RCCL on GPUs ("nccl" backend).  The reference's intent: torch DistributedSampler + DDP mean of the trainable
gradients (utils/base_trainer.py:192-211, 283-286, 483-484).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.distributed as dist


def shard_indices(n_items: int, world: int, rank: int, epoch: int = 0, seed: int = 0, shuffle: bool = True,
                  drop_last: bool = False) -> List[int]:
    """torch.utils.data.DistributedSampler rule: seeded (seed + epoch) permutation, padded by wrap-around to a
    multiple of `world`, rank r takes every world-th index starting at r."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n_items, generator=g).tolist()
    else:
        idx = list(range(n_items))
    if drop_last and n_items % world:
        total = (n_items // world) * world
        idx = idx[:total]
    else:
        total = math.ceil(n_items / world) * world
        pad = total - len(idx)
        if pad:
            idx += (idx * math.ceil(pad / max(1, len(idx))))[:pad]
    return idx[rank:total:world]


def allreduce_sum_(flat: torch.Tensor, group=None) -> int:
    """In-place SUM all-reduce of the flat gradient buffer; returns the world size (the mean is folded into the AdamW
    kernel as grad_mult = 1/world).  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return world


def broadcast_params_(flat: torch.Tensor, src: int = 0, group=None):
    """DDP's constructor broadcast: every rank starts from rank `src`'s trainable parameters."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
