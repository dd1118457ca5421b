"""Data-parallel pieces of the hot path: slide sharding and the one gradient collective.

Host/torch.distributed logic only, so it is exercised on CPU with gloo (tests/test_dp_cpu.py) and runs unchanged
over RCCL on GPUs ("nccl" backend).  The reference's intent: torch DistributedSampler + DDP mean of the trainable
gradients (utils/base_trainer.py:192-211, 283-286, 483-484).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence

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


def single_rank_rehearsal() -> bool:
    """MT_DP_REHEARSE=1: with an initialised process group of ONE rank, issue every collective of the data-parallel step anyway (a
    one-rank RCCL communicator runs them as copies): the only way to execute the `nccl` branches -- asynchronous all-reduce per
    bucket, reduce-scatter / all-gather of the sharded bucket, the flag MAX, the constructor broadcast -- on a one-GPU box."""
    return os.environ.get("MT_DP_REHEARSE") == "1" and dist.is_available() and dist.is_initialized()


def allreduce_sum_(flat: torch.Tensor, group=None) -> int:
    """In-place SUM all-reduce of the flat gradient buffer; returns the world size (the mean is folded into the AdamW
    kernel as grad_mult = 1/world).  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world > 1 or single_rank_rehearsal():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return world


def broadcast_params_(flat: torch.Tensor, src: int = 0, group=None):
    """DDP's constructor broadcast: every rank starts from rank `src`'s trainable parameters."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or single_rank_rehearsal()):
        dist.broadcast(flat, src=src, group=group)


# ------------------------------------------------------------------------------------------------
# Bucketed gradient all-reduce, started while the backward still runs (the reference wraps the model in
# DistributedDataParallel, whose reducer does the same per 25 MB bucket: utils/base_trainer.py:205-211).
#
# The backward of the hot path finalises parameter gradients in a fixed order: fusion head and interaction block 2
# first, then block 1, block 0, and last the gene encoder / gene_pe / task tokens (the gene tokens feed every block, so
# their gradient closes only at the very end).  The flat gradient buffer is in state_dict order, so each of these
# stages is a handful of contiguous ranges.
# ------------------------------------------------------------------------------------------------
def _merge_ranges(slots: Sequence[tuple]) -> List[tuple]:
    """[(offset, numel)] of 16-byte aligned slots -> maximal contiguous (offset, length) ranges."""
    out: List[list] = []
    for off, n in sorted(slots):
        n_al = (n + 3) // 4 * 4
        if out and out[-1][0] + out[-1][1] == off:
            out[-1][1] += n_al
        else:
            out.append([off, n_al])
    return [(a, b) for a, b in out]


def grad_buckets(slots: dict, n_interactions: int, n_flat: int) -> List[List[tuple]]:
    """Bucket b (b = 0 .. n_interactions) = ranges of the flat gradient buffer that are final when the backward has
    left interaction block n_interactions - 1 - b; the last bucket (everything else: gene encoder, gene_pe, task /
    clinical tokens) closes with the backward itself.  `slots`: ParamStore.slots (key -> (offset, numel, shape))."""
    stage_of = {}
    last = n_interactions
    for k in slots:
        st = last
        if k.startswith("interactions.") or k.startswith("prompt_selfattention."):
            st = n_interactions - 1 - int(k.split(".")[1])
        elif k.startswith("final_norm.") or k.startswith("final_project."):
            st = 0
        stage_of[k] = st
    buckets = []
    for b in range(last + 1):
        buckets.append(_merge_ranges([(slots[k][0], slots[k][1]) for k in slots if stage_of[k] == b]))
    covered = sum(n for bk in buckets for _, n in bk)
    if covered != n_flat:
        raise AssertionError(f"gradient buckets cover {covered} of {n_flat} elements")
    return buckets


class GradReducer:
    """start(b): launch the collective of bucket b's ranges (asynchronous: RCCL runs it on its own stream behind everything
    enqueued so far on the current stream); wait(): make the current stream wait for every collective in flight.  These are
    the `mt_grad_allreduce_start / _wait` of SURVEY §8b; they live above the C ABI because the communicator belongs to
    torch.distributed ("nccl" = RCCL over xGMI on the GPUs, gloo in the rehearsals).

    Buckets 0 .. last-1 are SUM all-reduced.  The LAST bucket (gene encoder + token parameters: 3 MB with toy pathways, 100 MB
    with the reference's 331) closes with the backward itself, so nothing can hide it; with `flat_param` given it is SHARDED
    instead (reduce-scatter + all-gather as separate exchanges): reduce-scatter of the gradients -> every rank runs
    AdamW on its 1/W of the bucket (`adam_pieces`) -> all-gather of the updated PARAMETERS (`start_param_gather`, waited for at
    the top of the next step by `wait_params`, i.e. under the next slide's input staging).  What is exposed in front of AdamW is
    the reduce-scatter alone -- half the bytes of the all-reduce -- and the optimiser touches 1/W of the bucket."""

    def __init__(self, flat_grad: torch.Tensor, buckets: List[List[tuple]], group=None, flat_param: Optional[torch.Tensor] = None,
                 shard_last: bool = True):
        self.flat, self.buckets, self.group = flat_grad, buckets, group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # `active`: collectives are issued (world > 1, or the one-rank rehearsal); `world` stays the arithmetic world size (1 / world)
        self.active = self.world > 1 or single_rank_rehearsal()
        self.rank = dist.get_rank(group) if self.active else 0
        self.views = [[flat_grad[o:o + n] for o, n in bk] for bk in buckets]
        self.pending: list = []
        self.started: set = set()
        self.param = flat_param
        self.sharded = bool(self.active and shard_last and flat_param is not None and len(buckets) > 0)
        self._param_pending: list = []
        self._rs: list = []                   # (range offset, elements per rank, reduce-scatter output buffer)
        self._tails: list = []                # (offset, length) of the few elements past W * s of a range: plain all-reduce
        if self.sharded:
            W = self.world
            for o, n in buckets[-1]:
                s = (n // (4 * W)) * 4        # elements per rank: 16-byte aligned shards, [o, o + W s) is reduce-scattered
                if s > 0:
                    self._rs.append((o, s, torch.empty(s, dtype=flat_grad.dtype, device=flat_grad.device)))
                if n - W * s > 0:
                    self._tails.append((o + W * s, n - W * s))
        self._host = self.active and dist.get_backend(group) == "gloo"      # rehearsal backend: RS / AG through host copies

    # -- gradient side
    def start(self, b: int):
        if not self.active or b in self.started:
            return
        self.started.add(b)
        if self.sharded and b == len(self.buckets) - 1:
            for o, s, out in self._rs:
                src = self.flat[o:o + self.world * s]
                if self._host:
                    h = torch.empty(s, dtype=src.dtype)
                    dist.reduce_scatter_tensor(h, src.cpu(), op=dist.ReduceOp.SUM, group=self.group)
                    out.copy_(h)
                else:
                    self.pending.append(dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for o, n in self._tails:
                self.pending.append(dist.all_reduce(self.flat[o:o + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        for v in self.views[b]:
            self.pending.append(dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def start_rest(self):
        for b in range(len(self.buckets)):
            self.start(b)

    def wait(self) -> int:
        """Returns the world size (the mean is folded into AdamW as grad_mult = 1 / world).  Sharded: this rank's slice of the
        last bucket now holds the summed gradient (the other slices keep their local values and are not read)."""
        for w in self.pending:
            w.wait()
        self.pending.clear()
        if self.sharded and (len(self.buckets) - 1) in self.started:
            for o, s, out in self._rs:
                self.flat[o + self.rank * s:o + (self.rank + 1) * s].copy_(out)
        self.started.clear()
        return self.world

    def sync_flag_(self, flag: torch.Tensor):
        """found_inf must be the same on every rank (GradScaler skips the WHOLE step): with a sharded bucket a rank only sees
        its own slice of the sums, so the flags are MAX-reduced (4 bytes)."""
        if self.sharded:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)

    def adam_pieces(self, n_flat: int) -> List[tuple]:
        """(offset, length) ranges of the flat buffers THIS rank's optimiser step updates: everything outside the sharded
        bucket, the tails, and its own shard of every reduce-scattered range."""
        if not self.sharded:
            return [(0, n_flat)]
        skip = sorted((o, o + self.world * s, s) for o, s, _ in self._rs)
        out, cur = [], 0
        for a, b, s in skip:
            if a > cur:
                out.append((cur, a - cur))
            out.append((a + self.rank * s, s))
            cur = b
        if cur < n_flat:
            out.append((cur, n_flat - cur))
        return out

    # -- parameter side (sharded bucket only)
    def start_param_gather(self):
        """After the optimiser step: every rank's updated shard -> all ranks (asynchronous)."""
        if not self.sharded:
            return
        W, r = self.world, self.rank
        for o, s, _ in self._rs:
            full = self.param[o:o + W * s]
            mine = self.param[o + r * s:o + (r + 1) * s]
            if self._host:
                h = torch.empty(W * s, dtype=full.dtype)
                dist.all_gather_into_tensor(h, mine.cpu(), group=self.group)
                full.copy_(h)
            else:       # gathered into a staging buffer (no aliasing of a collective's input and output), copied back in wait_params
                stage = torch.empty(W * s, dtype=full.dtype, device=full.device)
                work = dist.all_gather_into_tensor(stage, mine.clone(), group=self.group, async_op=True)
                self._param_pending.append((work, full, stage))

    def wait_params(self):
        for work, full, stage in self._param_pending:
            work.wait()
            full.copy_(stage)
        self._param_pending.clear()
