"""CPU (gloo, world_size 2): the N > 1 path of the train step -- slide sharding and the gradient collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from modaltune_amd import dp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
        params = torch.full((n,), float(rank))
        dp.broadcast_params_(params, 0)
        assert torch.equal(params, torch.zeros(n))
        w = dp.allreduce_sum_(flat)
        assert w == world
        want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        assert torch.equal(flat, want)
        # the mean DDP would have produced = sum * grad_mult
        assert torch.allclose(flat * (1.0 / w), torch.arange(n, dtype=torch.float32) * (world + 1) / 2)
        # sharding: ranks partition the padded permutation exactly like DistributedSampler
        from torch.utils.data import DistributedSampler
        ds = list(range(11))
        for epoch in (0, 3):
            samp = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=7)
            samp.set_epoch(epoch)
            assert list(samp) == dp.shard_indices(len(ds), world, rank, epoch=epoch, seed=7)
        gathered = [None] * world
        dist.all_gather_object(gathered, dp.shard_indices(11, world, rank, epoch=1, seed=7))
        allidx = sorted(i for g in gathered for i in g)
        assert set(allidx) == set(range(11)) and len(allidx) == 12      # padded to a multiple of world
    finally:
        dist.destroy_process_group()


def test_gloo_world2_allreduce_and_sharding():
    mp.spawn(_worker, args=(2, _free_port(), 1000), nprocs=2, join=True)


def test_single_process_paths():
    assert dp.allreduce_sum_(torch.ones(4)) == 1
    assert dp.shard_indices(5, 1, 0, shuffle=False) == [0, 1, 2, 3, 4]
    assert dp.shard_indices(5, 2, 1, shuffle=False) == [1, 3, 0]
    assert dp.shard_indices(5, 2, 0, shuffle=False, drop_last=True) == [0, 2]
