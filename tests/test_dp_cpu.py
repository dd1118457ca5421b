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
        # bucketed reducer: buckets started out of order / twice, the rest at the end; every element reduced exactly once
        from modaltune_amd import synth
        from modaltune_amd.config import ModelConfig
        from modaltune_amd.engine import ParamStore
        cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)))
        st = ParamStore(cfg, synth.toy_group_sizes(), "cpu")
        buckets = dp.grad_buckets(st.slots, 3, st.n_flat)
        assert len(buckets) == 4
        st.flat_grad.copy_(torch.arange(st.n_flat, dtype=torch.float32) % 1000 * (rank + 1))
        red = dp.GradReducer(st.flat_grad, buckets)
        red.start(0); red.start(0); red.start(2)
        red.start_rest()
        assert red.wait() == world and not red.pending
        want = torch.arange(st.n_flat, dtype=torch.float32) % 1000 * sum(r + 1 for r in range(world))
        assert torch.equal(st.flat_grad, want)
        # sharded last bucket (reduce-scatter -> AdamW on the shard -> all-gather of the parameters): every rank
        # ends with the parameters a full all-reduce + full update gives; a rank updates exactly n_flat - (W-1)/W of the
        # reduce-scattered ranges; the flag sync makes found_inf global
        g0 = torch.sin(torch.arange(st.n_flat, dtype=torch.float32) * 0.37) * (rank + 1)
        st.flat_grad.copy_(g0)
        st.flat.copy_(torch.cos(torch.arange(st.n_flat, dtype=torch.float32) * 0.11))
        p0 = st.flat.clone()
        red = dp.GradReducer(st.flat_grad, buckets, flat_param=st.flat)
        assert red.sharded and red._rs
        red.start_rest()
        assert red.wait() == world
        pieces = red.adam_pieces(st.n_flat)
        total = torch.sin(torch.arange(st.n_flat, dtype=torch.float32) * 0.37) * sum(r + 1 for r in range(world))
        covered = torch.zeros(st.n_flat, dtype=torch.bool)
        for o, k in pieces:
            assert not covered[o:o + k].any()
            covered[o:o + k] = True
            assert torch.allclose(st.flat_grad[o:o + k], total[o:o + k], rtol=1e-6, atol=1e-6)      # what the optimiser reads is the SUM
            st.flat[o:o + k] -= 0.1 * st.flat_grad[o:o + k]                                       # stand-in for AdamW on the piece
        sharded_elems = sum(world * s_ for _, s_, _ in red._rs)
        assert int(covered.sum()) == st.n_flat - sharded_elems + sharded_elems // world
        red.start_param_gather()
        red.wait_params()
        assert torch.allclose(st.flat, p0 - 0.1 * total, rtol=1e-6, atol=1e-6)                      # == all-reduce + full update
        flag = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)
        red.sync_flag_(flag)
        assert int(flag) == 1
        # The DEVICE-tensor branches (what RCCL runs: asynchronous reduce_scatter_tensor / all_reduce / all_gather_into_tensor with
        # work handles waited for later, no host staging; dp.py `else:` arms) -- gloo executes the same calls on CPU tensors, so
        # they are driven here with the host-staging switch forced off.  Same expectations as above.
        st.flat_grad.copy_(g0)
        st.flat.copy_(p0)
        red = dp.GradReducer(st.flat_grad, buckets, flat_param=st.flat)
        red._host = False
        red.start(1)
        assert red.pending and all(hasattr(w, "wait") for w in red.pending)           # async work handles, nothing waited for yet
        red.start_rest()
        n_async = len(red.pending)
        assert n_async == sum(len(v) for v in red.views[:-1]) + len(red._rs) + len(red._tails)
        assert red.wait() == world and not red.pending
        for o, k in red.adam_pieces(st.n_flat):
            assert torch.allclose(st.flat_grad[o:o + k], total[o:o + k], rtol=1e-6, atol=1e-6)
            st.flat[o:o + k] -= 0.1 * st.flat_grad[o:o + k]
        red.start_param_gather()
        assert len(red._param_pending) == len(red._rs)                                # staged, asynchronous: copied back in wait_params
        red.wait_params()
        assert not red._param_pending and torch.allclose(st.flat, p0 - 0.1 * total, rtol=1e-6, atol=1e-6)
        # sequence-parallel collectives (seqpar.py `else:` arms: all_gather_into_tensor / all_to_all_single on the tensors as they are)
        from modaltune_amd import seqpar
        sp = seqpar.SeqParallelAttention.__new__(seqpar.SeqParallelAttention)
        sp.dist, sp.group, sp.W = dist, None, world
        for staged in (True, False):
            sp._host_staged = lambda staged=staged: staged
            t = torch.arange(6, dtype=torch.float16).reshape(2, 3) + 10 * rank
            got = sp._all_gather(t)
            assert got.shape == (world, 2, 3) and all(torch.equal(got[r], torch.arange(6, dtype=torch.float16).reshape(2, 3) + 10 * r) for r in range(world))
            sizes = [3, 5] if rank == 0 else [5, 2]          # symmetric: what I send to r is as long as what r sends to me
            send = torch.arange(sum(sizes), dtype=torch.float16) + 100 * rank
            recv = sp._all_to_all(send, sizes)
            other = 1 - rank
            o_sizes = [3, 5] if other == 0 else [5, 2]
            o_send = torch.arange(sum(o_sizes), dtype=torch.float16) + 100 * other
            mine_from_me = send[:sizes[0]] if rank == 0 else send[sizes[0]:]
            from_other = o_send[sum(o_sizes[:rank]):sum(o_sizes[:rank + 1])]
            want_recv = torch.cat([mine_from_me, from_other]) if rank == 0 else torch.cat([from_other, mine_from_me])
            assert torch.equal(recv, want_recv), (rank, staged, recv, want_recv)
    finally:
        dist.destroy_process_group()


def _rehearsal_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MT_DP_REHEARSE="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from modaltune_amd import synth
        from modaltune_amd.config import ModelConfig
        from modaltune_amd.engine import ParamStore
        assert dp.single_rank_rehearsal()
        cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)))
        st = ParamStore(cfg, synth.toy_group_sizes(), "cpu")
        buckets = dp.grad_buckets(st.slots, 3, st.n_flat)
        g0 = torch.arange(st.n_flat, dtype=torch.float32) % 977
        p0 = torch.arange(st.n_flat, dtype=torch.float32) % 13
        for host in (True, False):
            st.flat_grad.copy_(g0); st.flat.copy_(p0)
            red = dp.GradReducer(st.flat_grad, buckets, flat_param=st.flat)
            assert red.active and red.world == 1 and red.rank == 0 and red.sharded      # collectives on, arithmetic world still 1
            red._host = host
            red.start(1); red.start_rest()
            assert len(red.started) == len(buckets) and (host or red.pending)
            assert red.wait() == 1
            assert torch.equal(st.flat_grad, g0)                                         # the sum over one rank
            pieces = red.adam_pieces(st.n_flat)
            assert sum(k for _, k in pieces) == st.n_flat                                # one rank owns every shard
            for o, k in pieces:
                st.flat[o:o + k] -= 0.5 * st.flat_grad[o:o + k]
            red.start_param_gather(); red.wait_params()
            assert torch.equal(st.flat, p0 - 0.5 * g0)
        flat = g0.clone()
        assert dp.allreduce_sum_(flat) == 1 and torch.equal(flat, g0)
        dp.broadcast_params_(flat, 0)
    finally:
        dist.destroy_process_group()


def test_single_rank_rehearsal_issues_the_collectives():
    """MT_DP_REHEARSE=1 on a one-rank group (what `bench.py --dp-rehearsal` and the one-rank RCCL tests of tests/test_dp_gpu.py use):
    the reducer is active and sharded with world == 1, every collective is issued and is the identity."""
    mp.spawn(_rehearsal_worker, args=(1, _free_port()), nprocs=1, join=True)


def test_grad_buckets_follow_the_backward_order():
    """Bucket 0 = fusion head + interaction block 2 (+ its prompt self-attention), ..., last = gene encoder / gene_pe /
    task tokens; together they tile the flat gradient buffer exactly once."""
    from modaltune_amd import synth
    from modaltune_amd.config import ModelConfig
    from modaltune_amd.engine import ParamStore
    for clinical in (False, True):
        cfg = ModelConfig(clinical=clinical)
        st = ParamStore(cfg, synth.toy_group_sizes(), "cpu")
        buckets = dp.grad_buckets(st.slots, len(cfg.interaction_indexes), st.n_flat)
        owner = torch.full((st.n_flat,), -1, dtype=torch.int64)
        for b, ranges in enumerate(buckets):
            for o, n in ranges:
                assert (owner[o:o + n] == -1).all()
                owner[o:o + n] = b
        assert (owner >= 0).all()

        def bucket_of(key):
            return int(owner[st.slots[key][0]])
        assert bucket_of("final_project.weight") == 0 and bucket_of("interactions.2.extra_extractors.1.ffn.linear1.weight") == 0
        assert bucket_of("prompt_selfattention.2.q_proj.weight") == 0
        assert bucket_of("interactions.1.injector.gamma") == 1 and bucket_of("prompt_selfattention.1.norm.bias") == 1
        assert bucket_of("interactions.0.extractor.attn.q_proj.weight") == 2
        assert bucket_of("gene_pe") == 3 and bucket_of("gene_encoder.pathway_compression.weight") == 3
        assert bucket_of("task_weight.0.weight") == 3
        if clinical:
            assert bucket_of("clinical_mlp.0.weight") == 3


def test_gloo_world2_allreduce_and_sharding():
    mp.spawn(_worker, args=(2, _free_port(), 1000), nprocs=2, join=True)


def test_single_process_paths():
    assert dp.allreduce_sum_(torch.ones(4)) == 1
    assert dp.shard_indices(5, 1, 0, shuffle=False) == [0, 1, 2, 3, 4]
    assert dp.shard_indices(5, 2, 1, shuffle=False) == [1, 3, 0]
    assert dp.shard_indices(5, 2, 0, shuffle=False, drop_last=True) == [0, 2]
