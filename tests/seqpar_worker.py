"""Rank process of the sequence-parallel attention test (tests/test_seqpar_gpu.py):
`python tests/seqpar_worker.py CASE RANK WORLD PORT OUT`.  Default: every rank uses GPU 0 over gloo (one GPU per box); with
MT_TEST_BACKEND=nccl (set by the tests when the box has >= WORLD GPUs) rank r takes GPU r and the collectives run over RCCL.
modaltune_amd/seqpar.py issues the same torch.distributed calls on both (gloo through host copies of the fp16 payloads)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import unit_inputs  # noqa: E402

QK = 0.14433756729740643 * 1.4426950408889634      # MT_QK_SCALE_LOG2

# (ranks, passes, local length, segment lengths, ratios): the golden cases of the reference + kernel-sized ones (several key
# tiles per sequence, query tiles that end inside the local part, 3 passes)
CASES = dict(unit_inputs.SEQPAR_CASES)
CASES["big2"] = (2, 3, 448, [64, 448, 896, 3584], [1, 2, 4, 8])
CASES["big4"] = (4, 1, 320, [160, 320, 640, 5120], [1, 1, 2, 4])


def inputs(case, seed=23):
    if case in unit_inputs.SEQPAR_CASES:
        return unit_inputs.seqpar_inputs(seed, case)
    W, B, L, _, _ = CASES[case]
    r = np.random.Generator(np.random.PCG64([seed, W, L, 7]))
    return (0.6 * r.standard_normal((W, B, L, 16, 48)), 0.6 * r.standard_normal((W, B, L, 16, 48)),
            r.standard_normal((W, B, L, 16, 48)), r.standard_normal((W, B, L, 768)))


def rounded(case):
    """fp16 kernel inputs of every rank and the exact fp64 values the kernels then see (q'/QK | k | v)."""
    q, k, v, dy = (torch.from_numpy(a) for a in inputs(case))
    q16 = (q.float() * QK).half()
    return q16, k.half(), v.half(), dy.half()


def main():
    case, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    backend = os.environ.get("MT_TEST_BACKEND", "gloo")
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
    from modaltune_amd.seqpar import SeqParallelAttention
    W, B, L, segs, ratios = CASES[case]
    assert W == world
    q16, k16, v16, dy16 = rounded(case)
    M = B * L
    tok = torch.cat([t[rank].reshape(M, 768) for t in (q16, k16, v16)], dim=-1).cuda()          # [M, 2304]
    qkv_hm = tok.view(M, 3, 16, 48).permute(1, 2, 0, 3).contiguous()
    sp = SeqParallelAttention(segs, ratios, B, L)
    ln_w, ln_b = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
    y, ctx = sp.forward(qkv_hm, ln_w, ln_b)
    dqkv = sp.backward(ctx, dy16[rank].reshape(M, 768).cuda())
    torch.cuda.synchronize()
    np.savez(out, y=y.float().cpu().numpy(), dqkv=dqkv.float().cpu().numpy(), nb_loc=sp.nb_loc, ngroups=len(sp.groups))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
