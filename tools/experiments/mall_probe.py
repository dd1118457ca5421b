"""Does the 256 MiB Infinity Cache (MALL) keep what a kernel has just written, and what is a hit worth to the FFN chain?

1. `ln_gelu_fwd` alone, back to back on the same rows, for M rows whose in + out footprint is below / above 256 MiB: ns per row.
2. The FFN forward chain fc1 -> LN(gelu) -> fc2 (feedforward_network.py:132-143) and the backward chain dX(fc2) -> LN-GELU backward ->
   dX(fc1) over M = 30 003 rows in 1, 2, 3, 4, 6, 8 row chunks (chunk c runs all three kernels before chunk c + 1 starts), each chain
   started cold (1 GiB fill in front), per-kernel HIP-event times summed over the chunks.
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import ops

M, D, F = 30003, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc)
x16 = rn(M, D, sc=0.5).half()
W1, b1 = rn(F, D, sc=0.05).half(), rn(F)
W2, b2 = rn(D, F, sc=0.02).half(), rn(D)
W2t, W1t = W2.t().contiguous(), W1.t().contiguous()       # dX GEMMs: [F, D] (N = F, K = D) and [D, F]
lnw, lnb = torch.ones(F, device="cuda"), torch.zeros(F, device="cuda")
a1 = torch.empty(M, F, dtype=torch.float16, device="cuda"); t16 = torch.empty_like(a1)
dt16 = torch.empty_like(a1); da1 = torch.empty_like(a1)
br = torch.empty(M, D, dtype=torch.float16, device="cuda"); dy16 = torch.empty_like(br)
dbr = rn(M, D, sc=0.1).half()
stats = torch.empty(M, 2, device="cuda")
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")


def ev():
    return torch.cuda.Event(enable_timing=True)


def bounds(chunks, align=768):
    per = -(-M // chunks)
    per = -(-per // align) * align
    out, r = [], 0
    while r < M:
        out.append((r, min(M, r + per))); r += per
    return out


print("== 1. ln_gelu_fwd alone, back to back (in + out footprint) ==", flush=True)
for m in (3750, 7500, 11250, 15000, 18750, 22500, 30003):
    for _ in range(3):
        ops.layernorm_fwd(a1[:m], lnw, lnb, t16[:m], stats[:m], m, F, gelu_in=True)
    e0, e1 = ev(), ev(); e0.record()
    for _ in range(10):
        ops.layernorm_fwd(a1[:m], lnw, lnb, t16[:m], stats[:m], m, F, gelu_in=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"M={m:6d} footprint {2 * m * F * 2 / 2**20:6.0f} MiB  {us:7.1f} us  {us * 1e3 / m:6.2f} ns/row  {2 * m * F * 2 / us / 1e6:5.2f} TB/s", flush=True)
# read-only re-read: in-place variant writes over its input (footprint = m * F * 2)
for m in (7500, 15000, 30003):
    e0, e1 = ev(), ev()
    for _ in range(3):
        ops.layernorm_fwd(a1[:m], lnw, lnb, a1[:m], stats[:m], m, F, gelu_in=False)
    e0.record()
    for _ in range(10):
        ops.layernorm_fwd(a1[:m], lnw, lnb, a1[:m], stats[:m], m, F, gelu_in=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"in place LN M={m:6d} footprint {m * F * 2 / 2**20:6.0f} MiB  {us:7.1f} us  {us * 1e3 / m:6.2f} ns/row", flush=True)
a1.copy_(rn(M, F, sc=1.0).half())


def chain_fwd(chunks, scratch):
    evs = []
    for r0, r1 in bounds(chunks):
        m = r1 - r0
        t = t16[:m] if scratch else t16[r0:r1]
        e = [ev() for _ in range(4)]
        e[0].record(); ops.gemm_nt(x16[r0:r1], W1, a1[r0:r1], m, F, D, bias=b1)
        e[1].record(); ops.layernorm_fwd(a1[r0:r1], lnw, lnb, t, stats[r0:r1], m, F, gelu_in=True)
        e[2].record(); ops.gemm_nt(t, W2, br[r0:r1], m, D, F, bias=b2)
        e[3].record(); evs.append(e)
    return evs


def chain_bwd(chunks, scratch):
    evs = []
    for r0, r1 in bounds(chunks):
        m = r1 - r0
        d_t = dt16[:m] if scratch else dt16[r0:r1]
        d_a = da1[:m] if scratch else da1[r0:r1]
        e = [ev() for _ in range(4)]
        e[0].record(); ops.gemm_nt(dbr[r0:r1], W2t, d_t, m, F, D, bias=None)
        e[1].record(); ops.layernorm_bwd(d_t, a1[r0:r1], lnw, stats[r0:r1], d_a, m, F, gelu_in=True)
        e[2].record(); ops.gemm_nt(d_a, W1t, dy16[r0:r1], m, D, F, bias=None)
        e[3].record(); evs.append(e)
    return evs


for name, chain in (("forward fc1 / ln_gelu / fc2", chain_fwd), ("backward dX_fc2 / ln_gelu_bwd / dX_fc1", chain_bwd)):
    print(f"== 2. {name}: chunks, scratch -> per-kernel us (sum over chunks) | total ==", flush=True)
    ref = None
    for chunks in (1, 2, 3, 4, 6, 8):
        for scratch in ((False,) if chunks == 1 else (False, True)):
            acc = []
            for rep in range(6):
                junk.fill_(float(rep))
                s0, s1 = ev(), ev(); s0.record()
                evs = chain(chunks, scratch)
                s1.record(); torch.cuda.synchronize()
                if rep >= 2:
                    k = [sum(e[i].elapsed_time(e[i + 1]) for e in evs) * 1e3 for i in range(3)]
                    acc.append(k + [s0.elapsed_time(s1) * 1e3])
            avg = [sum(a[i] for a in acc) / len(acc) for i in range(4)]
            chk = float(br.float().abs().sum()) if chain is chain_fwd else float(dy16.float().abs().sum())
            print(f"chunks {chunks} scratch {int(scratch)}: {avg[0]:7.1f} {avg[1]:7.1f} {avg[2]:7.1f} | {avg[3]:7.1f} us   checksum {chk:.6e}", flush=True)
