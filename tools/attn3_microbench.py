"""Per-kernel times of the dilated-attention kernels at the bench geometry (L = 10 000, 3 passes): forward, dK/dV, dQ, combine.
For same-box A/B runs of two library builds: tools/ab_lib.sh <libA> <libB> <reps> tools/attn3_microbench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
from modaltune_amd.config import branch_table, segment_lengths
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
B, N = 3, L + 1
M = B * N
plan = ops.make_plan(branch_table(N, segment_lengths()), N, B)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(M * 2304, device="cuda", generator=g) * 0.8).half()
dmixed = (torch.randn(M * 768, device="cuda", generator=g) * 0.1).half()
lse_tot = torch.full((M, 16), 6.0, device="cuda"); delta = torch.zeros(5, M, 16, device="cuda")
dqkv = torch.zeros(M, 2304, device="cuda", dtype=torch.float16)
wsb = torch.zeros(ops.dilated_attn_bwd_workspace_bytes(plan) // 4, device="cuda")
o_br = torch.zeros(5, M, 768, dtype=torch.float16, device="cuda"); lse_br = torch.zeros(5, M, 16, device="cuda")


def t(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {"fwd": t(lambda: ops.dilated_attn_fwd(qkv, plan, o_br, lse_br))}
for name, ph in (("kv", ops.ATTN_BWD_KV), ("q", ops.ATTN_BWD_Q), ("comb", ops.ATTN_BWD_COMBINE)):
    out[name] = t(lambda: ops.dilated_attn_bwd_phases(qkv, dmixed, lse_tot, delta, plan, wsb, dqkv, ph))
print(" ".join(f"{k} {v:.4f}" for k, v in out.items()))
