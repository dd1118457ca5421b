"""Branch mix + inner LayerNorm (forward, backward) and the backward's combine kernel at L = 10 000 (for rocprofv3 --pmc runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
from modaltune_amd.config import branch_table, segment_lengths
L = 10000; B, N = 3, L + 1; M = B * N
plan = ops.make_plan(branch_table(N, segment_lengths()), N, B)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(M * 2304, device="cuda", generator=g) * 0.8).half()
o_br = torch.zeros(5, M, 768, dtype=torch.float16, device="cuda"); lse_br = torch.zeros(5, M, 16, device="cuda")
ops.dilated_attn_fwd(qkv, plan, o_br, lse_br)
ln_w, ln_b = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
y = torch.zeros(M, 768, dtype=torch.float16, device="cuda"); stats = torch.zeros(M, 2, device="cuda"); lse_tot = torch.zeros(M, 16, device="cuda")
dy = (torch.randn(M, 768, device="cuda", generator=g) * 0.1).half()
dmixed = torch.zeros(16, M, 48, dtype=torch.float16, device="cuda"); delta = torch.zeros(5, M, 16, device="cuda")
ws = torch.zeros(ops.dilated_attn_bwd_workspace_bytes(plan) // 2, dtype=torch.float16, device="cuda")
dqkv = torch.zeros(M, 2304, dtype=torch.float16, device="cuda")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


fwd = t(lambda: ops.dilated_mix_ln_fwd(o_br, lse_br, plan, ln_w, ln_b, y, stats, lse_tot))
bwd = t(lambda: ops.dilated_mix_ln_bwd(dy, o_br, lse_br, lse_tot, plan, ln_w, stats, dmixed, delta))
comb = t(lambda: ops.dilated_attn_bwd_phases(qkv, dmixed, lse_tot, delta, plan, ws, dqkv, ops.ATTN_BWD_COMBINE))
print(f"mix_fwd {fwd:.4f} mix_bwd {bwd:.4f} combine {comb:.4f} ms  checksum {float(y.float().abs().sum()):.6e} {float(dmixed.float().abs().sum()):.6e} {float(delta.abs().sum()):.6e}")
