"""Dilated-attention forward + mix at the bench geometry (for rocprofv3 --pmc runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
from modaltune_amd.config import branch_table, segment_lengths
L = 10000; B, N = 3, L + 1; M = B * N
plan = ops.make_plan(branch_table(N, segment_lengths()), N, B)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(M * 2304, device="cuda", generator=g) * 0.8).half()
o_br = torch.zeros(5, M, 768, dtype=torch.float16, device="cuda")
lse_br = torch.zeros(5, M, 16, device="cuda")
ops.dilated_attn_fwd(qkv, plan, o_br, lse_br); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.dilated_attn_fwd(qkv, plan, o_br, lse_br)
e1.record(); torch.cuda.synchronize()
print("fwd ms/launch", e0.elapsed_time(e1) / 5)
