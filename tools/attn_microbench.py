"""Microbenchmark of the dilated-attention kernels at the bench geometry (for rocprofv3 --pmc runs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
from modaltune_amd.config import branch_table, segment_lengths, flops_per_slide_step

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "fwd"
B, N = 3, L + 1
M = B * N
bt = branch_table(N, segment_lengths())
plan = ops.make_plan(bt, N, B)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(M, 2304, device="cuda", generator=g) * 0.8).half()   # (any layout: random data)
o_br = torch.zeros(5, M, 768, dtype=torch.float16, device="cuda")
lse_br = torch.zeros(5, M, 16, device="cuda")
ops.dilated_attn_fwd(qkv, plan, o_br, lse_br)
y = torch.zeros(M, 768, dtype=torch.float16, device="cuda"); stats = torch.zeros(M, 2, device="cuda"); lse_tot = torch.zeros(M, 16, device="cuda")
w = torch.ones(768, device="cuda"); b = torch.zeros(768, device="cuda")
ops.dilated_mix_ln_fwd(o_br, lse_br, plan, w, b, y, stats, lse_tot)
dy = (torch.randn(M, 768, device="cuda", generator=g) * 0.1).half()
dmixed = torch.zeros(M, 768, dtype=torch.float16, device="cuda"); delta = torch.zeros(5, M, 16, device="cuda")
ops.dilated_mix_ln_bwd(dy, o_br, lse_br, lse_tot, plan, w, stats, dmixed, delta)
dqkv = torch.zeros(M, 2304, device="cuda", dtype=torch.float16)
wsb = torch.zeros(ops.dilated_attn_bwd_workspace_bytes(plan) // 2, device="cuda", dtype=torch.float16)
torch.cuda.synchronize()
fl = 3 * flops_per_slide_step(L, 65)["attn_layer"]
for name, fn, mult in (("fwd", lambda: ops.dilated_attn_fwd(qkv, plan, o_br, lse_br), 1.0),
                       ("bwd", lambda: ops.dilated_attn_bwd(qkv, dmixed, lse_tot, delta, plan, wsb, dqkv), 2.5)):
    if mode not in (name, "both"):
        continue
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name}: {ms:.3f} ms/launch  {fl * mult / ms / 1e9:.1f} TFLOP/s (algorithmic)")
