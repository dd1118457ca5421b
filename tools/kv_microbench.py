import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
from modaltune_amd.config import branch_table, segment_lengths
L=10000; B,N=3,L+1; M=B*N
plan=ops.make_plan(branch_table(N, segment_lengths()), N, B)
g=torch.Generator(device="cuda").manual_seed(0)
qkv=(torch.randn(M*2304, device="cuda", generator=g)*0.8).half()
dmixed=(torch.randn(M*768, device="cuda", generator=g)*0.1).half()
lse_tot=torch.full((M,16), 6.0, device="cuda"); delta=torch.zeros(5,M,16, device="cuda")
dqkv=torch.zeros(M,2304, device="cuda", dtype=torch.float16)
wsb=torch.zeros(ops.dilated_attn_bwd_workspace_bytes(plan)//4, device="cuda")
ops.dilated_attn_bwd(qkv, dmixed, lse_tot, delta, plan, wsb, dqkv); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): ops.dilated_attn_bwd(qkv, dmixed, lse_tot, delta, plan, wsb, dqkv)
e1.record(); torch.cuda.synchronize()
print("bwd total ms/launch", e0.elapsed_time(e1)/5)
