import os, sys
import torch
sys.path.insert(0, "/root/repo")
from modaltune_amd import ops
M = 30003
g = torch.Generator(device="cuda").manual_seed(0)
for N, K in [(3072, 768), (768, 3072), (2304, 768), (768, 2304)]:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    bias = torch.zeros(N, device="cuda")
    fn = lambda: ops.gemm_nt(A, W, C, M, N, K, bias=bias)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"MT_EXP={os.environ.get('MT_EXP','0')} N={N} K={K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s")
