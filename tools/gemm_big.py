import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in [(8192, 4096, 4096), (30003, 3072, 3072), (30003, 3072, 768)]:
    A = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).half()
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1).half()
    C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3): ops.gemm_nt(A, W, C, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm_nt(A, W, C, M, N, K)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.0f} TFLOP/s")
    if len(sys.argv) > 1:
        for _ in range(3): torch.matmul(A, W.t(), out=C)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): torch.matmul(A, W.t(), out=C)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"   matmul: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.0f} TFLOP/s")
