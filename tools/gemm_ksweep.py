"""gemm_nt time against K at the backbone's M and N: separates the per-tile fixed cost (prologue + epilogue) from the
per-K-step cost of the main loop (time per tile round = a + b * K / 64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 30003
g = torch.Generator(device="cuda").manual_seed(0)
for N in (3072, 768):
    for K in (192, 384, 768, 1536, 3072, 6144):
        A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
        C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
        bias = torch.zeros(N, device="cuda")
        fn = lambda: ops.gemm_nt(A, W, C, M, N, K, bias=bias)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s")
