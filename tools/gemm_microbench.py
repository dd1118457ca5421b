"""Microbenchmark of gemm_nt at the backbone shapes (for rocprofv3 --pmc runs); optional torch.matmul yardstick."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops

yard = len(sys.argv) > 1 and sys.argv[1] == "yard"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 30003      # python tools/gemm_microbench.py [yard|ours] [M]
shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768), (768, 2304)]
g = torch.Generator(device="cuda").manual_seed(0)
for N, K in shapes:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    bias = torch.zeros(N, device="cuda")
    fn = (lambda: torch.matmul(A, W.t(), out=C)) if yard else (lambda: ops.gemm_nt(A, W, C, M, N, K, bias=bias))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{'matmul' if yard else 'gemm_nt'} N={N} K={K}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s")
