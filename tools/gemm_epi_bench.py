"""Cost of the gemm_nt epilogues at the backbone shapes: fp16 out / fp32 out / fp32 out + fp32 residual."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops

M = 30003
g = torch.Generator(device="cuda").manual_seed(0)
for N, K in [(768, 768), (768, 3072), (3072, 768), (2304, 768)]:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    bias = torch.zeros(N, device="cuda")
    C16 = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    C32 = torch.zeros(M, N, device="cuda")
    R32 = torch.randn(M, N, device="cuda")
    variants = {
        "f16 out": lambda: ops.gemm_nt(A, W, C16, M, N, K, bias=bias),
        "f32 out": lambda: ops.gemm_nt(A, W, C32, M, N, K, bias=bias),
        "f32 out + resid": lambda: ops.gemm_nt(A, W, C32, M, N, K, bias=bias, epilogue=ops.EPI_BIAS_RESID, resid=R32, ldr=N),
        "f32 out + resid in place": lambda: ops.gemm_nt(A, W, R32, M, N, K, bias=bias, epilogue=ops.EPI_BIAS_RESID, resid=R32, ldr=N),
    }
    for name, fn in variants.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"N={N} K={K} {name:26s}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:5.0f} TFLOP/s")
