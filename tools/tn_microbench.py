"""gemm_tn (adapter weight gradients) at the step's shapes, for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
M = 30003
g = torch.Generator(device="cuda").manual_seed(0)
for N1, N2 in [(384, 768), (192, 768), (768, 192), (192, 192)]:
    A = (torch.randn(M, N1, device="cuda", generator=g) * 0.1).half()
    B = (torch.randn(M, N2, device="cuda", generator=g) * 0.1).half()
    C = torch.zeros(N1, N2, device="cuda")
    for _ in range(2):
        ops.gemm_tn(A, B, C, M, N1, N2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_tn(A, B, C, M, N1, N2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"gemm_tn {N1}x{N2}: {ms*1e3:.1f} us  {2.0*M*N1*N2/ms/1e9:.0f} TFLOP/s")
