"""fc1 (M = 30003, N = 3072, K = 768, bias) on the persistent kernel: regular build vs a -DPS_GELU_EPI build that applies erf-GELU where
the finished tile is converted (timing only: the pre-activation the backward needs is not written).  Run under tools/ab_lib.sh."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import ops
M, N, K = 30003, 3072, 768
g = torch.Generator(device="cuda").manual_seed(0)
A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half(); W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
bias = torch.randn(N, device="cuda", generator=g); C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
for _ in range(5):
    ops.gemm_nt(A, W, C, M, N, K, bias=bias)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gemm_nt(A, W, C, M, N, K, bias=bias)
e1.record(); torch.cuda.synchronize()
print(f"fc1 {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
