"""Which hipBLASLt kernel torch.matmul picks at the backbone GEMM shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
M = 30003
for N, K in [(3072, 768), (768, 3072), (2304, 768), (768, 768), (768, 2304)]:
    A = (torch.randn(M, K, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, device="cuda") * 0.05).half()
    C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    for _ in range(5):
        torch.matmul(A, W.t(), out=C)
    torch.cuda.synchronize()
