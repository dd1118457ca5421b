"""Phase durations of the ping-pong GEMM kernel (diagnostic build of gemm.hip with -DMT_GEMM_STAMP, loaded through
MODALTUNE_HIP_LIB): prologue / main loop / epilogue staging / epilogue stores, summed over the workgroups (s_memtime ticks
of wave 0 of each workgroup; 100 MHz).  The debug buffer rides in through the (unused) pos_table slot of the epilogue struct."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import ops
M = 30003
g = torch.Generator(device="cuda").manual_seed(0)
for N, K in [(3072, 768), (768, 3072), (2304, 768)]:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    bias = torch.zeros(N, device="cuda")
    dbg = torch.zeros(8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.gemm_nt(A, W, C, M, N, K, bias=bias, pos_table=dbg.view(torch.float32))
    dbg.zero_()
    reps = 5
    for _ in range(reps):
        ops.gemm_nt(A, W, C, M, N, K, bias=bias, pos_table=dbg.view(torch.float32))
    torch.cuda.synchronize()
    v = [int(x) for x in dbg.tolist()]
    nwg = v[5] / reps
    tick_us = 0.01
    print(f"N={N} K={K}: workgroups {nwg:.0f}  per workgroup us: prologue {v[0]/v[5]*tick_us:.2f}  main loop {v[1]/v[5]*tick_us:.2f}  "
          f"epilogue {v[4]/v[5]*tick_us:.2f} (staging+sync {v[2]/v[5]*tick_us:.2f}, read+store issue {v[3]/v[5]*tick_us:.2f})")
