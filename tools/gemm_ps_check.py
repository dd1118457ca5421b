"""Correctness + timing of the persistent GEMM (csrc/gemm_ps.hip) against torch.matmul in fp32 on the same fp16 operands, and -- in a
child process with MT_GEMM_PS=0 -- the ping-pong kernel's timing on the same shapes.  Usage: python tools/gemm_ps_check.py [time-only]"""
import os, subprocess, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops

shapes = [(30003, 3072, 768, "bias"), (30003, 768, 3072, "bias"), (30003, 2304, 768, "qkv"), (30003, 768, 768, "bias"), (30003, 768, 2304, "none"),
          (30003, 3072, 768, "none"), (8200, 768, 768, "bias"), (8449, 1024, 1536, "none")]
g = torch.Generator(device="cuda").manual_seed(0)
check = not (len(sys.argv) > 1 and sys.argv[1] == "time-only")
for M, N, K, kind in shapes:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(N, device="cuda", generator=g) if kind != "none" else None
    C = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    epi = ops.EPI_QKV_HM if kind == "qkv" else ops.EPI_BIAS
    fn = lambda: ops.gemm_nt(A, W, C, M, N, K, bias=bias, epilogue=epi)
    fn()
    torch.cuda.synchronize()
    msg = ""
    if check:
        ref = A.float() @ W.float().t()
        if bias is not None:
            ref += bias
        got = C.float()
        if kind == "qkv":
            got = C.view(N // 48, M, 48).permute(1, 0, 2).reshape(M, N).float()
        err = float((got - ref).abs().max() / ref.abs().max())
        nan = int(torch.isnan(C).sum())
        msg = f"  rel-max-err {err:.2e} nan {nan} {'OK' if err < 2e-3 and nan == 0 else 'FAIL'}"
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"MT_GEMM_PS={os.environ.get('MT_GEMM_PS', '1')} M={M} N={N} K={K} {kind}: {ms * 1e3:.1f} us  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s{msg}", flush=True)
if os.environ.get("MT_GEMM_PS", "1") != "0":
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "time-only"], env=dict(os.environ, MT_GEMM_PS="0")))
