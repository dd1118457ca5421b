"""gemm_nt at the backbone shapes with the operands COLD (a 1 GiB fill between launches evicts L2 / MALL, as the previous kernels of a
train step do) and warm (back-to-back), persistent kernel (MT_GEMM_PS=1) vs ping-pong (MT_GEMM_PS=0) in one process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 30003
shapes = [(3072, 768, "bias"), (768, 3072, "bias"), (2304, 768, "qkv"), (768, 768, "bias"), (768, 2304, "none")]
g = torch.Generator(device="cuda").manual_seed(0)
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
for N, K, kind in shapes:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(N, device="cuda", generator=g) if kind != "none" else None
    C = torch.zeros(M * N, device="cuda", dtype=torch.float16)
    epi = ops.EPI_QKV_HM if kind == "qkv" else ops.EPI_BIAS
    res = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["MT_GEMM_PS"] = mode
        for cold in (True, False):
            ts = []
            for it in range(8):
                if cold:
                    junk.fill_(float(it))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.gemm_nt(A, W, C, M, N, K, bias=bias, epilogue=epi); e1.record()
                torch.cuda.synchronize()
                if it >= 2:
                    ts.append(e0.elapsed_time(e1) * 1e3)
            res.setdefault((mode, cold), []).append(sum(ts) / len(ts))
    fmt = lambda k: "/".join(f"{v:.0f}" for v in res[k])
    print(f"M={M} N={N} K={K} {kind}: cold  pingpong {fmt(('0', True))} us  persistent {fmt(('1', True))} us | warm  pingpong {fmt(('0', False))}  persistent {fmt(('1', False))}", flush=True)
