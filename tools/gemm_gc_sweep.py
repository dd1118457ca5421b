"""Tile-order sweep of the persistent GEMMs: MT_GEMM_GC column tiles per group (see tile_of in csrc/gemm_ps.hip), warm and cold operands."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
M = 30003
shapes = [(3072, 768, "bias"), (2304, 768, "qkv"), (768, 768, "bias"), (768, 2304, "none"), (768, 3072, "none")]
g = torch.Generator(device="cuda").manual_seed(0)
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
kern = "ps"
os.environ["MT_GEMM_PS"] = "1"
for N, K, kind in shapes:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(N, device="cuda", generator=g) if kind != "none" else None
    C = torch.zeros(M * N, device="cuda", dtype=torch.float16)
    epi = ops.EPI_QKV_HM if kind == "qkv" else ops.EPI_BIAS
    out = []
    for gc in (0, 1, 2, 3, 4, 6):
        if gc > N // 256:
            continue
        os.environ["MT_GEMM_GC"] = str(gc)
        r = []
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
            r.append(sum(ts) / len(ts))
        out.append(f"gc={gc or 'all'}: {r[0]:.0f}/{r[1]:.0f}")
    print(f"{kern} N={N} K={K} (cold/warm us): " + "  ".join(out), flush=True)
