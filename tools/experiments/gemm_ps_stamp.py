"""Phase anatomy of the persistent GEMM's steady-state K-slice (csrc/gemm_ps.hip built with -DPS_STAMP into build_variants/gemm_ps_stamp.so):
s_memtime of wave 0 of every workgroup, summed over its plain slices.  Usage: python tools/experiments/gemm_ps_stamp.py [N K]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "build_variants", os.environ.get("PS_LIB", "gemm_ps_stamp.so")))
M = 30003
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3072, 768)
A = (torch.randn(M, K, device="cuda") * 0.5).half()
W = (torch.randn(N, K, device="cuda") * 0.05).half()
C = torch.zeros(M, N, device="cuda", dtype=torch.float16)
bias = torch.randn(N, device="cuda") if os.environ.get("PS_BIAS") else None
st = torch.zeros(256 * 8, device="cuda", dtype=torch.int64)
assert lib.mt_gemm_ps_set_stamps(ctypes.c_void_p(st.data_ptr())) == 0
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda") if os.environ.get("PS_COLD") else None
for it in range(3):
    if junk is not None:
        junk.fill_(float(it))          # evict L2 / MALL: operands come from HBM, as after the previous kernels of a train step
    rc = lib.mt_gemm_ps_stamp_launch(ctypes.c_void_p(A.data_ptr()), ctypes.c_long(K), ctypes.c_void_p(W.data_ptr()), M, N, K, ctypes.c_void_p(bias.data_ptr()) if bias is not None else None, ctypes.c_void_p(C.data_ptr()), ctypes.c_long(N))
    assert rc == 0
torch.cuda.synchronize()
ref = A[:512].float() @ W.float().t() + (bias if bias is not None else 0)
print("check", float((C[:512].float() - ref).abs().max() / ref.abs().max()))
s = st.view(256, 8).double().cpu()
S = K // 32
tiles = s[:, 7]
plain = tiles * (S - 12)                     # plain slices per workgroup (the stamped ones)
names = ["vmcnt wait", "barrier", "48 MFMAs + riders", "-", "-", "stamp cost"]
per = s[:, :6] / plain[:, None]
print(f"N={N} K={K}: tiles/WG min {tiles.min():.0f} max {tiles.max():.0f}; s_memtime ticks (= core-clock cycles) per plain slice, mean over workgroups:")
for i, n in enumerate(names):
    print(f"  {n:22s} {per[:, i].mean():8.2f}  (min {per[:, i].min():.2f} max {per[:, i].max():.2f})")
tot = per[:, :3].sum(1) - 3 * per[:, 5]
print(f"  slice total (stamps subtracted) {tot.mean():.2f} cycles; whole kernel {s[:, 6].max():.0f} cycles; MFMA floor 768 cycles per slice")
