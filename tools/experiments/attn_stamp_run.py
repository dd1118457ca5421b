"""Driver of the stamped forward attention kernel (tools/experiments/attn_fwd_stamp.hip)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import ops
from modaltune_amd.config import branch_table, segment_lengths
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "_C", "libattn_stamp.so"))
L = 10000; B, N = 3, L + 1; M = B * N
plan = ops.make_plan(branch_table(N, segment_lengths()), N, B)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(M * 2304, device="cuda", generator=g) * 0.8).half()
o_br = torch.zeros(5, M, 768, dtype=torch.float16, device="cuda"); lse_br = torch.zeros(5, M, 16, device="cuda")
dbg = torch.zeros(4, dtype=torch.int64, device="cuda")
for it in range(3):
    dbg.zero_()
    rc = lib.mt_dbg_attn_fwd_stamps(C.c_void_p(qkv.data_ptr()), C.byref(plan), C.c_void_p(o_br.data_ptr()), C.c_void_p(lse_br.data_ptr()),
                                    C.c_void_p(dbg.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    s, p_, b, n = (int(v) for v in dbg.tolist())
    tot = s + p_ + b
    print(f"rc {rc} tiles {n}: per tile-wave cycles  scores+max {s / n:.0f}  exp+PV {p_ / n:.0f}  wait+barrier {b / n:.0f}  total {tot / n:.0f}  "
          f"shares {100 * s / tot:.0f}/{100 * p_ / tot:.0f}/{100 * b / tot:.0f} %")
