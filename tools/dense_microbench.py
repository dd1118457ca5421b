"""Per-kernel times of the dense ALiBi attention kernels (TITAN configuration) at N = 4097 tokens, 3 passes, 12 heads."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
B, H = 3, 12
M, D = B * N, H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(M, 3 * D, device="cuda", generator=g) * 0.5).half()
d_o = (torch.randn(M, D, device="cuda", generator=g) * 0.1).half()
side = int(math.sqrt(N)) + 1
cells = torch.stack([torch.arange(N - 1) // side, torch.arange(N - 1) % side], 1).int().cuda()
nslope = torch.tensor([-(2.0 ** (-8.0 * (i + 1) / H)) * math.log2(math.e) for i in range(H)], device="cuda")
dist = torch.empty(ops.alibi_dist_halves(N), dtype=torch.float16, device="cuda")
ops.alibi_dist(cells, N, dist)
plan = ops.make_dense_plan(N, B, H, dist if os.environ.get("MT_DENSE_BIAS", "1") != "0" else None, nslope)
o = torch.empty(M, D, dtype=torch.float16, device="cuda"); lse = torch.empty(M, H, device="cuda")
delta = torch.empty(M, H, device="cuda"); dqkv = torch.empty_like(qkv)


def t(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {"fwd": t(lambda: ops.dense_attn_fwd(qkv, plan, o, lse))}
for name, ph in (("delta", ops.DENSE_BWD_DELTA), ("kv", ops.DENSE_BWD_KV), ("q", ops.DENSE_BWD_Q)):
    out[name] = t(lambda: ops._dense_attn_bwd_phase(qkv, o, d_o, lse, plan, delta, dqkv, ph))
fl = 4.0 * N * N * D * B
print(" ".join(f"{k} {v:.4f}" for k, v in out.items()), "| TF/s fwd %.0f kv %.0f q %.0f" % (fl / out["fwd"] / 1e9, 2 * fl / out["kv"] / 1e9, 1.5 * fl / out["q"] / 1e9))

# dK/dV and dQ are independent given delta: one after the other on one stream vs side by side on two streams (do the tail rounds of one
# launch fill with the other's workgroups?)
if os.environ.get("MT_DENSE_OVERLAP"):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def seq():
        ops._dense_attn_bwd_phase(qkv, o, d_o, lse, plan, delta, dqkv, ops.DENSE_BWD_KV)
        ops._dense_attn_bwd_phase(qkv, o, d_o, lse, plan, delta, dqkv, ops.DENSE_BWD_Q)

    def par():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ops._dense_attn_bwd_phase(qkv, o, d_o, lse, plan, delta, dqkv, ops.DENSE_BWD_KV)
        with torch.cuda.stream(s2):
            ops._dense_attn_bwd_phase(qkv, o, d_o, lse, plan, delta, dqkv, ops.DENSE_BWD_Q)
        cur.wait_stream(s1); cur.wait_stream(s2)
    print("kv then q: %.4f ms   kv || q: %.4f ms" % (t(seq), t(par)))
