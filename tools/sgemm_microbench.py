"""Per-launch time of the token-side fp32 GEMM (mt_sgemm_multi) at the shapes of one step, launched back to back inside a hipGraph
(what the step's replay sees): forward y = x W^T, dX = dy W, dW = dy^T x."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
dev = "cuda"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 195


def bench(name, make, n=50):
    probs, keep = make()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.sgemm_multi(probs)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                ops.sgemm_multi(probs)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s); torch.cuda.synchronize()
    print(f"{name:42s} {e0.elapsed_time(e1) / (5 * n) * 1e3:7.2f} us / launch")


def fwd(M, N, K):
    def mk():
        x, w, y = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        b = torch.randn(N, device=dev)
        return [ops.sgemm_problem(x, (K, 1), w, (K, 1), y, (N, 1), M, N, K, bias=b)], (x, w, y, b)
    return mk


def bwd(M, N, K):      # dX [M, K] = dy [M, N] W [N, K]  and  dW [N, K] = dy^T x  in one launch
    def mk():
        x, w, dy = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
        dx, dw, db = torch.empty(M, K, device=dev), torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        return [ops.sgemm_problem(dy, (N, 1), w, (1, K), dx, (K, 1), M, K, N),
                ops.sgemm_problem(dy, (1, N), x, (1, K), dw, (K, 1), N, K, M, accumulate=True, rowsum=db)], (x, w, dy, dx, dw, db)
    return mk


for (M, N, K) in ((T, 192, 768), (T, 768, 192), (T, 192, 192), (T, 576, 192), (T, 48, 192), (T, 768, 768), (3, 256, 768), (T, 192, 16)):
    bench(f"fwd  M={M} N={N} K={K}", fwd(M, N, K))
for (M, N, K) in ((T, 192, 768), (T, 768, 192), (T, 192, 192)):
    bench(f"bwd  M={M} N={N} K={K} (dX + dW)", bwd(M, N, K))


# ---- the step's other forms (tools/diag/token_launches.py prints where they come from)
def act_fwd(M, N, K, act, pre, drop):      # the mixer's first dense: GELU, saved pre-activation, element dropout
    def mk():
        x, w, y = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        b, p = torch.randn(N, device=dev), torch.empty(M, N, device=dev)
        rng = torch.tensor([1, 2, 3, 0], dtype=torch.int32, device=dev)
        d = ops.dropout_spec(rng, 5, 0.1) if drop else None
        return [ops.sgemm_problem(x, (K, 1), w, (K, 1), y, (N, 1), M, N, K, bias=b, act=act, pre_out=p if pre else None, c_drop=d)], (x, w, y, b, p, rng)
    return mk


def pass_sum(P, n):      # d(broadcast rows) += sum over the task passes: [1, n] += ones[1, P] . dy[P, n]
    def mk():
        dy, ones, out = torch.randn(P, n, device=dev), torch.ones(1, P, device=dev), torch.zeros(1, n, device=dev)
        return [ops.sgemm_problem(ones, (P, 1), dy, (1, n), out, (n, 1), 1, n, P, accumulate=True)], (dy, ones, out)
    return mk


for act, pre, drop in ((0, False, False), (ops.ACT_GELU, False, False), (ops.ACT_GELU, True, False), (ops.ACT_GELU, True, True), (ops.ACT_RELU, True, True)):
    bench(f"fwd  M=18 N=128 K=256 act={act} pre={int(pre)} drop={int(drop)}", act_fwd(18, 128, 256, act, pre, drop))
    bench(f"fwd  M={T} N=192 K=768 act={act} pre={int(pre)} drop={int(drop)}", act_fwd(T, 192, 768, act, pre, drop))
bench("pass sum [1, 49920] += 1[1,3] dy[3, 49920]", pass_sum(3, 49920))
