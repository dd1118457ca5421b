"""Per-launch time of the token-side stage kernels (csrc/token_stage.hip) at the shapes of one step, back to back inside a hipGraph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import ops
from modaltune_amd.tape import Param, Tape, Var
dev = "cuda"
B, T = 3, 65


def graph_time(fn, n=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


def case(name, K0, N1, N2, act, ln, pe, Nb=0, resid=None):
    R = B * T
    held = []

    def rn(*s):          # (the descriptor holds raw pointers: every tensor stays alive in `held`)
        t = torch.randn(*s, device=dev)
        held.append(t)
        return t
    x = rn(R, K0)
    h, pre, hb = rn(R, N1), rn(R, N1), rn(R, max(Nb, 1))
    st = ops.token_stage(R, K0, x, rn(N1, K0), N1, h, b1=rn(N1), act1=act, pre1=pre if act else None, ln_w=rn(K0) if ln else None,
                         ln_b=rn(K0) if ln else None, pe=rn(T, K0) if pe else None, pe_period=T if pe else 0, tn=rn(R, K0) if ln else None,
                         a=rn(R, K0) if pe else None, stats=rn(R, 2) if ln else None, W1b=rn(Nb, K0) if Nb else None, b1b=rn(Nb) if Nb else None,
                         N1b=Nb, hb=hb if Nb else None, W2=rn(N2, N1) if N2 else None, b2=rn(N2) if N2 else None, N2=N2,
                         resid=rn(R, N2) if resid else None, resid_scale=resid or 1.0, y=rn(R, N2) if N2 else None)
    keep = [x, h, pre, hb]
    f = graph_time(lambda: ops.token_stage_fwd(st))
    dy, dh, da, dx = rn(R, max(N2, 1)), rn(R, N1), rn(R, K0), rn(R, K0)
    dlw, dlb, dpe = torch.zeros(K0, device=dev), torch.zeros(K0, device=dev), torch.zeros(T, K0, device=dev)
    b = graph_time(lambda: ops.token_stage_bwd(st, dy=dy if N2 else None, dh=dh, dhb=hb if Nb else None, da=da, da_accumulate=False,
                                               dx=dx if ln else None, dlnw=dlw if ln else None, dlnb=dlb if ln else None, dpe=dpe if pe else None))
    print(f"{name:30s} K0={K0:4d} N1={N1:4d} N2={N2:4d}  fwd {f:7.2f} us   bwd {b:7.2f} us", flush=True)


case("extractor q (LN+pe, 2 lin)", 768, 192, 192, 0, True, True)
case("extractor out (2 lin + 2c)", 192, 192, 768, 0, False, False, resid=2.0)
case("extractor ffn (LN, relu)", 768, 192, 768, ops.ACT_RELU, True, False, resid=1.0)
case("inject k|v (LN+pe, sibling)", 768, 192, 0, 0, True, True, Nb=192)
case("head (LN, lin)", 768, 256, 0, 0, True, False)
case("lin only 768->192", 768, 192, 0, 0, False, False)
case("lin only 192->768", 192, 768, 0, 0, False, False)
