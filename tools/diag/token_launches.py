"""Per-launch cost of the token side: python tools/diag/token_launches.py [patches]
Records the token-side launches of one forward + backward (ops.RECORD), then times every recorded launch on its own: a hipGraph
holding 20 back-to-back copies of that ONE launch, replayed 5 times -> us per launch in a dependent chain of itself (what it costs
inside the step's chain).  Products print their (M, N, K, batch) and operand orientation; the table is grouped by signature."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import synth, ops
from modaltune_amd.config import ModelConfig
from modaltune_amd.engine import Engine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0)
cfg = ModelConfig(); sizes = synth.toy_group_sizes()
eng = Engine(cfg, sizes, dev)
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))
eng.set_stochastic(True, seed=1)
inp = synth.synth_inputs(L, sizes, seed=1, grid=128)
x = torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1).contiguous(); coords = torch.from_numpy(inp["coords"]).to(dev)
genes = [torch.from_numpy(a).to(dev) for a in inp["genes"]]
eye = torch.eye(3, device=dev)
dl = torch.randn(3, 256, device=dev) * 100.0


def step():
    eng.forward(x, coords, genes, eye, need_grad=True, fresh=True)
    eng.backward(dl, call=eng.last_call)


step(); step(); torch.cuda.synchronize()
ops.RECORD, ops.RECORD_KEEP[:] = [], []
step()
rec, ops.RECORD = ops.RECORD, None
torch.cuda.synchronize()


def signature(fn, a, k):
    name = fn.__name__
    if name == "sgemm_multi":
        parts = []
        for q in a[0]:
            kind = ("k" if q.as1 == 1 else "s") + ("k" if q.bs1 == 1 else "s")
            extra = "".join(t for t, on in (("+aux", bool(q.a_aux)), ("+rs", bool(q.rowsum)), ("+acc", bool(q.accumulate)),
                                            ("+act", q.act != 0), ("+res", bool(q.resid))) if on)
            parts.append(f"{q.M}x{q.N}x{q.K}" + (f"b{q.batch}" if q.batch > 1 else "") + kind + extra)
        return "sgemm[" + " | ".join(parts) + "]"
    return name


gs = torch.cuda.Stream()
rows = []
with torch.cuda.stream(gs):
    for fn, a, k in rec:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=gs):
            for _ in range(20):
                fn(*a, **k)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(gs)
        for _ in range(5):
            g.replay()
        e1.record(gs); torch.cuda.synchronize()
        rows.append((signature(fn, a, k), e0.elapsed_time(e1) * 1e3 / 100))
        if rows[-1][1] > 15 and fn.__name__ == "sgemm_multi":
            for q in a[0]:
                print("slow product:", {f: getattr(q, f) for f, _ in q._fields_ if not f.endswith("drop")}, "c_drop p", q.c_drop.p, q.c_drop.path_p,
                      "a_drop p", q.a_drop.p, q.a_drop.path_p)
        del g
tot = collections.OrderedDict()
for s, us in rows:
    n, t = tot.get(s, (0, 0.0))
    tot[s] = (n + 1, t + us)
print(f"{len(rows)} token-side launches, {sum(us for _, us in rows) / 1e3:.3f} ms as isolated dependent chains")
for s, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us  x{n:3d}  {t / n:6.2f} us each  {s}")
