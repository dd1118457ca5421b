"""Soak run on never-repeating bag lengths (the steady state of real data): TrainStep.step_graphed (every length new -> eager schedule)
and the drop-in nn.Module driven like the reference loop; watches device memory and host RSS for drift."""
import os, sys, time, random, resource, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import synth
from modaltune_amd.config import GIGAPATH_JSON, ModelConfig
from modaltune_amd.engine import Engine
from modaltune_amd.trainer import TrainStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = sys.argv[2] if len(sys.argv) > 2 else "trainstep"
dev = torch.device("cuda", 0)
cfg = ModelConfig(); sizes = synth.toy_group_sizes()
rnd = random.Random(5)
Lmax = 6000
inp = synth.synth_inputs(Lmax, sizes, seed=1, grid=128)
X = torch.from_numpy(inp["x"]).to(dev).half().reshape(Lmax, -1).contiguous(); C = torch.from_numpy(inp["coords"]).to(dev).reshape(-1, 2)
genes = [torch.from_numpy(a).to(dev) for a in inp["genes"]]; text = torch.from_numpy(inp["text"]).to(dev)
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20
if mode == "titan":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    import titan_standin, bench
    from modaltune_amd.titan import NativeBackbone, TitanEngine, titan_model_config
    vit = titan_standin.VisionTransformer(mlp_ratio=4.0); titan_standin.init_standin(vit, 0)
    tcfg = titan_model_config(bench.TITAN_JSON, 3, False, 6)
    inp_t = synth.synth_inputs_titan(Lmax, sizes, seed=3, grid=96)
    Xt = torch.from_numpy(inp_t["x"]).to(dev).reshape(Lmax, -1).contiguous(); Ct = torch.from_numpy(inp_t["coords"]).to(dev).reshape(Lmax, 2)
    gt = [torch.from_numpy(a).to(dev) for a in inp_t["genes"]]; tt = torch.from_numpy(inp_t["text"]).to(dev)
    eng = TitanEngine(tcfg, sizes, NativeBackbone(vit, dev), dev)
    eng.load_state_dict(synth.synth_state_dict(tcfg, sizes, seed=0)); eng.set_stochastic(True, seed=3)
    ts = TrainStep(eng); ts.set_projector(synth.projector_state(0))
    def one(L):
        return ts.step(Xt[:L], Ct[:L], gt, tt, update=True)
    val = lambda: ts.loss_value()
elif mode == "trainstep":
    eng = Engine(cfg, sizes, dev); eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0)); eng.set_stochastic(True, 7)
    ts = TrainStep(eng); ts.set_projector(synth.projector_state(0))
    def one(L):
        return ts.step_graphed(X[:L], C[:L], genes, text)
    val = lambda: ts.loss_value()
else:
    from modaltune_amd.aggregators import Aggregator
    import json
    from bench import ROOT  # noqa: F401
    groups = {i: [f"g{i}_{j}" for j in range(n)] for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3, **GIGAPATH_JSON).to(dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, sizes, seed=0).items()}, strict=True)
    model.train()
    from modaltune_amd.optim import AdamW          # (round 5: the fused drop-in; `torch` as a third argument keeps torch.optim.AdamW)
    Opt = torch.optim.AdamW if (len(sys.argv) > 3 and sys.argv[3] == "torch") else AdamW
    opt = Opt([p for p in model.parameters() if p.requires_grad], lr=1e-5)
    scaler = torch.amp.GradScaler("cuda")
    eye = torch.eye(3, device=dev)
    tgt = torch.softmax(torch.randn(3, 256, device=dev), dim=1)
    last = [0.0]
    def one(L):
        gd = {i: g for i, g in enumerate(genes)}
        xs, cs = X[:L].float().unsqueeze(0), C[:L].unsqueeze(0)          # ONE tensor for the three calls, as the reference passes `images`
        logits = torch.cat([model(x=xs, coords=cs, genes=gd, task_token=eye[t]) for t in range(3)])
        logits = logits / logits.norm(dim=-1, keepdim=True)
        loss = torch.nn.functional.kl_div(torch.log_softmax(logits, dim=1), tgt, reduction="batchmean")
        scaler.scale(loss).backward(); scaler.step(opt); scaler.update(); opt.zero_grad()
        last[0] = loss
        return loss
    val = lambda: float(last[0])
t0 = time.time(); marks = []
seen = set()
for i in range(steps):
    L = rnd.randrange(1500, Lmax)
    while L in seen:
        L = rnd.randrange(1500, Lmax)
    seen.add(L)
    one(L)
    if i % 25 == 24 or i == steps - 1:
        v = val()
        marks.append((torch.cuda.memory_reserved() / 2**30, rss()))
        print(f"step {i + 1}: loss {v:.5f} alloc {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved {marks[-1][0]:.2f} GiB host RSS {marks[-1][1]:.2f} GiB ({time.time() - t0:.1f} s)", flush=True)
assert marks[-1][0] <= marks[len(marks) // 2][0] * 1.05 + 0.25, "device memory keeps growing"
assert marks[-1][1] <= marks[len(marks) // 2][1] * 1.05 + 0.25, "host memory keeps growing"
if mode == "module":
    model._drain_decodes(block=True)
    print("read-back-free speculation:", model._nosync_rows, "fused optimiser steps:", getattr(opt, "last_step_fused", None))
print("soak ok")
