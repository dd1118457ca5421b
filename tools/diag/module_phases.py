"""Where the module-API step (bench.py --api module: the reference trainer's own loop on the drop-in nn.Module) spends its time:
wall time per phase with a device sync after each (so phases do not overlap: the sum is above the pipelined step time).
`python tools/diag/module_phases.py 10000 fused host`: HOST time per phase (no syncs: what the Python side of each phase costs; the GPU
runs behind) with modaltune_amd.optim.AdamW; second argument `torch` for torch.optim.AdamW."""
import os, sys, time, json
import torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import synth
from modaltune_amd.aggregators import Aggregator
from modaltune_amd.config import GIGAPATH_JSON
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
sizes = synth.toy_group_sizes(6)
groups = {i: ["g%d_%d" % (i, j) for j in range(n)] for i, n in enumerate(sizes)}
model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3, init_seed=0, **dict(GIGAPATH_JSON, pretrained=False)).to(dev)
params = [{"params": [p for p in model.parameters() if p.requires_grad], "lr": 5e-6}]
which = sys.argv[2] if len(sys.argv) > 2 else "torch"
HOST = len(sys.argv) > 3 and sys.argv[3] == "host"
if which == "fused":
    from modaltune_amd.optim import AdamW
    opt = AdamW(params, weight_decay=0.01)
else:
    opt = torch.optim.AdamW(params, weight_decay=0.01)
scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=2.0 ** 15)
inp = synth.synth_inputs(L, sizes, seed=1, grid=128)
x, coords = torch.from_numpy(inp["x"]).to(dev), torch.from_numpy(inp["coords"]).to(dev)
genes = {i: torch.from_numpy(a).to(dev) for i, a in enumerate(inp["genes"])}
text = torch.from_numpy(inp["text"]).to(dev)[:, :256]
text = text / text.norm(dim=-1, keepdim=True)
loss_fn, eye = nn.KLDivLoss(reduction="sum"), torch.eye(3, device=dev)
model.train()
acc = {}
def ph(name, t0):
    if not HOST:
        torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
for it in range(14):
    if it == 6: acc.clear(); torch.cuda.synchronize(); T0 = time.perf_counter()
    if not HOST: torch.cuda.synchronize()
    t = time.perf_counter()
    with torch.autocast("cuda", enabled=True):
        logit = torch.cat([model(x=x, coords=coords, genes=genes, clinical=[], task_token=eye[k]) for k in (0, 1, 2)], dim=0)
        t = ph("3 forward calls", t)
        logit = logit / logit.norm(dim=-1, keepdim=True)
        loss = loss_fn(F.log_softmax(logit, dim=1), F.softmax(text[[0, 1, 3], :], dim=1)) * 10
        t = ph("loss (torch)", t)
    scaler.scale(loss).backward(); t = ph("backward", t)
    scaler.step(opt); t = ph("scaler.step (unscale + inf check + AdamW)", t)
    scaler.update(); t = ph("scaler.update", t)
    opt.zero_grad(); t = ph("zero_grad", t)
torch.cuda.synchronize()
print(which, json.dumps({k: round(v / 8 * 1e3, 3) for k, v in acc.items()}), "ms per step", "(HOST time per phase, no syncs; step %.2f ms)" % ((time.perf_counter() - T0) / 8 * 1e3) if HOST else "(synchronised phases)")
