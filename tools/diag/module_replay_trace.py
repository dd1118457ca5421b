"""Per-step counters of the module bridge's graph replay under the reference loop (bench.py --api module's loop, small bag)."""
import json, os, sys
import torch, torch.nn as nn, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from modaltune_amd import synth
from modaltune_amd.aggregators import Aggregator
from modaltune_amd.config import GIGAPATH_JSON
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sizes = synth.toy_group_sizes()
groups = {i: ["g"] * n for i, n in enumerate(sizes)}
model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3, init_seed=0, **dict(GIGAPATH_JSON, pretrained=False)).cuda()
from modaltune_amd.optim import AdamW
opt = AdamW([{"params": [p for p in model.parameters() if p.requires_grad], "lr": 1e-5}], weight_decay=0.01)
scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=2.0 ** 15)
eye = torch.eye(3, device="cuda")
slides = []
for j in range(2):
    inp = synth.synth_inputs(L, sizes, seed=1000 + j, grid=128)
    slides.append((torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda(), {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}))
model.train()
rp = model._replay
for i in range(16):
    x, c, g = slides[i % 2]
    with torch.autocast("cuda", enabled=True):
        logit = torch.cat([model(x=x, coords=c, genes=g, clinical=[], task_token=eye[t]) for t in (0, 1, 2)], dim=0)
        loss = logit.float().square().mean()
    scaler.scale(loss).backward()
    scaler.step(opt); scaler.update(); opt.zero_grad()
    print(i, "primed", rp.primed, "captures", rp.captures, "replays", rp.replays, "fallbacks", rp.eager_fallbacks, "gen", model.engine.generation, rp.gen,
          "visits", dict(rp.visits), "nosync", model._nosync_rows is not None, flush=True)
