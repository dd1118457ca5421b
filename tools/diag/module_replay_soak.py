"""Module bridge under a rotation of more bag lengths than the replay cache holds (captures, evictions, recaptures): reserved memory must
plateau and every step must stay finite.  python tools/diag/module_replay_soak.py [rounds]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from modaltune_amd import synth
from modaltune_amd.aggregators import Aggregator
from modaltune_amd.config import GIGAPATH_JSON
from modaltune_amd.optim import AdamW
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 14
sizes = synth.toy_group_sizes()
groups = {i: ["g"] * n for i, n in enumerate(sizes)}
model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3, init_seed=0, **dict(GIGAPATH_JSON, pretrained=False)).cuda()
opt = AdamW([{"params": [p for p in model.parameters() if p.requires_grad], "lr": 1e-5}], weight_decay=0.01)
scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=2.0 ** 15)
eye = torch.eye(3, device="cuda")
lengths = [900, 1300, 1700, 2100, 2500, 2900, 3300]          # 7 geometries, cache of 4
slides = {}
for L in lengths:
    inp = synth.synth_inputs(L, sizes, seed=L, grid=128)
    slides[L] = (torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda(), {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])})
model.train()
rp = model._replay
res = []
for r in range(rounds):
    for L in lengths:
        x, c, g = slides[L]
        x = x.clone()                                  # (a new slide tensor per step, as a loader hands them over)
        with torch.autocast("cuda", enabled=True):
            logit = torch.cat([model(x=x, coords=c, genes=g, clinical=[], task_token=eye[t]) for t in (0, 1, 2)], dim=0)
            loss = logit.float().square().mean()
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update(); opt.zero_grad()
        assert torch.isfinite(loss), (r, L)
    torch.cuda.synchronize()
    res.append(torch.cuda.memory_reserved() / 2**30)
    print(f"round {r}: reserved {res[-1]:.2f} GiB  captures {rp.captures} replays {rp.replays} primed {rp.primed} fallbacks {rp.eager_fallbacks} loss {float(loss.detach()):.4f}", flush=True)
assert res[-1] <= res[len(res) // 2] * 1.02 + 0.05, res
print("plateau ok")
