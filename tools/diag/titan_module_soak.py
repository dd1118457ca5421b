"""TITAN configuration through the drop-in nn.Module (3 calls per slide, torch loss / AdamW) on never-repeating bag lengths:
device memory must stay flat (leased per-call workspaces, engine._Lease)."""
import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import titan_standin
from test_titan_cpu import TITAN_JSON
from modaltune_amd import synth
from modaltune_amd.aggregators import Aggregator
import modaltune_amd.titan  # noqa: F401
dev = torch.device("cuda", 0)
sizes = synth.toy_group_sizes()
vit = titan_standin.VisionTransformer(mlp_ratio=4.0); titan_standin.init_standin(vit, 0)
groups = {i: ["g"] * n for i, n in enumerate(sizes)}
model = Aggregator.create("titan_gene_adapter", gene_group_defination=groups, **TITAN_JSON, multi_task=3, backbone=vit, backbone_impl="native")
sd = synth.synth_state_dict(model.cfg, sizes, 0)
state = {k: torch.from_numpy(v) for k, v in sd.items() if k in dict(model._params)}; state.update(vit.state_dict())
model.load_state_dict(state, strict=True); model.train()
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5)
Lmax = 5000
inp = synth.synth_inputs_titan(Lmax, sizes, seed=3, grid=96)
X = torch.from_numpy(inp["x"]).to(dev).reshape(Lmax, -1); Cc = torch.from_numpy(inp["coords"]).to(dev).reshape(Lmax, 2)
genes = {i: torch.from_numpy(a).to(dev) for i, a in enumerate(inp["genes"])}
eye = torch.eye(3, device=dev); tgt = torch.softmax(torch.randn(3, 256, device=dev), dim=1)
rnd = random.Random(1); marks = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    L = rnd.randrange(1200, Lmax)
    x, c = X[:L].unsqueeze(0), Cc[:L].unsqueeze(0)
    logits = torch.cat([model(x=x, coords=c, genes=genes, task_token=eye[t]) for t in range(3)])
    logits = logits / logits.norm(dim=-1, keepdim=True)
    loss = torch.nn.functional.kl_div(torch.log_softmax(logits, dim=1), tgt, reduction="batchmean")
    loss.backward(); opt.step(); opt.zero_grad()
    if i % 25 == 24:
        marks.append(torch.cuda.memory_reserved() / 2**30)
        print(f"step {i + 1}: loss {float(loss):.5f} alloc {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved {marks[-1]:.2f} GiB "
              f"pool { {b: [st['cap'] for st in p] for b, p in model.engine._fresh_pool.items()} }", flush=True)
assert marks[-1] <= marks[len(marks) // 2] * 1.05 + 0.25, "device memory keeps growing"
print("soak ok")
