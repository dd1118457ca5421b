"""Train-step throughput with the inputs coming from HOST memory through CasePrefetcher (PCIe-inclusive rate), next to
the resident-input rate bench.py reports.  One slide per step, L patches x 1536 fp32 on the host (the reference format)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import data, synth
from modaltune_amd.config import ModelConfig
from modaltune_amd.engine import Engine
from modaltune_amd.trainer import TrainStep

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
cfg = ModelConfig()
sizes = synth.toy_group_sizes(6)
eng = Engine(cfg, sizes, dev)
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))
ts = TrainStep(eng)
ts.set_projector(synth.projector_state(0))
host = []
for j in range(4):
    inp = synth.synth_inputs(L, sizes, seed=1000 + j, grid=128 if L <= 128 * 128 else 512)
    host.append(dict(features=torch.from_numpy(inp["x"]).reshape(L, -1), coords=inp["coords"],
                     genes=[torch.from_numpy(a) for a in inp["genes"]], text=torch.from_numpy(inp["text"]), case_id=j))


def stream(n):
    for i in range(n):
        yield host[i % len(host)]


for s in data.CasePrefetcher(stream(4)):          # warm-up: eager steps + graph capture
    ts.step_graphed(s.x, s.coords, s.genes, s.text)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in data.CasePrefetcher(stream(steps)):
    ts.step_graphed(s.x, s.coords, s.genes, s.text)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"metric": "slides/sec (train step), inputs streamed from host memory (fp32 features, PCIe-inclusive)",
                  "value": 1.0 / dt, "ms_per_step": dt * 1e3, "patches": L, "host_bytes_per_slide": L * 1536 * 4}))
