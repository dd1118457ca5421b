"""Throughput of the eval / embedding-extraction pass: 3 task passes forward-only, one slide."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import synth
from modaltune_amd.config import ModelConfig, flops_per_slide_step
from modaltune_amd.engine import Engine
from modaltune_amd.evaluate import EmbeddingExtractor

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
cfg = ModelConfig()
sizes = synth.toy_group_sizes(6)
eng = Engine(cfg, sizes, dev)
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))
ex = EmbeddingExtractor(eng)
inp = synth.synth_inputs(L, sizes, seed=1000, grid=128 if L <= 128 * 128 else 512)
x = torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1).contiguous()
genes = [torch.from_numpy(a).to(dev) for a in inp["genes"]]
for _ in range(3):
    out = ex(x, inp["coords"], genes)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = ex(x, inp["coords"], genes)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
fl = 3 * flops_per_slide_step(L, eng.T)["fwd_pass"] - 2 * flops_per_slide_step(L, eng.T)["patch"]
print(json.dumps({"metric": "slides/sec (eval forward, 3 task passes)", "value": 1.0 / dt, "ms_per_slide": dt * 1e3, "patches": L,
                  "tflops": fl / 1e12, "mfma_frac": fl / dt / 2.5e15, "launch": "hipGraph replay"}))
