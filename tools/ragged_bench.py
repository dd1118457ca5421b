"""Eager train steps over bags of different lengths (the DataLoader case: one slide per step, ragged L)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import synth
from modaltune_amd.config import ModelConfig
from modaltune_amd.engine import Engine
from modaltune_amd.trainer import TrainStep

dev = torch.device("cuda", 0)
cfg = ModelConfig()
sizes = synth.toy_group_sizes()
eng = Engine(cfg, sizes, dev)
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))
eng.set_stochastic(True, seed=1)
ts = TrainStep(eng)
ts.set_projector(synth.projector_state(0))
lengths = [9000, 10000, 7000, 9500, 10000, 8000, 6000, 10000, 9000, 7500, 10000, 5000]
slides = {}
for L in sorted(set(lengths)):
    inp = synth.synth_inputs(L, sizes, seed=L, grid=128)
    slides[L] = (torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1).contiguous(), inp["coords"],
                 [torch.from_numpy(a).to(dev) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(dev))
for rep in range(2):
    for L in lengths:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ts.step(*slides[L])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"rep {rep} L={L:6d}: {dt*1e3:7.1f} ms  loss {float(ts.loss):.4f}  reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB "
              f"allocated {torch.cuda.memory_allocated()/2**30:.1f} GiB")
