"""Soak run: N train steps over a rotation of bag lengths (graph capture, LRU eviction, eager visits mixed), watching the loss,
the loss scale, the skipped steps and the allocator's high-water mark for drift."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import synth
from modaltune_amd.config import ModelConfig
from modaltune_amd.engine import Engine
from modaltune_amd.trainer import TrainStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
growth = int(sys.argv[2]) if len(sys.argv) > 2 else 2000      # GradScaler growth interval (small: overflow / skip / back-off get exercised)
dev = torch.device("cuda", 0)
cfg = ModelConfig(); sizes = synth.toy_group_sizes()
eng = Engine(cfg, sizes, dev); eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0)); eng.set_stochastic(True, 7)
ts = TrainStep(eng, graph_cache_size=4, growth_interval=growth); ts.set_projector(synth.projector_state(0))
lengths = [3000, 4096, 2500, 5000, 3000, 3500, 4096, 2800, 3000, 4500, 6000, 3000]      # 9 distinct lengths > the 4-entry graph LRU
slides = {}
for L in set(lengths):
    inp = synth.synth_inputs(L, sizes, seed=L, grid=128)
    slides[L] = (torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1).contiguous(), torch.from_numpy(inp["coords"]).to(dev),
                 [torch.from_numpy(a).to(dev) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(dev))
t0 = time.time(); marks = []
for i in range(steps):
    L = lengths[i % len(lengths)]
    loss = ts.step_graphed(*slides[L])
    if i % 50 == 49 or i == steps - 1:
        v = ts.loss_value()
        marks.append((i + 1, v, float(ts.scale), int(ts.step_dev), torch.cuda.max_memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
        print(f"step {i + 1}: loss {v:.5f} scale {float(ts.scale):.3g} optimiser steps {int(ts.step_dev)} replays {ts.graph_replays} eager {ts.eager_steps} "
              f"peak alloc {marks[-1][4]:.2f} GiB reserved {marks[-1][5]:.2f} GiB  ({time.time() - t0:.1f} s)", flush=True)
assert all(m[1] == m[1] and m[1] < 10 for m in marks), "loss is not finite"
assert marks[-1][5] <= marks[min(2, len(marks) - 1)][5] * 1.10 + 0.5, "reserved memory keeps growing"
print("soak ok")
