"""Reserved-memory growth under graph LRU thrash (small model): per round allocated / reserved / segments."""
import os, sys, gc, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import synth
from modaltune_amd.config import ModelConfig
from modaltune_amd.engine import Engine
from modaltune_amd.trainer import TrainStep
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2
after = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seed, ngrids = 43, 64
sizes = synth.toy_group_sizes()
cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)), slide_ngrids=ngrids)
sd = synth.synth_state_dict(cfg, sizes, seed)
lengths = [900, 333, 1200, 640, 1029]
slides = []
for L in lengths:
    inp = synth.synth_inputs(L, sizes, seed + L, grid=ngrids)
    slides.append((torch.from_numpy(inp["x"]).cuda().half().reshape(L, -1), torch.from_numpy(inp["coords"]).cuda(),
                   [torch.from_numpy(a).cuda() for a in inp["genes"]], torch.from_numpy(inp["text"]).cuda()))
eng = Engine(cfg, sizes, "cuda"); eng.load_state_dict(sd)
ts = TrainStep(eng, lr=0.0, weight_decay=0.0, capture_after=after, graph_cache_size=size)
ts.set_projector(synth.projector_state(seed))
for rnd in range(10):
    for s in slides:
        ts.step_graphed(*s)
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(f"round {rnd}: allocated {torch.cuda.memory_allocated() >> 20} MiB reserved {torch.cuda.memory_reserved() >> 20} MiB segments {st['segment.all.current']} "
          f"replays {ts.graph_replays} eager {ts.eager_steps} live graphs {sum(len(g) for g in (ts._graphs or []))} gc objects {len(gc.get_objects())}", flush=True)
