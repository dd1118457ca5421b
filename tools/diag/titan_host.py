"""TITAN configuration: host enqueue time of one eager step against its GPU time (is the step launch-bound?)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import titan_standin
import bench
from modaltune_amd import ops, synth
from modaltune_amd.titan import NativeBackbone, TitanEngine, titan_model_config
from modaltune_amd.trainer import TrainStep
dev = torch.device("cuda", 0)
vit = titan_standin.VisionTransformer(mlp_ratio=4.0); titan_standin.init_standin(vit, 0)
sizes = synth.toy_group_sizes(6)
cfg = titan_model_config(bench.TITAN_JSON, 3, False, 6)
eng = TitanEngine(cfg, sizes, NativeBackbone(vit, dev), dev)
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0)); eng.set_stochastic(True, seed=1)
ts = TrainStep(eng); ts.set_projector(synth.projector_state(0))
for want in [int(a) for a in sys.argv[1:]] or [2048, 4096, 6144]:
    L = want + want // 15
    inp = synth.synth_inputs_titan(L, sizes, seed=want, grid=96)
    a = (torch.from_numpy(inp["x"]).to(dev).reshape(L, -1).contiguous(), torch.from_numpy(inp["coords"]).to(dev).reshape(L, 2),
         [torch.from_numpy(g).to(dev) for g in inp["genes"]], torch.from_numpy(inp["text"]).to(dev))
    for _ in range(4):
        ts.step(*a, update=True)
    torch.cuda.synchronize()
    n = 12
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        ts.step(*a, update=True)
    e1.record(); host = time.perf_counter() - t0
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    calls0 = getattr(ops, "CALLS", None)
    print(f"cells {want}: host enqueue {host / n * 1e3:.2f} ms/step, gpu span {e0.elapsed_time(e1) / n:.2f} ms/step, wall {wall / n * 1e3:.2f} ms/step", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(6):
    ts.step(*a, update=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
