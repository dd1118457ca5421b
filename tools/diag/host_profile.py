"""cProfile of the host side of one eager train step (where the Python time of the launch schedule goes)."""
import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import synth
from modaltune_amd.config import ModelConfig
from modaltune_amd.engine import Engine
from modaltune_amd.trainer import TrainStep
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
cfg = ModelConfig(); sizes = synth.toy_group_sizes()
eng = Engine(cfg, sizes, dev); eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0)); eng.set_stochastic(True, 1)
ts = TrainStep(eng); ts.set_projector(synth.projector_state(0))
inp = synth.synth_inputs(L, sizes, seed=L, grid=128)
args = (torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1).contiguous(), torch.from_numpy(inp["coords"]).to(dev),
        [torch.from_numpy(a).to(dev) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(dev))
for _ in range(3):
    ts.step(*args)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    ts.step(*args)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
