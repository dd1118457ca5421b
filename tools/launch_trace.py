"""Ordered list of the C-ABI launches (and torch fills / copies) of one eager train step: which small launches exist, in what
order, from which engine phase -- the input for token-side fusion work.  Usage: python tools/launch_trace.py [L]"""
import os, sys, collections, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from modaltune_amd import synth, ops, _lib
from modaltune_amd.config import ModelConfig
from modaltune_amd.engine import Engine
from modaltune_amd.trainer import TrainStep

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
cfg = ModelConfig()
sizes = synth.toy_group_sizes()
eng = Engine(cfg, sizes, dev)
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed=0))
eng.set_stochastic(True, seed=1)
ts = TrainStep(eng)
ts.set_projector(synth.projector_state(0))
inp = synth.synth_inputs(L, sizes, seed=L, grid=128)
args = (torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1).contiguous(), inp["coords"],
        [torch.from_numpy(a).to(dev) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(dev))
for _ in range(2):
    ts.step(*args)
torch.cuda.synchronize()

lib = _lib.load()
trace = []


def where():
    for fr in reversed(traceback.extract_stack()[:-3]):
        fn = os.path.basename(fr.filename)
        if fn in ("engine.py", "trainer.py", "tape.py"):
            return f"{fn[:-3]}.{fr.name}:{fr.lineno}"
    return "?"


class Spy:
    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        f = getattr(self._real, name)
        if not name.startswith("mt_"):
            return f

        def call(*a):
            ints = [int(x) for x in a if isinstance(x, int) and 0 < x < 10 ** 6][:4]
            trace.append((name, tuple(ints), where()))
            return f(*a)
        return call


orig_load = _lib.load
spy = Spy(lib)
_lib.load = lambda: spy
ts.step(*args)
torch.cuda.synchronize()
_lib.load = orig_load
print(f"{len(trace)} C-ABI launches in one step")
cnt = collections.Counter((n, w.split(':')[0]) for n, s, w in trace)
for (n, w), c in cnt.most_common(80):
    print(f"{c:5d}  {n:28s} {w}")
if os.environ.get("MT_TRACE_FULL"):
    for n, s, w in trace:
        print(n, s, w)
