"""Train steps (graph replay) with evaluation passes interleaved -- as an epoch loop does (TM:187-257: train, then get_features on
train / val / test in eval mode) -- must give the same losses and weights as the same train steps alone."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import synth
from modaltune_amd.config import ModelConfig
from modaltune_amd.engine import Engine
from modaltune_amd.trainer import TrainStep
from modaltune_amd.evaluate import EmbeddingExtractor
dev = torch.device("cuda", 0)
sizes = synth.toy_group_sizes()
cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)), slide_ngrids=64, dropout=0.0, drop_path_rate=0.0)
sd = synth.synth_state_dict(cfg, sizes, 5)
def slide(L, seed):
    inp = synth.synth_inputs(L, sizes, seed, grid=64)
    return (torch.from_numpy(inp["x"]).to(dev).half().reshape(L, -1), torch.from_numpy(inp["coords"]).to(dev),
            [torch.from_numpy(a).to(dev) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(dev))
train = [slide(700, 1), slide(450, 2)]
evals = [slide(333, 3), slide(700, 4), slide(1200, 5)]
def run(interleave, stochastic):
    eng = Engine(cfg, sizes, dev); eng.load_state_dict(sd); eng.set_stochastic(stochastic, 3)
    ts = TrainStep(eng, lr=1e-3, capture_after=1); ts.set_projector(synth.projector_state(5))
    ex = EmbeddingExtractor(eng)
    losses, feats = [], []
    for it in range(10):
        losses.append(float(ts.step_graphed(*train[it % 2])))
        if interleave and it % 3 == 2:
            for e in evals:
                feats.append(ex(e[0], e[1], e[2]).float().cpu().numpy())
    return np.array(losses), eng.store.flat.clone(), feats
for stoch in (False, True):
    a, wa, _ = run(False, stoch)
    a2, wa2, _ = run(False, stoch)       # run-to-run noise of the same schedule (fp32 atomics in the weight-gradient GEMMs)
    b, wb, feats = run(True, stoch)
    noise = (float(np.abs(a - a2).max()), float((wa - wa2).abs().max()))
    diff = (float(np.abs(a - b).max()), float((wa - wb).abs().max()))
    print("stochastic", stoch, "same schedule twice: max |dloss| %.2e max |dw| %.2e" % noise, "| with evaluation interleaved: %.2e %.2e" % diff,
          "| eval finite", all(np.isfinite(f).all() for f in feats), flush=True)
    assert diff[0] <= 4 * noise[0] + 1e-6 and diff[1] <= 4 * noise[1] + 1e-6, (noise, diff)
print("interleave ok")
