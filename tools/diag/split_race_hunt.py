"""Hunt for a rare nondeterminism of the two-stream pass-group step: run it N times, compare every saved activation of both groups
with the first run's (the workspace keeps them all for the backward), report the FIRST buffer in forward order that differs.
    python tools/diag/split_race_hunt.py [L] [runs] [split|batched]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import synth  # noqa: E402
from modaltune_amd.config import ModelConfig  # noqa: E402
from modaltune_amd.engine import Engine  # noqa: E402
from modaltune_amd.trainer import TrainStep  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
RUNS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
MODE = sys.argv[3] if len(sys.argv) > 3 else "split"
seed = 91
sizes = synth.toy_group_sizes(6)
cfg = ModelConfig()
eng = Engine(cfg, sizes, "cuda")
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed))
ts = TrainStep(eng, lr=0.0, weight_decay=0.0)
ts.set_projector(synth.projector_state(seed))
inp = synth.synth_inputs(L, sizes, seed, grid=128)
x = torch.from_numpy(inp["x"]).cuda().half().reshape(L, -1)
genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
text = torch.from_numpy(inp["text"]).cuda()
ts.split_min_patches = 0 if MODE == "split" else 1 << 30

order = ["x0"]
for l in range(cfg.depth):
    order += [f"hin{l}", f"st1_{l}", f"qkv{l}", f"obr{l}", f"lsebr{l}", f"lsetot{l}", f"stin_{l}", f"hmid{l}", f"st2_{l}", f"a1_{l}", f"stf_{l}"]
    for i, (la, lb) in enumerate(cfg.interaction_indexes):
        if lb == l:
            order.append(f"hout{i}")


def run():
    ts.step(x, inp["coords"], genes, text, update=False)
    torch.cuda.synchronize()


def stores():
    return {key: {k: st["flat"][k] for k in order if k in st["flat"]} for key, st in eng._ws_store.items()}


run(); run()
ref = {key: {k: t.clone() for k, t in d.items()} for key, d in stores().items()}
ref_logits = ts.last_logits.clone()
bad_runs = 0
import time
SWEEP = os.environ.get("SWEEP_MS")          # e.g. "40": the host waits (it % 40) ms between the two groups' enqueues
for it in range(RUNS):
    if SWEEP:
        ts._group_hook = (lambda gi, d=(it % int(SWEEP)) * 1e-3: time.sleep(d) if gi == 0 else None)
    run()
    if torch.equal(ts.last_logits, ref_logits):
        continue
    bad_runs += 1
    print(f"run {it}: logits differ, max|d| per row {[f'{float(v):.2e}' for v in (ts.last_logits - ref_logits).abs().max(dim=1).values]}", flush=True)
    for key, d in stores().items():
        first = None
        for k in order:
            if k not in d:
                continue
            a, b = d[k], ref[key][k]
            if not torch.equal(a, b):
                ne = (a != b)
                idx = ne.nonzero()
                n = int(ne.sum())
                if first is None:
                    first = k
                    width = {"x0": 768}.get(k, None)
                    flat_idx = idx.reshape(-1)
                    print(f"  store {key}: FIRST differing buffer {k}: {n} of {a.numel()} elements, flat index range {int(flat_idx.min())}..{int(flat_idx.max())}"
                          f" max|d| {float((a.float() - b.float()).abs().max()):.3e}", flush=True)
                    # row / column picture for [M, C]-shaped buffers
                    for C in (768, 2304, 3072, 48, 16, 2):
                        if a.numel() % C == 0:
                            rows = torch.unique(flat_idx // C)
                            cols = torch.unique(flat_idx % C)
                            print(f"     as [.., {C}]: {rows.numel()} rows {rows[:12].tolist()}..{rows[-3:].tolist()}, {cols.numel()} cols {cols[:8].tolist()}..{cols[-3:].tolist()}", flush=True)
                else:
                    print(f"  store {key}: then {k}: {n} elements", flush=True)
                    break
    if bad_runs >= 3:
        break
print(f"{MODE}: {bad_runs} of {it + 1} runs differed from the reference", flush=True)
