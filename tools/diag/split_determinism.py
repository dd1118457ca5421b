"""Is the forward of the step bit-deterministic -- batched, and as two concurrent pass groups under different host interleavings?
    python tools/diag/split_determinism.py [L]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import synth  # noqa: E402
from modaltune_amd.config import ModelConfig  # noqa: E402
from modaltune_amd.engine import Engine  # noqa: E402
from modaltune_amd.trainer import TrainStep  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
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


def run():
    ts.step(x, inp["coords"], genes, text, update=False)
    torch.cuda.synchronize()
    return ts.last_logits.clone(), float(ts.loss), eng.store.flat_grad.clone()


def cmp(tag, a, b):
    d = (a[0] - b[0]).abs()
    print(f"{tag}: logits equal {torch.equal(a[0], b[0])} max|d| per row {[f'{float(v):.2e}' for v in d.max(dim=1).values]} "
          f"loss d {abs(a[1] - b[1]):.3e} grad rel {float((a[2] - b[2]).norm() / b[2].norm()):.2e}", flush=True)


ts.split_min_patches = 1 << 30
b0 = run()
for i in range(3):
    cmp(f"batched run {i + 1} vs 0", run(), b0)
ts.split_min_patches = 0
s0 = run()
cmp("split vs batched", s0, b0)
for d in (0.0, 0.0, 0.005, 0.02, 0.06, 0.2):
    ts._group_hook = (lambda gi, d=d: time.sleep(d) if gi == 0 else None)
    cmp(f"split delay {d}", run(), s0)
ts._group_hook = None
# one group at a time on its own stream (no concurrency at all): sync between
ts._group_hook = lambda gi: torch.cuda.synchronize()
s_serial = run()
cmp("split serialised (sync between groups) vs split", s_serial, s0)
for i in range(2):
    cmp(f"split serialised run {i + 1} vs serialised 0", run(), s_serial)

# ---- does any kernel of the forward read workspace memory it (or an earlier kernel of the SAME step) has not written?
# poison every workspace buffer with large finite garbage between two runs: the logits must not move
def poison(val16=3.0e4, val32=1.0e30):
    for key, st in eng._ws_store.items():
        for k, t in st["flat"].items():
            if t.dtype == torch.float16:
                t.fill_(val16)
            elif t.dtype == torch.float32:
                t.fill_(val32)
            else:
                t.fill_(123456)


ts._group_hook = None
for tag, smin in (("split", 0), ("batched", 1 << 30)):
    ts.split_min_patches = smin
    r0 = run()
    poison()
    cmp(f"{tag}: after poisoning the workspace vs before", run(), r0)
    poison(-2.0e4, -3.0e29)
    cmp(f"{tag}: after a second poison vs before", run(), r0)

# ---- the same question for the per-call temporaries (torch.empty inside the forward / backward): every device allocation
# comes back filled with garbage of our choosing; the logits must not depend on it
import collections
import traceback
_real_empty = torch.empty
_garbage = {"v16": 3.0e4, "v32": 1.0e30, "only": None, "sites": collections.Counter()}


def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "modaltune_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def _poisoned_empty(*a, **k):
    t = _real_empty(*a, **k)
    if t.is_cuda and t.numel() > 0:
        site = _site()
        _garbage["sites"][site] += 1
        if _garbage["only"] is None or site in _garbage["only"]:
            if t.dtype == torch.float16:
                t.fill_(_garbage["v16"])
            elif t.dtype == torch.float32:
                t.fill_(_garbage["v32"])
    return t


torch.empty = _poisoned_empty
for tag, smin in (("split", 0), ("batched", 1 << 30)):
    ts.split_min_patches = smin
    _garbage.update(v16=0.0, v32=0.0, only=None)
    r0 = run()
    _garbage.update(v16=3.0e4, v32=1.0e30)
    r1 = run()
    cmp(f"{tag}: temporaries filled with +garbage vs zeros", r1, r0)
    _garbage.update(v16=-2.0e4, v32=-3.0e29)
    cmp(f"{tag}: temporaries filled with -garbage vs zeros", run(), r0)
    if not torch.equal(r1[0], r0[0]):
        sites = sorted(_garbage["sites"])
        print("bisecting over", len(sites), "allocation sites", flush=True)
        for site in sites:
            _garbage.update(v16=3.0e4, v32=1.0e30, only={site})
            r = run()
            if not torch.equal(r[0], r0[0]):
                print("  logits depend on the initial contents of the buffer allocated at", site, flush=True)
torch.empty = _real_empty
