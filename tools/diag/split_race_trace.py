"""Follow-up of split_race_hunt.py: trace every ops.* call of pass group 1's FORWARD (the group whose logits move), keep its
per-call tensors alive, and on a run whose logits differ from the reference run report the first call (in launch order) that
touches a per-call tensor (tape / adapter temporaries: written once, never reused) whose contents differ.
    SWEEP_MS=40 python tools/diag/split_race_trace.py [L] [runs]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import ops, synth  # noqa: E402
from modaltune_amd.config import ModelConfig  # noqa: E402
from modaltune_amd.engine import Engine  # noqa: E402
from modaltune_amd.trainer import TrainStep  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
RUNS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
SWEEP = int(os.environ.get("SWEEP_MS", "40"))
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
ts.split_min_patches = 0

TRACE = None          # list of (name, [tensors]) while group 1's forward is being enqueued


def tensors_of(obj, out):
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            tensors_of(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            tensors_of(o, out)


for name in dir(ops):
    fn = getattr(ops, name)
    if callable(fn) and not isinstance(fn, type) and not name.startswith("_") and getattr(fn, "__module__", "") == ops.__name__ or name in (
            "gemm_nt", "layernorm_fwd", "sgemm_multi", "axpy", "copy_rows", "token_mha_fwd", "extract_attn_fwd", "inject_attn_fwd", "gene_snn_fwd"):
        def mk(fn, name):
            def w(*a, **k):
                if TRACE is not None:
                    ts_ = []
                    tensors_of(a, ts_); tensors_of(k, ts_)
                    TRACE.append((name, ts_))
                r = fn(*a, **k)
                if TRACE is not None and name == "token_mha_fwd":
                    # snapshots right behind the kernel, on its stream: inputs as the NEXT kernel would see them, and the output
                    TRACE.append((name + ".snap", [t.clone() for t in ts_]))
                return r
            return w
        if callable(fn) and not isinstance(fn, type):
            setattr(ops, name, mk(fn, name))

# sgemm problems hold raw pointers (ctypes structs), not tensors: keep the tensors by tracing the tape-level allocations instead
from modaltune_amd import tape as tape_mod  # noqa: E402
_new = tape_mod.Tape.new


def traced_new(self, *shape):
    t = _new(self, *shape)
    if TRACE is not None:
        TRACE.append(("tape.new" + str(tuple(shape)), [t]))
    return t


tape_mod.Tape.new = traced_new
_fwd = eng.forward


def traced_forward(*a, **k):
    global TRACE
    g1 = k.get("site_group") == 2
    if g1:
        TRACE = []
    try:
        return _fwd(*a, **k)
    finally:
        if g1:
            traced_forward.last, TRACE = TRACE, None


eng.forward = traced_forward
ws_ptrs = None


def in_workspace(t):
    p = t.untyped_storage().data_ptr()
    return p in ws_ptrs


def run(delay=0.0):
    ts._group_hook = (lambda gi: time.sleep(delay) if gi == 0 else None)
    ts.step(x, inp["coords"], genes, text, update=False)
    torch.cuda.synchronize()
    return traced_forward.last


run(); run()
ws_ptrs = {t.untyped_storage().data_ptr() for st in eng._ws_store.values() for t in st["flat"].values()}
ref_trace = run()
ref = [(name, [None if in_workspace(t) else t.clone() for t in tl]) for name, tl in ref_trace]
ref_logits = ts.last_logits.clone()
print("calls traced in group 1's forward:", len(ref), flush=True)
bad = 0
for it in range(RUNS):
    tr = run((it % SWEEP) * 1e-3)
    if torch.equal(ts.last_logits, ref_logits):
        continue
    bad += 1
    print(f"run {it} (delay {(it % SWEEP)} ms): logits differ {[f'{float(v):.2e}' for v in (ts.last_logits - ref_logits).abs().max(dim=1).values]}", flush=True)
    assert len(tr) == len(ref)
    shown = 0
    for ci, ((name, tl), (rname, rl)) in enumerate(zip(tr, ref)):
        assert name == rname
        for ai, (t, r) in enumerate(zip(tl, rl)):
            if r is None or t.shape != r.shape:
                continue
            if not torch.equal(t, r):
                ne = (t != r)
                print(f"   call {ci} {name} arg {ai} shape {tuple(t.shape)} {t.dtype}: {int(ne.sum())} of {t.numel()} differ, max|d| {float((t.float() - r.float()).abs().max()):.3e}"
                      f", first idx {ne.reshape(-1).nonzero()[0].item()}", flush=True)
                if name.startswith("token_mha_fwd") and t.dim() == 3:
                    idx = ne[0].nonzero()
                    print("      rows", torch.unique(idx[:, 0]).tolist(), "cols", torch.unique(idx[:, 1]).tolist(), flush=True)
                shown += 1
        if shown >= 12:
            break
    if bad >= 2:
        break
print(f"{bad} of {it + 1} runs differed", flush=True)
