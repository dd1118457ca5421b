"""victim_stress.py, second form: the aggressors are ALL ops.* launches of one full train step of a small-depth model (adapters, token
side, loss included), recorded once and replayed per op name on stream B while the victim (mt_token_mha_fwd) loops on stream A.
    python tools/diag/victim_stress2.py [L] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from modaltune_amd import ops, synth  # noqa: E402
from modaltune_amd.config import ModelConfig  # noqa: E402
from modaltune_amd.engine import Engine  # noqa: E402
from modaltune_amd.trainer import TrainStep  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
cfg = ModelConfig(depth=2, interaction_indexes=((0, 0), (1, 1)))
sizes = synth.toy_group_sizes(6)
eng = Engine(cfg, sizes, "cuda")
eng.load_state_dict(synth.synth_state_dict(cfg, sizes, 3))
ts = TrainStep(eng, lr=0.0, weight_decay=0.0, task_ids=(0, 1), text_rows=(0, 1), split_passes=False)
ts.set_projector(synth.projector_state(3))
inp = synth.synth_inputs(L, sizes, 3, grid=128)
x = torch.from_numpy(inp["x"]).cuda().half().reshape(L, -1)
genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
text = torch.from_numpy(inp["text"]).cuda()
ts.step(x, inp["coords"], genes, text, update=False)
torch.cuda.synchronize()

REC = []
KEEP = []
names = [n for n in dir(ops) if callable(getattr(ops, n)) and not isinstance(getattr(ops, n), type) and not n.startswith("_")
         and n not in ("check", "make_plan", "dropout_spec", "struct_of", "sgemm_problem", "make_dense_plan", "timer_summary", "measured_mfma_peak_tflops",
                       "dilated_attn_bwd_workspace_bytes", "alibi_dist_halves", "pool_attn_workspace_floats", "rowmap", "mfma_probe")]
orig = {n: getattr(ops, n) for n in names}


def keep(o):
    if torch.is_tensor(o):
        KEEP.append(o)
    elif isinstance(o, (list, tuple)):
        for v in o:
            keep(v)
    elif isinstance(o, dict):
        for v in o.values():
            keep(v)


depth = [0]
for n in names:
    def mk(n):
        def w(*a, **k):
            if depth[0] == 0:
                REC.append((n, a, k))
                keep(a); keep(k)
            depth[0] += 1
            try:
                return orig[n](*a, **k)
            finally:
                depth[0] -= 1
        return w
    setattr(ops, n, mk(n))
from modaltune_amd import tape as tape_mod  # noqa: E402
_new = tape_mod.Tape.new
tape_mod.Tape.new = lambda self, *s: (lambda t: (KEEP.append(t), t)[1])(_new(self, *s))
_zl = tape_mod.Tape.zeros_like
tape_mod.Tape.zeros_like = lambda self, t: (lambda z: (KEEP.append(z), z)[1])(_zl(self, t))
_empty = torch.empty
torch.empty = lambda *a, **k: (lambda t: (KEEP.append(t), t)[1])(_empty(*a, **k))
ts.step(x, inp["coords"], genes, text, update=False)
torch.cuda.synchronize()
torch.empty = _empty
for n in names:
    setattr(ops, n, orig[n])
by_name = {}
for n, a, k in REC:
    by_name.setdefault(n, []).append((a, k))
print("recorded", len(REC), "launches:", {n: len(v) for n, v in by_name.items()}, flush=True)

g = torch.Generator(device="cuda").manual_seed(1)
T, E, H = 65, 192, 12
VICTIM = os.environ.get("VICTIM", "token_mha_fwd")
q, k_, v = (torch.randn(1, T, E, generator=g, device="cuda") for _ in range(3))
out, probs = torch.empty(1, T, E, device="cuda"), torch.empty(1 * H * T * T, device="cuda")
if VICTIM == "token_mha_fwd":
    outs = [out, probs]

    def victim():
        ops.token_mha_fwd(q, k_, v, out, probs, 1, T, E, H)
elif VICTIM == "token_mha_bwd":
    ops.token_mha_fwd(q, k_, v, out, probs, 1, T, E, H)
    dout = torch.randn(1, T, E, generator=g, device="cuda")
    dq, dk, dv = (torch.empty(1, T, E, device="cuda") for _ in range(3))
    outs = [dq, dk, dv]

    def victim():
        ops.token_mha_bwd(q, k_, v, probs, dout, dq, dk, dv, 1, T, E, H)
elif VICTIM in ("gene_snn_fwd", "gene_snn_bwd"):
    # the recorded launch of the step (6 pathways: the eight-workgroups-per-pathway kernels), replayed on its own tensors; the backward
    # accumulates into the flat gradient buffer, so that is cleared in front of every run
    n_, a_, kw_ = next((n, a, k) for n, a, k in REC if n == VICTIM)
    if VICTIM == "gene_snn_fwd":
        outs = [a_[7], a_[8], a_[9]]

        def victim():
            orig[VICTIM](*a_, **kw_)
    else:
        gbuf = torch.zeros_like(a_[1])          # a gradient buffer of its own (the aggressor's launches add into the step's)
        a_ = (a_[0], gbuf) + tuple(a_[2:])
        outs = [gbuf]

        def victim():
            gbuf.zero_()
            orig[VICTIM](*a_, **kw_)
else:
    raise SystemExit("VICTIM = token_mha_fwd | token_mha_bwd | gene_snn_fwd | gene_snn_bwd")
victim()
torch.cuda.synchronize()
refs = [t.clone() for t in outs]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad = torch.zeros(2, dtype=torch.int64, device="cuda")


def trial(name, calls, iters):
    bad.zero_()
    torch.cuda.synchronize()
    ci = 0
    with torch.cuda.stream(sa):
        for i in range(iters):
            victim()
            bad[0] += (outs[0] != refs[0]).any()
            for t, r in zip(outs[1:], refs[1:]):
                bad[1] += (t != r).any()
            if calls and i % 4 == 0:
                with torch.cuda.stream(sb):
                    n, a, k = calls[ci % len(calls)]
                    orig[n](*a, **k)
                    ci += 1
    torch.cuda.synchronize()
    print(f"{name:28s} ({len(calls):3d} recorded launches): victim {VICTIM} wrong first output {int(bad[0])} / other outputs {int(bad[1])} of {iters}", flush=True)


ONLY = os.environ.get("ONLY")
trial("(alone)", [], ITERS)
if not ONLY:
    trial("(whole step in order)", [(n, a, k) for n, a, k in REC], max(ITERS, 4 * len(REC) * 3))
for n, lst in by_name.items():
    if ONLY and n != ONLY:
        continue
    trial(n, [(n, a, k) for a, k in lst], ITERS)
    if ONLY:
        for i, (a, k) in enumerate(lst):
            trial(f"{n}#{i} M={a[3]} N1={a[4]} N2={a[5]} cs={k.get('colsum') is not None}", [(n, a, k)], ITERS)
