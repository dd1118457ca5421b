"""GPU parity of the full HIP train step (3 task passes, loss, backward) against golden vectors produced by the
reference (tests/golden/model_*.npz) and against the CPU oracle.  Tolerance: 1e-3 relative on logits / loss
(BASELINE.json north_star); gradient norms and full tensors are checked at 1e-2 (four named gene-encoder tensors at
2.5e-2 / 3.5e-2: GRAD_TOL_NAMED; fp16 operands, scaled fp16 gradient stream)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from modaltune_amd import synth  # noqa: E402
from modaltune_amd.config import GIGAPATH_JSON, ModelConfig  # noqa: E402


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


# Gradient tolerances: 1 % on every norm, except the gene-encoder tensors whose gradients are sums over the gene tokens that cancel
# almost completely (norm ~1e-4 of the largest): the forward's fp16 operand rounding ALONE -- exact gradients of the network
# evaluated at fp16-rounded activations, no HIP kernel involved -- moves exactly these by 0.8-2.6 % in the fp64 oracle
# (tests/test_grad_rounding_cpu.py), and the HIP gradients lie 4 x closer to that emulation than to the fp64 golden
# (test_gradient_exceptions_follow_the_forward_fp16_rounding below).
GRAD_TOL_NAMED = {"gene_encoder.mlp_mixer.0.0.fn.0.bias": 2.5e-2, "gene_encoder.mlp_mixer.1.0.fn.0.bias": 2.5e-2,
                  "gene_encoder.mlp_mixer.2.0.fn.0.bias": 2.5e-2, "gene_encoder.pathway_compression.weight": 3.5e-2}


def _build(path):
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    g = np.load(path)
    L, depth, seed, ngrids = int(g["L"]), int(g["depth"]), int(g["seed"]), int(g["ngrids"])
    sizes = [int(s) for s in g["sizes"]]
    cfg = ModelConfig(depth=depth, interaction_indexes=tuple(tuple(int(i) for i in p) for p in g["inter"]), slide_ngrids=ngrids,
                      clinical=bool(int(g["clinical"])) if "clinical" in g.files else False,
                      token_agg=str(g["token_agg"]) if "token_agg" in g.files else "sum",
                      multi_task=int(g["multi_task"]) if "multi_task" in g.files else 3,
                      **(json.loads(str(g["extra_cfg"])) if "extra_cfg" in g.files else {}))
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed))
    ts = TrainStep(eng)
    ts.set_projector(synth.projector_state(seed))
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    return g, cfg, eng, ts, inp


@pytest.mark.parametrize("name", ["L37_d3", "L1500_d3", "L512_d12", "L37_d3_clin", "L37_d3_clin_cat", "L37_d3_cat",
                                  "L37_d3_pan", "L129_d3_pan", "L37_d3_single", "L37_d3_cls", "L37_d6_pre_gp", "L37_d3_clin_cls"])
def test_train_step_matches_reference_golden(golden_dir, name):
    """(pan: the pan-cancer trainer's shape, one-hot width 4 with task ids 0..2, train_modaltune_pancancer.py:50-134,537-542;
    single: multi_task = 1, one model call, the [1, O] logits against all three text rows.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(golden_dir, f"model_{name}.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    g, cfg, eng, ts, inp = _build(path)
    eng.collect_taps = True
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    clin = torch.from_numpy(inp["clinical"]).cuda() if cfg.clinical else None
    loss = ts.step(x, inp["coords"], genes, torch.from_numpy(inp["text"]), update=False, clinical=clin)
    torch.cuda.synchronize()
    logits = ts.last_logits.cpu().numpy()
    report = {}
    for i in range(len(cfg.interaction_indexes)):
        for t in range(logits.shape[0]):
            report[f"cls{i}/t{t}"] = _rel(eng.taps[f"cls{i}"][t].cpu().numpy(), g[f"f64_tap/task{t}/cls{i}"].reshape(-1))
            report[f"c{i}/t{t}"] = _rel(eng.taps[f"c{i}"][t].cpu().numpy(), g[f"f64_tap/task{t}/c{i}"][0])
    report["logits"] = _rel(logits, g["f64_logits"])
    report["loss"] = abs(float(loss) - float(g["f64_loss"])) / abs(float(g["f64_loss"]))
    print(name, {k: f"{v:.2e}" for k, v in report.items()})
    assert report["logits"] < 1e-3, report
    assert report["loss"] < 1e-3, report
    assert int(ts.found_inf) == 0
    grads = ts.unscaled_grads()
    names = [str(n) for n in g["f64_grad_names"]]
    ours = np.array([float(grads[n].double().norm()) for n in names])
    ref = g["f64_grad_norms"]
    # Tolerances (measured, tools/diag/grad_errors.py: every norm within 1 % and every full tensor within 0.4 % on all nine
    # fixtures, with two exceptions inside the gene encoder).  The exceptions do NOT come from the fp16 gradient stream: their
    # error is the same to three digits at loss scales 2^10 ... 2^24 -- it is the forward's fp16 operand rounding (activations
    # off by ~3e-4, as under the reference's own autocast) seen through gradient sums that cancel almost completely.
    loose = GRAD_TOL_NAMED
    bad = [(n, o, r) for n, o, r in zip(names, ours, ref) if abs(o - r) > loose.get(n, 1e-2) * r + 1e-6 * ref.max()]
    assert not bad, bad[:10]
    for k in g.files:
        if k.startswith("f64_grad/"):        # full tensors: relative L2 error
            key = k[len("f64_grad/"):]
            ours_k = grads[key].double().cpu().numpy()
            err = np.linalg.norm(ours_k - g[k]) / (np.linalg.norm(g[k]) + 1e-300)
            assert err < (3.5e-2 if key == "gene_encoder.pathway_compression.weight" else 1e-2), (k, err)


@pytest.mark.parametrize("name", ["L37_d3", "L1500_d3", "L37_d3_clin_cat"])
def test_pass_groups_on_two_streams_match_the_batched_step(golden_dir, name):
    """Round 5: TrainStep runs the task passes of a long bag as two concurrent groups (B = 2 and B = 1 on two HIP streams, own
    workspaces / tapes / gradient buffers, loss + backward per group, streams meeting in front of the optimiser).  Forced on at
    fixture size: same logits and loss as the batched B = 3 pass (the per-row arithmetic is identical), gradients equal to the
    rounding of two accumulation orders, the reference golden's tolerances hold, and the hipGraph replay of the forked schedule
    reproduces the eager one."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(golden_dir, f"model_{name}.npz")
    g, cfg, eng, ts, inp = _build(path)
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"])
    clin = torch.from_numpy(inp["clinical"]).cuda() if cfg.clinical else None
    ts.split_min_patches = 1 << 30                    # batched
    ts.step(x, inp["coords"], genes, text, update=False, clinical=clin)
    torch.cuda.synchronize()
    l0, loss0, g0 = ts.last_logits.clone(), float(ts.loss), {k: v.clone() for k, v in ts.unscaled_grads().items()}
    ts.split_min_patches = 0                          # two groups
    assert ts._split_now(int(g["L"]))
    ts.step(x, inp["coords"], genes, text, update=False, clinical=clin)
    torch.cuda.synchronize()
    l1, loss1, g1 = ts.last_logits.clone(), float(ts.loss), ts.unscaled_grads()
    assert ts._pass_streams is not None and torch.equal(l0, l1) and abs(loss0 - loss1) <= 1e-6 * abs(loss0)
    for k in g0:
        n = float(g0[k].norm())
        assert float((g0[k] - g1[k]).norm()) <= 2e-3 * n + 1e-7 * max(float(v.norm()) for v in g0.values()), k
    assert _rel(l1.cpu().numpy(), g["f64_logits"]) < 1e-3
    names = [str(n) for n in g["f64_grad_names"]]
    ours = np.array([float(g1[n].double().norm()) for n in names])
    ref = g["f64_grad_norms"]
    bad = [(n, o, r) for n, o, r in zip(names, ours, ref) if abs(o - r) > GRAD_TOL_NAMED.get(n, 1e-2) * r + 1e-6 * ref.max()]
    assert not bad, bad[:5]
    # hipGraph replay of the forked schedule (lr 0: the weights stay, every visit must reproduce the same loss)
    ts.set_lr(0.0); ts.wd = 0.0
    losses = []
    for _ in range(5):
        ts.step_graphed(x, inp["coords"], genes, text, clinical=clin)
        torch.cuda.synchronize()
        losses.append(float(ts.loss))
    assert ts.graph_replays >= 2 and max(losses) - min(losses) <= 1e-6 * abs(loss0) and abs(losses[-1] - loss0) <= 1e-6 * abs(loss0)


def test_pass_groups_with_per_bucket_joins_match_the_batched_step(golden_dir):
    """Round 6: the data-parallel form of the pass groups -- both groups' backwards advance stage by stage, each bucket's ranges of the
    two gradient sets are summed at ITS join and (world > 1) handed to the reducer there -- forced on one GPU: same logits and loss
    as the batched pass, gradients within the accumulation-order tolerance, three early buckets, and the captured step (cut at the
    three joins: four segments + nothing else) replays the eager loss."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(golden_dir, "model_L1500_d3.npz")
    g, cfg, eng, ts, inp = _build(path)
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"])
    ts.split_min_patches = 1 << 30
    ts.step(x, inp["coords"], genes, text, update=False)
    torch.cuda.synchronize()
    l0, loss0, g0 = ts.last_logits.clone(), float(ts.loss), {k: v.clone() for k, v in ts.unscaled_grads().items()}
    ts.split_min_patches, ts.force_bucket_joins = 0, True
    for _ in range(2):
        ts.step(x, inp["coords"], genes, text, update=False)
        torch.cuda.synchronize()
        l1, loss1, g1 = ts.last_logits.clone(), float(ts.loss), ts.unscaled_grads()
        assert ts.buckets_started_early == 3 and torch.equal(l0, l1) and abs(loss0 - loss1) <= 1e-6 * abs(loss0)
        for k in g0:
            assert float((g0[k] - g1[k]).norm()) <= 2e-3 * float(g0[k].norm()) + 1e-7 * max(float(v.norm()) for v in g0.values()), k
    ts.set_lr(0.0); ts.wd = 0.0
    losses = []
    for _ in range(5):
        ts.step_graphed(x, inp["coords"], genes, text)
        torch.cuda.synchronize()
        losses.append(float(ts.loss))
    assert ts.graph_replays >= 2 and max(len(sg) for sg in ts._graphs) == 4
    assert max(abs(v - loss0) for v in losses) <= 1e-6 * abs(loss0), (losses, loss0)
    assert torch.equal(ts.last_logits, l0)


def test_two_task_passes_as_two_single_pass_groups_match_the_batched_step(golden_dir):
    """B = 2 (two task ids): the groups are (0, 1) and (1, 2), BOTH of one pass -- the engine keys its workspace storage on the pass
    count, so the two concurrent groups need storage of their own (they used to share hin / qkv / dh / scratch and
    corrupt each other silently).  Same logits, loss and gradients as the batched B = 2 pass, eager and replayed."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.trainer import TrainStep
    path = os.path.join(golden_dir, "model_L1500_d3.npz")
    g, cfg, eng, _, inp = _build(path)
    ts = TrainStep(eng, task_ids=(0, 2), text_rows=(0, 3))
    ts.set_projector(synth.projector_state(int(g["seed"])))
    assert ts._groups == [(0, 1), (1, 2)] and ts._group_slots == [0, 1]
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"])
    ts.split_min_patches = 1 << 30
    ts.step(x, inp["coords"], genes, text, update=False)
    torch.cuda.synchronize()
    l0, loss0, g0 = ts.last_logits.clone(), float(ts.loss), {k: v.clone() for k, v in ts.unscaled_grads().items()}
    ts.split_min_patches = 0
    assert ts._split_now(int(g["L"]))
    for _ in range(3):                                # (repeated: an overlap that corrupts shows up as run-to-run noise too)
        ts.step(x, inp["coords"], genes, text, update=False)
        torch.cuda.synchronize()
        l1, loss1, g1 = ts.last_logits.clone(), float(ts.loss), ts.unscaled_grads()
        assert torch.equal(l0, l1) and abs(loss0 - loss1) <= 1e-6 * abs(loss0)
        for k in g0:
            assert float((g0[k] - g1[k]).norm()) <= 2e-3 * float(g0[k].norm()) + 1e-7 * max(float(v.norm()) for v in g0.values()), k
    ws_a, ws_b = eng._workspace(1, int(g["L"]), slot=0), eng._workspace(1, int(g["L"]), slot=1)
    assert ws_a["dh"].data_ptr() != ws_b["dh"].data_ptr()
    ts.set_lr(0.0); ts.wd = 0.0
    losses = []
    for _ in range(5):
        ts.step_graphed(x, inp["coords"], genes, text)
        torch.cuda.synchronize()
        losses.append(float(ts.loss))
    assert ts.graph_replays >= 2 and max(abs(v - loss0) for v in losses) <= 1e-6 * abs(loss0)


def test_pass_groups_draw_their_own_masks_and_stay_consistent_in_train_mode(golden_dir):
    """With Dropout / DropPath on, each group draws its own masks (site groups) and its backward regenerates them: with the step
    counter pinned the step repeats, and the analytic gradient predicts the loss change along itself, as for the batched pass."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(golden_dir, "model_L1500_d3.npz")
    g, cfg, eng, ts, inp = _build(path)
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"]).cuda()
    eng.set_stochastic(True, seed=77)
    ts.split_min_patches = 0

    def run(step_no):
        eng.rng[2] = step_no - 1                 # TrainStep.step advances the counter first
        return float(ts.step(x, inp["coords"], genes, text, update=False))
    l5 = run(5)
    grad = eng.store.flat_grad.clone() / float(ts.scale)
    assert run(5) == l5 and ts._pass_streams is not None
    gn = float(grad.norm())
    base = eng.store.flat.clone()
    eps = 2e-2 * l5 / gn
    vals = []
    for sgn in (+1.0, -1.0):
        eng.store.flat.copy_(base + sgn * eps * grad / gn)
        eng.refresh_trainable_caches()
        vals.append(run(5))
    eng.store.flat.copy_(base)
    eng.refresh_trainable_caches()
    assert abs((vals[0] - vals[1]) / (2 * eps) / gn - 1.0) < 0.05, (vals, l5, gn)
    ts.split_min_patches = 1 << 30               # the batched pass of the same step draws other masks (site group 0)
    assert run(5) != l5


def test_gradient_exceptions_follow_the_forward_fp16_rounding(golden_dir):
    """the named gradient exceptions of `test_train_step_matches_reference_golden` (three gene-encoder tensors whose
    gradients are sums that cancel almost completely) are measured against the oracle run with the patch-row products rounded to
    fp16 as under the reference's own autocast (oracle F16_PATCH_OPERANDS; tests/test_grad_rounding_cpu.py holds the CPU half):
    the HIP gradient of those tensors lies closer to that emulation than to the fp64 golden."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import test_oracle_golden as TOG
    from oracle import modaltune_oracle as O
    path = os.path.join(golden_dir, "model_L37_d3.npz")
    g, cfg, eng, ts, inp = _build(path)
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    ts.step(x, inp["coords"], genes, torch.from_numpy(inp["text"]), update=False)
    torch.cuda.synchronize()
    hip = {k: v.double().cpu() for k, v in ts.unscaled_grads().items()}
    exact = TOG._run_model_case(path, torch.float64)[4]
    O.F16_PATCH_OPERANDS = True
    try:
        emu = TOG._run_model_case(path, torch.float64)[4]
    finally:
        O.F16_PATCH_OPERANDS = False
    rows = []
    for k in hip:
        n = float(exact[k].norm())
        if n < 1e-12:
            continue
        rows.append((k, float((hip[k] - exact[k]).norm()) / n, float((hip[k] - emu[k]).norm()) / n, float((emu[k] - exact[k]).norm()) / n))
    rows.sort(key=lambda r: -r[1])
    for r in rows[:8]:
        print("%-58s hip-exact %.3e  hip-emulated %.3e  emulated-exact %.3e" % r)
    # every tensor whose HIP gradient is more than 0.5 % (relative L2) off the fp64 golden is a gene-encoder tensor, and its HIP
    # gradient agrees with the fp16-operand emulation at least twice as well as with the golden: the deviation IS the forward rounding
    off = [r for r in rows if r[1] > 8e-3]
    assert off and all(r[0] in GRAD_TOL_NAMED for r in off), off[:5]
    assert all(r[2] < 0.5 * r[1] for r in off), off[:5]
    gene = [r for r in rows if r[0].startswith("gene_encoder.") and r[1] > 4e-3]
    assert gene and all(r[2] < 0.6 * r[1] for r in gene), gene[:8]


@pytest.mark.parametrize("L", [1, 2, 7, 63, 129])
def test_tiny_bags_against_the_oracle(L):
    """Edge sizes the fixtures do not hold: a single patch (N = 2: every dilated branch degenerates to one or two keys,
    most of each sparse sequence is padding), bags below one 64-key tile, one row past a tile.  Reference: the fp64
    oracle (pinned to the reference's golden vectors in test_oracle_golden.py) on the same seeded inputs."""
    from modaltune_amd.config import segment_lengths
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    from oracle import modaltune_oracle as O
    seed, ngrids = 300 + L, 16
    sizes = synth.toy_group_sizes()
    cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)), slide_ngrids=ngrids)
    cfg.validate()
    sd_np = synth.synth_state_dict(cfg, sizes, seed)
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    # oracle (fp64, CPU)
    F64 = torch.float64
    sd = {k: torch.from_numpy(v).to(F64) for k, v in sd_np.items()}
    psd = {k: torch.from_numpy(v).to(F64) for k, v in synth.projector_state(seed).items()}
    ref_logits, ref_loss, ref_grads = O.train_step_loss_and_grads(
        sd, cfg, synth.trainable_keys(cfg, sizes), torch.from_numpy(inp["x"]).to(F64), torch.from_numpy(inp["coords"]).to(F64),
        [torch.from_numpy(a).to(F64) for a in inp["genes"]], torch.from_numpy(inp["text"]).to(F64), psd, segment_lengths())
    # HIP
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(sd_np)
    ts = TrainStep(eng)
    ts.set_projector(synth.projector_state(seed))
    loss = ts.step(torch.from_numpy(inp["x"]).cuda(), inp["coords"], [torch.from_numpy(a).cuda() for a in inp["genes"]],
                   torch.from_numpy(inp["text"]), update=False)
    torch.cuda.synchronize()
    assert int(ts.found_inf) == 0
    assert _rel(ts.last_logits.cpu().numpy(), ref_logits.numpy()) < 1e-3
    assert abs(float(loss) - float(ref_loss)) < 1e-3 * abs(float(ref_loss))
    grads = ts.unscaled_grads()
    ref_max = max(float(v.norm()) for v in ref_grads.values())
    bad = []
    for n, r in ref_grads.items():
        o, rn = float(grads[n].double().norm()), float(r.norm())
        # (L = 1: softmax over a single patch is exactly 1, so the extractor's query-side gradients are exactly zero in
        # fp64 and rounding noise, ~1e-5 of the largest gradient, here: hence the absolute term)
        if abs(o - rn) > GRAD_TOL_NAMED.get(n, 1e-2) * rn + 1e-4 * ref_max:
            bad.append((n, o, rn))
    assert not bad, (ref_max, bad[:10])


def test_empty_bag_is_refused_with_a_clear_error():
    """Zero patches: a ValueError on the host from every entry (engine, TrainStep eager / graphed, module), no kernel launched on it."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    sizes = synth.toy_group_sizes()
    cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)), slide_ngrids=32)
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, 3))
    ts = TrainStep(eng)
    ts.set_projector(synth.projector_state(3))
    inp = synth.synth_inputs(8, sizes, 3, grid=32)
    x0 = torch.zeros(0, 1536, device="cuda")
    c0 = torch.zeros(0, 2, device="cuda")
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"]).cuda()
    with pytest.raises(ValueError, match="empty bag"):
        eng.forward(x0, c0, genes, torch.eye(3, device="cuda"), need_grad=False)
    with pytest.raises(ValueError, match="empty bag"):
        ts.step(x0, c0, genes, text, update=False)
    with pytest.raises(ValueError, match="empty bag"):
        ts.step_graphed(x0, c0, genes, text)
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3, init_seed=0,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=32, interaction_indexes=[[0, 0], [1, 1], [2, 2]], pretrained=False))
    with pytest.raises(ValueError, match="empty bag"):
        model(x=x0.reshape(1, 0, 1536), coords=c0.reshape(1, 0, 2), genes={i: g for i, g in enumerate(genes)}, clinical=[],
              task_token=torch.eye(3, device="cuda")[0])
    # ... and the engine still works afterwards
    x = torch.from_numpy(inp["x"]).cuda()
    assert torch.isfinite(ts.step(x, torch.from_numpy(inp["coords"]).cuda(), genes, text, update=False)).all()


def test_optimizer_step_matches_oracle_adamw(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import modaltune_oracle as O
    g, cfg, eng, ts, inp = _build(os.path.join(golden_dir, "model_L37_d3.npz"))
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    before = {k: v.clone() for k, v in eng.store.tensors.items() if k in eng.store.grads}
    ts.step(x, inp["coords"], genes, torch.from_numpy(inp["text"]), update=True)
    torch.cuda.synchronize()
    assert int(ts.step_dev) == 1
    for k in ("interactions.0.injector.gamma", "final_project.bias", "gene_pe"):
        gk = (eng.store.grads[k] / 2.0 ** 15).double().cpu()
        want, _, _ = O.adamw_update(before[k].double().cpu(), gk, torch.zeros_like(gk), torch.zeros_like(gk), 1, ts.lr)
        got = eng.store.tensors[k].double().cpu()
        assert float((got - want).abs().max()) < 1e-6 * float(want.abs().max()) + 1e-9, k
    # second step runs on the refreshed fp16 weight caches and stays finite
    loss2 = ts.step(x, inp["coords"], genes, torch.from_numpy(inp["text"]), update=True)
    torch.cuda.synchronize()
    assert np.isfinite(float(loss2)) and int(ts.step_dev) == 2


def test_nn_module_dropin_api_matches_reference_golden(golden_dir):
    """The reference's own call pattern (TM:123-126,172-177,225-235): Aggregator.create(...), 3x model(...), torch loss,
    loss.backward() -> param.grad; state_dict keys round-trip."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    from oracle import modaltune_oracle as O
    g = np.load(os.path.join(golden_dir, "model_L37_d3.npz"))
    L, seed, ngrids = int(g["L"]), int(g["seed"]), int(g["ngrids"])
    sizes = [int(s) for s in g["sizes"]]
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    json_cfg = dict(in_chans=1536, embed_dim=768, depth=3, slide_ngrids=ngrids, interaction_indexes=[[0, 0], [1, 1], [2, 2]],
                    num_heads=12, output_dim=256, init_values=0.0, geneclass_name="gene_mixer_group", with_cffn=True,
                    cffn_ratio=0.25, add_prompt_feature=True, use_extra_extractor=True, freeze_vit=True, with_cp=False,
                    use_prompt_sa=True, prompt_dropout=0.0, prompt_agg="avg", token_agg="sum", pretrained=False,
                    dropout=0.0, drop_path_rate=0.0, mlp_ratio=4, global_pool=False, tile_size=256, max_wsi_size=262144,
                    clinfeat_dim=5)        # (parity configuration: the stochastic ops are defined only at p = 0)
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, **json_cfg, multi_task=3).to("cuda")
    assert model.is_multi
    cfg = ModelConfig.from_json(json_cfg, multi_task=3)
    sd = synth.synth_state_dict(cfg, sizes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert list(model.state_dict().keys()) == list(sd.keys())
    trainable = [p for p in model.parameters() if p.requires_grad]
    assert sum(p.numel() for p in trainable) == 9229831 and len(trainable) == 244
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    x = torch.from_numpy(inp["x"]).cuda()
    coords = torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    model.train()
    logits = torch.cat([model(x=x, coords=coords, genes=genes, clinical=[], task_token=torch.eye(3)[t].cuda()) for t in (0, 1, 2)], dim=0)
    assert _rel(logits.detach().cpu().numpy(), g["f64_logits"]) < 1e-3
    text = torch.from_numpy(g["f64_text"]).float().cuda()
    loss = O.distill_loss(logits, text)            # torch ops on the GPU, exactly as the reference trainer computes it
    assert abs(float(loss.detach()) - float(g["f64_loss"])) < 1e-3 * float(g["f64_loss"])
    loss.backward()
    torch.cuda.synchronize()
    names = [str(n) for n in g["f64_grad_names"]]
    params = dict(model.named_parameters())
    ours = np.array([float(params[n].grad.double().norm()) for n in names])
    ref = g["f64_grad_norms"]
    bad = [(n, o, r) for n, o, r in zip(names, ours, ref) if abs(o - r) > GRAD_TOL_NAMED.get(n, 1e-2) * r + 1e-6 * ref.max()]
    assert not bad, bad[:10]
    # a torch optimiser over model.parameters() works on the flat-buffer views, and the next forward sees the update
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    opt.step()
    opt.zero_grad()
    with torch.no_grad():
        model.eval()
        l2 = model(x=x, coords=coords, genes=genes, task_token=torch.eye(3)[0].cuda())
    assert torch.isfinite(l2).all() and float((l2 - logits[:1].detach()).abs().max()) > 0


def test_module_bridge_chains_the_calls_of_a_step(golden_dir):
    """aggregators._StepGroup: the grad-mode calls of one slide share one backward hand-over.  Whatever subset of the outputs
    the loss uses, and whatever was abandoned before, param.grad equals the gradient of exactly what was backpropagated."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    g = np.load(os.path.join(golden_dir, "model_L37_d3.npz"))
    L, seed, ngrids = int(g["L"]), int(g["seed"]), int(g["ngrids"])
    sizes = [int(s) for s in g["sizes"]]
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=ngrids, interaction_indexes=[[0, 0], [1, 1], [2, 2]], dropout=0.0,
                                     drop_path_rate=0.0))
    sd = synth.synth_state_dict(model.cfg, sizes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    x, coords = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    w = torch.randn(3, 256, generator=torch.Generator().manual_seed(1)).cuda()
    eye = torch.eye(3).cuda()
    model.train()
    model.speculate = False                   # this test is about the chain; the batched prediction has its own below
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    params = dict(model.named_parameters())

    def grads():
        torch.cuda.synchronize()
        out = torch.cat([(params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])).reshape(-1).double() for k in names])
        for k in names:
            params[k].grad = None
        return out

    def call(t):
        return model(x=x, coords=coords, genes=genes, clinical=[], task_token=eye[t])

    single = []
    for t in (0, 1, 2):                       # one call, one backward: three groups of one
        (call(t) * w[t]).sum().backward()
        single.append(grads())
    ys = [call(t) for t in (0, 1, 2)]         # the trainer's pattern: one group, one hand-over
    assert model._group is not None and model._group.count == 3 and "x0" in model._group.share      # patch embedding computed once
    sum((y * w[t]).sum() for t, y in enumerate(ys)).backward()
    assert model._group is None
    tot = grads()
    ref = single[0] + single[1] + single[2]
    assert float((tot - ref).norm() / ref.norm()) < 2e-2
    ys = [call(t) for t in (0, 1, 2)]         # only the middle output reaches the loss
    (ys[1] * w[1]).sum().backward()
    assert float((grads() - single[1]).norm() / single[1].norm()) < 2e-2
    _ = [call(t) for t in (0, 1, 2)]          # a whole step abandoned (never backpropagated) ...
    ys = [call(t) for t in (0, 1, 2)]         # ... does not leak into the next one
    (ys[2] * w[2]).sum().backward()
    assert float((grads() - single[2]).norm() / single[2].norm()) < 2e-2
    with pytest.raises(RuntimeError, match="second backward"):
        (ys[2] * w[2]).sum().backward()

    # -- speculative batching (aggregators._forward_one_task): after one slide served with task ids 0, 1, 2 the next first call
    # runs ONE B = 3 engine pass and the later calls get their rows of it; gradients are those of the per-call path
    model.speculate = True
    calls = {"n": 0}
    real = model._apply_bridge

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    model._apply_bridge = counting
    ys = [call(t) for t in (0, 1, 2)]         # learning pass: three engine calls
    assert calls["n"] == 3 and model._spec is None
    sum((y * w[t]).sum() for t, y in enumerate(ys)).backward()
    assert float((grads() - ref).norm() / ref.norm()) < 2e-2
    x2 = x.clone()                            # "the next slide"
    calls["n"] = 0
    ys2 = [model(x=x2, coords=coords, genes=genes, clinical=[], task_token=eye[t]) for t in (0, 1, 2)]
    assert calls["n"] == 1 and model._spec is not None and model._spec["used"] == {0, 1, 2}
    for a, b in zip(ys, ys2):
        assert float((a - b).abs().max()) < 1e-4 * float(a.abs().max())      # batched passes == one-by-one (dropout off)
    sum((y * w[t]).sum() for t, y in enumerate(ys2)).backward()
    assert float((grads() - ref).norm() / ref.norm()) < 2e-2
    x3 = x.clone()                            # a slide that breaks the pattern: task 1 first -> runs on its own, pattern relearnt
    calls["n"] = 0
    y1 = model(x=x3, coords=coords, genes=genes, clinical=[], task_token=eye[1])
    assert calls["n"] == 1 and model._spec is None
    (y1 * w[1]).sum().backward()
    assert float((grads() - single[1]).norm() / single[1].norm()) < 2e-2
    with torch.no_grad():                     # eval loops (TM:252-327) go through the same prediction
        model.eval()
        x4, x5 = x.clone(), x.clone()
        e1 = [model(x=x4, coords=coords, genes=genes, task_token=eye[t]) for t in (0, 1, 2)]
        calls["n"] = 0
        e2 = [model(x=x5, coords=coords, genes=genes, task_token=eye[t]) for t in (0, 1, 2)]
        assert calls["n"] == 1
        for a, b in zip(e1, e2):
            assert float((a - b).abs().max()) < 1e-4 * float(a.abs().max())


def test_speculation_stops_reading_task_tokens_back_once_the_pattern_holds(golden_dir):
    """Round 5: after `nosync_after` slides served in full by the same prediction the module answers every call
    without a host read-back of the task token -- batched pass at the slide's first call, this call's row by a device-side index.
    Same logits, same gradients, any call order; the one-hots still reach the host asynchronously: a shrunken pattern sends the
    module back to learning, a task id outside the learnt rows raises (late, never silently)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    g = np.load(os.path.join(golden_dir, "model_L37_d3.npz"))
    L, seed, ngrids = int(g["L"]), int(g["seed"]), int(g["ngrids"])
    sizes = [int(s) for s in g["sizes"]]
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=ngrids, interaction_indexes=[[0, 0], [1, 1], [2, 2]], dropout=0.0,
                                     drop_path_rate=0.0))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(model.cfg, sizes, seed).items()}, strict=True)
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    x, coords = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    w = torch.randn(3, 256, generator=torch.Generator().manual_seed(1)).cuda()
    eye = torch.eye(3).cuda()
    model.train()
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    params = dict(model.named_parameters())
    calls = {"n": 0}
    real = model._apply_bridge

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    model._apply_bridge = counting

    def grads():
        torch.cuda.synchronize()
        out = torch.cat([params[k].grad.reshape(-1).double() for k in names])
        for k in names:
            params[k].grad = None
        return out

    def slide(order=(0, 1, 2), backward=True):
        xs = x.clone()
        calls["n"] = 0
        ys = {t: model(x=xs, coords=coords, genes=genes, clinical=[], task_token=eye[t].clone()) for t in order}
        if backward:
            sum((ys[t] * w[t]).sum() for t in order).backward()
        return ys, calls["n"]

    ys0, n = slide()
    ref_y, ref_g = {t: y.detach().clone() for t, y in ys0.items()}, grads()
    assert n == 3 and model._nosync_rows is None                      # learning
    for k in range(2):                                                # two slides served in full by the prediction (host-checked)
        _, n = slide(); grads()
        assert n == 1 and model._nosync_rows is None
    ys, n = slide()                                                   # the switch
    assert n == 1 and model._nosync_rows == [0, 1, 2]
    assert float((grads() - ref_g).norm() / ref_g.norm()) < 2e-2
    for t in range(3):
        assert float((ys[t].detach() - ref_y[t]).abs().max()) < 1e-4 * float(ref_y[t].abs().max())
    ys, n = slide(order=(2, 0, 1))                                    # any order: the row is picked by the token, not by position
    assert n == 1 and model._nosync_rows == [0, 1, 2]
    assert float((grads() - ref_g).norm() / ref_g.norm()) < 2e-2
    for t in range(3):
        assert float((ys[t].detach() - ref_y[t]).abs().max()) < 1e-4 * float(ref_y[t].abs().max())
    torch.cuda.synchronize()
    model._drain_decodes(block=True)
    assert model._nosync_rows == [0, 1, 2]
    # a loop that now asks for ONE task per slide: still correct (its row of the batched pass), and once the read-backs have
    # arrived the module notices and goes back to learning instead of computing three passes for one
    ys, n = slide(order=(1,), backward=False)
    assert n == 1 and float((ys[1].detach() - ref_y[1]).abs().max()) < 1e-4 * float(ref_y[1].abs().max())
    slide(order=(1,), backward=False)
    torch.cuda.synchronize()
    model._drain_decodes(block=True)
    assert model._nosync_rows is None
    # a task id outside the learnt rows is refused loudly when its read-back arrives
    model._nosync_rows = [0, 1]
    y, _ = slide(order=(2,), backward=False)
    torch.cuda.synchronize()
    assert bool(torch.isnan(y[2]).all())                              # ... and the call itself came back poisoned, not as rows[0]'s logits
    with pytest.raises(RuntimeError, match="nosync_after = 0"):
        model._drain_decodes(block=True)
    assert model._nosync_rows is None
    model._nosync_rows = [0, 1]                                       # the LAST call of a loop: eval() / state_dict() wait for the check
    slide(order=(2,), backward=False)
    with pytest.raises(RuntimeError, match="nosync_after = 0"):
        model.eval()
    model.train()
    model.nosync_after = 0                                            # opt out: every token is read back, as in round 4
    for k in range(5):
        _, n = slide(); grads()
    assert model._nosync_rows is None and n == 1


def test_module_batched_pass_runs_as_two_pass_groups(golden_dir):
    """Round 5: the drop-in module's batched (speculative) pass over a long bag runs as two concurrent pass groups, like
    TrainStep's (forced on at fixture size): same logits as the single batched pass, same parameter gradients, one hand-over."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    g = np.load(os.path.join(golden_dir, "model_L1500_d3.npz"))
    L, seed, ngrids = int(g["L"]), int(g["seed"]), int(g["ngrids"])
    sizes = [int(s) for s in g["sizes"]]
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=ngrids, interaction_indexes=[[0, 0], [1, 1], [2, 2]], dropout=0.0,
                                     drop_path_rate=0.0))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(model.cfg, sizes, seed).items()}, strict=True)
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    x, coords = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    w = torch.randn(3, 256, generator=torch.Generator().manual_seed(1)).cuda()
    eye = torch.eye(3).cuda()
    model.train()
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    params = dict(model.named_parameters())

    def grads():
        torch.cuda.synchronize()
        out = torch.cat([params[k].grad.reshape(-1).double() for k in names])
        for k in names:
            params[k].grad = None
        return out

    def slide():
        xs = x.clone()
        ys = [model(x=xs, coords=coords, genes=genes, clinical=[], task_token=eye[t].clone()) for t in (0, 1, 2)]
        sum((y * w[t]).sum() for t, y in enumerate(ys)).backward()
        return torch.cat([y.detach() for y in ys]), grads()
    model.split_min_patches = 1 << 30
    slide()                                      # learning
    y_ref, g_ref = slide()                       # one batched pass
    assert model._split is None
    model.split_min_patches = 0
    y_two, g_two = slide()                       # two groups on two streams
    assert model._split is not None and torch.equal(y_ref, y_two)
    assert float((g_two - g_ref).norm() / g_ref.norm()) < 2e-3
    y_again, g_again = slide()
    assert torch.equal(y_again, y_two) and float((g_again - g_ref).norm() / g_ref.norm()) < 2e-3


@pytest.mark.parametrize("split", [False, True])
def test_module_steady_state_runs_as_graph_replays(golden_dir, split):
    """Round 6: a bag geometry that keeps coming back under the reference trainer's loop (three model(...) calls, loss.backward()) is
    served by TWO hipGraph replays per step -- the batched forward inside the slide's first call, the backward inside the autograd node
    (module_graph.ModuleReplay) -- after one eager visit of the batched pass, two priming visits and the capture.  Same logits (bitwise: same
    kernels) and the same parameter gradients as the eager bridge, on the batched pass and on the two pass groups; a second forward
    before the first one's backward falls back to the eager bridge (the captured pair owns the long-lived workspaces), and both
    backwards deliver."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    g = np.load(os.path.join(golden_dir, "model_L1500_d3.npz"))
    L, seed, ngrids = int(g["L"]), int(g["seed"]), int(g["ngrids"])
    sizes = [int(s) for s in g["sizes"]]
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=ngrids, interaction_indexes=[[0, 0], [1, 1], [2, 2]], dropout=0.0,
                                     drop_path_rate=0.0))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(model.cfg, sizes, seed).items()}, strict=True)
    model.split_min_patches = 0 if split else 1 << 30
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    x, coords = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    w = torch.randn(3, 256, generator=torch.Generator().manual_seed(1)).cuda()
    eye = torch.eye(3).cuda()
    model.train()
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    params = dict(model.named_parameters())
    rp = model._replay
    assert rp.enabled and rp.capture_after == 2

    def grads():
        torch.cuda.synchronize()
        out = torch.cat([params[k].grad.reshape(-1).double() for k in names])
        for k in names:
            params[k].grad = None
        return out

    def fwd():
        xs = x.clone()
        return [model(x=xs, coords=coords, genes=genes, clinical=[], task_token=eye[t].clone()) for t in (0, 1, 2)]

    def slide():
        ys = fwd()
        sum((y * w[t]).sum() for t, y in enumerate(ys)).backward()
        return torch.cat([y.detach() for y in ys]), grads()
    rp.enabled = False
    slide()                                      # learning the task-id pattern
    y_ref, g_ref = slide()                       # the eager bridge, one batched pass (or two pass groups)
    rp.enabled = True
    seen = []
    for i in range(7):                           # 2 eager visits, 2 priming visits, capture (+ replay), 2 replays
        y, gr = slide()
        seen.append((rp.primed, rp.captures, rp.replays))
        assert torch.equal(y, y_ref), i
        assert float((gr - g_ref).norm() / g_ref.norm()) < 2e-3, (i, float((gr - g_ref).norm() / g_ref.norm()))
    assert seen == [(0, 0, 0), (0, 0, 0), (1, 0, 0), (2, 0, 0), (2, 1, 1), (2, 1, 2), (2, 1, 3)], seen
    assert (model._split is not None) == split
    # a no-grad forward of another slide between a replayed forward and its backward (a validation batch inside the step): it runs on
    # the engine's slot-0 workspaces, the replayed pair on its own slots -- the saved activations survive
    n0 = rp.replays
    ys = fwd()
    with torch.no_grad():
        xo = torch.roll(x, 7, 0).clone()
        [model(x=xo, coords=coords, genes=genes, clinical=[], task_token=eye[t].clone()) for t in (0, 1, 2)]
    sum((y * w[t]).sum() for t, y in enumerate(ys)).backward()
    gr = grads()
    assert rp.replays == n0 + 1 and torch.equal(torch.cat([y.detach() for y in ys]), y_ref)
    assert float((gr - g_ref).norm() / g_ref.norm()) < 2e-3
    # two forwards before the first backward: the second one cannot use the captured pair
    ya = fwd()
    yb = fwd()
    assert rp.replays == 5 and rp.eager_fallbacks == 1
    sum((y * w[t]).sum() for t, y in enumerate(yb)).backward()
    gb = grads()
    sum((y * w[t]).sum() for t, y in enumerate(ya)).backward()
    ga = grads()
    assert torch.equal(torch.cat([y.detach() for y in ya]), y_ref) and torch.equal(torch.cat([y.detach() for y in yb]), y_ref)
    assert float((ga - g_ref).norm() / g_ref.norm()) < 2e-3 and float((gb - g_ref).norm() / g_ref.norm()) < 2e-3
    # a replayed forward that is never backpropagated: the module's speculation cache keeps its autograd graph (and with it the
    # pair's lease) alive until the next slide's pass has replaced it -- that slide runs on the eager bridge, the one after replays again
    del ya, yb
    fwd()
    assert rp.replays == 6
    y, gr = slide()
    assert rp.replays == 6 and rp.eager_fallbacks == 2 and torch.equal(y, y_ref) and float((gr - g_ref).norm() / g_ref.norm()) < 2e-3
    y, gr = slide()
    assert rp.replays == 7 and torch.equal(y, y_ref) and float((gr - g_ref).norm() / g_ref.norm()) < 2e-3


def test_module_graph_replays_draw_fresh_dropout_masks(golden_dir):
    """Dropout / DropPath under the module's graph replays: the forward graph advances the engine's device-side Philox state, so two
    replays of the same slide give different logits (fresh masks) and finite gradients; eval() goes back to the deterministic pass."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    sizes = synth.toy_group_sizes()
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3, init_seed=0,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=32, interaction_indexes=[[0, 0], [1, 1], [2, 2]], pretrained=False))
    inp = synth.synth_inputs(300, sizes, 3, grid=32)
    x, coords = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    eye = torch.eye(3).cuda()
    model.train()
    assert model.engine.stochastic
    outs = []
    for i in range(8):
        xs = x.clone()                           # (a new slide tensor per step, as a loader hands them over)
        ys = torch.cat([model(x=xs, coords=coords, genes=genes, clinical=[], task_token=eye[t]) for t in (0, 1, 2)])
        ys.square().sum().backward()
        gn = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
        assert torch.isfinite(ys).all() and torch.isfinite(gn).all() and float(gn.norm()) > 0
        for p in model.parameters():
            p.grad = None
        outs.append(ys.detach().clone())
    assert model._replay.replays >= 2 and model._replay.captures == 1
    assert not torch.equal(outs[-1], outs[-2]) and not torch.equal(outs[-2], outs[-3])      # replays: fresh masks each
    model.eval()
    with torch.no_grad():
        a = torch.cat([model(x=x, coords=coords, genes=genes, clinical=[], task_token=eye[t]) for t in (0, 1, 2)])
        b = torch.cat([model(x=x, coords=coords, genes=genes, clinical=[], task_token=eye[t]) for t in (0, 1, 2)])
    assert torch.equal(a, b)


def test_module_replay_cache_grows_when_more_geometries_recur_than_it_holds():
    """Three bag lengths in rotation through a replay cache of two: the first geometry that has to be captured a SECOND time (it was
    evicted in between) doubles the cache, so a rotation does not recapture on every visit (tools/diag/module_replay_soak.py is the long
    form: 7 lengths, reserved memory flat)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    sizes = synth.toy_group_sizes()
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3, init_seed=0,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=32, interaction_indexes=[[0, 0], [1, 1], [2, 2]], pretrained=False,
                                     dropout=0.0, drop_path_rate=0.0))
    rp = model._replay
    rp.cache_size = 2
    eye = torch.eye(3).cuda()
    slides = {}
    for L in (120, 190, 260):
        inp = synth.synth_inputs(L, sizes, L, grid=32)
        slides[L] = (torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda(), {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])})
    model.train()
    caps = []
    for r in range(10):
        for L, (x, c, g) in slides.items():
            xs = x.clone()
            ys = torch.cat([model(x=xs, coords=c, genes=g, clinical=[], task_token=eye[t]) for t in (0, 1, 2)])
            ys.square().sum().backward()
            assert torch.isfinite(ys).all()
            for p in model.parameters():
                p.grad = None
        caps.append(rp.captures)
    assert rp.cache_size == 4 and caps[-1] == caps[-3] and caps[-1] <= 6, (rp.cache_size, caps)      # the rotation settled into replays
    assert rp.replays >= 3 * 4


def test_graph_replay_matches_eager(golden_dir):
    """hipGraph replay of the whole train step reproduces the eager step (same kernels, same order; the fp32-atomic
    weight-gradient reductions make two runs agree to rounding, not bitwise, and AdamW's normalised update amplifies
    that on near-zero gradients -- hence lr-scale tolerances)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(golden_dir, "model_L37_d3.npz")
    g, cfg, eng_a, ts_a, inp = _build(path)
    _, _, eng_b, ts_b, _ = _build(path)
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"]).cuda()
    la, lb = [], []
    for i in range(5):
        la.append(float(ts_a.step(x, inp["coords"], genes, text, update=True)))
        lb.append(float(ts_b.step_graphed(x, inp["coords"], genes, text)))     # 2 eager warm-ups, capture, replays
    torch.cuda.synchronize()
    assert ts_b._graphs is not None
    assert int(ts_a.step_dev) == int(ts_b.step_dev) == 5
    assert np.allclose(la, lb, rtol=5e-4, atol=0), (la, lb)
    assert la[0] > la[-1] or abs(la[0] - la[-1]) < 1e-3          # training moves the loss, nothing diverges
    for k in ("interactions.0.injector.gamma", "final_project.weight", "gene_pe"):
        a, b = eng_a.store.tensors[k], eng_b.store.tensors[k]
        assert float((a - b).abs().max()) <= 2.5 * 5 * ts_a.lr, k


def test_clinical_module_and_pancancer_task_width(golden_dir):
    """(a) registry name longnetvit_gene_clinical_adapter through the nn.Module API vs the reference golden;
    (b) pan-cancer trainer shape (train_modaltune_pancancer.py:537-542: num_tasks = 4, task ids 0..2) vs the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    from modaltune_amd.config import segment_lengths
    from modaltune_amd.engine import Engine
    from oracle import modaltune_oracle as O
    g = np.load(os.path.join(golden_dir, "model_L37_d3_clin.npz"))
    L, seed, ngrids = int(g["L"]), int(g["seed"]), int(g["ngrids"])
    sizes = [int(s) for s in g["sizes"]]
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_clinical_adapter", gene_group_defination=groups, multi_task=3,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=ngrids, interaction_indexes=[[0, 0], [1, 1], [2, 2]], token_agg="sum"))
    cfg = model.cfg
    assert cfg.clinical and cfg.num_tokens == 66
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, sizes, seed).items()}, strict=True)
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    x, coords = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    clin = torch.from_numpy(inp["clinical"]).cuda()
    with torch.no_grad():
        logits = torch.cat([model(x=x, coords=coords, genes=genes, clinical=clin, task_token=torch.eye(3)[t].cuda()) for t in (0, 1, 2)])
    assert _rel(logits.cpu().numpy(), g["f64_logits"]) < 1e-3
    # (b) multi_task = 4, three task passes
    cfg4 = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)), slide_ngrids=ngrids, multi_task=4)
    sd4 = synth.synth_state_dict(cfg4, sizes, 21)
    eng = Engine(cfg4, sizes, "cuda")
    eng.load_state_dict(sd4)
    oh = torch.eye(4)[[0, 1, 2]].cuda()
    out = eng.forward(x, inp["coords"], [torch.from_numpy(a).cuda() for a in inp["genes"]], oh, need_grad=False)
    torch.cuda.synchronize()
    sdt = {k: torch.from_numpy(v) for k, v in sd4.items()}
    with torch.no_grad():
        ref = O.multitask_logits(sdt, cfg4, torch.from_numpy(inp["x"]), torch.from_numpy(inp["coords"]),
                                 [torch.from_numpy(a) for a in inp["genes"]], segment_lengths(), task_ids=(0, 1, 2))
    assert _rel(out.cpu().numpy(), ref.numpy()) < 1e-3


def test_eval_embedding_path_matches_training_forward(golden_dir):
    """the forward-only pass (batched task passes, hipGraph replay) gives the train step's logits
    (reference: the same model call under eval()/no_grad, train_modaltune.py:252-327)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.engine import Engine
    from modaltune_amd.evaluate import EmbeddingExtractor, get_features
    g = np.load(os.path.join(golden_dir, "model_L1500_d3.npz"))
    L, seed, ngrids = int(g["L"]), int(g["seed"]), int(g["ngrids"])
    sizes = [int(s) for s in g["sizes"]]
    cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)), slide_ngrids=ngrids)
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed))
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    ex = EmbeddingExtractor(eng)
    outs = [ex(x, inp["coords"], genes).clone() for _ in range(3)]          # eager warm-up, capture, replay
    torch.cuda.synchronize()
    for o in outs:
        assert _rel(o.cpu().numpy(), g["f64_logits"]) < 1e-3
    assert torch.equal(outs[1], outs[2])
    feats, ids = get_features(EmbeddingExtractor(eng, graphed=False),
                              [dict(x=x, coords=inp["coords"], genes=genes, case_id="c0")])
    assert feats.shape == (1, 3, 256) and ids == ["c0"]
    assert _rel(feats[0], g["f64_logits"]) < 1e-3
    # long bags run the task passes as two concurrent groups (forced on here): the same logits, eager and replayed
    ex2 = EmbeddingExtractor(eng)
    ex2.split_min_patches = 0
    outs2 = [ex2(x, inp["coords"], genes).clone() for _ in range(3)]
    torch.cuda.synchronize()
    assert ex2._streams is not None and ex2.graph_replays >= 1 and all(torch.equal(o, outs[1]) for o in outs2)


def test_full_size_properties_L10000():
    _full_size_properties(10000)


def test_full_size_properties_at_the_reference_threshold_L25000():
    """The largest bag the reference's loader hands over (`threshold` = 25 000 patches, scripts/submit_modaltune.sh:47-49,
    data_utils/datasets.py:274-281: longer slides are subsampled to it): M = 75 003 rows per batched pass, the same properties."""
    _full_size_properties(25000)


def _full_size_properties(L):
    """BASELINE config 2 geometry (10 000 patches, 12 layers, T = 65) is out of the CPU oracle's reach for a test, so the
    full size is held to size-independent properties: (a) the batched task passes equal the passes run one by one
    (what the reference does, TM:175-177); (b) the analytic gradient of the whole step predicts the measured change
    of the loss along the gradient direction (central difference through three full forward passes); (c) a
    graph-replayed step reproduces the eager loss."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    seed = 77
    sizes = synth.toy_group_sizes(6)
    cfg = ModelConfig()
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed))
    ts = TrainStep(eng)
    ts.set_projector(synth.projector_state(seed))
    inp = synth.synth_inputs(L, sizes, seed, grid=128 if L <= 128 * 128 else 512)
    x = torch.from_numpy(inp["x"]).cuda().half().reshape(L, -1)
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"]).cuda()
    # (a) batched == one by one
    oh = torch.eye(3, device="cuda")
    with torch.no_grad():
        batched = eng.forward(x, inp["coords"], genes, oh, need_grad=False).clone()
        single = torch.cat([eng.forward(x, inp["coords"], genes, oh[t:t + 1], need_grad=False).clone() for t in range(3)])
    assert torch.isfinite(batched).all()
    # (B = 3 and B = 1 pick different GEMM kernels for some shapes -- M = 30 003 runs the persistent kernel, which takes the bias as the
    # first slice's C operand, M = 10 001 partly the ping-pong kernel, which adds it last: the same fp32 products in another order, one
    # fp16 rounding apart here and there.  Before gemm_ps.hip both sides ran identical arithmetic and this read 0.)
    assert _rel(batched.cpu().numpy(), single.cpu().numpy()) < 2e-4
    # (b) directional derivative along the gradient
    loss0 = float(ts.step(x, inp["coords"], genes, text, update=False))
    assert int(ts.found_inf) == 0
    g = eng.store.flat_grad.clone() / float(ts.scale)
    gn = float(g.norm())
    assert gn > 0 and np.isfinite(gn)
    base = eng.store.flat.clone()
    eps = 2e-2 * loss0 / gn ** 2 * gn          # predicted loss change 2e-2 * loss0 per side
    vals = []
    for sgn in (+1.0, -1.0):
        eng.store.flat.copy_(base + sgn * eps * g / gn)
        eng.refresh_trainable_caches()
        vals.append(float(ts.step(x, inp["coords"], genes, text, update=False)))
    eng.store.flat.copy_(base)
    eng.refresh_trainable_caches()
    measured = (vals[0] - vals[1]) / (2 * eps)
    assert abs(measured / gn - 1.0) < 0.05, (measured, gn, loss0, vals)
    # (c) graph replay reproduces the eager loss (no update between: 2 eager warm-ups + capture + replay all update,
    # so compare the first warm-up with the eager value above)
    ts2 = TrainStep(eng, lr=0.0, weight_decay=0.0)
    ts2.set_projector(synth.projector_state(seed))
    lg = [float(ts2.step_graphed(x, inp["coords"], genes, text)) for _ in range(4)]
    assert ts2._graphs is not None
    assert all(abs(v - loss0) < 2e-4 * abs(loss0) for v in lg), (lg, loss0)


def test_train_mode_stochastic_step_gradient_is_consistent(golden_dir):
    """Dropout / DropPath on (Engine.set_stochastic): with the step counter pinned the masks repeat, the step is
    deterministic, and the analytic gradient predicts the loss change along itself (i.e. forward and backward apply the
    same masks); a different step draws different masks; need_grad=False (eval) ignores them."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(golden_dir, "model_L1500_d3.npz")
    g, cfg, eng, ts, inp = _build(path)
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"]).cuda()
    base_loss = float(ts.step(x, inp["coords"], genes, text, update=False))
    eng.set_stochastic(True, seed=1234)

    def run(step_no):
        eng.rng[2] = step_no - 1                 # TrainStep.step advances the counter first
        return float(ts.step(x, inp["coords"], genes, text, update=False))
    l5 = run(5)
    grad = eng.store.flat_grad.clone() / float(ts.scale)
    assert run(5) == l5                                                                  # same masks, same step
    g2 = eng.store.flat_grad / float(ts.scale)           # (fp32-atomic reductions: equal to rounding, not bitwise)
    assert float((grad - g2).norm()) < 1e-4 * float(grad.norm())
    l6 = run(6)
    assert l6 != l5 and abs(l5 - base_loss) > 1e-6                                       # masks matter and change
    gn = float(grad.norm())
    base = eng.store.flat.clone()
    eps = 2e-2 * l5 / gn
    vals = []
    for sgn in (+1.0, -1.0):
        eng.store.flat.copy_(base + sgn * eps * grad / gn)
        eng.refresh_trainable_caches()
        vals.append(run(5))
    eng.store.flat.copy_(base)
    eng.refresh_trainable_caches()
    assert abs((vals[0] - vals[1]) / (2 * eps) / gn - 1.0) < 0.05, (vals, l5, gn)
    # eval path: no masks
    oh = torch.eye(3, device="cuda")
    with torch.no_grad():
        ev = eng.forward(x, inp["coords"], genes, oh, need_grad=False)
    assert _rel(ev.cpu().numpy(), g["f64_logits"]) < 1e-3


def test_two_adamw_steps_match_reference_trainer_golden(golden_dir):
    """Post-AdamW weights: the reference trainer's torch.optim.AdamW stepped twice on the reference model
    (tests/golden/make_golden.py `adamw`) vs two TrainStep steps (device GradScaler + fused AdamW).  AdamW's normalised
    update moves every weight by ~lr per step, so an element whose gradient is rounding noise may land up to 2 lr per step
    away; all others must agree to a small fraction of the distance travelled."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(golden_dir, "model_L37_d3_adamw.npz")
    g, cfg, eng, ts, inp = _build(path)
    lr, steps = float(g["adamw_lr"]), int(g["adamw_steps"])
    ts.set_lr(lr)
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"])
    before = {k: v.clone() for k, v in eng.store.tensors.items() if k in eng.store.grads}
    losses = [float(ts.step(x, inp["coords"], genes, text, update=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    assert int(ts.step_dev) == steps and int(ts.found_inf) == 0
    assert np.allclose(losses, g["f64_adamw_losses"], rtol=1e-3, atol=0), (losses, g["f64_adamw_losses"])
    n = 0
    for k in g.files:
        if not k.startswith("f64_adamw/"):
            continue
        key = k[len("f64_adamw/"):]
        ref, got, old = g[k].reshape(-1), eng.store.tensors[key].double().cpu().numpy().reshape(-1), before[key].double().cpu().numpy().reshape(-1)
        travelled = np.abs(ref - old)
        assert travelled.mean() > 0.5 * lr, key                      # (the reference really moved these weights)
        d = np.abs(got - ref)
        assert d.max() <= 2.0 * steps * lr * 1.01, (key, d.max())
        assert np.mean(d > 0.1 * steps * lr) < 0.01, (key, float(np.mean(d > 0.1 * steps * lr)))      # (measured: <= 0.13 %)
        n += 1
    assert n >= 6


def test_ragged_lengths_through_one_trainstep():
    """Real data: a new bag length almost every slide (datasets.py:274-281 subsamples only above 25 000).  Six lengths
    through ONE TrainStep (lr 0, so every step sees the same weights): each loss equals a fresh engine's eager step on that
    slide; a length that comes back is captured and replayed; after the first pass over the lengths nothing is allocated
    on the device any more and nothing synchronises the stream."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    seed, ngrids = 41, 64
    sizes = synth.toy_group_sizes()
    cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)), slide_ngrids=ngrids)
    sd = synth.synth_state_dict(cfg, sizes, seed)
    lengths = [700, 333, 1500, 64, 1029, 512]          # the largest is not first: the workspace grows twice
    slides = []
    for L in lengths:
        inp = synth.synth_inputs(L, sizes, seed + L, grid=ngrids)
        slides.append((torch.from_numpy(inp["x"]).cuda().half().reshape(L, -1), torch.from_numpy(inp["coords"]).cuda(),
                       [torch.from_numpy(a).cuda() for a in inp["genes"]], torch.from_numpy(inp["text"]).cuda()))
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(sd)
    ts = TrainStep(eng, lr=0.0, weight_decay=0.0, capture_after=2)
    ts.set_projector(synth.projector_state(seed))
    want = []
    for x, coords, genes, text in slides:               # fresh engine per slide, plain eager step
        e2 = Engine(cfg, sizes, "cuda")
        e2.load_state_dict(sd)
        t2 = TrainStep(e2, lr=0.0, weight_decay=0.0)
        t2.set_projector(synth.projector_state(seed))
        want.append(float(t2.step(x, coords, genes, text, update=True)))
        del t2, e2
    got = [[float(ts.step_graphed(*s)) for s in slides] for _ in range(2)]      # two eager passes over the six lengths
    for row in got:
        assert np.allclose(row, want, rtol=2e-4, atol=0), (row, want)
    assert ts.graph_replays == 0 and ts._graphs is None
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_reserved()
    row = [float(ts.step_graphed(*s)) for s in slides]                          # third visit: captured ...
    row2 = [float(ts.step_graphed(*s)) for s in slides]                         # ... and replayed
    assert np.allclose(row, want, rtol=2e-4, atol=0) and np.allclose(row2, want, rtol=2e-4, atol=0)
    assert ts.graph_replays == 12 and len(ts._graphs) == 6
    # steady state of the eager schedule (capture off): no allocator growth, same losses
    ts2 = TrainStep(eng, lr=0.0, weight_decay=0.0, capture_after=1 << 30)
    ts2.set_projector(synth.projector_state(seed))
    for s in slides:
        ts2.step_graphed(*s)
    torch.cuda.synchronize()
    mem1 = torch.cuda.memory_reserved()
    row3 = [float(ts2.step_graphed(*s)) for s in slides]
    torch.cuda.synchronize()
    assert torch.cuda.memory_reserved() == mem1, (mem1, torch.cuda.memory_reserved())
    assert np.allclose(row3, want, rtol=2e-4, atol=0)
    assert int(ts.step_dev) == 24 and mem0 > 0
    eng.check_inputs()


def test_graph_lru_thrash_keeps_reserved_memory_flat_and_losses_right():
    """More bag lengths in rotation than the graph LRU holds: every step evicts one captured geometry and captures another.  All
    captures share one memory pool, so the evicted graphs' memory is reused -- with a pool per capture the reserved memory grew by
    one step's temporaries per recapture (tools/soak.py: +0.24 GiB per step at L ~ 4 000) -- and a geometry's replay after other
    geometries were captured into the same pool still gives that slide's loss."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
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
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(sd)
    ts = TrainStep(eng, lr=0.0, weight_decay=0.0, capture_after=1, graph_cache_size=2)
    ts.set_projector(synth.projector_state(seed))
    want = [float(ts.step(*s, update=True)) for s in slides]                    # eager reference (lr 0: the weights never move)
    rows, reserved = [], []
    for rnd in range(8):
        rows.append([float(ts.step_graphed(*s)) for s in slides])
        torch.cuda.synchronize()
        reserved.append(torch.cuda.memory_reserved())
    for row in rows:
        assert np.allclose(row, want, rtol=2e-4, atol=0), (row, want)
    assert ts.graph_replays >= 5 * 6 and len(ts._graphs) <= 2                   # rounds 2.. recapture every step
    assert reserved[-1] <= reserved[3] + (8 << 20), [r >> 20 for r in reserved]


def test_module_api_recycles_its_per_call_workspaces():
    """The reference loop on the drop-in module over never-repeating bag lengths: every call leases its workspace from the engine's
    pool and the lease comes back when the step's backward has run (no reference cycle keeps a step's tapes alive, no exact-size
    allocations per call) -- the allocated device memory at the end of a step stops moving after the first steps."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    seed, ngrids = 47, 64
    sizes = synth.toy_group_sizes()
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=ngrids, interaction_indexes=[[0, 0], [1, 1], [2, 2]]))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(model.cfg, sizes, seed).items()}, strict=True)
    model.train()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5)
    inp = synth.synth_inputs(1400, sizes, seed, grid=ngrids)
    X, C = torch.from_numpy(inp["x"]).cuda().reshape(1400, -1), torch.from_numpy(inp["coords"]).cuda().reshape(1400, 2)
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    eye = torch.eye(3, device="cuda")
    alloc = []
    for i, L in enumerate([700, 1333, 410, 1200, 999, 640, 1400, 520, 1111, 860, 777, 1250]):
        x, c = X[:L].unsqueeze(0), C[:L].unsqueeze(0)
        logits = torch.cat([model(x=x, coords=c, genes=genes, task_token=eye[t]) for t in range(3)])
        logits.square().sum().backward()
        opt.step()
        opt.zero_grad()
        torch.cuda.synchronize()
        alloc.append(torch.cuda.memory_allocated())
        pool = model.engine._fresh_pool
        assert sum(len(v) for v in pool.values()) <= 5, {b: len(v) for b, v in pool.items()}
    assert max(alloc[7:]) <= max(alloc[3:7]) + (32 << 20), [a >> 20 for a in alloc]


def test_fused_adamw_is_torch_adamw_on_the_models_flat_buffers():
    """`modaltune_amd.optim.AdamW` -- the second import of INTEGRATION.md section 1 -- runs ONE mt_adamw_step over the
    model's flat parameter / gradient buffers when the trainer's loop hands it the module's gradients (train_modaltune.py:139-149,
    235-238), and equals `torch.optim.AdamW` fed the same gradients to 1e-6 after three steps: one plain, one with GradScaler's
    `grad_scale`, one skipped by `found_inf`; the per-parameter state (views of the flat moments, lazily synchronised step count)
    round-trips through state_dict() into a torch.optim.AdamW."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    from modaltune_amd.optim import AdamW
    seed, ngrids = 53, 64
    sizes = synth.toy_group_sizes()
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=ngrids, interaction_indexes=[[0, 0], [1, 1], [2, 2]], dropout=0.0,
                                     drop_path_rate=0.0))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(model.cfg, sizes, seed).items()}, strict=True)
    model.train()
    train = [p for p in model.parameters() if p.requires_grad]
    twins = [torch.nn.Parameter(p.detach().clone()) for p in train]
    kw = dict(lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999))
    opt = AdamW([{"params": train, "lr": 1e-3}], **kw)
    ref = torch.optim.AdamW([{"params": twins, "lr": 1e-3}], **kw)
    inp = synth.synth_inputs(300, sizes, seed, grid=ngrids)
    x, c = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    eye = torch.eye(3, device="cuda")
    for it, (scale, inf) in enumerate([(None, None), (64.0, 0.0), (8.0, 1.0), (8.0, 0.0)]):
        logits = torch.cat([model(x=x, coords=c, genes=genes, task_token=eye[t]) for t in range(3)])
        (logits.square().sum() * (scale or 1.0)).backward()
        v0 = train[0]._version
        for p, q in zip(train, twins):
            q.grad = p.grad.detach().clone() / (scale or 1.0)
        if scale is not None:                    # what GradScaler.step leaves on an optimiser with _step_supports_amp_scaling
            opt.grad_scale, opt.found_inf = torch.full((), scale, device="cuda"), torch.full((), inf, device="cuda")
        opt.step()
        if scale is not None:
            del opt.grad_scale, opt.found_inf
        assert opt.last_step_fused is True, "the module's gradients are views of one flat buffer in the parameters' layout"
        if not inf:
            ref.step()
            assert train[0]._version > v0          # the weight caches of the module key on the version counters
        opt.zero_grad(); ref.zero_grad()
    for (k, _), p, q in zip(((k, p) for k, p in model.named_parameters() if p.requires_grad), train, twins):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-6, k
    # the per-task calls run one by one (no batched pass): autograd hands every parameter its own gradient tensor -- gathered by one
    # multi-tensor copy, still ONE fused launch
    model.speculate = False
    logits = torch.cat([model(x=x, coords=c, genes=genes, task_token=eye[t]) for t in range(3)])
    logits.square().sum().backward()
    for p, q in zip(train, twins):
        q.grad = p.grad.detach().clone()
    opt.step(); ref.step()
    assert opt.last_step_fused is True
    opt.zero_grad(); ref.zero_grad()
    model.speculate = True
    for (k, _), p, q in zip(((k, p) for k, p in model.named_parameters() if p.requires_grad), train, twins):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-6, k
    sd = opt.state_dict()
    assert {float(s["step"]) for s in sd["state"].values()} == {4.0}           # the skipped step does not count
    ref_sd = ref.state_dict()
    for i in sd["state"]:
        for nm in ("exp_avg", "exp_avg_sq"):      # (torch forms them by lerp_ / addcmul_: other fp32 roundings of the same sums)
            a, b = sd["state"][i][nm], ref_sd["state"][i][nm]
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-30, (i, nm)
    # the real GradScaler drives it without an unscale pass or a read-back, and an optimiser over foreign tensors is torch's own
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10)
    logits = torch.cat([model(x=x, coords=c, genes=genes, task_token=eye[t]) for t in range(3)])
    scaler.scale(logits.square().sum()).backward()
    before = train[5].detach().clone()
    scaler.step(opt); scaler.update()
    assert opt.last_step_fused is True and float((train[5].detach() - before).abs().max()) > 0 and float(scaler.get_scale()) == 2.0 ** 10
    other = [torch.nn.Parameter(torch.randn(4, 3, device="cuda")), torch.nn.Parameter(torch.randn(5, device="cuda"))]
    o2, r2 = AdamW(other, lr=1e-2), torch.optim.AdamW([torch.nn.Parameter(t.detach().clone()) for t in other], lr=1e-2)
    for a, b in zip(other, r2.param_groups[0]["params"]):
        a.grad = torch.ones_like(a); b.grad = torch.ones_like(b)
    o2.step(); r2.step()
    assert o2.last_step_fused is False          # (torch's own single-tensor step on this object's state; the twin runs torch's foreach flavour)
    assert all(torch.allclose(a, b, rtol=1e-6, atol=1e-7) for a, b in zip(other, r2.param_groups[0]["params"]))


def test_lr_schedule_reaches_captured_graphs_and_eager_steps(golden_dir):
    """the learning rate is a device scalar.  The reference steps GradualWarmupScheduler + CosineAnnealingLR every
    epoch (TM:151-154,242): set_lr() between replays must change the update of an already captured graph."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = os.path.join(golden_dir, "model_L37_d3.npz")
    g, cfg, eng, ts, inp = _build(path)
    ts.capture_after = 1
    x = torch.from_numpy(inp["x"]).cuda()
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"]).cuda()
    ts.set_lr(1e-4)
    key = "interactions.0.injector.gamma"

    def moved(fn):
        w0 = eng.store.tensors[key].clone()
        fn()
        torch.cuda.synchronize()
        return float((eng.store.tensors[key] - w0).abs().mean())
    step = lambda: ts.step_graphed(x, inp["coords"], genes, text)
    d_eager = moved(step)                   # eager visit
    d_cap = moved(step)                     # capture + first replay
    assert ts._graphs is not None
    d_rep = moved(step)                     # replay
    ts.set_lr(1e-3)
    assert ts.lr == 1e-3
    d_big = moved(step)                     # replay of the SAME graph, 10x the learning rate
    ts.lr = 0.0                             # attribute form
    d_zero = moved(step)
    d_eager0 = moved(lambda: ts.step(x, inp["coords"], genes, text))
    for d in (d_eager, d_cap, d_rep):
        assert 0.3e-4 < d < 1.5e-4, (d_eager, d_cap, d_rep)
    assert 5.0 < d_big / d_rep < 15.0, (d_big, d_rep)
    assert d_zero < 1e-9 and d_eager0 < 1e-9          # (weight decay 0.01 x lr 0 = 0 as well)
    assert ts.graph_replays == 4


def test_captured_graphs_are_retired_when_their_buffers_move(golden_dir):
    """the trainer and the embedding extractor share the engine's B = 3 storage and fp16 weight caches.  A
    workspace that grows under one of them, or load_state_dict(), must retire the other's captured graph."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.engine import Engine
    from modaltune_amd.evaluate import EmbeddingExtractor
    from modaltune_amd.trainer import TrainStep
    seed, ngrids = 11, 128
    sizes = synth.toy_group_sizes()
    cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)), slide_ngrids=ngrids)
    eng = Engine(cfg, sizes, "cuda")
    sd = synth.synth_state_dict(cfg, sizes, seed)
    eng.load_state_dict(sd)
    small, big = synth.synth_inputs(37, sizes, seed, grid=ngrids), synth.synth_inputs(900, sizes, seed + 1, grid=ngrids)
    dev = lambda inp: (torch.from_numpy(inp["x"]).cuda(), inp["coords"], [torch.from_numpy(a).cuda() for a in inp["genes"]])
    ex = EmbeddingExtractor(eng)
    ref_small = EmbeddingExtractor(eng, graphed=False)(*dev(small)).clone()
    outs = [ex(*dev(small)) for _ in range(3)]              # eager, capture, replay
    assert ex._graph is not None and all(_rel(o.cpu().numpy(), ref_small.cpu().numpy()) < 1e-6 for o in outs)
    assert outs[1].data_ptr() != outs[2].data_ptr()         # callers get their own tensor, not the static buffer
    gen0 = eng.generation
    ts = TrainStep(eng, lr=0.0, weight_decay=0.0)
    ts.set_projector(synth.projector_state(seed))
    xb, cb, gb = dev(big)
    ts.step_graphed(xb, cb, gb, torch.from_numpy(big["text"]).cuda())        # a bigger bag: the shared storage is reallocated
    assert eng.generation > gen0
    again = ex(*dev(small))                                  # must NOT replay against the freed buffers
    assert _rel(again.cpu().numpy(), ref_small.cpu().numpy()) < 1e-6
    # new weights: the fp16 caches are rebuilt, captured graphs of the old ones are retired
    for _ in range(3):
        ex(*dev(small))
    assert ex._graph is not None
    sd2 = synth.synth_state_dict(cfg, sizes, seed + 7)
    eng.load_state_dict(sd2)
    new = ex(*dev(small))
    want = EmbeddingExtractor(eng, graphed=False)(*dev(small))
    assert _rel(new.cpu().numpy(), want.cpu().numpy()) < 1e-6
    assert _rel(new.cpu().numpy(), ref_small.cpu().numpy()) > 1e-3
    from modaltune_amd.evaluate import multitask_forward

    model_cfg = cfg

    class _M:            # task_ids=None means "all tasks" (TM:156-179)
        is_multi, cfg = True, model_cfg

        def forward_tasks(self, x, coords, genes, onehots, clinical=None):
            return onehots
    assert multitask_forward(_M(), task_ids=None, x=0, coords=0, genes=0).shape == (3, 3)


def test_fresh_constructor_trains_under_the_reference_loop(tmp_path, monkeypatch):
    """SURVEY §8b "train_modaltune.py drops them in": Aggregator.create with the SHIPPED JSON unmodified (pretrained: true ->
    {GIGAPATH_WEIGHT_LOC}/slide_encoder.pth, written here from synth) and NO load_state_dict, driven by the reference trainer's own
    loop (TM:123-149,195-240: 3 model calls under autocast, KL loss, GradScaler, torch.optim.AdamW): finite, decreasing loss."""
    import torch.nn as nn
    import torch.nn.functional as F
    from modaltune_amd.aggregators import Aggregator
    from test_init_cpu import SHIPPED_JSON
    sizes = synth.toy_group_sizes(6)
    groups = {i: ["g%d_%d" % (i, j) for j in range(n)] for i, n in enumerate(sizes)}
    cfg = ModelConfig.from_json(SHIPPED_JSON, multi_task=3)
    sd = synth.synth_state_dict(cfg, sizes, 41)
    frozen = [k for k, _, _, t in synth.param_specs(cfg, sizes) if not t]
    torch.save({"model": {k: torch.from_numpy(sd[k]) for k in frozen}}, tmp_path / "slide_encoder.pth")
    monkeypatch.setenv("GIGAPATH_WEIGHT_LOC", str(tmp_path))
    torch.manual_seed(0)
    model = Aggregator.create(subclass_name="longnetvit_gene_adapter", gene_group_defination=groups, **SHIPPED_JSON, multi_task=3).to("cuda")
    assert model.pretrained_report == ([], [])
    msd = model.state_dict()
    assert torch.equal(msd["encoder.layers.11.ffn.fc2.weight"].cpu(), torch.from_numpy(sd["encoder.layers.11.ffn.fc2.weight"]))
    params = [{"params": list(filter(lambda p: p.requires_grad, model.parameters())), "lr": 1e-3}]         # TM:139-149 (lr raised: 4 steps)
    opt = torch.optim.AdamW(params, weight_decay=0.01, betas=(0.9, 0.999))
    scaler = torch.amp.GradScaler("cuda", enabled=True, init_scale=2.0 ** 15)                              # TM:107
    inp = synth.synth_inputs(700, sizes, 5, grid=128)
    images, coords = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
    gene_data = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    text = torch.from_numpy(inp["text"]).cuda()[:, :256]
    text = text / text.norm(dim=-1, keepdim=True)
    loss_fn, eye = nn.KLDivLoss(reduction="sum"), torch.eye(3).cuda()
    model.train()
    losses = []
    for _ in range(4):
        with torch.autocast("cuda", enabled=True):
            logit = torch.cat([model(x=images, coords=coords, genes=gene_data, clinical=[], task_token=eye[t]) for t in (0, 1, 2)], dim=0)
            logit = logit / logit.norm(dim=-1, keepdim=True)
            loss = loss_fn(F.log_softmax(logit, dim=1), F.softmax(text[[0, 1, 3], :], dim=1)) * 10
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(v) for v in losses), losses
    assert losses[-1] < losses[0] and losses[1] < losses[0], losses
    # gamma left its init (0): the Injector path receives gradients from the first step on (adapter_modules.py:357,367)
    assert float(model.state_dict()["interactions.0.injector.gamma"].abs().max()) > 0


def test_three_forwards_then_three_separate_backwards(golden_dir):
    """A trainer that backpropagates the task losses one by one (not the reference's single loss.backward()): every pass hands
    over exactly what it replayed and param.grad accumulates, as with independent calls."""
    from modaltune_amd.aggregators import Aggregator
    g = np.load(os.path.join(golden_dir, "model_L37_d3.npz"))
    L, seed, ngrids = int(g["L"]), int(g["seed"]), int(g["ngrids"])
    sizes = [int(s) for s in g["sizes"]]
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, multi_task=3,
                              **dict(GIGAPATH_JSON, depth=3, slide_ngrids=ngrids, pretrained=False, interaction_indexes=[[0, 0], [1, 1], [2, 2]],
                                     dropout=0.0, drop_path_rate=0.0))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(model.cfg, sizes, seed).items()}, strict=True)
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    x, coords = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    w = torch.randn(3, 256, generator=torch.Generator().manual_seed(1)).cuda()
    eye = torch.eye(3).cuda()
    model.train()
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    params = dict(model.named_parameters())

    def grads():
        torch.cuda.synchronize()
        out = torch.cat([(params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])).reshape(-1).double() for k in names])
        for k in names:
            params[k].grad = None
        return out

    model.speculate = False
    for _ in range(2):
        ys = [model(x=x, coords=coords, genes=genes, clinical=[], task_token=eye[t]) for t in (0, 1, 2)]
        sum((y * w[t]).sum() for t, y in enumerate(ys)).backward()
        ref = grads()
        ys = [model(x=x, coords=coords, genes=genes, clinical=[], task_token=eye[t]) for t in (0, 1, 2)]
        for t in (2, 0, 1):                    # any order, one backward per task
            (ys[t] * w[t]).sum().backward(retain_graph=True)
        got = grads()
        assert float((got - ref).norm() / ref.norm()) < 2e-2
        with pytest.raises(RuntimeError, match="second backward"):
            (ys[1] * w[1]).sum().backward()
        grads()
    # speculative batching answers the three calls with ONE engine pass: its single tape serves a single backward, and the
    # error says how to run per-task backwards
    model.speculate = True
    for _ in range(2):
        ys = [model(x=x, coords=coords, genes=genes, clinical=[], task_token=eye[t]) for t in (0, 1, 2)]
        sum((y * w[t]).sum() for t, y in enumerate(ys)).backward()
        assert float((grads() - ref).norm() / ref.norm()) < 2e-2
    ys = [model(x=x, coords=coords, genes=genes, clinical=[], task_token=eye[t]) for t in (0, 1, 2)]
    (ys[0] * w[0]).sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="speculate = False"):
        (ys[1] * w[1]).sum().backward()
    grads()


def test_recurring_geometry_picks_its_pass_schedule_by_trial():
    """Round 6: a bag length that is about to be captured is captured both ways (batched / two pass groups), each capture is replayed a few
    times and the faster one is kept; the trial's replays are real steps, so the training state they touch (weights, moments, step count,
    loss scale, dropout counter) is restored behind them: bit-identical before and after."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.engine import Engine
    from modaltune_amd.trainer import TrainStep
    L, seed = 3000, 13
    sizes = synth.toy_group_sizes(6)
    cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)))
    eng = Engine(cfg, sizes, "cuda")
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed))
    eng.set_stochastic(True, seed=5)
    ts = TrainStep(eng, capture_after=1)
    ts.set_projector(synth.projector_state(seed))
    inp = synth.synth_inputs(L, sizes, seed, grid=128)
    x = torch.from_numpy(inp["x"]).cuda().half().reshape(L, -1)
    genes = [torch.from_numpy(a).cuda() for a in inp["genes"]]
    text = torch.from_numpy(inp["text"]).cuda()
    assert ts.auto_split
    ts.step_graphed(x, inp["coords"], genes, text)                     # eager visit
    torch.cuda.synchronize()
    assert L not in ts.split_decisions
    state = [t.clone() for t in (eng.store.flat, ts.m, ts.v, ts.step_dev, ts.scale, ts.tracker, eng.rng)]
    ts._trial_split(x, inp["coords"], genes, text, None, L, (L, L, 1, True, None, False))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(state, (eng.store.flat, ts.m, ts.v, ts.step_dev, ts.scale, ts.tracker, eng.rng)))
    tr = ts.split_trials[L]
    assert set(tr) == {"batched", "groups"} and ts.split_decisions[L] == (tr["groups"] < tr["batched"])
    del ts.split_decisions[L]
    ts._gcache.clear()
    ts.step_graphed(x, inp["coords"], genes, text)                     # this visit captures: the trial runs first
    assert L in ts.split_decisions and ts._split_now(L) == ts.split_decisions[L]
    losses = [float(ts.step_graphed(x, inp["coords"], genes, text)) for _ in range(3)]
    assert ts.graph_replays >= 3 and all(np.isfinite(losses))
    assert (ts._pass_streams is not None) or not ts.split_decisions[L]
