"""TITAN configuration on the GPU (BASELINE config 4 family): modaltune_amd.titan with the stand-in backbone
plugged in, against fixtures produced by the REFERENCE's titan_adapter.py on the same stand-in (tests/golden/make_golden.py
`titan`) -- with the frozen blocks on the HIP kernels (backbone_impl "native": dense ALiBi attention, GEMMs, LayerNorms, GELU,
attentional pooling, grid-free feature gridding) and as the module's own torch code ("torch").  Parity against the REAL snapshot
is unpinned (its source is not in the reference tree); the native path checks itself against the supplied module at load."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from modaltune_amd import synth  # noqa: E402
from test_titan_cpu import TITAN_JSON, _case  # noqa: E402

import titan_standin  # noqa: E402


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def _model(golden_dir, name, impl="native"):
    from modaltune_amd.aggregators import Aggregator
    import modaltune_amd.titan  # noqa: F401
    from oracle import modaltune_oracle as O
    g, cfg, sizes, inp, seed, clinical = _case(golden_dir, name)
    vit = titan_standin.VisionTransformer()
    titan_standin.init_standin(vit, seed)
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    model = Aggregator.create("titan_gene_clinical_adapter" if clinical else "titan_gene_adapter", gene_group_defination=groups,
                              **TITAN_JSON, multi_task=3, backbone=vit, backbone_impl=impl)
    sd = synth.synth_state_dict(model.cfg, sizes, seed)
    state = {k: torch.from_numpy(v) for k, v in sd.items() if k in dict(model._params)}
    state.update(vit.state_dict())
    model.load_state_dict(state, strict=True)
    assert set(model.state_dict().keys()) == set(state.keys())
    assert model.backbone_impl == impl
    if impl == "native":      # every piece of the stand-in is recognised and reproduced on the probe slide
        rep = model.engine.backbone.report
        assert rep["embed"] == rep["pool"] == rep["blocks"] == "native", rep
        assert rep["block_err"] < 1e-2 and rep["embed_err"] < 1e-2 and rep["pool_err"] < 1e-2, rep
    return g, model, sizes, inp, seed, clinical, O


@pytest.mark.parametrize("impl", ["native", "torch"])
@pytest.mark.parametrize("name", ["titan_L300", "titan_L170_clin"])
def test_titan_adapter_train_step_matches_reference_golden(golden_dir, name, impl):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g, model, sizes, inp, seed, clinical, O = _model(golden_dir, name, impl)
    x = torch.from_numpy(inp["x"]).cuda()
    coords = torch.from_numpy(inp["coords"]).cuda()
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    kw = dict(clinical=torch.from_numpy(inp["clinical"]).cuda()) if clinical else {}
    model.train()
    trainable = [p for p in model.parameters() if p.requires_grad]
    assert len(trainable) == len(g["f64_grad_names"])
    logits = torch.cat([model(x=x, coords=coords, genes=genes, task_token=torch.eye(3)[t].cuda(), **kw) for t in (0, 1, 2)], dim=0)
    assert _rel(logits.detach().cpu().numpy(), g["f64_logits"]) < 1e-3
    psd = {k: torch.from_numpy(v).cuda() for k, v in synth.projector_state(seed).items()}
    loss = O.distill_loss(logits, O.projector_forward(torch.from_numpy(inp["text"]).cuda(), psd))
    assert abs(float(loss.detach()) - float(g["f64_loss"])) < 1e-3 * float(g["f64_loss"])
    loss.backward()
    torch.cuda.synchronize()
    names = [str(n) for n in g["f64_grad_names"]]
    params = dict(model.named_parameters())
    ours = np.array([float(params[n].grad.double().norm()) for n in names])
    ref = g["f64_grad_norms"]
    bad = [(n, o, r) for n, o, r in zip(names, ours, ref) if abs(o - r) > 2e-2 * r + 1e-6 * ref.max()]
    assert not bad, bad[:10]
    for k in g.files:
        if k.startswith("f64_grad/"):
            ours_k = params[k[len("f64_grad/"):]].grad.double().cpu().numpy()
            err = np.linalg.norm(ours_k - g[k]) / (np.linalg.norm(g[k]) + 1e-300)
            assert err < 4e-2, (k, err)


@pytest.mark.parametrize("impl", ["native", "torch"])
def test_titan_gridding_and_ragged_bags(golden_dir, impl):
    """preprocess_features on the device vs the reference's grid; bags of different sizes through one model (config 4:
    "mixed bag lengths"): each forward equals a fresh model's."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.titan import device_tokens, preprocess_features
    g, model, sizes, inp, seed, clinical, O = _model(golden_dir, "titan_L300", impl)
    x16, cells, dims, Lv = device_tokens(torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda(), 1024)
    assert tuple(dims.cpu().tolist()) == tuple(g["grid_hw"]) and Lv == int(g["bg_mask"].sum())      # the grid-free form agrees
    assert np.array_equal(cells.cpu().numpy(), np.argwhere(g["bg_mask"][0]))
    fg, cg, bgm = preprocess_features(torch.from_numpy(inp["x"]).cuda(), inp["coords"], 1024)
    assert tuple(fg.shape[-2:]) == tuple(g["grid_hw"])
    assert np.array_equal(bgm.cpu().numpy(), g["bg_mask"]) and np.array_equal(cg.cpu().numpy(), g["coords_grid"])
    assert np.allclose(fg.sum(dim=1).cpu().numpy(), g["grid_feature_sum"], rtol=1e-5, atol=1e-5)
    fg2, _, _ = preprocess_features(torch.from_numpy(inp["x"]).cuda(), inp["coords"], 1024)
    assert torch.equal(fg, fg2)                      # cells shared by several patches: summed in patch order, every time
    model.eval()
    outs = {}
    from modaltune_amd.evaluate import EmbeddingExtractor
    with torch.no_grad():      # the eval / embedding pass (TM:252-327): 3 task passes in one batched forward == one call per task
        ext = EmbeddingExtractor(model.engine, task_ids=(0, 1, 2))
        xs, cs = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
        gl = [torch.from_numpy(a).cuda() for a in inp["genes"]]
        emb = ext(xs, cs, gl)
        if impl == "native":      # a geometry that comes back is captured per (patches, tokens) and replayed (gridding + read-back stay eager)
            again = [ext(xs, cs, gl) for _ in range(3)]
            assert ext.graph_replays >= 2 and all(float((a - emb).abs().max()) < 1e-5 * float(emb.abs().max()) for a in again)
            other = synth.synth_inputs_titan(211, sizes, seed + 5, grid=24)
            eo = ext(torch.from_numpy(other["x"]).cuda(), torch.from_numpy(other["coords"]).cuda(), [torch.from_numpy(a).cuda() for a in other["genes"]])
            assert torch.isfinite(eo).all() and float((ext(xs, cs, gl) - emb).abs().max()) < 1e-5 * float(emb.abs().max())
        model.speculate = False
        one = torch.cat([model(x=xs, coords=cs, genes=gl, task_token=torch.eye(3)[t].cuda()) for t in (0, 1, 2)], dim=0)
        model.speculate = True
        assert emb.shape == (3, 256) and float((emb - one).abs().max()) < 1e-4 * float(one.abs().max())
        for rep in range(2):
            for L in (300, 90, 520, 33):
                inp_l = synth.synth_inputs_titan(L, sizes, seed + L, grid=24)
                o = model(x=torch.from_numpy(inp_l["x"]).cuda(), coords=torch.from_numpy(inp_l["coords"]).cuda(),
                          genes={i: torch.from_numpy(a).cuda() for i, a in enumerate(inp_l["genes"])}, task_token=torch.eye(3)[1].cuda())
                assert torch.isfinite(o).all()
                if L in outs:
                    assert torch.equal(o, outs[L])          # same slide later, other lengths in between: same result
                outs[L] = o.clone()


def test_backbone_with_live_dropout_runs_natively_in_eval_and_refuses_train():
    """a slide encoder that carries Dropout / DropPath with p > 0 is usable for inference on the native blocks (those
    layers are identities in eval mode); what is refused -- with the `backbone_impl="torch"` hint -- is putting the model in train
    mode, where the reference keeps them active and the HIP blocks have no such layers.  Construction warns and
    leaves the model in eval()."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    import modaltune_amd.titan  # noqa: F401
    sizes = synth.toy_group_sizes()
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    vit = titan_standin.VisionTransformer()
    titan_standin.init_standin(vit, 9)
    vit.blocks.modules_list[2].extra_drop = torch.nn.Dropout(0.1)        # (registered, p > 0: a stochastic layer inside the backbone)
    with pytest.warns(UserWarning, match="left in eval"):
        model = Aggregator.create("titan_gene_adapter", gene_group_defination=groups, **TITAN_JSON, multi_task=3, backbone=vit)
    assert model.backbone_impl == "native" and not model.training
    inp = synth.synth_inputs_titan(150, sizes, 2)
    with torch.no_grad():
        out = model(x=torch.from_numpy(inp["x"]).cuda(), coords=torch.from_numpy(inp["coords"]).cuda(),
                    genes=[torch.from_numpy(a).cuda() for a in inp["genes"]], task_token=torch.eye(3)[1].cuda())
    assert out.shape == (1, 256) and bool(torch.isfinite(out).all())
    with pytest.raises(RuntimeError, match='backbone_impl="torch"'):
        model.train()
    assert model.eval() is model
    vit2 = titan_standin.VisionTransformer()
    titan_standin.init_standin(vit2, 9)
    vit2.blocks.modules_list[2].extra_drop = torch.nn.Dropout(0.1)
    m2 = Aggregator.create("titan_gene_adapter", gene_group_defination=groups, **TITAN_JSON, multi_task=3, backbone=vit2, backbone_impl="torch")
    assert m2.backbone_impl == "torch" and m2.training


def test_titan_without_backbone_fails_loudly():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    import modaltune_amd.titan  # noqa: F401
    sizes = synth.toy_group_sizes()
    model = Aggregator.create("titan_gene_adapter", gene_group_defination={i: ["g"] * n for i, n in enumerate(sizes)}, **TITAN_JSON, multi_task=3)
    inp = synth.synth_inputs_titan(40, sizes, 1)
    with pytest.raises(RuntimeError, match="TITAN slide encoder"):
        model(x=torch.from_numpy(inp["x"]).cuda(), coords=torch.from_numpy(inp["coords"]).cuda(),
              genes=[torch.from_numpy(a).cuda() for a in inp["genes"]], task_token=torch.eye(3)[0].cuda())


def test_native_blocks_match_module_at_4096_cells_mixed_lengths():
    """BASELINE config 4's shape: ~4k-cell slides of mixed lengths through ONE model, 3 task calls + loss + backward each (the
    reference trainer's loop, TM:172-177,225-238), frozen blocks on the HIP kernels vs the same module's torch code between the
    same adapter kernels: logits within 1e-3 (relative to the largest logit), every gradient norm within 2 %."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.aggregators import Aggregator
    import modaltune_amd.titan  # noqa: F401
    from oracle import modaltune_oracle as O
    seed = 4
    sizes = synth.toy_group_sizes()
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    models = {}
    for impl in ("native", "torch"):
        vit = titan_standin.VisionTransformer()
        titan_standin.init_standin(vit, seed)
        m = Aggregator.create("titan_gene_adapter", gene_group_defination=groups, **TITAN_JSON, multi_task=3, backbone=vit, backbone_impl=impl)
        sd = synth.synth_state_dict(m.cfg, sizes, seed)
        state = {k: torch.from_numpy(v) for k, v in sd.items() if k in dict(m._params)}
        state.update(vit.state_dict())
        m.load_state_dict(state, strict=True)
        for k, p in m._params.items():          # Injector gammas start at 0 (AM:349): give the backward something to carry
            if k.endswith("gamma"):
                with torch.no_grad():
                    p.copy_(0.1 * torch.randn(p.shape, generator=torch.Generator().manual_seed(len(k))).cuda())
        m.train()
        models[impl] = m
    psd = {k: torch.from_numpy(v).cuda() for k, v in synth.projector_state(seed).items()}
    for L in (4300, 2300, 6100):                 # -> 4096 / ~2200 / ~5800 foreground cells on an 80 x 80 lattice
        inp = synth.synth_inputs_titan(L, sizes, seed + L, grid=80)
        x, coords = torch.from_numpy(inp["x"]).cuda(), torch.from_numpy(inp["coords"]).cuda()
        genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
        target = O.projector_forward(torch.from_numpy(inp["text"]).cuda(), psd)
        res = {}
        for impl, m in models.items():
            for p in m.parameters():
                p.grad = None
            logits = torch.cat([m(x=x, coords=coords, genes=genes, task_token=torch.eye(3)[t].cuda()) for t in (0, 1, 2)], dim=0)
            O.distill_loss(logits, target).backward()
            torch.cuda.synchronize()
            res[impl] = (logits.detach().double().cpu(), {k: float(p.grad.double().norm()) for k, p in m.named_parameters() if p.requires_grad})
        ln, lt = res["native"][0], res["torch"][0]
        assert float((ln - lt).abs().max() / lt.abs().max()) < 1e-3, L
        gn, gt = res["native"][1], res["torch"][1]
        top = max(gt.values())
        bad = [(k, gn[k], gt[k]) for k in gt if abs(gn[k] - gt[k]) > 2e-2 * gt[k] + 1e-6 * top]
        assert not bad, (L, bad[:8])


def test_titan_degenerate_slides(golden_dir):
    """One foreground cell (three patches that all fall into it: the sequence is cls + 1 token), and a slide whose patches sit
    on one grid row: native blocks == the module's torch blocks on both."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _, m_nat, sizes, inp, seed, _, _ = _model(golden_dir, "titan_L300", "native")
    _, m_tor, _, _, _, _, _ = _model(golden_dir, "titan_L300", "torch")
    genes = {i: torch.from_numpy(a).cuda() for i, a in enumerate(inp["genes"])}
    g = torch.Generator().manual_seed(3)
    cases = {"one_cell": (torch.randn(1, 3, 768, generator=g), torch.tensor([[[5000, 7000], [5100, 7100], [5900, 7900]]])),
             "one_row": (torch.randn(1, 40, 768, generator=g), torch.stack([torch.full((40,), 9000), 3000 + 1024 * torch.arange(40)], 1)[None])}
    for name, (x, coords) in cases.items():
        outs = []
        for m in (m_nat, m_tor):
            m.eval()
            with torch.no_grad():
                outs.append(m(x=x.cuda(), coords=coords.cuda(), genes=genes, task_token=torch.eye(3)[2].cuda()))
        assert torch.isfinite(outs[0]).all(), name
        assert float((outs[0] - outs[1]).abs().max()) < 1e-3 * float(outs[1].abs().max()), name


def test_titan_graph_replay_matches_eager_over_mixed_bag_lengths():
    """TrainStep.step_graphed on the TITAN configuration: the gridding and the token-count read-back stay eager, the rest of the
    step is captured per (patches, tokens) geometry and replayed.  Two engines with the same weights walk the same rotation of three
    bag lengths -- one eagerly, one through step_graphed: same losses and the same parameters after every step up to the run-to-run noise of
    the atomically accumulated weight gradients (dropout off), and the rotation's later visits are replays."""
    from modaltune_amd.titan import NativeBackbone, TitanEngine, titan_model_config
    from modaltune_amd.trainer import TrainStep
    seed = 6
    sizes = synth.toy_group_sizes()
    engs, steps = [], []
    for _ in range(2):
        vit = titan_standin.VisionTransformer()
        titan_standin.init_standin(vit, seed)
        cfg = titan_model_config(dict(TITAN_JSON, drop_path_rate=0.0), 3, False, 6)
        eng = TitanEngine(cfg, sizes, NativeBackbone(vit, "cuda"), "cuda")
        eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed))
        ts = TrainStep(eng, lr=1e-5)
        ts.set_projector(synth.projector_state(seed))
        engs.append(eng); steps.append(ts)
    slides = []
    for j, L in enumerate((900, 520, 1300)):
        inp = synth.synth_inputs_titan(L, sizes, seed + j, grid=40)
        slides.append((torch.from_numpy(inp["x"]).cuda().reshape(L, -1), torch.from_numpy(inp["coords"]).cuda().reshape(L, 2),
                       [torch.from_numpy(a).cuda() for a in inp["genes"]], torch.from_numpy(inp["text"]).cuda()))
    for i in range(12):
        x, coords, genes, text = slides[i % 3]
        le = float(steps[0].step(x, coords, genes, text, update=True))
        lg = float(steps[1].step_graphed(x, coords, genes, text))
        assert np.isfinite(le) and abs(le - lg) <= 1e-3 * abs(le), (i, le, lg)      # (fp32 atomics in the weight-gradient kernels: not bit-stable,
        #                                                                                       and AdamW's first steps are +-lr per element)
        a, b = engs[0].store.flat, engs[1].store.flat
        # AdamW turns the sign of a near-zero gradient into a +-lr step: compare as test_two_adamw_steps_... does -- almost every
        # element within a tenth of the distance the optimiser has moved it
        far = float(((a - b).abs() > 0.1 * 1e-5 * (i + 1)).float().mean())
        assert far < 1e-2, (i, far)
    assert steps[1].graph_replays >= 3 and steps[1].eager_steps == 6      # two eager visits per bag length, then capture + replays
    engs[1].check_inputs()
