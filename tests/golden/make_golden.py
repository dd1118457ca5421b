"""Golden-vector generator: runs the REFERENCE (imported read-only from /root/reference) on seeded
synthetic weights/inputs and stores its outputs as small .npz fixtures next to this script.

Run in the build container only:   python tests/golden/make_golden.py
Nothing from the reference is copied: the script imports it, loads our synthetic state_dict
(modaltune_amd/synth.py, strict=True) and records what the reference computes.  Weights and inputs
are NOT stored -- they are regenerated from (config, seed) -- only reference outputs are.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shims  # noqa: E402
import unit_inputs  # noqa: E402

ref_shims.install()

from models.aggregators import Aggregator  # noqa: E402  (reference)
from models.vitadapter.adapter_modules import Injector, Extractor, SelfAttentionLayer  # noqa: E402
from models.genomic_utils import GeneBaseClass  # noqa: E402
from models.prov_gigapath.gigapath.torchscale.model.LongNet import make_longnet_from_name  # noqa: E402
import train_modaltune as TM  # noqa: E402  (reference trainer module: Projection_layer)

from modaltune_amd.config import ModelConfig  # noqa: E402
from modaltune_amd import synth  # noqa: E402

REF_CFG = json.load(open("/root/reference/model_configs/modaltune_gigapath_config.json"))


def tt(a, dtype):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def sub_state(sd, prefix, dtype):
    return {k[len(prefix):]: tt(v, dtype) for k, v in sd.items() if k.startswith(prefix)}


def unit_adapter(seed=1):
    """Injector / Extractor / SelfAttentionLayer at L=37 with T=65 and T=7 (float64)."""
    cfg = ModelConfig.from_json(REF_CFG, depth=3, interaction_indexes=[[0, 0], [1, 1], [2, 2]])
    sd = synth.synth_state_dict(cfg, synth.toy_group_sizes(), seed)
    out = {}
    dt = torch.float64
    for T in (65, 7):
        x, c, pe = unit_inputs.adapter_inputs(seed, T)
        inj = Injector(dim=768, num_heads=12, init_values=0.0, with_cffn=True, cffn_ratio=0.25).double()
        inj.load_state_dict(sub_state(sd, "interactions.0.injector.", dt), strict=True)
        out[f"T{T}_injector"] = inj(query=tt(x, dt), feat=tt(c, dt), pos=tt(pe, dt)).detach().numpy()
        ext = Extractor(dim=768, num_heads=12, with_cffn=True, cffn_ratio=0.25, drop=0.0, drop_path=0.0).double()
        ext.load_state_dict(sub_state(sd, "interactions.0.extractor.", dt), strict=True)
        out[f"T{T}_extractor"] = ext(query=tt(c, dt), feat=tt(x, dt), pos=tt(pe, dt)).detach().numpy()
        sa = SelfAttentionLayer(d_model=768, nheads=12, dropout=0.0, normalize_before=True, with_cffn=True,
                                cffn_ratio=0.25).double()
        sa.load_state_dict(sub_state(sd, "prompt_selfattention.1.", dt), strict=True)
        out[f"T{T}_selfattn"] = sa(tt(c, dt), tt(pe, dt)).detach().numpy()
    np.savez_compressed(os.path.join(HERE, "unit_adapter.npz"), seed=seed, **out)


def unit_layer(seed=2):
    """One LongNet EncoderLayer (768-d, 16 heads) with small custom segments so that several segments,
    a ragged last segment and non-multiple-of-ratio lengths (zero-pad keys) are all exercised."""
    cfg = ModelConfig.from_json(REF_CFG, depth=3, interaction_indexes=[[0, 0], [1, 1], [2, 2]])
    sd = synth.synth_state_dict(cfg, synth.toy_group_sizes(), seed)
    out = {}
    dt = torch.float64
    cases = unit_inputs.LAYER_CASES
    for name, (N, segs, ratios) in cases.items():
        enc = make_longnet_from_name("LongNet_3_layers_768_dim", dilated_ratio=str(ratios), segment_length=str(segs),
                                     drop_path_rate=0.0, dropout=0.0).double()
        layer = enc.layers[0]
        layer.load_state_dict(sub_state(sd, "encoder.layers.0.", dt), strict=True)
        ref_shims.zero_dropout(layer)
        x = unit_inputs.layer_inputs(seed, N)
        y, _ = layer(tt(x, dt), encoder_padding_mask=None)
        out[f"{name}_y"] = y.detach().numpy().astype(np.float64 if N < 64 else np.float32)
        out[f"{name}_segs"], out[f"{name}_ratios"] = np.array(segs), np.array(ratios)
        # raw dilated attention (before inner LN / out_proj) on q=k=v projections of a scaled input, so the
        # softmax is far from uniform
        at = layer.self_attn
        h = tt(x, dt) * unit_inputs.ATTN_INPUT_SCALE
        q, k, v = at.q_proj(h), at.k_proj(h), at.v_proj(h)
        from einops import rearrange
        qh, kh, vh = (rearrange(t, "b l (h d) -> b l h d", h=16) for t in (q, k, v))
        outs, lses = [], []
        for sl, dr in zip(segs, ratios):
            ki = at.gathering(rearrange(kh, "b l h d -> b l h d"), dr, sl, is_causal=False, offset=0, is_kv=True, seq_parall=False)
            vi = at.gathering(vh, dr, sl, is_causal=False, offset=0, is_kv=True, seq_parall=False)
            qi = at.gathering(qh, dr, sl, is_causal=False, offset=0, is_kv=False, seq_parall=False)
            o, lse = at.attention_ops(qi, ki, vi)
            outs.append(o); lses.append(lse)
        attn = at.scattering(outs, lses, N, 2, offset=0)
        out[f"{name}_attn"] = attn.detach().numpy().astype(np.float64 if N < 64 else np.float32)
    np.savez_compressed(os.path.join(HERE, "unit_layer.npz"), seed=seed, **out)


def unit_seqpar(seed=23):
    """The reference's DilatedAttention with args.seq_parallel (DA:61-111) on W simulated ranks: one thread per rank runs the
    reference's own gathering / attention_ops / scattering on its chunk; the module-level hooks of the reference's
    dilated_attention module (world size, rank, all_gather_func) are pointed at an in-process exchange that concatenates the
    ranks' LIVE tensors along dim 0 -- exactly what Allgather.forward returns -- so one backward over the joint graph gives
    every rank the sum over ranks of its chunk's gradient, which is Allgather.backward's reduce-scatter
    (TS/component/utils.py:43-82).  Records per rank: the mixed attention output and dq, dk, dv for a fixed cotangent."""
    import threading
    out = {}
    dt = torch.float64
    DA = saved = None
    for name, (W, B, L, segs, ratios) in unit_inputs.SEQPAR_CASES.items():
        enc = make_longnet_from_name("LongNet_3_layers_768_dim", dilated_ratio=str(ratios), segment_length=str(segs),
                                     drop_path_rate=0.0, dropout=0.0).double()
        at = enc.layers[0].self_attn
        if DA is None:       # the module object the class was defined in (LongNet.py imports it as torchscale.component...)
            DA = sys.modules[type(at).__module__]
            saved = DA.get_data_parallel_world_size, DA.get_data_parallel_rank, DA.all_gather_func
        q, k, v, dy = (tt(a, dt) for a in unit_inputs.seqpar_inputs(seed, name))
        tl = threading.local()
        barrier = threading.Barrier(W)
        slots = [None] * W

        def exchange(x):
            slots[tl.rank] = x
            barrier.wait()
            res = torch.cat(list(slots), 0)
            barrier.wait()
            return res

        DA.get_data_parallel_world_size = lambda: W
        DA.get_data_parallel_rank = lambda: tl.rank
        DA.all_gather_func = exchange
        leaves, attns, errs = [None] * W, [None] * W, []

        def run(r):
            try:
                tl.rank = r
                qr, kr, vr = (t[r].clone().requires_grad_(True) for t in (q, k, v))
                outs, lses = [], []
                for sl, dr in zip(segs, ratios):
                    ki = at.gathering(kr, dr, sl, is_causal=False, offset=0, is_kv=True, seq_parall=True)
                    vi = at.gathering(vr, dr, sl, is_causal=False, offset=0, is_kv=True, seq_parall=True)
                    qi = at.gathering(qr, dr, sl, is_causal=False, offset=0, is_kv=False, seq_parall=True)
                    o, lse = at.attention_ops(qi, ki, vi)
                    outs.append(o); lses.append(lse)
                leaves[r], attns[r] = (qr, kr, vr), at.scattering(outs, lses, L, B, offset=0)
            except Exception as e:      # a failed rank must not leave the others at the barrier
                errs.append(e)
                barrier.abort()

        th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        sum((attns[r] * dy[r]).sum() for r in range(W)).backward()
        for r in range(W):
            out[f"{name}_attn{r}"] = attns[r].detach().numpy().astype(np.float32)      # (fp32 storage keeps the fixture small)
            for nm, leaf in zip("qkv", leaves[r]):
                out[f"{name}_d{nm}{r}"] = leaf.grad.numpy().astype(np.float32)
    DA.get_data_parallel_world_size, DA.get_data_parallel_rank, DA.all_gather_func = saved
    np.savez_compressed(os.path.join(HERE, "unit_seqpar.npz"), seed=seed, **out)


def unit_gene(seed=3):
    cfg = ModelConfig.from_json(REF_CFG, depth=3, interaction_indexes=[[0, 0], [1, 1], [2, 2]])
    out = {}
    dt = torch.float64
    for name, sizes in {"toy6": synth.toy_group_sizes(6), "g21": [3 + (7 * i) % 11 for i in range(21)]}.items():
        sd = synth.synth_state_dict(cfg, sizes, seed)
        groups = {i: ["g"] * n for i, n in enumerate(sizes)}
        ge = GeneBaseClass.create("gene_mixer_group", latent_dim=256, depth=3, expansion_groups=0.5, expansion_dim=0.5,
                                  dropout=0.25, cls_token=False, n_classes=2, final_groups=64, output_dim=768,
                                  mode="feature", group_sizes=groups, n_groups=len(groups)).double()
        ge.load_state_dict(sub_state(sd, "gene_encoder.", dt), strict=True)
        ref_shims.zero_dropout(ge)
        ge.eval()
        genes = synth.synth_inputs(8, sizes, seed)["genes"]
        y = ge({i: tt(g, dt) for i, g in enumerate(genes)})
        out[f"{name}_sizes"] = np.array(sizes)
        out[f"{name}_y"] = y.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "unit_gene.npz"), seed=seed, **out)


def unit_gene331(seed=5):
    """The reference-default pathway grouping: sizes = genes per pathway in the reference's own grouping table
    (dataset/gene_pathway_processed_v2.csv, 331 pathways, 1..199 genes) -> reference GeneEncoder_Group output."""
    import pandas as pd
    table = pd.read_csv("/root/reference/dataset/gene_pathway_processed_v2.csv")
    sizes = [int(v) for v in table.drop(columns=["gene"]).sum(0).values]
    json.dump(sizes, open(os.path.join(HERE, "pathway_sizes_331.json"), "w"))
    cfg = ModelConfig.from_json(REF_CFG, depth=3, interaction_indexes=[[0, 0], [1, 1], [2, 2]])
    dt = torch.float64
    sd = synth.synth_state_dict(cfg, sizes, seed)
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    ge = GeneBaseClass.create("gene_mixer_group", latent_dim=256, depth=3, expansion_groups=0.5, expansion_dim=0.5,
                              dropout=0.25, cls_token=False, n_classes=2, final_groups=64, output_dim=768,
                              mode="feature", group_sizes=groups, n_groups=len(groups)).double()
    ge.load_state_dict(sub_state(sd, "gene_encoder.", dt), strict=True)
    ref_shims.zero_dropout(ge)
    ge.eval()
    genes = synth.synth_inputs(8, sizes, seed)["genes"]
    y = ge({i: tt(g, dt) for i, g in enumerate(genes)})
    np.savez_compressed(os.path.join(HERE, "unit_gene331.npz"), seed=seed, y=y.detach().numpy().astype(np.float32))


def unit_dataset(seed=7):
    """Case assembly + subsampling of the reference dataset class (data_utils/datasets.py:213-285) on three tiny
    synthetic slides; the object is built without its __init__ (which wants the TCGA csv files)."""
    import tempfile
    import pandas as pd
    from data_utils.datasets import FeaturesGeneTextDataset
    r = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp()
    lens, C = [23, 40, 11], 16
    paths, slides = [], {}
    for i, n in enumerate(lens):
        f = torch.from_numpy(r.standard_normal((n, C)).astype(np.float32))
        c = torch.from_numpy((r.integers(0, 40, size=(n, 2)) * 256).astype(np.int64))
        pth = os.path.join(tmp, f"s{i}.pt")
        torch.save({"features": f, "coords": c}, pth)
        paths.append(pth)
        slides[f"s{i}_features"], slides[f"s{i}_coords"] = f.numpy(), c.numpy()
    ds = object.__new__(FeaturesGeneTextDataset)
    ds.case_wise, ds.return_images, ds.return_case = True, True, True
    ds.df = pd.DataFrame({"case_id": ["c0"] * 3, "case_submitter_id": ["c0"] * 3, "features_path": paths, "lab": [1, 1, 1]})
    ds.case_ids = ds.df["case_id"].unique()
    ds.labelset = "lab"
    ds.textembeddings = {"c0": torch.zeros(4, 8)}
    ds.clinicaldata = None
    ds.gene_df = pd.DataFrame({"case_id": ["c0"], "g0": [0.5], "g1": [-1.0], "g2": [2.0]})
    ds.gene_group_defination = {0: ["g0", "g2"], 1: ["g1"]}
    out = {}
    for thr in (1000, 50):
        ds.threshold = thr
        torch.manual_seed(seed)
        image, coords, text, clinical, gene_data, label, case_id = ds[0]
        out[f"thr{thr}_image"], out[f"thr{thr}_coords"] = image.numpy(), coords.numpy()
        out[f"thr{thr}_genes"] = np.concatenate([np.atleast_1d(gene_data[k].numpy()) for k in sorted(gene_data)])
    np.savez_compressed(os.path.join(HERE, "unit_dataset.npz"), seed=seed, nslides=len(lens), **slides, **out)


GRAD_KEYS_FULL = ["interactions.0.injector.gamma", "interactions.1.injector.gamma", "final_project.bias",
                  "interactions.0.injector.attn.q_proj.bias", "task_weight.0.weight",
                  "gene_encoder.pathway_compression.weight", "interactions.2.extractor.ffn.linear1.bias"]


def model_case(name, L, depth, inter, seed, dtypes=(torch.float64, torch.float32), ngrids=128, time_it=False, clinical=False,
               token_agg=None, multi_task=3, adamw_steps=0, lr=1e-3, extra=None):
    """multi_task: width of the task one-hot (3: train_modaltune.py; 4: train_modaltune_pancancer.py:537-542, task ids 0..2
    either way; 1: the single-task path, one model call, TM:172-179).  adamw_steps > 0: after the backward, take that many
    torch.optim.AdamW steps exactly as the trainer does (TM:139-149,235-238; no GradScaler on the CPU) and record the
    post-step weights of GRAD_KEYS_FULL and the loss of every step."""
    cfg_kw = dict(REF_CFG)
    cfg_kw.update(depth=depth, interaction_indexes=inter, slide_ngrids=ngrids, pretrained=False)
    if token_agg:
        cfg_kw["token_agg"] = token_agg
    if extra:      # config branches outside the two shipped JSONs: prompt_agg "cls", use_prompt_sa False, global_pool True, ...
        cfg_kw.update(extra)
    cfg = ModelConfig.from_json(cfg_kw, multi_task=multi_task, clinical=clinical)
    sizes = synth.toy_group_sizes(6)
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    sd = synth.synth_state_dict(cfg, sizes, seed)
    inp = synth.synth_inputs(L, sizes, seed, grid=ngrids)
    psd = synth.projector_state(seed)
    out = {"L": L, "depth": depth, "inter": np.array(inter), "seed": seed, "ngrids": ngrids, "sizes": np.array(sizes),
           "clinical": int(clinical), "token_agg": cfg.token_agg, "multi_task": multi_task, "extra_cfg": json.dumps(extra or {})}
    for dt in dtypes:
        tag = "f64" if dt == torch.float64 else "f32"
        model = Aggregator.create("longnetvit_gene_clinical_adapter" if clinical else "longnetvit_gene_adapter",
                                  gene_group_defination=groups, **cfg_kw, multi_task=multi_task)
        model.load_state_dict({k: tt(v, torch.float32) for k, v in sd.items()}, strict=True)
        model = model.to(dt)
        ref_shims.zero_dropout(model)
        model.train()
        proj = TM.Projection_layer(512, 256)
        proj.load_state_dict({k: tt(v, torch.float32) for k, v in psd.items()}, strict=True)
        proj = proj.to(dt)
        for prm in proj.parameters():      # frozen random projector (TM:114-116)
            prm.requires_grad = False
        taps = {}

        def hook(i):
            def f(mod, args, res):
                x, c, cls = res
                taps.setdefault(i, []).append((cls.detach().numpy().copy(), c.detach().numpy().copy(),
                                               x[:, :8].detach().numpy().copy()))
            return f
        hs = [m.register_forward_hook(hook(i)) for i, m in enumerate(model.interactions)]
        x, coords = tt(inp["x"], dt), tt(inp["coords"], dt)
        genes = {i: tt(g, dt) for i, g in enumerate(inp["genes"])}
        t0 = time.time()
        text = proj(tt(inp["text"], dt)); text = text / text.norm(dim=-1, keepdim=True)
        clin = tt(inp["clinical"], dt) if clinical else []
        def multitask_forward():      # TM:156-179
            if model.is_multi:
                return torch.cat([model(x=x, coords=coords, genes=genes, clinical=clin, task_token=torch.eye(multi_task, dtype=dt)[t])
                                  for t in (0, 1, 2)], dim=0)
            return model(x=x, coords=coords, genes=genes, clinical=clin)

        def loss_of(logits):          # TM:225-233
            logit = logits / logits.norm(dim=-1, keepdim=True)
            return torch.nn.KLDivLoss(reduction="sum")(torch.nn.functional.log_softmax(logit / 1.0, dim=1),
                                                       torch.nn.functional.softmax(text[[0, 1, 3], :] / 1.0, dim=1)) * 10
        logits = multitask_forward()
        t1 = time.time()
        loss = loss_of(logits)
        loss.backward()
        t2 = time.time()
        for h in hs:
            h.remove()
        out[f"{tag}_logits"] = logits.detach().numpy()
        out[f"{tag}_loss"] = loss.detach().numpy()
        out[f"{tag}_text"] = text.detach().numpy()
        names, norms = [], []
        for k, p in model.named_parameters():
            if p.requires_grad:
                names.append(k); norms.append(float(p.grad.double().norm()))
        out[f"{tag}_grad_names"] = np.array(names)
        out[f"{tag}_grad_norms"] = np.array(norms)
        if dt == torch.float64:
            for k in GRAD_KEYS_FULL:
                kk = k.replace("interactions.2.", f"interactions.{len(inter) - 1}.")
                if kk in dict(model.named_parameters()):      # (no task_weight in the single-task model)
                    out["f64_grad/" + kk] = dict(model.named_parameters())[kk].grad.numpy().copy()
            for i, lst in taps.items():
                for t, (cls, c, xh) in enumerate(lst):
                    out[f"f64_tap/task{t}/cls{i}"] = cls
                    out[f"f64_tap/task{t}/c{i}"] = c.astype(np.float32)
                    out[f"f64_tap/task{t}/x{i}_head"] = xh
        if adamw_steps and dt == torch.float64:
            opt = torch.optim.AdamW([{"params": filter(lambda p: p.requires_grad, model.parameters()), "lr": lr}],
                                    weight_decay=0.01, betas=(0.9, 0.999))
            losses = [float(loss)]
            for it in range(adamw_steps):
                opt.step()
                opt.zero_grad()
                if it + 1 < adamw_steps:
                    l2 = loss_of(multitask_forward())
                    l2.backward()
                    losses.append(float(l2))
            out["adamw_lr"], out["adamw_steps"], out["f64_adamw_losses"] = lr, adamw_steps, np.array(losses)
            for k in GRAD_KEYS_FULL:
                kk = k.replace("interactions.2.", f"interactions.{len(inter) - 1}.")
                if kk in dict(model.named_parameters()):
                    out["f64_adamw/" + kk] = dict(model.named_parameters())[kk].detach().numpy().copy()
        if time_it:
            out[f"{tag}_time_fwd_bwd"] = np.array([t1 - t0, t2 - t1])
        print(name, tag, "loss", float(loss), "fwd %.2fs bwd %.2fs" % (t1 - t0, t2 - t1), flush=True)
    np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **out)


def titan_case(name, L, seed, clinical=False, grid=24):
    """The reference's TITAN adapter (titan_gene_adapter / titan_gene_clinical_adapter, titan_adapter.py) run end to end on
    the stand-in backbone of tests/golden/titan_standin.py (the real snapshot is absent: SURVEY §8c): 3 task passes, loss,
    backward -- fp64.  Pins the adapter-side flow: feature gridding, background masking, interaction blocks, cat head."""
    import titan_standin
    from modaltune_amd.titan import titan_model_config
    cfg_kw = json.load(open("/root/reference/model_configs/modaltune_titan_config.json"))
    cfg_kw.update(pretrained=False, drop_path_rate=0.0)
    sizes = synth.toy_group_sizes(6)
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    dt = torch.float64
    model = Aggregator.create("titan_gene_clinical_adapter" if clinical else "titan_gene_adapter", gene_group_defination=groups,
                              **cfg_kw, multi_task=3)
    titan_standin.init_standin(model, seed)
    cfg = titan_model_config(cfg_kw, 3, clinical, depth=6)
    adapter_sd = {k: tt(v, torch.float32) for (k, _, _, train), v in
                  zip(synth.param_specs(cfg, sizes), synth.synth_state_dict(cfg, sizes, seed).values()) if train}
    res = model.load_state_dict(adapter_sd, strict=False)
    backbone_keys = set(titan_standin.VisionTransformer().state_dict().keys())
    assert not res.unexpected_keys and set(res.missing_keys) <= backbone_keys, (res.unexpected_keys[:5], [k for k in res.missing_keys if k not in backbone_keys][:5])
    assert sorted(k for k, p in model.named_parameters() if p.requires_grad) == sorted(adapter_sd.keys())
    model = model.to(dt)
    ref_shims.zero_dropout(model)
    model.train()
    inp = synth.synth_inputs_titan(L, sizes, seed, grid=grid)
    psd = synth.projector_state(seed)
    proj = TM.Projection_layer(512, 256)
    proj.load_state_dict({k: tt(v, torch.float32) for k, v in psd.items()}, strict=True)
    proj = proj.to(dt)
    for prm in proj.parameters():
        prm.requires_grad = False
    x, coords = tt(inp["x"], dt), torch.from_numpy(inp["coords"])
    genes = {i: tt(g, dt) for i, g in enumerate(inp["genes"])}
    text = proj(tt(inp["text"], dt)); text = text / text.norm(dim=-1, keepdim=True)
    kw = dict(clinical=tt(inp["clinical"], dt)) if clinical else {}
    torch.set_default_dtype(dt)      # (preprocess_features builds its grid with torch.zeros(...): default dtype)
    fg, cg, bgm = model.preprocess_features(x, coords, 1024)
    logits = torch.cat([model(x=x, coords=coords, genes=genes, task_token=torch.eye(3, dtype=dt)[t], **kw) for t in (0, 1, 2)], dim=0)
    logit = logits / logits.norm(dim=-1, keepdim=True)
    loss = torch.nn.KLDivLoss(reduction="sum")(torch.nn.functional.log_softmax(logit, dim=1),
                                              torch.nn.functional.softmax(text[[0, 1, 3], :], dim=1)) * 10
    loss.backward()
    torch.set_default_dtype(torch.float32)
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.requires_grad:
            names.append(k); norms.append(float(p.grad.double().norm()))
    out = {"L": L, "seed": seed, "grid": grid, "sizes": np.array(sizes), "clinical": int(clinical),
           "grid_hw": np.array(fg.shape[-2:]), "n_foreground": int(bgm.sum()), "bg_mask": bgm.numpy(),
           "grid_feature_sum": fg.sum(dim=1).numpy().astype(np.float32), "coords_grid": cg.numpy(),
           "f64_logits": logits.detach().numpy(), "f64_loss": loss.detach().numpy(), "f64_grad_names": np.array(names),
           "f64_grad_norms": np.array(norms)}
    params = dict(model.named_parameters())
    for k in GRAD_KEYS_FULL:
        if k in params:
            out["f64_grad/" + k] = params[k].grad.numpy().copy()
    print(name, "loss", float(loss), "grid", tuple(fg.shape[-2:]), "foreground", int(bgm.sum()), flush=True)
    np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **out)


def init_case():
    """What the REFERENCE's constructor leaves behind (longvit_adapter.py:75-77,162,176-203): per state_dict key the mean / std /
    min / max of a fresh model (both registry names, the shipped JSON with depth / ngrids reduced), and the outcome of its
    `pretrained=True` path on a slide_encoder.pth we write (synth backbone weights + one unexpected key, one key left out):
    which keys end up equal to the file.  The statistics pin modaltune_amd/init.py's families; no weights are stored."""
    import tempfile
    import models.aggregators.longvit_adapter as LVA
    out = {}
    sizes = synth.toy_group_sizes(6)
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    for name, clinical in (("longnetvit_gene_adapter", False), ("longnetvit_gene_clinical_adapter", True)):
        cfg_kw = dict(REF_CFG)
        cfg_kw.update(slide_ngrids=128, pretrained=False)
        torch.manual_seed(0)
        model = Aggregator.create(name, gene_group_defination=groups, **cfg_kw, multi_task=3)
        stats = {}
        for k, v in model.state_dict().items():
            v = v.double()
            stats[k] = [float(v.mean()), float(v.std()) if v.numel() > 1 else 0.0, float(v.min()), float(v.max()), int(v.numel())]
        out[name] = {"stats": stats, "trainable": [k for k, p in model.named_parameters() if p.requires_grad]}
    # pretrained=True against a file
    cfg_kw = dict(REF_CFG)
    cfg_kw.update(slide_ngrids=128, pretrained=True)
    cfg = ModelConfig.from_json(cfg_kw, multi_task=3)
    sd = synth.synth_state_dict(cfg, sizes, 31)
    frozen = [k for k, _, _, t in synth.param_specs(cfg, sizes) if not t]
    left_out = "encoder.layers.3.ffn.fc1.bias"
    with tempfile.TemporaryDirectory() as d:
        blob = {k: tt(sd[k], torch.float32) for k in frozen if k != left_out}
        blob["some.unexpected.key"] = torch.zeros(3)
        torch.save({"model": blob}, os.path.join(d, "slide_encoder.pth"))
        old = LVA.GIGAPATH_WEIGHT_LOC
        LVA.GIGAPATH_WEIGHT_LOC = d
        try:
            torch.manual_seed(0)
            model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, **cfg_kw, multi_task=3)
        finally:
            LVA.GIGAPATH_WEIGHT_LOC = old
    msd = model.state_dict()
    out["pretrained"] = {"seed": 31, "left_out": left_out,
                         "equal_to_file": [k for k in frozen if k != left_out and bool((msd[k] == tt(sd[k], torch.float32)).all())],
                         "left_out_equals_file": bool((msd[left_out] == tt(sd[left_out], torch.float32)).all()),
                         "requires_grad_frozen": sorted({bool(p.requires_grad) for k, p in model.named_parameters() if k in set(frozen)})}
    json.dump(out, open(os.path.join(HERE, "init_stats.json"), "w"))
    print("wrote init_stats.json:", {k: len(v.get("stats", v)) for k, v in out.items()})


def ctor_case():
    """The architecture the REFERENCE's constructors build when the caller passes only what has no usable default there
    (longvit_adapter.py:35-53: use_prompt_sa False, prompt_agg "cls", token_agg "cat"): state_dict key -> shape, trainable keys."""
    out = {}
    sizes = synth.toy_group_sizes(6)
    groups = {i: ["g"] * n for i, n in enumerate(sizes)}
    minimal = dict(embed_dim=768, depth=3, slide_ngrids=128, interaction_indexes=[[0, 0], [1, 1], [2, 2]], pretrained=False)
    for name in ("longnetvit_gene_adapter", "longnetvit_gene_clinical_adapter"):
        torch.manual_seed(0)
        model = Aggregator.create(name, gene_group_defination=groups, **minimal, multi_task=3)
        out[name] = {"kwargs": minimal, "shapes": {k: list(v.shape) for k, v in model.state_dict().items()},
                     "trainable": [k for k, p in model.named_parameters() if p.requires_grad]}
    json.dump(out, open(os.path.join(HERE, "ctor_defaults.json"), "w"))
    print("wrote ctor_defaults.json:", {k: len(v["shapes"]) for k, v in out.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["adapter", "layer", "gene", "m37", "m1500", "m512", "clin"]
    if "clin" in which:   # clinical-prior variant (T = 66), sum and cat fusion heads
        model_case("L37_d3_clin", 37, 3, [[0, 0], [1, 1], [2, 2]], seed=14, clinical=True)
        model_case("L37_d3_clin_cat", 37, 3, [[0, 0], [1, 1], [2, 2]], seed=15, clinical=True, token_agg="cat")
        model_case("L37_d3_cat", 37, 3, [[0, 0], [1, 1], [2, 2]], seed=16, token_agg="cat")
    if "init" in which:    # constructor-time state of the reference (init families, pretrained loading)
        init_case()
    if "ctor" in which:    # the reference constructor's own defaults for omitted keys
        ctor_case()
    if "titan" in which:   # TITAN configuration (BASELINE config 4 family) on the stand-in backbone
        titan_case("titan_L300", 300, seed=21)
        titan_case("titan_L170_clin", 170, seed=22, clinical=True, grid=16)
    if "branches" in which or "branches2" in which:   # reference config branches outside the shipped JSONs (longvit_adapter.py:146,259-261,269-281,309-320)
        if "branches2" not in which:
            model_case("L37_d3_cls", 37, 3, [[0, 0], [1, 1], [2, 2]], seed=31, dtypes=(torch.float64,),
                       extra=dict(prompt_agg="cls", use_prompt_sa=False))
        model_case("L37_d6_pre_gp", 37, 6, [[2, 2], [3, 4], [5, 5]], seed=32, dtypes=(torch.float64,), token_agg="cat",
                   extra=dict(global_pool=True))
        model_case("L37_d3_clin_cls", 37, 3, [[0, 0], [1, 1], [2, 2]], seed=33, dtypes=(torch.float64,), clinical=True, token_agg="cat",
                   extra=dict(prompt_agg="cls"))
    if "pan" in which:    # pan-cancer trainer shape: one-hot width 4, task ids 0..2 (train_modaltune_pancancer.py:50-134,537-542)
        model_case("L37_d3_pan", 37, 3, [[0, 0], [1, 1], [2, 2]], seed=17, multi_task=4)
        model_case("L129_d3_pan", 129, 3, [[0, 0], [1, 1], [2, 2]], seed=18, multi_task=4, dtypes=(torch.float64,))
    if "single" in which:  # single-task path (multi_task = 1, is_multi False: one model call, no task token)
        model_case("L37_d3_single", 37, 3, [[0, 0], [1, 1], [2, 2]], seed=19, multi_task=1, dtypes=(torch.float64,))
    if "adamw" in which:   # two optimiser steps of the reference trainer: post-AdamW weights
        model_case("L37_d3_adamw", 37, 3, [[0, 0], [1, 1], [2, 2]], seed=11, dtypes=(torch.float64,), adamw_steps=2, lr=1e-3)
    if "adapter" in which:
        unit_adapter()
    if "layer" in which:
        unit_layer()
    if "gene" in which:
        unit_gene()
    if "seqpar" in which:   # LongNet sequence parallelism
        unit_seqpar()
    if "gene331" in which:
        unit_gene331()
    if "dataset" in which:
        unit_dataset()
    if "m37" in which:
        model_case("L37_d3", 37, 3, [[0, 0], [1, 1], [2, 2]], seed=11)
    if "m1500" in which:
        model_case("L1500_d3", 1500, 3, [[0, 0], [1, 1], [2, 2]], seed=12)
    if "m512" in which:   # BASELINE config 1 shape: 512 patches, full 12-layer Prov-GigaPath geometry
        model_case("L512_d12", 512, 12, [[0, 3], [4, 7], [8, 11]], seed=13, time_it=True)
