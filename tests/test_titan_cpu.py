"""TITAN configuration, CPU: the oracle's restatement of the TITAN adapter flow and the host-side gridding of
modaltune_amd.titan against fixtures produced by the REFERENCE's titan_adapter.py run on the stand-in backbone
(tests/golden/make_golden.py `titan`; the real TITAN snapshot is absent from the reference tree: backbone parity unpinned)."""
import os

import numpy as np
import pytest
import torch

from modaltune_amd import synth
from modaltune_amd.titan import grid_index, titan_model_config
from oracle import modaltune_oracle as O

import titan_standin

TITAN_JSON = dict(num_heads=12, output_dim=256, init_values=0.0, interaction_indexes=[[0, 1], [2, 3], [4, 5]],
                  geneclass_name="gene_mixer_group", with_cffn=True, cffn_ratio=0.25, add_prompt_feature=True, use_extra_extractor=True,
                  freeze_vit=True, with_cp=False, use_prompt_sa=True, prompt_dropout=0.0, prompt_agg="avg", token_agg="cat",
                  pretrained=False, drop_path_rate=0.0, clinfeat_dim=5)      # model_configs/modaltune_titan_config.json


def _case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"model_{name}.npz"))
    L, seed, grid, clinical = int(g["L"]), int(g["seed"]), int(g["grid"]), bool(int(g["clinical"]))
    sizes = [int(s) for s in g["sizes"]]
    cfg = titan_model_config(TITAN_JSON, 3, clinical, depth=6)
    inp = synth.synth_inputs_titan(L, sizes, seed, grid=grid)
    return g, cfg, sizes, inp, seed, clinical


@pytest.mark.parametrize("name", ["titan_L300", "titan_L170_clin"])
def test_oracle_titan_flow_matches_reference_on_the_standin_backbone(golden_dir, name):
    g, cfg, sizes, inp, seed, clinical = _case(golden_dir, name)
    F64 = torch.float64
    vit = titan_standin.VisionTransformer()
    titan_standin.init_standin(vit, seed)
    vit = vit.double()
    sd = {k: torch.from_numpy(v).to(F64) for k, v in synth.synth_state_dict(cfg, sizes, seed).items()}
    trainable = synth.trainable_keys(cfg, sizes)
    sd = {k: (v.clone().requires_grad_(True) if k in set(trainable) else v) for k, v in sd.items()}
    x, coords = torch.from_numpy(inp["x"]).to(F64), torch.from_numpy(inp["coords"])
    fg, cg, bgm = O.titan_gridding(x, coords, 1024)
    assert tuple(fg.shape[-2:]) == tuple(g["grid_hw"]) and int(bgm.sum()) == int(g["n_foreground"])
    assert np.array_equal(bgm.numpy(), g["bg_mask"]) and np.array_equal(cg.numpy(), g["coords_grid"])
    assert np.allclose(fg.sum(dim=1).numpy(), g["grid_feature_sum"], rtol=1e-6, atol=1e-6)
    genes = [torch.from_numpy(a).to(F64) for a in inp["genes"]]
    clin = torch.from_numpy(inp["clinical"]).to(F64) if clinical else None
    logits = torch.cat([O.titan_model_forward(sd, cfg, vit, x, coords, genes, torch.eye(3, dtype=F64)[t], clinical=clin) for t in range(3)])
    psd = {k: torch.from_numpy(v).to(F64) for k, v in synth.projector_state(seed).items()}
    loss = O.distill_loss(logits, O.projector_forward(torch.from_numpy(inp["text"]).to(F64), psd))
    loss.backward()
    assert float((logits.detach() - torch.from_numpy(g["f64_logits"])).abs().max()) < 1e-9 * float(np.abs(g["f64_logits"]).max())
    assert abs(float(loss) - float(g["f64_loss"])) < 1e-9 * abs(float(g["f64_loss"]))
    names = [str(n) for n in g["f64_grad_names"]]
    assert sorted(names) == sorted(trainable)
    ours = np.array([float(sd[n].grad.norm()) for n in names])
    assert np.abs(ours - g["f64_grad_norms"]).max() <= 1e-8 * g["f64_grad_norms"].max()


def test_grid_index_is_the_reference_rule(golden_dir):
    g, cfg, sizes, inp, seed, clinical = _case(golden_dir, "titan_L300")
    idx, H, W = grid_index(inp["coords"], 1024)
    assert (H, W) == tuple(int(v) for v in g["grid_hw"])
    occ = np.zeros(H * W, dtype=bool)
    occ[idx] = True
    assert np.array_equal(occ.reshape(1, H, W), g["bg_mask"])
    cg = np.zeros((H * W, 2), dtype=np.int64)
    np.add.at(cg, idx, inp["coords"].reshape(-1, 2))
    assert np.array_equal(cg.reshape(H, W, 2).transpose(2, 0, 1)[None], g["coords_grid"])


def test_registry_has_the_titan_names():
    from modaltune_amd.aggregators import Aggregator
    import modaltune_amd.titan  # noqa: F401
    for name in ("longnetvit_gene_adapter", "longnetvit_gene_clinical_adapter", "titan_gene_adapter", "titan_gene_clinical_adapter"):
        assert name in Aggregator.subclasses
