"""Deterministic synthetic weights and inputs (no checkpoints or datasets exist offline).

Weights are drawn with numpy's PCG64, one independent stream per state_dict key (seeded by the key's
CRC32), so a (config, seed) pair identifies a full model bit-for-bit on any machine.  Key names and
shapes are the reference's state_dict layout (observed from LongNetGeneAdapter built
in the build container) so reference checkpoints and ours interchange with strict=True.

Per SURVEY A.10 we do NOT mirror the reference's init RNG order; the golden generator loads these
tensors into the reference model instead.  The Injector gamma is drawn N(0, 0.1) (not the shipped
init 0) so the injector backward is exercised.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np

from .config import ModelConfig


def _cross_attn_keys(p: str, D: int, E: int) -> List[Tuple[str, tuple, str]]:
    return [
        (p + "q_proj.weight", (E, D), "w"), (p + "q_proj.bias", (E,), "b"),
        (p + "output_proj.weight", (D, E), "w"), (p + "output_proj.bias", (D,), "b"),
        (p + "multihead_attn.q_proj_weight", (E, E), "w"),
        (p + "multihead_attn.k_proj_weight", (E, D), "w"),
        (p + "multihead_attn.v_proj_weight", (E, D), "w"),
        (p + "multihead_attn.in_proj_bias", (3 * E,), "b"),
        (p + "multihead_attn.out_proj.weight", (E, E), "w"),
        (p + "multihead_attn.out_proj.bias", (E,), "b"),
        (p + "norm_kq.weight", (D,), "lnw"), (p + "norm_kq.bias", (D,), "lnb"),
        (p + "norm.weight", (D,), "lnw"), (p + "norm.bias", (D,), "lnb"),
    ]


def _extractor_keys(p: str, D: int, E: int):
    return _cross_attn_keys(p + "attn.", D, E) + [
        (p + "ffn.linear1.weight", (E, D), "w"), (p + "ffn.linear1.bias", (E,), "b"),
        (p + "ffn.linear2.weight", (D, E), "w"), (p + "ffn.linear2.bias", (D,), "b"),
        (p + "ffn.norm.weight", (D,), "lnw"), (p + "ffn.norm.bias", (D,), "lnb"),
    ]


def param_specs(cfg: ModelConfig, group_sizes: Sequence[int]) -> List[Tuple[str, tuple, str, bool]]:
    """[(key, shape, kind, trainable)] in the reference's state_dict order.

    kind: w (matrix), b (bias), lnw / lnb (LayerNorm affine), gamma, tok (token-like parameter).
    """
    D, E, F, O = cfg.embed_dim, cfg.adapter_dim, cfg.ffn_dim, cfg.output_dim
    G = len(group_sizes)
    T = cfg.num_tokens
    s: List[Tuple[str, tuple, str, bool]] = []

    def add(items, trainable):
        s.extend((k, sh, kind, trainable) for k, sh, kind in items)

    add([("cls_token", (1, 1, D), "tok")], False)
    if cfg.has_gene_cls:                     # prompt_agg == "cls" (longvit_adapter.py:146-149)
        add([("gene_cls", (1, 1, D), "tok")], True)
    add([("gene_pe", (T, D), "tok")], True)
    add([("patch_embed.proj.weight", (D, cfg.in_chans), "w"), ("patch_embed.proj.bias", (D,), "b")], False)
    for l in range(cfg.depth):
        p = f"encoder.layers.{l}."
        items = []
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            items += [(p + f"self_attn.{nm}.weight", (D, D), "w"), (p + f"self_attn.{nm}.bias", (D,), "b")]
        items += [(p + "self_attn.inner_attn_ln.weight", (D,), "lnw"), (p + "self_attn.inner_attn_ln.bias", (D,), "lnb"),
                  (p + "self_attn_layer_norm.weight", (D,), "lnw"), (p + "self_attn_layer_norm.bias", (D,), "lnb"),
                  (p + "ffn.fc1.weight", (F, D), "w"), (p + "ffn.fc1.bias", (F,), "b"),
                  (p + "ffn.fc2.weight", (D, F), "w"), (p + "ffn.fc2.bias", (D,), "b"),
                  (p + "ffn.ffn_layernorm.weight", (F,), "lnw"), (p + "ffn.ffn_layernorm.bias", (F,), "lnb"),
                  (p + "final_layer_norm.weight", (D,), "lnw"), (p + "final_layer_norm.bias", (D,), "lnb")]
        add(items, False)
    # dead weights on the adapter path but present in the state_dict
    add([("encoder.layer_norm.weight", (D,), "lnw"), ("encoder.layer_norm.bias", (D,), "lnb"),
         ("norm.weight", (D,), "lnw"), ("norm.bias", (D,), "lnb")], False)
    nint = len(cfg.interaction_indexes)
    for i in range(nint):
        p = f"interactions.{i}."
        add([(p + "injector.gamma", (D,), "gamma")], True)
        add(_cross_attn_keys(p + "injector.attn.", D, E), True)
        add(_extractor_keys(p + "extractor.", D, E), True)
        if i == nint - 1 and cfg.use_extra_extractor:
            for j in range(2):
                add(_extractor_keys(p + f"extra_extractors.{j}.", D, E), True)
    for i in range(1, nint if cfg.use_prompt_sa else 1):      # (use_prompt_sa False: Identity_mod, no parameters)
        p = f"prompt_selfattention.{i}."
        add([(p + "q_proj.weight", (E, D), "w"), (p + "q_proj.bias", (E,), "b"),
             (p + "output_proj.weight", (D, E), "w"), (p + "output_proj.bias", (D,), "b"),
             (p + "self_attn.q_proj_weight", (E, E), "w"), (p + "self_attn.k_proj_weight", (E, D), "w"),
             (p + "self_attn.v_proj_weight", (E, D), "w"), (p + "self_attn.in_proj_bias", (3 * E,), "b"),
             (p + "self_attn.out_proj.weight", (E, E), "w"), (p + "self_attn.out_proj.bias", (E,), "b"),
             (p + "norm.weight", (D,), "lnw"), (p + "norm.bias", (D,), "lnb")], True)
    g = cfg.gene
    Hd = g.latent_dim
    for i, n in enumerate(group_sizes):
        p = f"gene_encoder.gene_networks.{i}."
        add([(p + "0.0.weight", (Hd, int(n)), "w"), (p + "0.0.bias", (Hd,), "b"),
             (p + "1.0.weight", (Hd, Hd), "w"), (p + "1.0.bias", (Hd,), "b")], True)
    Gi, Hi = int(G * g.expansion_groups), int(Hd * g.expansion_dim)
    for k in range(g.depth):
        p = f"gene_encoder.mlp_mixer.{k}."
        add([(p + "0.fn.0.weight", (Gi, G, 1), "w"), (p + "0.fn.0.bias", (Gi,), "b"),
             (p + "0.fn.3.weight", (G, Gi, 1), "w"), (p + "0.fn.3.bias", (G,), "b"),
             (p + "0.norm.weight", (Hd,), "lnw"), (p + "0.norm.bias", (Hd,), "lnb"),
             (p + "1.fn.0.weight", (Hi, Hd), "w"), (p + "1.fn.0.bias", (Hi,), "b"),
             (p + "1.fn.3.weight", (Hd, Hi), "w"), (p + "1.fn.3.bias", (Hd,), "b"),
             (p + "1.norm.weight", (Hd,), "lnw"), (p + "1.norm.bias", (Hd,), "lnb")], True)
    p = f"gene_encoder.mlp_mixer.{g.depth}."
    add([(p + "weight", (Hd,), "lnw"), (p + "bias", (Hd,), "lnb")], True)
    p = f"gene_encoder.mlp_mixer.{g.depth + 1}."
    add([(p + "weight", (D, Hd), "w"), (p + "bias", (D,), "b")], True)
    add([("gene_encoder.pathway_compression.weight", (g.final_groups, G), "w"),
         ("gene_encoder.pathway_compression.bias", (g.final_groups,), "b")], True)
    if cfg.is_multi:
        add([("task_weight.0.weight", (D, cfg.multi_task), "w"), ("task_weight.0.bias", (D,), "b"),
             ("task_weight.1.weight", (D,), "lnw"), ("task_weight.1.bias", (D,), "lnb")], True)
    if cfg.clinical:      # clinical_mlp (longvit_adapter.py:486-491): Linear(5, 384) -> ReLU -> Linear(384, 768) -> LayerNorm
        add([("clinical_mlp.0.weight", (D // 2, cfg.clinfeat_dim), "w"), ("clinical_mlp.0.bias", (D // 2,), "b"),
             ("clinical_mlp.2.weight", (D, D // 2), "w"), ("clinical_mlp.2.bias", (D,), "b"),
             ("clinical_mlp.3.weight", (D,), "lnw"), ("clinical_mlp.3.bias", (D,), "lnb")], True)
    fin = D * ((2 + int(cfg.is_multi) + int(cfg.clinical)) if cfg.token_agg == "cat" else 1)
    add([("final_norm.weight", (fin,), "lnw"), ("final_norm.bias", (fin,), "lnb"),
         ("final_project.weight", (O, fin), "w"), ("final_project.bias", (O,), "b")], True)
    return s


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(key.encode())]))


def synth_tensor(key: str, shape: tuple, kind: str, seed: int) -> np.ndarray:
    r = _rng(seed, key)
    if kind == "w":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        fan_out = shape[0]
        if key.startswith(("encoder.", "patch_embed.")):
            std = 0.02                      # what the reference's effective backbone init draws (A.10)
        else:
            std = float(np.sqrt(2.0 / (fan_in + fan_out)))   # xavier-scale for adapter/gene/head matrices
        a = r.standard_normal(shape) * std
    elif kind == "b":
        a = r.standard_normal(shape) * 0.02
    elif kind == "lnw":
        a = 1.0 + 0.1 * r.standard_normal(shape)
    elif kind == "lnb":
        a = 0.05 * r.standard_normal(shape)
    elif kind == "gamma":
        a = 0.1 * r.standard_normal(shape)
    elif kind == "tok":
        a = 0.02 * r.standard_normal(shape)
    else:
        raise ValueError(kind)
    return a.astype(np.float32)


def synth_state_dict(cfg: ModelConfig, group_sizes: Sequence[int], seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    return OrderedDict((k, synth_tensor(k, sh, kind, seed)) for k, sh, kind, _ in param_specs(cfg, group_sizes))


def trainable_keys(cfg: ModelConfig, group_sizes: Sequence[int]) -> List[str]:
    return [k for k, _, _, t in param_specs(cfg, group_sizes) if t]


def toy_group_sizes(n_groups: int = 6) -> List[int]:
    """6 toy pathways of sizes 5..10 (SURVEY §8d synthetic inputs)."""
    return [5 + i for i in range(n_groups)]


def synth_inputs(L: int, group_sizes: Sequence[int], seed: int = 0, grid: int = 128, in_chans: int = 1536,
                 text_dim: int = 512) -> Dict[str, np.ndarray]:
    """One synthetic slide: x ~ N(0,1) [1,L,in_chans]; L distinct cells of a grid x grid lattice (x256 px);
    z-scored genes per pathway; 4 text-prompt embeddings [4, text_dim] (targets; TM:211-213)."""
    r = _rng(seed, f"inputs/{L}")
    x = r.standard_normal((1, L, in_chans)).astype(np.float32)
    cells = r.choice(grid * grid, size=L, replace=False)
    rows, cols = cells // grid, cells % grid
    jitter = r.integers(0, 256, size=(L, 2))
    coords = (np.stack([rows, cols], 1) * 256 + jitter).astype(np.float32)[None]
    genes = [r.standard_normal((1, int(n))).astype(np.float32) for n in group_sizes]
    text = r.standard_normal((4, text_dim)).astype(np.float32)
    clinical = r.standard_normal((1, 5)).astype(np.float32)      # z-scored clinical features [1, clinfeat_dim]
    return {"x": x, "coords": coords, "genes": genes, "text": text, "clinical": clinical}


def synth_inputs_titan(L: int, group_sizes: Sequence[int], seed: int = 0, grid: int = 24, feat_dim: int = 768,
                        patch_size_lv0: int = 1024, text_dim: int = 512) -> Dict[str, np.ndarray]:
    """One synthetic slide for the TITAN configuration (titan_adapter.py:329-353): tile embeddings [1, L, 768], integer
    level-0 pixel coordinates [1, L, 2] of L patches on a `grid` x `grid` lattice of 1024-px cells (a few patches share a
    cell: the gridding sums them), offset so that the minimum is not 0."""
    r = _rng(seed, f"titan_inputs/{L}")
    x = r.standard_normal((1, L, feat_dim)).astype(np.float32)
    cells = r.choice(grid * grid, size=L - L // 16, replace=False)
    cells = np.concatenate([cells, r.choice(cells, size=L // 16)])           # duplicates
    rows, cols = cells // grid, cells % grid
    coords = (np.stack([rows, cols], 1) * patch_size_lv0 + r.integers(0, patch_size_lv0, size=(L, 2)) + 3 * patch_size_lv0 + 17).astype(np.int64)[None]
    genes = [r.standard_normal((1, int(n))).astype(np.float32) for n in group_sizes]
    text = r.standard_normal((4, text_dim)).astype(np.float32)
    clinical = r.standard_normal((1, 5)).astype(np.float32)
    return {"x": x, "coords": coords, "genes": genes, "text": text, "clinical": clinical}


def projector_state(seed: int = 0, in_dim: int = 512, out_dim: int = 256) -> Dict[str, np.ndarray]:
    """Frozen random text projector (train_modaltune.py:44-59): Conv1x1 -> LayerNorm([C,1,1]) -> ReLU -> Conv1x1."""
    spec = [("conv1.0.weight", (out_dim, in_dim, 1, 1), "w"), ("conv1.0.bias", (out_dim,), "b"),
            ("conv1.1.weight", (out_dim, 1, 1), "lnw"), ("conv1.1.bias", (out_dim, 1, 1), "lnb"),
            ("conv1.3.weight", (out_dim, out_dim, 1, 1), "w"), ("conv1.3.bias", (out_dim,), "b")]
    return OrderedDict((k, synth_tensor("projector." + k, sh, kind, seed)) for k, sh, kind in spec)
