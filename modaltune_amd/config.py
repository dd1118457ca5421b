"""Model configuration and the integer index arithmetic of the hot path.

Mirrors the config surface of the reference (keys of model_configs/modaltune_gigapath_config.json:1-30
plus model_configs/other_configs.py:10-24) without importing it.  Everything here is host-side,
torch-free integer/shape logic shared by the HIP engine, the oracle and the tests.
"""
from __future__ import annotations

import dataclasses
import json
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

# LongNet architecture table for the names LongNetViT builds ("LongNet_{depth}_layers_{dim}_dim",
# reference slide_encoder.py:122, torchscale/model/LongNetConfig.py:121-179): all 768-d variants use
# 16 heads, ffn 3072, dilated ratios [1,2,4,8,16].
BACKBONE_HEADS = 16
DILATED_RATIOS = (1, 2, 4, 8, 16)
LN_EPS = 1e-5  # torchscale EncoderConfig.layernorm_eps (architecture/config.py:43); nn.LayerNorm default too


# model_configs/modaltune_gigapath_config.json:1-30 -- the configuration the reference trainer passes to the constructor
GIGAPATH_JSON = {"in_chans": 1536, "embed_dim": 768, "depth": 12, "slide_ngrids": 1000, "tile_size": 256, "max_wsi_size": 262144,
                 "global_pool": False, "dropout": 0.25, "drop_path_rate": 0.1, "mlp_ratio": 4, "num_heads": 12, "output_dim": 256,
                 "init_values": 0.0, "geneclass_name": "gene_mixer_group", "interaction_indexes": [[0, 3], [4, 7], [8, 11]],
                 "with_cffn": True, "cffn_ratio": 0.25, "add_prompt_feature": True, "use_extra_extractor": True, "freeze_vit": True,
                 "with_cp": False, "use_prompt_sa": True, "prompt_dropout": 0.0, "prompt_agg": "avg", "token_agg": "sum",
                 "pretrained": True, "clinfeat_dim": 5}
# What the reference CONSTRUCTOR assumes for a key the caller leaves out, where that differs from the shipped JSON
# (LongNetGeneAdapter.__init__, longvit_adapter.py:35-53; LongNetViT.__init__, slide_encoder.py:87-97): a bare
# `Aggregator.create("longnetvit_gene_adapter", ...)` must build the architecture (and state_dict key set) the reference builds.
LONGNET_CTOR_DEFAULTS = {"use_prompt_sa": False, "prompt_agg": "cls", "token_agg": "cat", "embed_dim": 256, "interaction_indexes": None}


@dataclasses.dataclass
class GeneConfig:
    """model_configs/other_configs.py:12-21 ("gene_mixer_group")."""
    latent_dim: int = 256
    depth: int = 3
    expansion_groups: float = 0.5
    expansion_dim: float = 0.5
    final_groups: int = 64
    dropout: float = 0.25       # AlphaDropout in the pathway networks, Dropout in the mixer feed-forwards (train mode)


@dataclasses.dataclass
class ModelConfig:
    """Keys of model_configs/modaltune_gigapath_config.json (same names, same defaults)."""
    in_chans: int = 1536
    embed_dim: int = 768
    depth: int = 12
    slide_ngrids: int = 1000
    tile_size: int = 256
    max_wsi_size: int = 262144
    global_pool: bool = False
    dropout: float = 0.25
    drop_path_rate: float = 0.1
    mlp_ratio: float = 4
    num_heads: int = 12
    output_dim: int = 256
    init_values: float = 0.0
    geneclass_name: str = "gene_mixer_group"
    interaction_indexes: Sequence[Sequence[int]] = ((0, 3), (4, 7), (8, 11))
    with_cffn: bool = True
    cffn_ratio: float = 0.25
    add_prompt_feature: bool = True
    use_extra_extractor: bool = True
    freeze_vit: bool = True
    with_cp: bool = False
    use_prompt_sa: bool = True
    prompt_dropout: float = 0.0
    prompt_agg: str = "avg"
    token_agg: str = "sum"
    pretrained: bool = True
    clinfeat_dim: int = 5
    multi_task: int = 3
    clinical: bool = False      # LongNetGeneSimpleClinicalAdapter (longvit_adapter.py:350-672): + 1 clinical token
    gene: GeneConfig = dataclasses.field(default_factory=GeneConfig)

    # ---- derived ----
    @property
    def adapter_dim(self) -> int:          # E = int(768 * 0.25) (adapter_modules.py:153)
        return int(self.embed_dim * self.cffn_ratio)

    @property
    def ffn_dim(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // BACKBONE_HEADS

    @property
    def is_multi(self) -> bool:            # longvit_adapter.py:88
        return self.multi_task > 1

    @property
    def has_gene_cls(self) -> bool:        # prompt_agg == "cls": a learned token in front of the gene tokens (longvit_adapter.py:146,259-261)
        return self.prompt_agg == "cls"

    @property
    def num_tokens(self) -> int:           # T = final_groups (+ gene_cls) + task token (+ clinical token) (longvit_adapter.py:146-154,470-477)
        return self.gene.final_groups + int(self.has_gene_cls) + int(self.is_multi) + int(self.clinical)

    @property
    def first_interaction_layer(self) -> int:   # layers below it run on the embedded slide before any adapter (longvit_adapter.py:269-281)
        return int(self.interaction_indexes[0][0])

    def validate(self):
        if self.embed_dim != 768 or self.head_dim != 48:
            raise ValueError("the HIP path is built for the Prov-GigaPath geometry (768-d, 16 heads x 48)")
        if self.prompt_agg not in ("avg", "cls") or self.token_agg not in ("sum", "cat"):
            raise NotImplementedError("prompt_agg must be 'avg' or 'cls'; token_agg 'sum' or 'cat' (the reference raises for anything else too)")
        if not self.add_prompt_feature:
            raise ValueError("add_prompt_feature=False: the reference's own forward fails on this configuration (`outcome` is only bound "
                             "inside `if self.add_prompt_feature`, longvit_adapter.py:315-346): there is no behaviour to reproduce")
        if not self.with_cffn:
            raise NotImplementedError("with_cffn=False widens the adapter attention to 768 (12 heads x 64): the adapter kernels are built "
                                      "for cffn_ratio 0.25 (12 x 16), as in both shipped ModalTune configurations")
        if not self.freeze_vit:
            raise NotImplementedError("freeze_vit=False: the backbone's weight gradients are not computed (selective backward)")
        last = self.first_interaction_layer - 1
        if last < -1:
            raise ValueError("interaction_indexes must start at a layer >= 0")
        for a, b in self.interaction_indexes:
            if a != last + 1 or b < a:
                raise ValueError("interaction_indexes must tile the layers contiguously")
            last = b
        if last != self.depth - 1:
            raise ValueError("interaction_indexes must cover the layers up to depth - 1")

    @staticmethod
    def from_longnet_ctor(kwargs, **overrides) -> "ModelConfig":
        """The configuration `LongNetGeneAdapter(**kwargs)` builds in the reference: omitted keys take the reference CONSTRUCTOR's
        defaults (prompt_agg "cls", token_agg "cat", no prompt self-attention, embed_dim 256 -- which `validate` then refuses), not
        the shipped JSON's.  `interaction_indexes` has no usable default there either (None: longvit_adapter.py:40,100-129 iterate it)."""
        d = dict(LONGNET_CTOR_DEFAULTS)
        d.update(kwargs)
        if d["interaction_indexes"] is None:
            raise TypeError("interaction_indexes is required (the reference constructor's default None cannot be iterated, "
                            "longvit_adapter.py:40,100-129); model_configs/modaltune_gigapath_config.json passes [[0, 3], [4, 7], [8, 11]]")
        return ModelConfig.from_json(d, **overrides)

    @staticmethod
    def from_json(path_or_dict, **overrides) -> "ModelConfig":
        d = dict(path_or_dict) if isinstance(path_or_dict, dict) else json.load(open(path_or_dict))
        d.update(overrides)
        names = {f.name for f in dataclasses.fields(ModelConfig)}
        kw = {k: v for k, v in d.items() if k in names and k != "gene"}
        if "interaction_indexes" in kw:
            kw["interaction_indexes"] = tuple(tuple(int(i) for i in p) for p in kw["interaction_indexes"])
        cfg = ModelConfig(**kw)
        if isinstance(d.get("gene"), dict):
            cfg.gene = GeneConfig(**d["gene"])
        return cfg


def segment_lengths(max_wsi_size: int = 262144, tile_size: int = 256) -> List[int]:
    """LongNetViT.get_optimal_segment_length (reference slide_encoder.py:163-182).

    Five log-spaced lengths from 1024 to (max_wsi_size/tile_size)^2, truncated to int exactly like
    the reference (np.power(2, linspace).astype(int)) -> [1024, 5792, 32768, 185363, 1048576].
    """
    max_seq_len = (max_wsi_size // tile_size) ** 2
    e = np.linspace(np.log2(1024), int(np.log2(max_seq_len)), 5)
    return [int(v) for v in np.power(2, e).astype(int)]


@dataclasses.dataclass(frozen=True)
class Branch:
    """One dilated-attention branch at sequence length N (dilated_attention.py:82-111, 212-253)."""
    seg: int      # s = min(segment_length, N)
    ratio: int    # dilation r
    nseg: int     # ceil(N / s)
    n: int        # sparse sequence length per (segment, head) = ceil(s / r)  (zero padded)


def branch_table(N: int, seg_lengths: Sequence[int], ratios: Sequence[int] = DILATED_RATIOS) -> List[Branch]:
    out = []
    for sl, dr in zip(seg_lengths, ratios):
        s = min(int(sl), N)
        out.append(Branch(seg=s, ratio=int(dr), nseg=-(-N // s), n=-(-s // int(dr))))
    return out


def coords_to_rowcol(coords: np.ndarray, tile: float = 256.0) -> Tuple[np.ndarray, np.ndarray]:
    """LongNetViT.coords_to_pos (slide_encoder.py:198-211) split into (row, col) grid indices.

    pos = floor(c0/256)*ngrids + floor(c1/256) + 1; we keep (row, col) because the table row
    pos_embed[pos] = concat(sincos(col), sincos(row)) (pos_embed.py:42-59: the w-meshgrid goes first).
    """
    g = np.floor(np.asarray(coords, dtype=np.float32) / np.float32(tile))
    return g[..., 0].astype(np.int64), g[..., 1].astype(np.int64)


def sincos_1d_table(ngrids: int, dim: int) -> np.ndarray:
    """1-D sin-cos table [ngrids, dim] in float64 -> float32, following pos_embed.py:62-81.

    The reference's full 2-D table row for grid cell (row, col) is concat(T[col], T[row]) with
    T = this table at dim = embed_dim/2, so a [ngrids, 384] table replaces the 3 GB buffer.
    """
    omega = np.arange(dim // 2, dtype=float)
    omega /= dim / 2.0
    omega = 1.0 / 10000 ** omega
    pos = np.arange(ngrids, dtype=np.float32).reshape(-1)
    out = np.einsum("m,d->md", pos, omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1).astype(np.float32)


def flops_per_slide_step(L: int, T: int, depth: int = 12, seg=None, tasks: int = 3) -> Dict[str, float]:
    """Algorithmic FLOPs: 2mnk GEMMs, 4*nq*nk*d attention, no recompute counted."""
    seg = seg or segment_lengths()
    N, D, F, H, d, E = L + 1, 768, 3072, 16, 48, 192
    patch = 2.0 * L * 1536 * D
    gemm_layer = 2.0 * N * (4 * D * D + 2 * D * F)
    attn_layer = sum(b.nseg * H * b.n * b.n * d * 4.0 for b in branch_table(N, seg))
    # what the kernels execute: the zero padding at a segment / sequence end is skipped (attn.hip), so a sparse
    # sequence with nv real entries costs nv^2 instead of n^2
    attn_exec = 0.0
    for b in branch_table(N, seg):
        for j in range(b.nseg):
            lim = min(b.seg, N - j * b.seg)
            for r in range(b.ratio):
                nv = min(b.n, max(0, -(-(lim - r) // b.ratio)))
                attn_exec += (H // b.ratio) * nv * nv * d * 4.0
    inj = 2.0 * L * (D * E + E * E + E * E + E * D) + 4.0 * L * T * E + 2.0 * T * 2 * D * E
    ext = 2.0 * T * (D * E + E * E + E * E + E * D) + 4.0 * L * T * E + 2.0 * L * 2 * D * E + 2.0 * T * 2 * D * E
    adapter = 3 * inj + 5 * ext
    fwd = patch + depth * (gemm_layer + attn_layer) + adapter
    bwd = depth * (gemm_layer + 2.5 * attn_layer) + 2 * adapter
    step = tasks * (fwd + bwd) - (tasks - 1) * patch
    return {"fwd_pass": fwd, "bwd_pass": bwd, "step": step, "gemm_layer": gemm_layer,
            "attn_layer": attn_layer, "attn_layer_executed": attn_exec, "adapter": adapter, "patch": patch}


def flops_per_titan_step(Lv: int, T: int, depth: int = 6, D: int = 768, F: int = 3072, C: int = 768, tasks: int = 3) -> Dict[str, float]:
    """Algorithmic FLOPs of one TITAN-configuration slide step with Lv foreground cells (same counting rules as
    flops_per_slide_step: 2mnk GEMMs, 4 nq nk d attention, frozen blocks dX-only in the backward, flash backward = 2.5 x
    forward, adapters 2 x): dense ViT blocks qkv + proj + fc1 + fc2 (TA:359-361), patch-embedding MLP once per slide, attentional
    pooling K|V projection (TA:401-402), Injector x3 / Extractor x5 as in the LongNet path."""
    N, E = Lv + 1, 192
    patch = 2.0 * Lv * (C * D + D * D)
    gemm_layer = 2.0 * N * (4 * D * D + 2 * D * F)
    attn_layer = 4.0 * N * N * D
    pool = 2.0 * N * D * 2 * D
    inj = 2.0 * Lv * (D * E + E * E + E * E + E * D) + 4.0 * Lv * T * E + 2.0 * T * 2 * D * E
    ext = 2.0 * T * (D * E + E * E + E * E + E * D) + 4.0 * Lv * T * E + 2.0 * Lv * 2 * D * E + 2.0 * T * 2 * D * E
    adapter = 3 * inj + 5 * ext
    fwd = depth * (gemm_layer + attn_layer) + adapter + pool
    bwd = depth * (gemm_layer + 2.5 * attn_layer) + 2 * adapter + pool
    return {"fwd_pass": fwd, "bwd_pass": bwd, "step": tasks * (fwd + bwd) + patch, "gemm_layer": gemm_layer, "attn_layer": attn_layer,
            "adapter": adapter, "patch": patch}
