"""TITAN configuration of the Modal Adapter (BASELINE config 4): `titan_gene_adapter` /
`titan_gene_clinical_adapter` (reference models/aggregators/titan_adapter.py:42-438, 441-...).

What is native here: everything the reference's TITAN adapter adds around the slide encoder -- the feature gridding of
`preprocess_features` (TA:295-327), the background masking of `prepare_forward_features` (TA:253-293), the interaction
blocks `InteractionBlockWithCls_TITAN` (adapter_modules.py:526-558: Injector -> [cls | patches] through the backbone
blocks -> Extractor (+ extra extractors)), prompt self-attention, gene encoder, task tokens and the fusion head on the
attentionally pooled image token (TA:399-437) -- on the same HIP kernels and tape as the Prov-GigaPath path
(engine.Engine), trainable parameters in the same flat buffers.

What is NOT native, and why: the TITAN slide encoder itself (HF MahmoodLab/TITAN @ b2fb4f47, utils/constants.py:22-23) --
its source and weights are absent from the reference tree, so its arithmetic cannot be restated or pinned (PARITY
UNPINNED).  It is taken behind an interface instead: any object with the surface the reference uses (`patch_embed`,
`_pos_embed`, `norm_pre`, `get_alibi`, `blocks.modules_list[i](x, attn_bias, bg_mask)`, `norm`, `forward_attn_pool`) -- in
practice the user's own `VisionTransformer` instance from the TITAN snapshot.  The frozen blocks run as that object's
torch code between our kernels; their activation gradients come from `torch.autograd.grad` on the recorded block call.
tests/test_titan_gpu.py pins the native part against the REFERENCE's adapter code run on a stand-in backbone
(tests/golden/titan_standin.py).

Bags are ragged (a different number of foreground cells per slide): one slide per call, any length; a batch of slides is a
Python loop over `forward` (the reference's TITAN path is batch-1 too: TA:258-267 uses the per-slide ALiBi only for B == 1).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import rowmap
from .aggregators import Aggregator, LongNetGeneAdapter, _ModelFn
from .config import ModelConfig
from .engine import Engine, F32
from .tape import Var


# ------------------------------------------------------------------------------------------------ feature gridding
def grid_index(coords: np.ndarray, patch_size_lv0: int) -> Tuple[np.ndarray, int, int]:
    """TA:304-312: cell (row, col) of every patch = floor((coords - min) / patch_size_lv0), shifted to start at 0.
    Returns (flat cell index row * W + col per patch, H, W)."""
    c = np.asarray(coords).reshape(-1, 2).astype(np.int64)
    g = np.floor_divide(c - c.min(axis=0), int(patch_size_lv0))
    g = g - g.min(axis=0)
    H, W = (int(v) + 1 for v in g.max(axis=0))
    return g[:, 0] * W + g[:, 1], H, W


def preprocess_features(features: torch.Tensor, coords, patch_size_lv0: int):
    """`TITANGeneAdapter.preprocess_features` (TA:295-327) on the device: scatter-ADD of the patch features (and level-0
    coordinates) into their grid cells (mt_scatter_rows_f32; index arithmetic on the host, integers only).
    Returns (feature_grid [1, C, H, W], coords_grid [1, 2, H, W] int64, bg_mask [1, H, W] bool) like the reference."""
    f = features.reshape(-1, features.shape[-1]).to(dtype=F32).contiguous()
    cnp = coords.detach().cpu().numpy() if torch.is_tensor(coords) else np.asarray(coords)
    cnp = cnp.reshape(-1, 2)
    idx, H, W = grid_index(cnp, patch_size_lv0)
    dev = f.device
    grid = torch.zeros(H * W, f.shape[1], dtype=F32, device=dev)
    # one pass per occurrence rank: the k-th patch of every cell goes in pass k, so a pass touches distinct cells and the
    # sums come out in patch order (what index_add_ does on the CPU), bitwise the same on every run
    order = np.argsort(idx, kind="stable")
    sidx = idx[order]
    first = np.r_[True, sidx[1:] != sidx[:-1]]
    rank = np.arange(len(sidx)) - np.maximum.accumulate(np.where(first, np.arange(len(sidx)), 0))
    for r in range(int(rank.max()) + 1):
        sel = order[rank == r]
        ops.scatter_rows(f, torch.from_numpy(idx[sel].astype(np.int32)).to(dev), grid, len(sel), f.shape[1], accumulate=True,
                         src_idx=torch.from_numpy(sel.astype(np.int32)).to(dev))
    cg = np.zeros((H * W, 2), dtype=np.int64)
    np.add.at(cg, idx, cnp.astype(np.int64))
    occupied = torch.zeros(H * W, dtype=F32, device=dev)
    ops.row_absmax(grid, occupied, H * W, f.shape[1])          # bg_mask = any(feature != 0) per cell (TA:326)
    fg = grid.view(H, W, -1).permute(2, 0, 1).unsqueeze(0)
    return fg, torch.from_numpy(cg).view(H, W, 2).permute(2, 0, 1).unsqueeze(0).to(dev), (occupied > 0).view(1, H, W)


# ------------------------------------------------------------------------------------------------ backbone interface
class TorchBackbone:
    """Adapter around a TITAN-like `VisionTransformer` (torch).  Frozen: no weight gradients; activation gradients through a
    block come from torch.autograd.grad on the recorded call.  This is the ONLY place torch arithmetic runs on this path."""

    def __init__(self, vit: nn.Module):
        self.vit = vit
        for p in vit.parameters():
            p.requires_grad_(False)

    @property
    def depth(self) -> int:
        return len(self.vit.blocks.modules_list)

    @torch.no_grad()
    def embed(self, feature_grid, coords_grid, bg_mask):
        """prepare_forward_features for B = 1 (TA:253-293): tokens [1, 1 + Lv, D] (cls first, background dropped), attention
        bias, token mask."""
        v = self.vit
        B, nc, w, h = feature_grid.shape
        x = feature_grid.flatten(2, 3).transpose(1, 2)
        attn_bias = None
        if getattr(v, "pos_encode_type", None) == "alibi":
            attn_bias = v.get_alibi(w, h, bg_mask).to(dtype=x.dtype, device=x.device)
        x = v.norm_pre(v._pos_embed(v.patch_embed(x), coords_grid, w, h))
        m = torch.cat((torch.ones((1, 1), dtype=torch.bool, device=x.device), bg_mask.view(1, -1)), dim=1)
        return x[m].unsqueeze(0), attn_bias, m

    def block(self, l: int, h: torch.Tensor, attn_bias, mask, need_grad: bool):
        """Returns (output, handle); handle (None without gradients) is what block_backward takes -- it belongs to THIS
        call, so several forwards may precede one backward (the reference calls the model 3x per step, TM:175-177)."""
        blk = self.vit.blocks.modules_list[l]
        bias = None if attn_bias is None else attn_bias.expand(h.shape[0], -1, -1, -1)
        if not need_grad:
            with torch.no_grad():
                return blk(h, bias, mask), None
        with torch.enable_grad():
            x = h.detach().requires_grad_(True)
            y = blk(x, bias, mask)
        return y.detach(), (x, y)

    @staticmethod
    def backward(handle, dy: torch.Tensor) -> torch.Tensor:
        """fp32 activation gradient of a recorded call (the block may have run under autocast: dy is cast to the output's
        dtype, dx comes back in the input's and is returned as fp32)."""
        x, y = handle
        (dx,) = torch.autograd.grad(y, x, dy.to(y.dtype))
        return dx.to(F32)

    def pool(self, h: torch.Tensor, mask, need_grad: bool):
        """norm + forward_attn_pool (TA:400-402): [B, N, D] -> (image token [B, D], handle)."""
        def run(x):
            img, _ = self.vit.forward_attn_pool(self.vit.norm(x), bg_mask=mask)
            return img
        if not need_grad:
            with torch.no_grad():
                return run(h), None
        with torch.enable_grad():
            x = h.detach().requires_grad_(True)
            y = run(x)
        return y.detach(), (x, y)


# ------------------------------------------------------------------------------------------------ engine
class TitanEngine(Engine):
    """Engine with the frozen image side delegated to a backbone object; adapters / tokens / head as in Engine."""

    def __init__(self, cfg: ModelConfig, group_sizes: Sequence[int], backbone: Optional[TorchBackbone], device="cuda"):
        super().__init__(cfg, group_sizes, device)
        self.backbone = backbone
        self._tok = None

    def _build_caches(self):         # no LongNet weights to pack: only the trainable big-M adapter linears
        t, dev = self.store.tensors, self.device
        from .engine import _W16
        self._frozen16 = {}
        self._train16 = {}
        self._pack_table = None
        for pref in self._cross_attn_prefixes():
            self._train16[pref + "q_proj"] = _W16([t[pref + "q_proj.weight"]], dev)
            self._train16[pref + "q_in"] = _W16([t[pref + "multihead_attn.q_proj_weight"]], dev)
            self._train16[pref + "kv"] = _W16([t[pref + "multihead_attn.k_proj_weight"], t[pref + "multihead_attn.v_proj_weight"]], dev)
            self._train16[pref + "out_in"] = _W16([t[pref + "multihead_attn.out_proj.weight"]], dev)
            self._train16[pref + "output_proj"] = _W16([t[pref + "output_proj.weight"]], dev)
        self._caches_ready = True
        self.generation += 1

    def forward_slide(self, x, coords, genes, task_onehots, patch_size_lv0: int = 1024, need_grad: bool = True, fresh: bool = False,
                      clinical=None) -> torch.Tensor:
        """x [1, L, C] tile embeddings, coords [1, L, 2] level-0 pixels (TA:329-353) -> logits [B, output_dim]."""
        if self.backbone is None:
            raise RuntimeError("titan_gene_adapter needs the TITAN slide encoder: pass backbone=<VisionTransformer from the "
                               "MahmoodLab/TITAN snapshot> (its source is not part of ModalTune; parity unpinned)")
        fg, cg, bgm = preprocess_features(x.to(self.device), coords, patch_size_lv0)
        tok, bias, mask = self.backbone.embed(fg, cg, bgm)
        self._tok, self._bias, self._mask = tok.to(F32).contiguous(), bias, mask
        self._need = need_grad
        patches = self._tok[0, 1:]
        if patches.shape[0] < 1:
            raise ValueError("slide has no foreground cell")
        return self.forward(patches, None, genes, task_onehots, need_grad=need_grad, fresh=fresh, clinical=clinical)

    # -- image-side hooks
    def _embed_patches(self, x, coords, ws, staged, L):
        ops.copy_rows(self._tok[0, 1:], ws["x0"], L, self.cfg.embed_dim)

    def _cls_source(self) -> torch.Tensor:
        return self._tok[0, 0]

    def _layer(self, l: int, out: torch.Tensor, pend=None, defer: bool = False):
        ctx, ws, D = self._ctx, self._ctx["ws"], self.cfg.embed_dim
        B, N, M = ctx["B"], ctx["N"], ctx["M"]
        bb, need = self.backbone, self._need
        # the reference hands the cls-prefixed bg_mask to every block: blk(x, attn_bias, bg_mask) (adapter_modules.py:535)
        y, handle = bb.block(l, ws[f"hin{l}"].view(B, N, D), self._bias, self._mask, need)
        ops.copy_rows(y.to(F32).reshape(M, D).contiguous(), out, M, D)      # (a half-precision block output is widened first)

        def bwd():
            dh = ws["dh"]
            dx = bb.backward(handle, dh.view(B, N, D))
            ops.copy_rows(dx.reshape(M, D).contiguous(), dh, M, D)
            ctx["dh16_valid"] = False
        self.tape.record(bwd)
        return None

    def _image_token(self, hout: torch.Tensor) -> Var:
        ctx, D = self._ctx, self.cfg.embed_dim
        B, N, M, ws = ctx["B"], ctx["N"], ctx["M"], ctx["ws"]
        bb = self.backbone
        pooled, handle = bb.pool(hout.view(B, N, D), self._mask, self._need)      # forward_attn_pool(x, bg_mask=bg_mask), TA:402
        img = Var(pooled.to(F32).contiguous())

        def bwd():
            dh = ws["dh"]
            ctx["dh16_valid"] = False
            if img.grad is None:
                dh.zero_()
                return
            dx = bb.backward(handle, img.grad)
            ops.copy_rows(dx.reshape(M, D).contiguous(), dh, M, D)
        self.tape.record(bwd)
        return img


# ------------------------------------------------------------------------------------------------ nn.Module surface
def titan_model_config(kwargs: Dict[str, Any], multi_task: int, clinical: bool, depth: int) -> ModelConfig:
    kw = dict(kwargs)
    kw.setdefault("interaction_indexes", [[0, 1], [2, 3], [4, 5]])
    cfg = ModelConfig.from_json(kw, multi_task=multi_task, clinical=clinical, depth=depth, in_chans=768, embed_dim=768,
                                dropout=0.0)
    return cfg


@Aggregator.register("titan_gene_adapter")
class TITANGeneAdapter(LongNetGeneAdapter):
    """Drop-in for the reference's TITANGeneAdapter (TA:42-438): same registry name, ctor kwargs (keys of
    model_configs/modaltune_titan_config.json + gene_group_defination, multi_task), forward signature
    (x, coords, genes, task_token, patch_size_lv0), `is_multi`.  `backbone`: the TITAN VisionTransformer instance
    (required to run; see the module docstring).  state_dict holds the adapter-side keys under the reference's names; the
    backbone's own tensors are exposed un-prefixed after them, as in the reference (which inherits from the backbone)."""
    CLINICAL = False

    def __init__(self, gene_group_defination: Dict[Any, Sequence[str]] = None, multi_task: int = 1, backbone: Optional[nn.Module] = None,
                 device="cuda", **kwargs):
        nn.Module.__init__(self)
        gene_group_defination = gene_group_defination or {}
        depth = len(backbone.blocks.modules_list) if backbone is not None else 6
        cfg = titan_model_config(kwargs, multi_task, self.CLINICAL, depth)
        self.cfg = cfg
        self.is_multi = multi_task > 1
        if backbone is not None:
            backbone = backbone.to(device)
        object.__setattr__(self, "_backbone_module", backbone)
        self.engine = TitanEngine(cfg, [len(v) for v in gene_group_defination.values()],
                                  TorchBackbone(backbone) if backbone is not None else None, device)
        self._params = OrderedDict()
        for k, shape, kind, train in self.engine.store.specs:
            if train:
                self._params[k] = nn.Parameter(self.engine.store.tensors[k], requires_grad=True)
        self._trainable = OrderedDict(self._params)
        self._slots = [self.engine.store.slots[k] for k in self._trainable]
        self._versions = None
        self.training_grad = True
        self.train(True)

    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True):
        bb = self._backbone_module
        if bb is not None:
            for k, p in bb.named_parameters():
                yield (prefix + ("." if prefix else "") + k, p)
        for k, p in self._params.items():
            yield (prefix + ("." if prefix else "") + k, p)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else OrderedDict()
        bb = self._backbone_module
        if bb is not None:
            for k, v in bb.state_dict().items():
                out[prefix + k] = v
        for k, p in self._params.items():
            out[prefix + k] = p if keep_vars else p.detach()
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        bb = self._backbone_module
        own = {k: v for k, v in state_dict.items() if k in self._params}
        rest = {k: v for k, v in state_dict.items() if k not in self._params}
        missing = [k for k in self._params if k not in own]
        if strict and missing:
            raise KeyError(f"state_dict mismatch: missing {missing[:5]}")
        full = {k: (own[k] if k in own else self.engine.store.tensors[k]) for k in self.engine.store.tensors}
        self.engine.load_state_dict(full, strict=False)
        if bb is not None and rest:
            bb.load_state_dict(rest, strict=strict)
        elif strict and rest:
            raise KeyError(f"unexpected keys (no backbone attached): {list(rest)[:5]}")
        self._versions = None
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def forward(self, x, coords, genes, task_token=None, patch_size_lv0=1024, clinical=None, **kwargs):
        if self.is_multi:
            if task_token is None:
                raise ValueError("task_token is required when multi_task > 1")
            onehots = task_token.reshape(1, -1)
        else:
            onehots = torch.zeros(1, 1, device=self.engine.device)
        return self.forward_tasks(x, coords, genes, onehots, clinical=clinical, patch_size_lv0=patch_size_lv0)

    def forward_tasks(self, x, coords, genes, task_onehots, clinical=None, patch_size_lv0=1024):
        self._sync_weight_caches()
        if isinstance(genes, dict):
            genes = [genes[k] for k in sorted(genes.keys())] if all(isinstance(k, int) for k in genes) else list(genes.values())
        need = torch.is_grad_enabled() and self.training_grad
        if not self.CLINICAL:
            clinical = None
        return _TitanFn.apply(self, x, coords, genes, task_onehots.to(self.engine.device, F32), need, clinical, int(patch_size_lv0),
                              *self._trainable.values())


class _TitanFn(torch.autograd.Function):
    """Same bridge as aggregators._ModelFn with the TITAN entry point."""

    @staticmethod
    def forward(ctx, module, x, coords, genes, onehots, need, clinical, psz, *params):
        eng = module.engine
        logits = eng.forward_slide(x, coords, genes, onehots, patch_size_lv0=psz, need_grad=need, fresh=need, clinical=clinical)
        ctx.module, ctx.call = module, (eng.last_call if need else None)
        return logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        grads = _ModelFn.backward(ctx, dlogits)
        return (None,) * 8 + grads[7:]


@Aggregator.register("titan_gene_clinical_adapter")
class TITANGeneSimpleClinicalAdapter(TITANGeneAdapter):
    """Clinical-prior variant (TA:441-...): one extra clinical token, as in longnetvit_gene_clinical_adapter."""
    CLINICAL = True
