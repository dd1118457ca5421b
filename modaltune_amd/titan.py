"""TITAN configuration of the Modal Adapter (BASELINE config 4): `titan_gene_adapter` /
`titan_gene_clinical_adapter` (reference models/aggregators/titan_adapter.py:42-438, 441-...).

Native (HIP kernels, same tape and flat parameter buffers as the Prov-GigaPath path of engine.Engine):
  * everything the reference's TITAN adapter adds around the slide encoder: feature gridding + background drop of
    `preprocess_features` / `prepare_forward_features` (TA:295-327, 253-293) on the device (csrc/titan.hip), the interaction
    blocks `InteractionBlockWithCls_TITAN` (adapter_modules.py:526-558), prompt self-attention, gene encoder, task tokens,
    the fusion head on the pooled image token (TA:399-437);
  * the frozen slide encoder itself, when the supplied `VisionTransformer` has the standard pre-norm ViT structure
    (`NativeBackbone`): patch-embedding MLP + cls + `norm_pre`, blocks = LayerNorm -> qkv -> dense attention with the 2-D ALiBi
    bias from one fp16 cell-distance table per slide (csrc/dense_attn.hip) -> proj (+ layer scale) -> LayerNorm -> fc1 -> GELU -> fc2, with
    activation-gradient (dX-only) backward through all of it, and the attentional pooling.

The TITAN snapshot's source and weights (HF MahmoodLab/TITAN @ b2fb4f47, utils/constants.py:22-23) are NOT in the reference
tree: its arithmetic cannot be restated from source, so BACKBONE PARITY IS UNPINNED against the real snapshot.  What replaces the
pin: `NativeBackbone` reads the structure off the module the user supplies, derives the ALiBi slopes from the module's own
`get_alibi`, and at construction runs every native piece (embedding, each block, pooling) against that module's torch code on a
probe slide; anything that does not reproduce it RAISES (`backbone_impl="native"`, the default since round 4).  The module's own
torch code between our kernels (`TorchBackbone`, gradients by `torch.autograd.grad` on the recorded call) is an explicit opt-in:
`backbone_impl="torch"`, or "auto" (native, else torch with a `warnings.warn` naming the reason).
tests/test_titan_gpu.py pins the adapter flow against the REFERENCE's titan_adapter.py run on a stand-in backbone
(tests/golden/titan_standin.py) with either implementation of the frozen blocks.

Bags are ragged (a different number of foreground cells per slide): one slide per call, any length; a batch of slides is a
Python loop over `forward` (the reference's TITAN path is batch-1 too: TA:258-267 uses the per-slide ALiBi only for B == 1).
"""
from __future__ import annotations

import math
import warnings
from collections import OrderedDict
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import init, ops
from .aggregators import Aggregator, LongNetGeneAdapter, _bridge_backward
from .config import ModelConfig
from .engine import Engine, F32, H16, _W16
from .tape import Param, Var

I32 = torch.int32


# ------------------------------------------------------------------------------------------------ feature gridding (host indices)
def grid_index(coords: np.ndarray, patch_size_lv0: int) -> Tuple[np.ndarray, int, int]:
    """TA:304-312: cell (row, col) of every patch = floor((coords - min) / patch_size_lv0), shifted to start at 0.
    Returns (flat cell index row * W + col per patch, H, W)."""
    c = np.asarray(coords).reshape(-1, 2).astype(np.int64)
    g = np.floor_divide(c - c.min(axis=0), int(patch_size_lv0))
    g = g - g.min(axis=0)
    H, W = (int(v) + 1 for v in g.max(axis=0))
    return g[:, 0] * W + g[:, 1], H, W


def preprocess_features(features: torch.Tensor, coords, patch_size_lv0: int):
    """`TITANGeneAdapter.preprocess_features` (TA:295-327) with the DENSE outputs of the reference -- (feature_grid [1, C, H, W],
    coords_grid [1, 2, H, W] int64, bg_mask [1, H, W] bool) -- for the bring-your-own-backbone path, whose module expects them
    (index arithmetic on the host, scatter-ADD by mt_scatter_rows_f32, one pass per occurrence rank of a cell so that the sums are
    bitwise reproducible).  The native backbone never builds the grid: `device_tokens`."""
    f = features.reshape(-1, features.shape[-1]).to(dtype=F32).contiguous()
    cnp = coords.detach().cpu().numpy() if torch.is_tensor(coords) else np.asarray(coords)
    cnp = cnp.reshape(-1, 2)
    idx, H, W = grid_index(cnp, patch_size_lv0)
    dev = f.device
    grid = torch.zeros(H * W, f.shape[1], dtype=F32, device=dev)
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


# ------------------------------------------------------------------------------------------------ feature gridding (device)
class GridStage:
    """Static buffers of the on-device gridding for hipGraph replay (TrainStep.step_graphed): the five gridding kernels and the one
    host read-back (the token count) run eagerly into these; the captured step starts at the token gather and reads them.  Sized
    for the largest bag seen (grown by >= 25 %; the owner bumps its generation when they move)."""

    def __init__(self, device):
        self.dev, self.cap, self.C = device, 0, 0
        self.L = self.Lv = 0

    def ensure(self, L: int, C: int) -> bool:
        if L <= self.cap and C == self.C:
            return False
        cap = self.cap if L <= self.cap else -(-max(L, self.cap + self.cap // 4) // 256) * 256     # (a new feature width alone keeps the capacity)
        dev = self.dev
        self.f = torch.empty(cap, C, dtype=F32, device=dev)
        self.c = torch.empty(cap, 2, dtype=F32, device=dev)
        self.cells, self.cells_tok = torch.empty(cap, 2, dtype=I32, device=dev), torch.empty(cap, 2, dtype=I32, device=dev)
        self.dims, self.count = torch.empty(2, dtype=I32, device=dev), torch.empty(1, dtype=I32, device=dev)
        self.first, self.nxt, self.nz, self.pos = (torch.empty(cap, dtype=I32, device=dev) for _ in range(4))
        self.sums = torch.empty(cap, C, dtype=F32, device=dev)
        self.cap, self.C = cap, C
        return True

    def run(self, features: torch.Tensor, coords, patch_size_lv0, err: Optional[torch.Tensor]) -> int:
        """Gridding of one slide (eager) -> token count (one host read-back, as device_tokens)."""
        f = features.reshape(-1, features.shape[-1])
        L, C = f.shape
        c = (coords if torch.is_tensor(coords) else torch.as_tensor(np.asarray(coords))).reshape(-1, 2)
        if c.shape[0] != L:
            raise ValueError(f"coords has {c.shape[0]} rows for {L} patches")
        self.f[:L].copy_(f, non_blocking=True)
        self.c[:L].copy_(c.to(self.dev, non_blocking=True))
        ops.titan_grid(self.c, L, float(patch_size_lv0), self.cells, self.dims, err)
        ops.titan_cell_sums(self.f, self.cells, L, C, self.first, self.nxt, self.sums, self.nz)
        ops.titan_token_order(self.cells, self.first, self.nz, L, self.pos, self.cells_tok, self.count)
        Lv = int(self.count)
        if Lv < 1:
            raise ValueError("slide has no foreground cell")
        self.L, self.Lv = L, Lv
        return Lv

    def tokens(self):
        """Capturable half: gather the staged slide's tokens (fresh fp16 operand) -> (x16, cells [Lv, 2], dims, Lv)."""
        x16 = torch.empty(self.Lv, self.C, dtype=H16, device=self.dev)
        ops.titan_gather_tokens(self.sums, self.pos, self.L, self.C, x16)
        return x16, self.cells_tok[:self.Lv], self.dims, self.Lv


def device_tokens(features: torch.Tensor, coords, patch_size_lv0, err: Optional[torch.Tensor] = None):
    """The tokens of one slide without the H x W grid (csrc/titan.hip): occupied cells in row-major order, each the sum of its
    patches' features in patch order -- exactly the rows `x[bg_mask]` keeps of the reference's gridded tensor (TA:295-327,
    282-291).  Returns (x16 [Lv, C] fp16, cells [Lv, 2] int32 (row, col), dims int32[2] = (H, W) on the device, Lv).
    One host read-back (the token count: every shape downstream depends on it; the reference synchronises on H, W here)."""
    f = features.reshape(-1, features.shape[-1])
    dev = f.device
    f = f.to(F32).contiguous()
    c = (coords if torch.is_tensor(coords) else torch.as_tensor(np.asarray(coords))).reshape(-1, 2).to(dev, F32).contiguous()
    L, C = f.shape
    if c.shape[0] != L:
        raise ValueError(f"coords has {c.shape[0]} rows for {L} patches")
    cells, dims = torch.empty(L, 2, dtype=I32, device=dev), torch.empty(2, dtype=I32, device=dev)
    ops.titan_grid(c, L, float(patch_size_lv0), cells, dims, err)
    first, nxt, nz, pos = (torch.empty(L, dtype=I32, device=dev) for _ in range(4))
    sums = torch.empty(L, C, dtype=F32, device=dev)
    ops.titan_cell_sums(f, cells, L, C, first, nxt, sums, nz)
    cells_tok, count = torch.empty(L, 2, dtype=I32, device=dev), torch.empty(1, dtype=I32, device=dev)
    ops.titan_token_order(cells, first, nz, L, pos, cells_tok, count)
    Lv = int(count)
    if Lv < 1:
        raise ValueError("slide has no foreground cell")
    x16 = torch.empty(Lv, C, dtype=H16, device=dev)
    ops.titan_gather_tokens(sums, pos, L, C, x16)
    return x16, cells_tok[:Lv], dims, Lv


# ------------------------------------------------------------------------------------------------ backbone: the module's torch code
class TorchBackbone:
    """Adapter around a TITAN-like `VisionTransformer` (torch).  Frozen: no weight gradients; activation gradients through a
    block come from torch.autograd.grad on the recorded call.  This is the ONLY place torch arithmetic runs on this path."""
    kind = "torch"

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


# ------------------------------------------------------------------------------------------------ backbone: HIP kernels
class Unsupported(Exception):
    """The supplied module does not have the structure / arithmetic the native backbone implements."""


def _is_identity(m) -> bool:
    return m is None or isinstance(m, nn.Identity)


def _linear_of(m, what: str) -> nn.Linear:
    if not isinstance(m, nn.Linear):
        raise Unsupported(f"{what} is {type(m).__name__}, not nn.Linear")
    return m


def _norm_of(m, what: str, D: int) -> nn.LayerNorm:
    if not isinstance(m, nn.LayerNorm) or tuple(m.normalized_shape) != (D,) or m.weight is None or m.bias is None:
        raise Unsupported(f"{what} is not an affine nn.LayerNorm({D})")
    return m


def _ln_params(m: nn.LayerNorm, dev):
    return m.weight.detach().to(dev, F32).contiguous(), m.bias.detach().to(dev, F32).contiguous(), float(m.eps)


class _Lin:
    """fp16 caches (as stored + transposed) and fp32 bias of one frozen nn.Linear, with an optional per-output-row scale folded
    in (layer scale: x + gamma * f(x) == x + f'(x) with W' = diag(gamma) W, b' = gamma b; the attention's q rows: the softmax
    scale in log2 units)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], dev, row_scale: Optional[torch.Tensor] = None, need_t: bool = True):
        W = weight.detach().to(dev, F32)
        b = bias.detach().to(dev, F32) if bias is not None else torch.zeros(W.shape[0], dtype=F32, device=dev)
        if row_scale is not None:
            rs = row_scale.to(dev, F32)
            W, b = W * rs[:, None], b * rs
        self.N, self.K = W.shape
        if self.K % 64 or self.N % 8:
            raise Unsupported(f"nn.Linear {self.K} -> {self.N}: the GEMM kernels need K % 64 == 0 and N % 8 == 0")
        w16 = _W16([W.contiguous()], dev, need_t=need_t)
        self.w, self.wt, self.b = w16.w, w16.wt, b.contiguous()


def _is_erf_gelu(m) -> bool:
    return isinstance(m, nn.GELU) and getattr(m, "approximate", "none") == "none"


class NativeBackbone:
    """The frozen TITAN-style ViT on the HIP kernels (module docstring).  Construct with the user's `VisionTransformer`; raises
    `Unsupported` (with the reason) when a structural or numerical check fails."""
    kind = "native"
    alibi_table_budget_bytes = 16 << 30      # refuse (clearly) a slide whose fp16 distance table would exceed this (N ~ 92 k tokens)
    PROBE_TOL = 2e-2          # fp16-operand kernels vs the module's fp32 torch code, relative max error on the probe slide

    def __init__(self, vit: nn.Module, device, self_check: bool = True):
        self.vit, self.dev = vit.to(device), torch.device(device)
        for p in vit.parameters():
            p.requires_grad_(False)
        dev = self.dev
        try:
            blocks = list(vit.blocks.modules_list)
        except AttributeError:
            raise Unsupported("no blocks.modules_list")
        if not blocks:
            raise Unsupported("no blocks")
        self.depth = len(blocks)
        # The native blocks are deterministic; the reference keeps the frozen backbone's stochastic layers alive in train mode
        # (SURVEY fact 3).  A module that HAS such layers with p > 0 is not what the kernels implement.
        # Such a module still runs natively in eval mode, where those layers are identities (embedding extraction, TM:252-327);
        # putting the MODEL in train mode is what is refused (TITANGeneAdapter.train), not its construction.
        self.stochastic_layers = []
        for nm, m in vit.named_modules():
            p = getattr(m, "p", None) if isinstance(m, (nn.Dropout, nn.AlphaDropout)) else getattr(m, "drop_prob", None)
            if isinstance(p, (int, float)) and p > 0:
                self.stochastic_layers.append(f"{nm}: {type(m).__name__}(p = {p})")
        D = None
        self.blocks: List[dict] = []
        for l, blk in enumerate(blocks):
            attn, mlp = getattr(blk, "attn", blk), getattr(blk, "mlp", blk)
            qkv, proj = _linear_of(getattr(attn, "qkv", None), f"block {l} qkv"), _linear_of(getattr(attn, "proj", None), f"block {l} proj")
            fc1, fc2 = _linear_of(getattr(mlp, "fc1", None), f"block {l} fc1"), _linear_of(getattr(mlp, "fc2", None), f"block {l} fc2")
            D = qkv.in_features if D is None else D
            if qkv.in_features != D or qkv.out_features != 3 * D or proj.in_features != D or proj.out_features != D:
                raise Unsupported(f"block {l}: qkv / proj shapes")
            if fc1.in_features != D or fc2.out_features != D or fc2.in_features != fc1.out_features:
                raise Unsupported(f"block {l}: MLP shapes")
            for nm in ("q_norm", "k_norm"):
                if not _is_identity(getattr(attn, nm, None)):
                    raise Unsupported(f"block {l}: {nm} is not Identity")
            act = getattr(mlp, "act", None)
            if act is not None and not _is_erf_gelu(act):
                raise Unsupported(f"block {l}: MLP activation {type(act).__name__} is not erf-GELU")
            heads = getattr(attn, "num_heads", None) or getattr(blk, "heads", None) or getattr(vit, "num_heads", None)
            if not heads or D % int(heads) or D // int(heads) != 64:
                raise Unsupported(f"block {l}: head dim {D}/{heads} (the dense attention kernels are built for 64)")
            n1, n2 = _norm_of(getattr(blk, "norm1", None), f"block {l} norm1", D), _norm_of(getattr(blk, "norm2", None), f"block {l} norm2", D)
            gam = []
            for nm in ("ls1", "ls2"):
                ls = getattr(blk, nm, None)
                if _is_identity(ls):
                    gam.append(None)
                elif isinstance(getattr(ls, "gamma", None), torch.Tensor) and ls.gamma.numel() == D:
                    gam.append(ls.gamma.detach().reshape(D))
                else:
                    raise Unsupported(f"block {l}: {nm} is {type(ls).__name__}")
            qscale = torch.ones(3 * D)
            qscale[:D] = ops.DENSE_QK_SCALE_LOG2          # q' = 64^-1/2 log2(e) q, baked before the fp16 rounding
            n1w, n1b, n1e = _ln_params(n1, dev)
            n2w, n2b, n2e = _ln_params(n2, dev)
            blkw = dict(qkv=_Lin(qkv.weight, qkv.bias, dev, qscale), proj=_Lin(proj.weight, proj.bias, dev, gam[0]),
                        fc1=_Lin(fc1.weight, fc1.bias, dev), fc2=_Lin(fc2.weight, fc2.bias, dev, gam[1]),
                        n1w=n1w, n1b=n1b, n1e=n1e, n2w=n2w, n2b=n2b, n2e=n2e)
            blkw["cw"] = ops.struct_of(      # pointer table of the composite launcher (mt_vit_block_fwd / _bwd)
                ops.MtVitBlockWeights, n1_w=n1w, n1_b=n1b, n2_w=n2w, n2_b=n2b, n1_eps=n1e, n2_eps=n2e, b_qkv=blkw["qkv"].b, b_proj=blkw["proj"].b,
                b_fc1=blkw["fc1"].b, b_fc2=blkw["fc2"].b, w_qkv=blkw["qkv"].w, w_proj=blkw["proj"].w, w_fc1=blkw["fc1"].w, w_fc2=blkw["fc2"].w,
                wt_qkv=blkw["qkv"].wt, wt_proj=blkw["proj"].wt, wt_fc1=blkw["fc1"].wt, wt_fc2=blkw["fc2"].wt)
            self.blocks.append(blkw)
            self.H, self.F = int(heads), fc1.out_features
        if D != 768:
            raise Unsupported(f"embed dim {D}: the fused residual + LayerNorm kernel is built for 768")
        if any(b["fc1"].N != self.F for b in self.blocks):
            raise Unsupported("blocks with different MLP widths")
        self.D = D
        self.nfw, self.nfb, self.nfe = _ln_params(_norm_of(getattr(vit, "norm", None), "norm", D), dev)      # final norm (TA:401)
        # ALiBi slopes from the module's own get_alibi
        self.alibi = getattr(vit, "pos_encode_type", None) == "alibi"
        self.nslope = self._derive_slopes() if self.alibi else None
        # optional native pieces (each verified by the probe; None = the module's torch code does it)
        self.embed_w = self._find_embed()
        self.pool_w = self._find_pool()
        self.report: Dict[str, Any] = {"blocks": "native", "embed": "native" if self.embed_w else "torch",
                                       "pool": "native" if self.pool_w else "torch"}
        if self_check:
            was_training = vit.training
            vit.eval()                      # the probe compares against the module's DETERMINISTIC arithmetic
            try:
                self._self_check()
            finally:
                vit.train(was_training)

    # -- structure discovery
    def _derive_slopes(self) -> torch.Tensor:
        """slope_h from `get_alibi` on a 4 x 3 probe grid with two background cells; raises unless the whole tensor is
        -slope_h * euclidean cell distance with zeros to / from cls."""
        v = self.vit
        mask = torch.ones(1, 4, 3, dtype=torch.bool, device=self.dev)
        mask[0, 1, 1] = mask[0, 3, 0] = False
        with torch.no_grad():
            bias = v.get_alibi(4, 3, mask).to(self.dev, torch.float64)
        pos = torch.nonzero(mask[0]).to(torch.float64)                     # row-major foreground cells
        T = pos.shape[0] + 1
        if bias.dim() != 4 or tuple(bias.shape[-2:]) != (T, T) or bias.shape[1] != self.H:
            raise Unsupported(f"get_alibi returned {tuple(bias.shape)} for {T} tokens / {self.H} heads")
        dist = torch.cdist(pos, pos)
        slopes = -bias[0, :, 1, 2] / dist[0, 1]
        want = torch.zeros(self.H, T, T, dtype=torch.float64, device=self.dev)
        want[:, 1:, 1:] = -slopes.view(-1, 1, 1) * dist
        if not torch.allclose(bias[0], want, rtol=1e-4, atol=1e-5) or not bool((slopes >= 0).all()):
            raise Unsupported("get_alibi is not -slope_h * euclidean cell distance with a zero cls row / column")
        return (-slopes * math.log2(math.e)).to(F32).contiguous()

    def _find_embed(self):
        """patch_embed = Sequential(Linear, GELU, Linear[, GELU, Linear ...]) + cls concat + norm_pre, or None."""
        v = self.vit
        pe = getattr(v, "patch_embed", None)
        try:
            if not isinstance(pe, nn.Sequential) or len(pe) < 1 or len(pe) % 2 == 0:
                return None
            lins = []
            for i, m in enumerate(pe):
                if i % 2 == 0:
                    m = _linear_of(m, "patch_embed")
                    lins.append(_Lin(m.weight, m.bias, self.dev, need_t=False))
                elif not _is_erf_gelu(m):
                    return None
            cls = getattr(v, "cls_token", None)
            w, b, eps = _ln_params(_norm_of(getattr(v, "norm_pre", None), "norm_pre", self.D), self.dev)
            if cls is None or cls.numel() != self.D or lins[-1].N != self.D:
                return None
        except Unsupported:
            return None
        return dict(lins=lins, cls=cls.detach().to(self.dev, F32).reshape(1, self.D).contiguous(), w=w, b=b, eps=eps)

    def _find_pool(self):
        """One-query attentional pooling: (query, nn.MultiheadAttention over the normed tokens, LayerNorm on the pooled row)
        under the attribute names of the stand-in or of an open_clip-style AttentionalPooler, or None."""
        v, D, dev = self.vit, self.D, self.dev
        cands = [("pool_query", "pool_attn", None, None, "pool_norm"),
                 ("attn_pool_contrast.query", "attn_pool_contrast.attn", "attn_pool_contrast.ln_q", "attn_pool_contrast.ln_k", "ln_contrast")]

        def get(path):
            if not path:
                return None
            o = v
            for part in path.split("."):
                o = getattr(o, part, None)
                if o is None:
                    return None
            return o
        for qn, an, lqn, lkn, pn in cands:
            q, mha, lq, lk, post = get(qn), get(an), get(lqn), get(lkn), get(pn)
            if q is None or not isinstance(mha, nn.MultiheadAttention) or mha.in_proj_weight is None or mha.embed_dim != D:
                continue
            if q.numel() != D or not isinstance(post, nn.LayerNorm) or (lk is not None and not isinstance(lk, nn.LayerNorm)):
                continue
            E, heads = mha.embed_dim, mha.num_heads
            if (E // heads) % 8 or E // heads > 128:
                continue
            try:
                with torch.no_grad():
                    qv = q.detach().reshape(1, D).to(dev, F32)
                    if lq is not None:
                        qv = lq.to(dev)(qv)
                    Wi, bi = mha.in_proj_weight.detach().to(dev, F32), mha.in_proj_bias.detach().to(dev, F32)
                    qp = torch.nn.functional.linear(qv, Wi[:E], bi[:E]).contiguous()     # frozen query: projected once, here
                pre = [(self.nfw, self.nfb, self.nfe)]
                if lk is not None:
                    pre.append(_ln_params(_norm_of(lk, "pool ln_k", D), dev))
                pw, pb, peps = _ln_params(_norm_of(post, "pool norm", D), dev)
                return dict(q=qp, kv=_Lin(Wi[E:], bi[E:], dev), heads=heads, E=E, pre=pre,
                            wo=mha.out_proj.weight.detach().to(dev, F32).contiguous(), bo=mha.out_proj.bias.detach().to(dev, F32).contiguous(),
                            pw=pw, pb=pb, peps=peps)
            except Unsupported:
                continue
        return None

    # -- workspace of one (B, N) geometry
    def ws_spec(self, B: int, Lx: int) -> Dict[str, tuple]:
        D, Fd, H = self.D, self.F, self.H
        M = B * (Lx + 1)
        sp = {}
        for l in range(self.depth):
            sp[f"hmid{l}"] = (F32, (M, D)); sp[f"qkv{l}"] = (H16, (M, 3 * D)); sp[f"o{l}"] = (H16, (M, D)); sp[f"lse{l}"] = (F32, (M, H))
            sp[f"a1_{l}"] = (H16, (M, Fd)); sp[f"st1_{l}"] = (F32, (M, 2)); sp[f"st2_{l}"] = (F32, (M, 2))
        for nm in ("u16", "br16", "dy16", "dh16"):
            sp[nm] = (H16, (M, D))
        for nm in ("t16", "dt16", "da1"):
            sp[nm] = (H16, (M, Fd))
        sp["dqkv16"] = (H16, (M, 3 * D)); sp["delta"] = (F32, (M, H))
        if self.pool_w:
            E = self.pool_w["E"]
            hp = self.pool_w["heads"]
            sp["pool_kv"] = (H16, (M, 2 * E)); sp["pool_dkv"] = (H16, (M, 2 * E)); sp["pool_probs"] = (F32, (B * hp * (Lx + 1),))
            sp["pool_lse"] = (F32, (B * hp,)); sp["pool_part"] = (F32, (ops.pool_attn_workspace_floats(B, Lx + 1, hp, 1),))
            for j in range(len(self.pool_w["pre"])):
                sp[f"pool_st{j}"] = (F32, (M, 2))
                if j > 0:
                    sp[f"pool_x{j}"] = (F32, (M, D))
        return sp

    # -- embedding: patch-embedding MLP + cls + norm_pre (TA:276-280)
    def embed(self, x16: torch.Tensor, Lv: int) -> torch.Tensor:
        """x16 [Lv, C] fp16 summed cell features -> tokens [1 + Lv, D] fp32 (cls first)."""
        e, D, dev = self.embed_w, self.D, self.dev
        cur = x16
        pre = torch.empty(Lv + 1, D, dtype=F32, device=dev)
        for i, lin in enumerate(e["lins"]):
            if i == len(e["lins"]) - 1:
                ops.gemm_nt(cur, lin.w, pre[1:], Lv, lin.N, lin.K, bias=lin.b)
            else:
                h = torch.empty(Lv, lin.N, dtype=H16, device=dev)
                ops.gemm_nt(cur, lin.w, h, Lv, lin.N, lin.K, bias=lin.b)
                g = torch.empty_like(h)
                ops.gelu_f16_fwd(h, g)
                cur = g
        ops.copy_rows(e["cls"], pre, 1, D)
        tok = torch.empty(Lv + 1, D, dtype=F32, device=dev)
        st = torch.empty(Lv + 1, 2, dtype=F32, device=dev)
        ops.layernorm_fwd(pre, e["w"], e["b"], tok, st, Lv + 1, D, eps=e["eps"])
        return tok

    # -- one frozen pre-norm block
    def block_fwd(self, l: int, ws: Dict[str, torch.Tensor], M: int, plan, hin: torch.Tensor, out: torch.Tensor, pend=None, defer: bool = False):
        """hin -> out (fp32 [M, D]).  The residual adds ride on the LayerNorm that consumes the sum (mt_add_layernorm_fwd), as in
        Engine._layer: `pend` = (stream, branch) of the block below whose fc2 add is outstanding (hin is then WRITTEN here); with
        `defer` this block leaves its own fc2 add to the block above and returns such a pair."""
        w, D, Fd = self.blocks[l], self.D, self.F
        u16, br16, t16 = ws["u16"], ws["br16"], ws["t16"]
        hmid, qkv, o16, lse, a1 = ws[f"hmid{l}"], ws[f"qkv{l}"], ws[f"o{l}"], ws[f"lse{l}"], ws[f"a1_{l}"]
        if ops.TIMER is None:      # one C call (csrc/layer.hip enqueues exactly the launches spelled out below)
            ops.vit_block_fwd(w["cw"], self._cbuf(l, ws, hin), plan, M, D, Fd, out, pend=pend, defer=defer)
            return (hmid, br16) if defer else None
        if pend is None:
            ops.layernorm_fwd(hin, w["n1w"], w["n1b"], u16, ws[f"st1_{l}"], M, D, eps=w["n1e"])
        else:
            ops.add_layernorm_fwd(pend[0], pend[1], w["n1w"], w["n1b"], hin, u16, ws[f"st1_{l}"], M, D, eps=w["n1e"])
        ops.gemm_nt(u16, w["qkv"].w, qkv, M, 3 * D, D, bias=w["qkv"].b)
        ops.dense_attn_fwd(qkv, plan, o16, lse)
        ops.gemm_nt(o16, w["proj"].w, br16, M, D, D, bias=w["proj"].b)
        ops.add_layernorm_fwd(hin, br16, w["n2w"], w["n2b"], hmid, u16, ws[f"st2_{l}"], M, D, eps=w["n2e"])      # hmid = hin + attention branch
        ops.gemm_nt(u16, w["fc1"].w, a1, M, Fd, D, bias=w["fc1"].b)
        ops.gelu_f16_fwd(a1, t16, M * Fd)
        if defer:
            ops.gemm_nt(t16, w["fc2"].w, br16, M, D, Fd, bias=w["fc2"].b)
            return (hmid, br16)
        ops.gemm_nt(t16, w["fc2"].w, out, M, D, Fd, epilogue=ops.EPI_BIAS_RESID, bias=w["fc2"].b, resid=hmid, ldr=D)
        return None

    def _cbuf(self, l: int, ws, hin: torch.Tensor, dh: Optional[torch.Tensor] = None):
        """Pointer table of block l's buffers in workspace `ws` (cached in it; `dh` is known from the first backward on)."""
        cb = ws.get(("_cb", l))
        if cb is None:
            cb = ws[("_cb", l)] = ops.struct_of(
                ops.MtVitBlockBuffers, hin=hin, hmid=ws[f"hmid{l}"], qkv=ws[f"qkv{l}"], o16=ws[f"o{l}"], lse=ws[f"lse{l}"], a1=ws[f"a1_{l}"],
                st1=ws[f"st1_{l}"], st2=ws[f"st2_{l}"], u16=ws["u16"], br16=ws["br16"], t16=ws["t16"], dy16=ws["dy16"], dh16=ws["dh16"],
                dt16=ws["dt16"], da1=ws["da1"], dqkv16=ws["dqkv16"], delta=ws["delta"])
        if dh is not None:
            cb.dh = dh.data_ptr()
        return cb

    def block_bwd(self, l: int, ws: Dict[str, torch.Tensor], M: int, plan, hin: torch.Tensor, dh: torch.Tensor, dh16_valid: bool, feeds_lower: bool):
        """dh (fp32 [M, D], gradient of the block's output) -> gradient of its input, in place; activation gradients only."""
        w, D, Fd = self.blocks[l], self.D, self.F
        dy16, dt16, da1, u16 = ws["dy16"], ws["dt16"], ws["da1"], ws["u16"]
        if ops.TIMER is None:
            return ops.vit_block_bwd(w["cw"], self._cbuf(l, ws, hin, dh), plan, M, D, Fd, dh16_valid, feeds_lower)
        if dh16_valid:
            src16 = ws["dh16"]
        else:
            ops.cast_f32_to_f16(dh, dy16, M * D)
            src16 = dy16
        ops.gemm_nt(src16, w["fc2"].wt, dt16, M, Fd, D)
        ops.gelu_f16_bwd(ws[f"a1_{l}"], dt16, da1, M * Fd)
        ops.gemm_nt(da1, w["fc1"].wt, dy16, M, D, Fd)
        ops.layernorm_bwd(dy16, ws[f"hmid{l}"], w["n2w"], ws[f"st2_{l}"], dh, M, D, accumulate=True, dx16=ws["dh16"])
        ops.gemm_nt(ws["dh16"], w["proj"].wt, u16, M, D, D)                                   # dO
        ops.dense_attn_bwd(ws[f"qkv{l}"], ws[f"o{l}"], u16, ws[f"lse{l}"], plan, ws["delta"], ws["dqkv16"])
        ops.gemm_nt(ws["dqkv16"], w["qkv"].wt, dy16, M, D, 3 * D)
        ops.layernorm_bwd(dy16, hin, w["n1w"], ws[f"st1_{l}"], dh, M, D, accumulate=True, dx16=ws["dh16"] if feeds_lower else None)

    # -- final norm + attentional pooling (TA:400-402)
    def pool_fwd(self, tape, ws: Dict[str, torch.Tensor], B: int, N: int, hout: torch.Tensor) -> Tuple[Var, Var]:
        p, D, M = self.pool_w, self.D, B * N
        E, heads = p["E"], p["heads"]
        cur = hout
        for j, (w_, b_, e_) in enumerate(p["pre"]):          # norm (+ ln_k): the last one writes the fp16 GEMM operand
            dst = ws["u16"] if j == len(p["pre"]) - 1 else ws[f"pool_x{j + 1}"]
            ops.layernorm_fwd(cur, w_, b_, dst, ws[f"pool_st{j}"], M, D, eps=e_)
            cur = dst
        ops.gemm_nt(ws["u16"], p["kv"].w, ws["pool_kv"], M, 2 * E, D, bias=p["kv"].b)
        pooled = Var(tape.new(B, 1, E))
        ops.pool_attn_fwd(p["q"], ws["pool_kv"], pooled.data, ws["pool_probs"], ws["pool_lse"], ws["pool_part"], B, N, E, heads, 1)
        y = tape.linear(pooled, Param(p["wo"], None), Param(p["bo"], None))
        # the pooled row's LayerNorm may carry its own eps: launched here (frozen affine), not through tape.layernorm
        img = Var(tape.new(B, D))
        st = tape.new(B, 2)
        ops.layernorm_fwd(y.data.view(B, D), p["pw"], p["pb"], img.data, st, B, D, eps=p["peps"])

        def bwd_ln():
            if img.grad is not None:
                ops.layernorm_bwd(img.grad, y.data.view(B, D), p["pw"], st, y.g().view(B, D), B, D, accumulate=True)
        tape.record(bwd_ln)
        return img, pooled

    def pool_bwd(self, ws: Dict[str, torch.Tensor], B: int, N: int, hout: torch.Tensor, pooled: Var, dh: torch.Tensor):
        """dh (overwritten) = gradient of the block stack's output through pooling and the final norm(s)."""
        p, D, M = self.pool_w, self.D, B * N
        E, heads = p["E"], p["heads"]
        if pooled.grad is None:
            dh.zero_()
            return
        ops.pool_attn_bwd(p["q"], ws["pool_kv"], ws["pool_probs"], ws["pool_lse"], pooled.data, pooled.grad, ws["pool_dkv"], B, N, E, heads, 1)
        ops.gemm_nt(ws["pool_dkv"], p["kv"].wt, ws["dy16"], M, D, 2 * E)
        dcur = ws["dy16"]
        for j in range(len(p["pre"]) - 1, -1, -1):
            w_, b_, e_ = p["pre"][j]
            if j == 0:
                ops.layernorm_bwd(dcur, hout, w_, ws[f"pool_st{j}"], dh, M, D)
            else:
                tmp = torch.empty(M, D, dtype=F32, device=self.dev)
                ops.layernorm_bwd(dcur, ws[f"pool_x{j}"], w_, ws[f"pool_st{j}"], tmp, M, D)
                dcur = tmp

    def plan_with(self, keep, N: int, B: int):
        """A plan for another pass count on the slide's existing distance table (`keep` as returned by make_plan)."""
        if not self.alibi or not keep:
            return ops.make_dense_plan(N, B, self.H)
        return ops.make_dense_plan(N, B, self.H, keep[0], self.nslope)

    def make_plan(self, cells: torch.Tensor, dims: torch.Tensor, N: int, B: int, err: Optional[torch.Tensor] = None):
        """Dense-attention plan of one slide: the fp16 distance table of the token cells (one per slide: every head, pass, block
        and kernel of the step reads the same one).  Returns (plan, tensors to keep alive)."""
        if not self.alibi:
            return ops.make_dense_plan(N, B, self.H), ()
        need = 2 * ops.alibi_dist_halves(N)          # the table is O(N^2): 34 MB at N = 4097, 0.8 GB at 20 k, 4.3 GB at 46 k tokens
        # static budget first (the only check inside a captured region); then the allocation itself decides -- the driver's "free"
        # figure does not count blocks the caching allocator holds (the previous slide's table among them), so a slide that fits
        # would be refused after a few ragged ones
        if need > self.alibi_table_budget_bytes:
            raise ValueError(f"TITAN slide with {N - 1} foreground cells: the ALiBi distance table needs {need / 2**30:.2f} GiB of fp16 "
                             f"(2 N^2 bytes; budget `backbone.alibi_table_budget_bytes` = {self.alibi_table_budget_bytes / 2**30:.1f} GiB): "
                             "subsample the slide or raise the budget")
        try:
            dist = torch.empty(ops.alibi_dist_halves(N), dtype=H16, device=self.dev)
        except torch.OutOfMemoryError as e:
            raise ValueError(f"TITAN slide with {N - 1} foreground cells: the ALiBi distance table needs {need / 2**30:.2f} GiB of fp16 "
                             f"(2 N^2 bytes) and the device has no room for it: subsample the slide") from e
        ops.alibi_dist(cells.contiguous(), N, dist)
        return ops.make_dense_plan(N, B, self.H, dist, self.nslope), (dist,)

    # -- construction-time check of every native piece against the module's torch code
    def _probe_inputs(self, seed: int = 7):
        g = torch.Generator().manual_seed(seed)
        grid, psz, L = 9, 1024, 70
        cells = torch.randperm(grid * grid, generator=g)[:L - 6]
        cells = torch.cat([cells, cells[:6]])                                  # six cells hold two patches
        rc = torch.stack([cells // grid, cells % grid], 1)
        coords = rc * psz + torch.randint(0, psz, (L, 2), generator=g) + 5 * psz + 11
        pe = getattr(self.vit, "patch_embed", None)
        C = pe[0].in_features if isinstance(pe, nn.Sequential) and isinstance(pe[0], nn.Linear) else self.D
        x = torch.randn(L, C, generator=g)
        return x.to(self.dev), coords.to(self.dev), psz

    def _self_check(self):
        dev, D = self.dev, self.D
        tb = TorchBackbone(self.vit)
        x, coords, psz = self._probe_inputs()
        fg, cg, bgm = preprocess_features(x, coords, psz)
        with torch.no_grad():
            tok_t, bias_t, mask_t = tb.embed(fg, cg, bgm)
        tok_t = tok_t[0].to(F32)
        x16, cells, dims, Lv = device_tokens(x, coords, psz)
        if Lv + 1 != tok_t.shape[0]:
            raise Unsupported(f"probe: {Lv} device tokens vs {tok_t.shape[0] - 1} from the module")
        rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
        if self.embed_w is not None:
            err = rel(self.embed(x16, Lv), tok_t)
            self.report["embed_err"] = err
            if not err < self.PROBE_TOL:
                warnings.warn(f"modaltune_amd.titan: native patch embedding differs from the module's by {err:.2e} on the probe slide "
                              "-> the module's torch embedding is used")
                self.embed_w, self.report["embed"] = None, "torch"
        N, B = Lv + 1, 2
        M = B * N
        plan, keep = self.make_plan(cells, dims, N, B)
        ws = {k: torch.empty(shape, dtype=dt, device=dev) for k, (dt, shape) in self.ws_spec(B, Lv).items()}
        gen = torch.Generator().manual_seed(11)
        hin = torch.cat([tok_t, tok_t + 0.1 * torch.randn(N, D, generator=gen).to(dev)]).contiguous()      # two different "passes"
        dout = torch.randn(M, D, generator=gen).to(dev)
        worst = 0.0
        for l in range(self.depth):
            out = torch.empty(M, D, dtype=F32, device=dev)
            self.block_fwd(l, ws, M, plan, hin, out)
            dh = dout.clone()
            self.block_bwd(l, ws, M, plan, hin, dh, False, False)
            y_t, handle = tb.block(l, hin.view(B, N, D), bias_t, mask_t, True)
            dx_t = tb.backward(handle, dout.view(B, N, D))
            e1, e2 = rel(out, y_t.reshape(M, D)), rel(dh, dx_t.reshape(M, D))
            worst = max(worst, e1, e2)
            if not (e1 < self.PROBE_TOL and e2 < 2 * self.PROBE_TOL):
                raise Unsupported(f"block {l}: native forward / input gradient differ from the module's by {e1:.2e} / {e2:.2e} on the probe slide")
        self.report["block_err"] = worst
        if self.pool_w is not None:
            from .tape import Tape
            tape = Tape(dev)
            tape.reset()
            img, pooled = self.pool_fwd(tape, ws, B, N, hin)
            gi = torch.randn(B, D, generator=gen).to(dev)
            img.grad = gi.clone()
            tape.run_backward()
            dh = torch.empty(M, D, dtype=F32, device=dev)
            self.pool_bwd(ws, B, N, hin, pooled, dh)
            y_t, handle = tb.pool(hin.view(B, N, D), mask_t, True)
            dx_t = tb.backward(handle, gi)
            e1, e2 = rel(img.data, y_t.reshape(B, D)), rel(dh, dx_t.reshape(M, D))
            self.report["pool_err"] = max(e1, e2)
            if not (e1 < self.PROBE_TOL and e2 < 2 * self.PROBE_TOL):
                warnings.warn(f"modaltune_amd.titan: native attentional pooling differs from the module's by {e1:.2e} / {e2:.2e} on the "
                              "probe slide -> the module's torch pooling is used")
                self.pool_w, self.report["pool"] = None, "torch"
        torch.cuda.synchronize()
        del keep


# ------------------------------------------------------------------------------------------------ engine
class TitanEngine(Engine):
    """Engine with the frozen image side on a TITAN backbone (NativeBackbone or TorchBackbone); adapters / tokens / head as in
    Engine."""

    def __init__(self, cfg: ModelConfig, group_sizes: Sequence[int], backbone, device="cuda"):
        super().__init__(cfg, group_sizes, device)
        self.backbone = backbone
        self._tok = None
        self._titan_err = torch.zeros(1, dtype=I32, device=self.device)

    @property
    def native(self) -> bool:
        return self.backbone is not None and self.backbone.kind == "native"

    def _build_caches(self):         # no LongNet weights to pack: only the trainable big-M adapter linears
        t, dev = self.store.tensors, self.device
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

    def _ws_spec(self, B: int, Lx: int) -> Dict[str, tuple]:
        cfg, D = self.cfg, self.cfg.embed_dim
        M, Mp = B * (Lx + 1), B * Lx
        sp = {"x0": (F32, (Lx, D)), "dh": (F32, (M, D)), "scratch32": (F32, (Mp, D))}
        for l in range(cfg.depth):
            sp[f"hin{l}"] = (F32, (M, D))
        for i in range(len(cfg.interaction_indexes)):
            sp[f"hout{i}"] = (F32, (M, D))
        if self.stochastic:
            sp["x0d"] = (F32, (4,))       # (Engine._workspace keys its rebuild on this name; the TITAN config has no input dropout)
        if self.native:
            sp.update(self.backbone.ws_spec(B, Lx))
        return sp

    def _attention_plan(self, N: int, B: int):
        return self._plan if self.native else None

    def check_inputs(self):
        if int(self._titan_err) != 0:
            self._titan_err.zero_()
            raise ValueError("TITAN slide: non-finite or out-of-range coordinates")
        super().check_inputs()

    def stage_slide(self, x, coords, patch_size_lv0: int = 1024) -> int:
        """Eager half of a hipGraph-replayed step: the slide's gridding into static buffers and the one host read-back -> token count
        (the key of the captured geometry).  forward_slide(..., staged=True) continues from the token gather."""
        if self.backbone is None or not self.native or self.backbone.embed_w is None:
            raise NotImplementedError("hipGraph replay of the TITAN configuration needs the native backbone with its native embedding")
        if getattr(self, "_grid_stage", None) is None:
            self._grid_stage = GridStage(self.device)
        x = x.reshape(-1, x.shape[-1])
        if x.shape[0] < 1:
            raise ValueError("empty bag: 0 tile embeddings (a slide needs at least one)")
        if self._grid_stage.ensure(x.shape[0], x.shape[1]):
            self.generation += 1            # captured graphs read the old buffers
        return self._grid_stage.run(x, coords, patch_size_lv0, self._titan_err)      # (ONE copy: host / device, any float dtype -> the static fp32 buffer)

    def forward_slide(self, x, coords, genes, task_onehots, patch_size_lv0: int = 1024, need_grad: bool = True, fresh: bool = False,
                      clinical=None, share: Optional[dict] = None, staged: bool = False, tape=None, site_group: int = 0,
                      prologue_only: bool = False, ws_slot: int = 0) -> Optional[torch.Tensor]:
        """x [1, L, C] tile embeddings, coords [1, L, 2] level-0 pixels (TA:329-353) -> logits [B, output_dim].
        share: see Engine.forward -- here the whole task-independent prologue (gridding, embedding, the ALiBi distance table) is
        reused by the later calls of one slide; a call with another pass count builds its plan on the shared table.
        prologue_only: run just that prologue into `share` (TrainStep's pass groups fork behind it); tape / site_group: Engine.forward."""
        bb = self.backbone
        if bb is None:
            raise RuntimeError("titan_gene_adapter needs the TITAN slide encoder: pass backbone=<VisionTransformer from the "
                               "MahmoodLab/TITAN snapshot> (its source is not part of ModalTune; parity unpinned)")
        B = int(task_onehots.shape[0])
        self._need = need_grad
        if not staged and x is not None and x.reshape(-1, x.shape[-1]).shape[0] < 1:
            raise ValueError("empty bag: 0 tile embeddings (a slide needs at least one)")
        if not staged and x is not None:
            x = x.to(self.device)
        if share is not None and "tok" in share:
            self._tok = share["tok"]
            self._bias, self._mask = share.get("bias"), share.get("mask")
            if share.get("titan_B") == B:
                self._plan, self._plan_keep = share["plan"], share["keep"]
            elif self.native:
                self._plan_keep = share["keep"]
                self._plan = bb.plan_with(share["keep"], self._tok.shape[0], B)
            else:
                self._plan = self._plan_keep = None
        elif self.native:
            if bb.embed_w is not None:
                if staged:          # stage_slide() has run the gridding: start at the token gather (capturable)
                    x16, cells, dims, Lv = self._grid_stage.tokens()
                else:
                    x16, cells, dims, Lv = device_tokens(x, coords, patch_size_lv0, self._titan_err)
                tok = bb.embed(x16, Lv)
            else:          # the module's own embedding (dense grid), then the cells of the kept tokens
                fg, cg, bgm = preprocess_features(x, coords, patch_size_lv0)
                t3, _, _ = TorchBackbone(bb.vit).embed(fg, cg, bgm)
                tok = t3[0].to(F32).contiguous()
                cells = torch.nonzero(bgm[0]).to(I32).contiguous()
                dims = torch.tensor(list(bgm.shape[-2:]), dtype=I32, device=self.device)
                Lv = tok.shape[0] - 1
            self._tok = tok
            self._plan, self._plan_keep = bb.make_plan(cells, dims, Lv + 1, B, self._titan_err)
        else:
            fg, cg, bgm = preprocess_features(x, coords, patch_size_lv0)
            tok, bias, mask = bb.embed(fg, cg, bgm)
            self._tok, self._bias, self._mask = tok[0].to(F32).contiguous(), bias, mask
            self._plan = self._plan_keep = None
        if share is not None and "tok" not in share:
            share.update(titan_B=B, tok=self._tok, plan=self._plan, keep=self._plan_keep, bias=getattr(self, "_bias", None),
                         mask=getattr(self, "_mask", None))
        patches = self._tok[1:]
        if patches.shape[0] < 1:
            raise ValueError("slide has no foreground cell")
        if prologue_only:
            return None
        return self.forward(patches, None, genes, task_onehots, need_grad=need_grad, fresh=fresh, clinical=clinical, share=share,
                            tape=tape, site_group=site_group, ws_slot=ws_slot)

    # -- image-side hooks
    def _embed_patches(self, x, coords, ws, staged, L):
        ops.copy_rows(self._tok[1:], ws["x0"], L, self.cfg.embed_dim)

    def _cls_source(self) -> torch.Tensor:
        return self._tok[0]

    def _layer(self, l: int, out: torch.Tensor, pend=None, defer: bool = False):
        ctx, ws, D = self._ctx, self._ctx["ws"], self.cfg.embed_dim
        B, N, M = ctx["B"], ctx["N"], ctx["M"]
        bb, need = self.backbone, self._need
        hin = ws[f"hin{l}"]
        if self.native:
            plan, keep = ctx["plan"], self._plan_keep      # (the plan points into `keep`: alive as long as this call's tape)
            nxt = bb.block_fwd(l, ws, M, plan, hin, out, pend=pend, defer=defer)
            feeds_lower = all(l != a for a, _ in self.cfg.interaction_indexes)

            def bwd():
                bb.block_bwd(l, ws, M, plan, hin, ws["dh"], bool(ctx.get("dh16_valid")), feeds_lower)
                ctx["dh16_valid"] = feeds_lower
                return keep and None
            self.tape.record(bwd)
            return nxt
        # the reference hands the cls-prefixed bg_mask to every block: blk(x, attn_bias, bg_mask) (adapter_modules.py:535)
        y, handle = bb.block(l, hin.view(B, N, D), self._bias, self._mask, need)
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
        if self.native and bb.pool_w is not None:
            holder = {}

            def bwd_patch():      # recorded FIRST so that it runs LAST of the pooling closures: d pooled is complete by then
                ctx["dh16_valid"] = False
                bb.pool_bwd(ws, B, N, hout, holder["pooled"], ws["dh"])
            self.tape.record(bwd_patch)
            img, holder["pooled"] = bb.pool_fwd(self.tape, ws, B, N, hout)
            return img
        if self.native:      # native blocks, the module's own pooling: its mask is the cls-prefixed mask of the KEPT tokens
            tb, mask = TorchBackbone(bb.vit), torch.ones(1, N, dtype=torch.bool, device=self.device)
        else:
            tb, mask = bb, self._mask
        pooled, handle = tb.pool(hout.view(B, N, D), mask, self._need)      # forward_attn_pool(x, bg_mask=bg_mask), TA:402
        img = Var(pooled.to(F32).contiguous())

        def bwd():
            dh = ws["dh"]
            ctx["dh16_valid"] = False
            if img.grad is None:
                dh.zero_()
                return
            dx = tb.backward(handle, img.grad)
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


def make_backbone(vit: Optional[nn.Module], device, impl: str = "auto"):
    """impl: "native" (raise if the module is not supported), "torch" (the module's own code), "auto" (native when every check
    passes, else torch with a warning that says why)."""
    if vit is None:
        return None
    if impl not in ("auto", "native", "torch"):
        raise ValueError(f"backbone_impl {impl!r}")
    if impl == "torch":
        return TorchBackbone(vit)
    try:
        return NativeBackbone(vit, device)
    except Unsupported as e:
        if impl == "native":
            raise Unsupported(f"{e} -- pass backbone_impl=\"torch\" to run the module's own blocks between the HIP kernels (slower; "
                              f"the adapter side is unaffected)") from e
        warnings.warn(f"modaltune_amd.titan: the supplied backbone runs as its own torch code between the HIP kernels "
                      f"(NativeBackbone: {e})")
        return TorchBackbone(vit)


@Aggregator.register("titan_gene_adapter")
class TITANGeneAdapter(LongNetGeneAdapter):
    """Drop-in for the reference's TITANGeneAdapter (TA:42-438): same registry name, ctor kwargs (keys of
    model_configs/modaltune_titan_config.json + gene_group_defination, multi_task), forward signature
    (x, coords, genes, task_token, patch_size_lv0), `is_multi`.  `backbone`: the TITAN VisionTransformer instance
    (required to run; see the module docstring); `backbone_impl`: "native" (default since round 4: the frozen blocks run on the HIP
    kernels, and a module the native path does not implement or reproduce RAISES, saying which check failed -- no silent second
    path), "torch" (explicit opt-in: the module's own torch code between our kernels), "auto" (native, else torch with a warning);
    `.backbone_impl` afterwards says
    which one runs).  state_dict holds the adapter-side keys under the reference's names; the backbone's own tensors are exposed
    un-prefixed after them, as in the reference (which inherits from the backbone)."""
    CLINICAL = False

    def __init__(self, gene_group_defination: Dict[Any, Sequence[str]] = None, multi_task: int = 1, backbone: Optional[nn.Module] = None,
                 device="cuda", backbone_impl: str = "native", init_seed: Optional[int] = None, **kwargs):
        nn.Module.__init__(self)
        gene_group_defination = gene_group_defination or {}
        self.backbone_source = "backbone= argument"
        if backbone is None:
            # The reference IS the snapshot's VisionTransformer (TA:42: it inherits from it): it imports the class from
            # TITAN_CODE_PATH / TITAN_SNAPSHOT_ID, builds it from TitanConfig().vision_config and loads model.safetensors when
            # `pretrained` (TA:16-37,88-107,233-247).  Same route here ($TITAN_CODE_PATH / $TITAN_SNAPSHOT_ID override the
            # reference's constants); `backbone=` is the extra door for a module the caller already holds.
            try:
                backbone = init.build_titan_backbone(bool(kwargs.get("pretrained", True)), device)
                self.backbone_source = "TITAN snapshot package"
            except (ImportError, FileNotFoundError) as e:
                warnings.warn(f"titan_gene_adapter: no slide encoder attached -- {e}.  The adapter side is initialised; forward() raises "
                              f"until a backbone is given.")
                self.backbone_source = None
        depth = len(backbone.blocks.modules_list) if backbone is not None else 6
        cfg = titan_model_config(kwargs, multi_task, self.CLINICAL, depth)
        self.cfg = cfg
        self.is_multi = multi_task > 1
        if backbone is not None:
            backbone = backbone.to(device)
        object.__setattr__(self, "_backbone_module", backbone)
        object.__setattr__(self, "_backbone_impl_req", backbone_impl)
        sizes = [len(v) for v in gene_group_defination.values()]
        self.engine = TitanEngine(cfg, sizes, None, device)
        # trainable side initialised as the reference's constructor leaves it (TA:195-203: same families as the LongNet adapter)
        self.engine.load_state_dict(init.init_state_dict(cfg, sizes, init_seed, trainable_only=True), strict=False)
        self._rebuild_backbone()
        self._params = OrderedDict()
        for k, shape, kind, train in self.engine.store.specs:
            if train:
                self._params[k] = nn.Parameter(self.engine.store.tensors[k], requires_grad=True)
        self._trainable = OrderedDict(self._params)
        self._slots = [self.engine.store.slots[k] for k in self._trainable]
        self._versions = None
        self.training_grad = True
        self._group = self._token = None
        self._spec = self._hist = self._spec_rows = None
        self.speculate = True
        self._init_nosync()
        self._call_psz = 1024
        try:
            self.train(True)
        except RuntimeError as e:      # a backbone with live Dropout / DropPath on the native path: usable for inference only
            warnings.warn(f"{e} -- the model is left in eval() mode")
            self.train(False)

    def train(self, mode: bool = True):
        """The reference adapter IS the backbone module (TA:42), so model.train() / .eval() reach its blocks; here the module is
        held outside nn.Module's registry, so the mode is forwarded by hand (it matters to the TorchBackbone path)."""
        eng_bb = getattr(getattr(self, "engine", None), "backbone", None)
        live = getattr(eng_bb, "stochastic_layers", None)
        if mode and live:
            raise RuntimeError("model.train(): the slide encoder has live stochastic layers (" + "; ".join(live[:3]) + (" ..." if len(live) > 3 else "")
                               + ") which the reference keeps active in train mode and the native HIP blocks do not implement; the model runs "
                               "natively in eval() -- for training pass backbone_impl=\"torch\" (the module's own blocks between the HIP "
                               "kernels, slower) or set those probabilities to 0")
        super().train(mode)
        bb = getattr(self, "_backbone_module", None)
        if bb is not None:
            bb.train(mode)
        return self

    def _rebuild_backbone(self):
        """(Re)derive the backbone implementation from the module's CURRENT weights (the fp16 caches of the native path are
        copies: load_state_dict calls this again)."""
        eng = self.engine
        eng.backbone = make_backbone(self._backbone_module, eng.device, self._backbone_impl_req)
        eng._ws.clear(); eng._ws_store.clear()
        eng.generation += 1

    @property
    def backbone_impl(self) -> Optional[str]:
        return None if self.engine.backbone is None else self.engine.backbone.kind

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
        """Adapter-side keys go to the engine, the rest to the attached slide-encoder module; returns the missing / unexpected keys
        of both halves (strict: raises on any, like nn.Module)."""
        bb = self._backbone_module
        own = {k: v for k, v in state_dict.items() if k in self._params}
        rest = {k: v for k, v in state_dict.items() if k not in self._params}
        missing = [k for k in self._params if k not in own]
        unexpected = []
        if bb is not None:
            have = bb.state_dict()
            missing = [k for k in have if k not in rest] + missing
            unexpected = [k for k in rest if k not in have]
        else:
            unexpected = list(rest)
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for {}: missing keys {}, unexpected keys {}{}".format(
                type(self).__name__, missing, unexpected, "" if bb is not None else " (no backbone attached)"))
        full = {k: (own[k] if k in own else self.engine.store.tensors[k]) for k in self.engine.store.tensors}
        self.engine.load_state_dict(full, strict=False)
        if bb is not None and rest:
            bb.load_state_dict({k: v for k, v in rest.items() if k not in unexpected}, strict=False)
            self._rebuild_backbone()
        self._versions = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def forward(self, x, coords, genes, task_token=None, patch_size_lv0=1024, clinical=None, **kwargs):
        self._call_psz = int(patch_size_lv0)
        if self.is_multi:
            if task_token is None:
                raise ValueError("task_token is required when multi_task > 1")
            return self._forward_one_task(x, coords, genes, task_token.reshape(1, -1), clinical)
        onehots = torch.zeros(1, 1, device=self.engine.device)
        return self.forward_tasks(x, coords, genes, onehots, clinical=clinical)

    def forward_tasks(self, x, coords, genes, task_onehots, clinical=None, patch_size_lv0=None):
        if patch_size_lv0 is not None:
            self._call_psz = int(patch_size_lv0)
        return super().forward_tasks(x, coords, genes, task_onehots, clinical=clinical)

    def _extra_key(self):
        return self._call_psz

    def _apply_bridge(self, x, coords, genes, onehots, need, clinical, token):
        return _TitanFn.apply(self, x, coords, genes, onehots, need, clinical, self._call_psz, token, *self._trainable.values())


class _TitanFn(torch.autograd.Function):
    """Same bridge as aggregators._ModelFn (chained calls of a step: aggregators._StepGroup) with the TITAN entry point."""

    @staticmethod
    def forward(ctx, module, x, coords, genes, onehots, need, clinical, psz, token, *params):
        eng = module.engine
        ctx.set_materialize_grads(False)
        grp = module._group if need else None
        logits = eng.forward_slide(x, coords, genes, onehots, patch_size_lv0=psz, need_grad=need, fresh=need, clinical=clinical,
                                   share=grp.share if grp is not None else None)
        ctx.module, ctx.call, ctx.group, ctx.has_pred = module, (eng.last_call if need else None), grp, token is not None
        ctx.batched = module.is_multi and logits.shape[0] > 1
        return logits.clone(), torch.zeros((), dtype=F32, device=logits.device)

    @staticmethod
    def backward(ctx, dlogits, dtoken):
        tok, grads = _bridge_backward(ctx, dlogits)
        return (None,) * 8 + (tok,) + grads


@Aggregator.register("titan_gene_clinical_adapter")
class TITANGeneSimpleClinicalAdapter(TITANGeneAdapter):
    """Clinical-prior variant (TA:441-...): one extra clinical token, as in longnetvit_gene_clinical_adapter."""
    CLINICAL = True
