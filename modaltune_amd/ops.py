"""Thin torch-tensor front end over the C-ABI launchers (device pointers in, status checked).

Each function maps 1:1 to an entry of include/modaltune_hip.h; PyTorch only supplies device memory and
the current HIP stream.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import MtLongNetLayerBuffers, MtLongNetLayerWeights, MtVitBlockBuffers, MtVitBlockWeights
from ._lib import MT_SGEMM_MAX, MtDensePlan, MtDilatedPlan, MtDropout, MtGemmEpilogue, MtRowMap, MtSgemm, check, rowmap

F16, F32 = 0, 1
EPI_BIAS, EPI_BIAS_RESID, EPI_INJECT, EPI_POSEMB, EPI_QKV_HM = 0, 1, 2, 3, 4
ACT_NONE, ACT_RELU, ACT_GELU, ACT_ELU = 0, 1, 2, 3


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()      # (ctypes converts the int for c_void_p parameters / struct fields)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _s():
    """The current HIP stream of the current device as a raw handle.  (torch.cuda.current_stream() builds a Stream object and
    re-validates the device on every call: a third of the host time of an eager step, tools/diag/host_profile.py.)"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.float32:
        return F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _rm(m):
    return None if m is None else C.byref(m)


def _need(t, dtype, name):
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise TypeError(f"{name}: expected contiguous cuda {dtype}, got {t.dtype} cuda={t.is_cuda} contig={t.is_contiguous()}")


def make_plan(branches, N: int, B: int, qlimit=None) -> MtDilatedPlan:
    """Plan of the dilated-attention launches; qlimit[i] (optional): only the first qlimit[i] sparse entries of branch i's
    sequences act as queries (sequence parallelism: local queries over gathered keys)."""
    p = MtDilatedPlan()
    p.nbranch, p.N, p.B = len(branches), N, B
    for i, b in enumerate(branches):
        p.seg[i], p.ratio[i], p.nseg[i], p.n[i] = b.seg, b.ratio, b.nseg, b.n
        p.qlimit[i] = 0 if qlimit is None else int(qlimit[i])
    return p


def dropout_spec(rng, site=0, p=0.0, path_site=0, path_p=0.0, rows_per_pass=1) -> MtDropout:
    """MtDropout for one stochastic site (rng: device uint32[4] tensor {seed_lo, seed_hi, step, 0})."""
    d = MtDropout()
    d.rng, d.site, d.p, d.path_site, d.path_p, d.rows_per_pass = rng.data_ptr(), site, p, path_site, path_p, rows_per_pass
    return d


def _dr(d):
    return C.byref(d) if d is not None else None


def gemm_nt(A, W, out, M, N, K, *, lda=None, amap=None, ldc=None, cmap=None, epilogue=EPI_BIAS, bias=None, resid=None,
            ldr=0, rmap=None, colscale=None, pos_table=None, pos_row=None, pos_col=None, drop=None):
    """out = epilogue(A[M,K] @ W[N,K]^T) (include/modaltune_hip.h: mt_gemm_nt_f16)."""
    epi = MtGemmEpilogue()
    if drop is not None:
        epi.drop = drop
    epi.bias, epi.resid, epi.ldr = (bias.data_ptr() if bias is not None else None,
                                    resid.data_ptr() if resid is not None else None, ldr)
    if rmap is not None:
        epi.rmap = rmap
    epi.colscale = colscale.data_ptr() if colscale is not None else None
    epi.pos_table = pos_table.data_ptr() if pos_table is not None else None
    epi.pos_row = pos_row.data_ptr() if pos_row is not None else None
    epi.pos_col = pos_col.data_ptr() if pos_col is not None else None
    check(_lib.load().mt_gemm_nt_f16(_p(A), lda if lda is not None else K, _rm(amap), _p(W), M, N, K, epilogue,
                                     C.byref(epi), _p(out), ldc if ldc is not None else N, _rm(cmap), _dt(out), _s()),
          "gemm_nt")


def gemm_tn(A, B, out, M, N1, N2, *, lda=None, amap=None, ldb=None, bmap=None, ldc=None, colsum=None):
    """out[N1,N2] += A^T B over M rows (weight gradient); colsum[N1] += column sums of A (the bias gradient) on the same pass."""
    check(_lib.load().mt_gemm_tn_f16(_p(A), lda if lda is not None else N1, _rm(amap), _p(B), ldb if ldb is not None else N2,
                                     _rm(bmap), M, N1, N2, _p(out), ldc if ldc is not None else N2, _p(colsum), _s()), "gemm_tn")


def colsum(A, out, M, N, *, lda=None, amap=None):
    check(_lib.load().mt_colsum_f16(_p(A), lda if lda is not None else N, _rm(amap), M, N, _p(out), _s()), "colsum")


def sgemm_problem(A, a_str, B, b_str, Cm, c_str, M, N, K, *, bias=None, bias_on_m=False, act=ACT_NONE, accumulate=False,
                  batch=1, a_bs=0, b_bs=0, c_bs=0, rowsum=None, pre_out=None, resid=None, c_drop=None, a_aux=None,
                  a_act=ACT_NONE, a_drop=None, resid_scale=1.0, a_ld=0) -> MtSgemm:
    """One product of an mt_sgemm_multi launch (include/modaltune_hip.h: MtSgemm).  The tensors must outlive the launch."""
    q = MtSgemm()
    q.A, q.as0, q.as1, q.a_bs = _p(A), a_str[0], a_str[1], a_bs
    q.B, q.bs0, q.bs1, q.b_bs = _p(B), b_str[0], b_str[1], b_bs
    q.bias, q.bias_on_m = _p(bias), int(bias_on_m)
    q.C, q.cs0, q.cs1, q.c_bs = _p(Cm), c_str[0], c_str[1], c_bs
    q.M, q.N, q.K, q.batch, q.act, q.accumulate = M, N, K, batch, act, int(accumulate)
    q.rowsum, q.pre_out, q.resid = _p(rowsum), _p(pre_out), _p(resid)
    if c_drop is not None:
        q.c_drop = c_drop
    q.a_aux, q.a_act = _p(a_aux), a_act
    if a_drop is not None:
        q.a_drop = a_drop
    q.resid_scale, q.a_ld = float(resid_scale), int(a_ld)
    return q


def sgemm_multi(problems):
    """Up to MT_SGEMM_MAX independent small products in one launch (more are split over launches, in order)."""
    for i in range(0, len(problems), MT_SGEMM_MAX):
        chunk = problems[i:i + MT_SGEMM_MAX]
        arr = (MtSgemm * len(chunk))(*chunk)
        check(_lib.load().mt_sgemm_multi(arr, len(chunk), _s()), "sgemm_multi")


def sgemm(A, a_str, B, b_str, Cm, c_str, M, N, K, **k):
    """C(m,n) = act(sum_k A(m,k) B(n,k) + bias) with explicit (row, col) element strides; rowsum[m] += sum_k A(m,k).
    Keyword arguments as in sgemm_problem (the fused nn.Linear forward / backward forms included)."""
    sgemm_multi([sgemm_problem(A, a_str, B, b_str, Cm, c_str, M, N, K, **k)])


def layernorm_fwd(x, w, b, y, stats, M, D, *, ldx=None, xmap=None, ldy=None, ymap=None, gelu_in=False, add_rows=None,
                  add_period=0, eps=1e-5):
    check(_lib.load().mt_layernorm_fwd_eps(_p(x), ldx if ldx is not None else D, _rm(xmap), _dt(x), int(gelu_in), _p(w), _p(b),
                                           _p(add_rows), add_period, _p(y), ldy if ldy is not None else D, _rm(ymap), _dt(y),
                                           _p(stats), M, D, float(eps), _s()), "layernorm_fwd")


def add_layernorm_fwd(x, branch, w, b, h, y, stats, M, D, drop=None, eps=1e-5):
    """h = x + drop(branch); y = fp16(LN(h) * w + b) (include/modaltune_hip.h: mt_add_layernorm_fwd)."""
    check(_lib.load().mt_add_layernorm_fwd_eps(_p(x), _p(branch), _dr(drop), _p(w), _p(b), _p(h), _p(y), _p(stats), M, D,
                                               float(eps), _s()), "add_layernorm_fwd")


def layernorm_bwd(dy, x, w, stats, dx, M, D, *, lddy=None, dymap=None, ldx=None, xmap=None, lddx=None, dxmap=None,
                  gelu_in=False, accumulate=False, dw=None, db=None, dx16=None, dx16_drop=None):
    check(_lib.load().mt_layernorm_bwd(_p(dy), lddy if lddy is not None else D, _rm(dymap), _dt(dy), _p(x),
                                       ldx if ldx is not None else D, _rm(xmap), _dt(x), int(gelu_in), _p(w), _p(stats),
                                       _p(dx), lddx if lddx is not None else D, _rm(dxmap), _dt(dx), int(accumulate),
                                       _p(dw), _p(db), _p(dx16), _dr(dx16_drop), M, D, _s()), "layernorm_bwd")


def dilated_attn_fwd(qkv, plan, o_br, lse_br):
    check(_lib.load().mt_dilated_attn_fwd(_p(qkv), C.byref(plan), _p(o_br), _p(lse_br), _s()), "dilated_attn_fwd")


def dilated_mix_ln_fwd(o_br, lse_br, plan, ln_w, ln_b, y, stats, lse_tot):
    check(_lib.load().mt_dilated_mix_ln_fwd(_p(o_br), _p(lse_br), C.byref(plan), _p(ln_w), _p(ln_b), _p(y), _p(stats),
                                            _p(lse_tot), _s()), "dilated_mix_ln_fwd")


def dilated_mix_ln_bwd(dy, o_br, lse_br, lse_tot, plan, ln_w, stats, dmixed, delta_br):
    check(_lib.load().mt_dilated_mix_ln_bwd(_p(dy), _p(o_br), _p(lse_br), _p(lse_tot), C.byref(plan), _p(ln_w), _p(stats),
                                            _p(dmixed), _p(delta_br), _s()), "dilated_mix_ln_bwd")


def dilated_attn_bwd_workspace_bytes(plan) -> int:
    n = _lib.load().mt_dilated_attn_bwd_workspace_bytes(C.byref(plan))
    if n < 0:
        check(int(n), "dilated_attn_bwd_workspace_bytes")
    return int(n)


ATTN_BWD_KV, ATTN_BWD_Q, ATTN_BWD_COMBINE, ATTN_BWD_ALL = 1, 2, 4, 7


def _dilated_attn_bwd_phase(qkv, dmixed, lse_tot, delta_br, plan, workspace, dqkv16, phases):
    check(_lib.load().mt_dilated_attn_bwd(_p(qkv), _p(dmixed), _p(lse_tot), _p(delta_br), C.byref(plan), _p(workspace),
                                          _p(dqkv16), phases, _s()), "dilated_attn_bwd")


def dilated_attn_bwd_phases(qkv, dmixed, lse_tot, delta_br, plan, workspace, dqkv16, phases):
    """Selected phases (ATTN_BWD_KV | ATTN_BWD_Q | ATTN_BWD_COMBINE) of the backward: the sequence-parallel path fills parts of
    the workspace from another plan's launches before the combine."""
    _dilated_attn_bwd_phase(qkv, dmixed, lse_tot, delta_br, plan, workspace, dqkv16, phases)


def dilated_attn_bwd(qkv, dmixed, lse_tot, delta_br, plan, workspace, dqkv16):
    if TIMER is None:
        return _dilated_attn_bwd_phase(qkv, dmixed, lse_tot, delta_br, plan, workspace, dqkv16, ATTN_BWD_ALL)
    for name, ph in (("dilated_attn_bwd_kv", ATTN_BWD_KV), ("dilated_attn_bwd_q", ATTN_BWD_Q), ("dilated_attn_bwd_combine", ATTN_BWD_COMBINE)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _dilated_attn_bwd_phase(qkv, dmixed, lse_tot, delta_br, plan, workspace, dqkv16, ph)
        e1.record()
        TIMER.setdefault(name, []).append((e0, e1))
        if TIMELINE is not None:
            TIMELINE.append((name, e0, e1, torch.cuda.current_stream().cuda_stream))


# ---- dense attention with the 2-D ALiBi bias (one fp16 distance table per slide) + the other TITAN-side launchers (include/modaltune_hip.h)
DENSE_QK_SCALE_LOG2 = 0.125 * 1.4426950408889634
DENSE_BWD_DELTA, DENSE_BWD_KV, DENSE_BWD_Q, DENSE_BWD_ALL = 1, 2, 4, 7


def make_dense_plan(N: int, B: int, H: int, dist=None, nslope=None) -> MtDensePlan:
    """dist: the fp16 distance table of alibi_dist (or None: no bias); nslope: fp32 [H] = -slope_h * log2(e).
    The tensors must outlive every launch made with the plan."""
    p = MtDensePlan()
    p.N, p.B, p.H = N, B, H
    p.dist = dist.data_ptr() if dist is not None else None
    p.nslope = nslope.data_ptr() if nslope is not None else None
    return p


def alibi_dist_halves(N: int) -> int:
    return int(_lib.load().mt_alibi_dist_halves(N))


def alibi_dist(cells, N, table):
    """cells: int32 [N - 1, 2] (row, col) on the device; table: fp16, alibi_dist_halves(N) elements."""
    assert table.numel() >= alibi_dist_halves(N) and table.dtype == torch.float16
    check(_lib.load().mt_alibi_dist(_p(cells), N, _p(table), _s()), "alibi_dist")


def dense_attn_fwd(qkv, plan, o, lse):
    check(_lib.load().mt_dense_attn_fwd(_p(qkv), C.byref(plan), _p(o), _p(lse), _s()), "dense_attn_fwd")


def _dense_attn_bwd_phase(qkv, o, d_o, lse, plan, delta, dqkv, phases):
    check(_lib.load().mt_dense_attn_bwd(_p(qkv), _p(o), _p(d_o), _p(lse), C.byref(plan), _p(delta), _p(dqkv), phases, _s()),
          "dense_attn_bwd")


def dense_attn_bwd(qkv, o, d_o, lse, plan, delta, dqkv):
    if TIMER is None:
        return _dense_attn_bwd_phase(qkv, o, d_o, lse, plan, delta, dqkv, DENSE_BWD_ALL)
    for name, ph in (("dense_attn_delta", DENSE_BWD_DELTA), ("dense_attn_bwd_kv", DENSE_BWD_KV), ("dense_attn_bwd_q", DENSE_BWD_Q)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _dense_attn_bwd_phase(qkv, o, d_o, lse, plan, delta, dqkv, ph)
        e1.record()
        TIMER.setdefault(name, []).append((e0, e1))


def gelu_f16_fwd(x, y, n=None):
    check(_lib.load().mt_gelu_f16_fwd(_p(x), _p(y), n if n is not None else x.numel(), _s()), "gelu_f16_fwd")


def gelu_f16_bwd(x, dy, dx, n=None):
    check(_lib.load().mt_gelu_f16_bwd(_p(x), _p(dy), _p(dx), n if n is not None else x.numel(), _s()), "gelu_f16_bwd")


def pool_attn_workspace_floats(B, N, heads, nq) -> int:
    return int(_lib.load().mt_pool_attn_workspace_floats(B, N, heads, nq))


def pool_attn_fwd(q, kv, out, scores, lse, workspace, B, N, E, heads, nq):
    check(_lib.load().mt_pool_attn_fwd(_p(q), _p(kv), B, N, E, heads, nq, _p(out), _p(scores), _p(lse), _p(workspace), _s()), "pool_attn_fwd")


def pool_attn_bwd(q, kv, scores, lse, out, dout, dkv, B, N, E, heads, nq):
    check(_lib.load().mt_pool_attn_bwd(_p(q), _p(kv), _p(scores), _p(lse), _p(out), _p(dout), B, N, E, heads, nq, _p(dkv), _s()), "pool_attn_bwd")


def titan_grid(coords, L, patch, cells, dims, err=None):
    check(_lib.load().mt_titan_grid(_p(coords), L, float(patch), _p(cells), _p(dims), _p(err), _s()), "titan_grid")


def titan_cell_sums(feat, cells, L, Cc, first, nxt, sums, nz, ldf=None):
    check(_lib.load().mt_titan_cell_sums(_p(feat), ldf if ldf is not None else Cc, _p(cells), L, Cc, _p(first), _p(nxt), _p(sums),
                                         _p(nz), _s()), "titan_cell_sums")


def titan_token_order(cells, first, nz, L, pos, cells_tok, count):
    check(_lib.load().mt_titan_token_order(_p(cells), _p(first), _p(nz), L, _p(pos), _p(cells_tok), _p(count), _s()),
          "titan_token_order")


def titan_gather_tokens(sums, pos, L, Cc, x16):
    check(_lib.load().mt_titan_gather_tokens(_p(sums), _p(pos), L, Cc, _p(x16), _s()), "titan_gather_tokens")


# ---- composite launchers: one frozen backbone layer per call (include/modaltune_hip.h; csrc/layer.hip)
def struct_of(cls, **tensors):
    """A ctypes struct of device pointers from tensors (floats pass through); the caller keeps the tensors alive."""
    st = cls()
    for k, v in tensors.items():
        setattr(st, k, v.data_ptr() if torch.is_tensor(v) else v)
    return st


def longnet_layer_fwd(w, b, plan, M, D, Fd, out, pend=None, defer=False, drop_attn=None, drop_ffn=None):
    px, pb, pd = (pend[0], pend[1], pend[2]) if pend is not None else (None, None, None)
    check(_lib.load().mt_longnet_layer_fwd(C.byref(w), C.byref(b), C.byref(plan), M, D, Fd, _p(px), _p(pb), _dr(pd), int(defer), _p(out),
                                           _dr(drop_attn), _dr(drop_ffn), _s()), "longnet_layer_fwd")


def longnet_layer_bwd(w, b, plan, M, D, Fd, dh16_valid, feeds_lower, drop_attn=None, drop_ffn=None, drop_lower_ffn=None):
    check(_lib.load().mt_longnet_layer_bwd(C.byref(w), C.byref(b), C.byref(plan), M, D, Fd, int(dh16_valid), int(feeds_lower), _dr(drop_attn),
                                           _dr(drop_ffn), _dr(drop_lower_ffn), _s()), "longnet_layer_bwd")


def vit_block_fwd(w, b, plan, M, D, Fd, out, pend=None, defer=False):
    px, pb = (pend[0], pend[1]) if pend is not None else (None, None)
    check(_lib.load().mt_vit_block_fwd(C.byref(w), C.byref(b), C.byref(plan), M, D, Fd, _p(px), _p(pb), int(defer), _p(out), _s()), "vit_block_fwd")


def vit_block_bwd(w, b, plan, M, D, Fd, dh16_valid, feeds_lower):
    check(_lib.load().mt_vit_block_bwd(C.byref(w), C.byref(b), C.byref(plan), M, D, Fd, int(dh16_valid), int(feeds_lower), _s()), "vit_block_bwd")


def gene_snn_fwd(params, offs, sizes, goff, genes, G, latent, a1, a2, z, alpha_drop=None, passes=1):
    check(_lib.load().mt_gene_snn_fwd(_p(params), _p(offs), _p(sizes), _p(goff), _p(genes), G, latent, passes, _p(a1), _p(a2), _p(z),
                                      _dr(alpha_drop), _s()), "gene_snn_fwd")


def gene_snn_bwd(params, grads, offs, sizes, goff, genes, G, latent, a1, a2, dz, alpha_drop=None, passes=1):
    check(_lib.load().mt_gene_snn_bwd(_p(params), _p(grads), _p(offs), _p(sizes), _p(goff), _p(genes), G, latent, passes, _p(a1),
                                      _p(a2), _p(dz), _dr(alpha_drop), _s()), "gene_snn_bwd")


def inject_attn_fwd(q, k, v, a, M, rows_per_pass, T, lse=None):
    check(_lib.load().mt_inject_attn_fwd(_p(q), M, rows_per_pass, _p(k), _p(v), T, _p(a), _p(lse), _s()), "inject_attn_fwd")


def inject_attn_bwd(q, a, lse, da, k, v, dq, dk, dv, M, rows_per_pass, T):
    check(_lib.load().mt_inject_attn_bwd(_p(q), _p(a), _p(lse), _p(da), M, rows_per_pass, _p(k), _p(v), T, _p(dq), _p(dk),
                                         _p(dv), _s()), "inject_attn_bwd")


def extract_attn_fwd(q, kv, out, lse, part_acc, part_ml, B, T, L, nsplit):
    check(_lib.load().mt_extract_attn_fwd(_p(q), _p(kv), B, T, L, _p(out), _p(lse), _p(part_acc), _p(part_ml), nsplit,
                                          _s()), "extract_attn_fwd")


def extract_attn_bwd(q, kv, out, lse, dout, dq, dkv, B, T, L):
    check(_lib.load().mt_extract_attn_bwd(_p(q), _p(kv), _p(out), _p(lse), _p(dout), B, T, L, _p(dq), _p(dkv), _s()),
          "extract_attn_bwd")


def token_mha_fwd(q, k, v, out, probs, B, T, E, heads):
    check(_lib.load().mt_token_mha_fwd(_p(q), _p(k), _p(v), B, T, E, heads, _p(out), _p(probs), _s()), "token_mha_fwd")


def token_mha_bwd(q, k, v, probs, dout, dq, dk, dv, B, T, E, heads):
    check(_lib.load().mt_token_mha_bwd(_p(q), _p(k), _p(v), _p(probs), _p(dout), B, T, E, heads, _p(dq), _p(dk), _p(dv),
                                       _s()), "token_mha_bwd")


def cast_f32_to_f16(x, y, n=None, drop=None, D=0):
    check(_lib.load().mt_cast_f32_to_f16(_p(x), _p(y), n if n is not None else x.numel(), _dr(drop), D, _s()), "cast")


def rng_advance(rng):
    check(_lib.load().mt_rng_advance(_p(rng), _s()), "rng_advance")


def dropout_f32(x, y, M, D, drop, *, ldx=None, xmap=None):
    check(_lib.load().mt_dropout_f32(_p(x), ldx if ldx is not None else D, _rm(xmap), _p(y), M, D, _dr(drop), _s()), "dropout")


def droppath_rows(x, M, D, drop):
    check(_lib.load().mt_droppath_rows_f32(_p(x), M, D, _dr(drop), _s()), "droppath_rows")


def cast_f16_to_f32(x, y, n=None):
    check(_lib.load().mt_cast_f16_to_f32(_p(x), _p(y), n if n is not None else x.numel(), _s()), "cast")


def pack_weight(src, dst, R, C, transpose=False):
    check(_lib.load().mt_pack_weight_f16(_p(src), R, C, _p(dst), int(transpose), _s()), "pack_weight")


def pack_weights(items, n_items):
    """items: device int64 [n_items, 8] records (include/modaltune_hip.h: mt_pack_weights_f16)."""
    check(_lib.load().mt_pack_weights_f16(_p(items), n_items, _s()), "pack_weights")


def act_fwd(x, y, act, n=None):
    check(_lib.load().mt_act_fwd(_p(x), _p(y), n if n is not None else x.numel(), act, _s()), "act_fwd")


def act_bwd(x, dy, dx, act, n=None):
    check(_lib.load().mt_act_bwd(_p(x), _p(dy), _p(dx), n if n is not None else x.numel(), act, _s()), "act_bwd")


def axpy(a, b, alpha, y, n=None):
    check(_lib.load().mt_axpy(_p(a), _p(b), float(alpha), _p(y), n if n is not None else a.numel(), _s()), "axpy")


def axpy_bcast(a, b, alpha, y, period, n=None):
    check(_lib.load().mt_axpy_bcast(_p(a), _p(b), float(alpha), _p(y), n if n is not None else a.numel(), period, _s()),
          "axpy_bcast")


def copy_rows(src, dst, M, D, *, lds=None, smap=None, ldd=None, dmap=None, accumulate=False):
    check(_lib.load().mt_copy_rows_f32(_p(src), lds if lds is not None else D, _rm(smap), _p(dst),
                                       ldd if ldd is not None else D, _rm(dmap), M, D, int(accumulate), _s()), "copy_rows")


def inject_resid_bwd(dy, x, proj, gamma, dx, dproj, dgamma, M, D, *, lddy=None, dymap=None, ldx=None, xmap=None, lddx=None,
                     dxmap=None, dx_accumulate=False):
    check(_lib.load().mt_inject_resid_bwd(_p(dy), lddy if lddy is not None else D, _rm(dymap), _p(x),
                                          ldx if ldx is not None else D, _rm(xmap), _p(proj), _p(gamma), _p(dx),
                                          lddx if lddx is not None else D, _rm(dxmap), int(dx_accumulate), _p(dproj),
                                          _p(dgamma), M, D, _s()), "inject_resid_bwd")


def l2norm_row(x, y, O):
    check(_lib.load().mt_l2norm_rows(_p(x), _p(y), 1, O, _s()), "l2norm_rows")


def distill_loss(logits, target, loss, dlogits, R, O, loss_scale=1.0, scale_dev=None):
    check(_lib.load().mt_distill_loss(_p(logits), _p(target), R, O, float(loss_scale), _p(scale_dev), _p(loss),
                                      _p(dlogits), _s()), "distill_loss")


def adamw_step(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step_count, scale=None, found_inf=None, grad_mult=1.0,
               step_dev=None, lr_dev=None):
    check(_lib.load().mt_adamw_step(_p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, weight_decay, step_count,
                                    _p(step_dev), float(grad_mult), _p(scale), _p(found_inf), _p(lr_dev), _s()), "adamw_step")


def absmax_scale(x, s, target=1024.0, n=None):
    """s[0] = target / max|x|, s[1] = 1 / s[0] on the device (include/modaltune_hip.h: mt_absmax_scale)."""
    check(_lib.load().mt_absmax_scale(_p(x), n if n is not None else x.numel(), float(target), _p(s), _s()), "absmax_scale")


def fold_rows(x, reps, period, out):
    """out[i] += sum_r x[r * period + i] (the adjoint of axpy_bcast)."""
    check(_lib.load().mt_fold_rows(_p(x), reps, period, _p(out), _s()), "fold_rows")


def axpy_dev(a, b, alpha_dev, y, n=None):
    """y = a + (*alpha_dev) * b; a may be None."""
    check(_lib.load().mt_axpy_dev(_p(a), _p(b), _p(alpha_dev), _p(y), n if n is not None else b.numel(), _s()), "axpy_dev")


def scatter_rows(src, idx, dst, M, D, accumulate=True, src_idx=None):
    """dst[idx[m], :] (+)= src[src_idx[m], :] (include/modaltune_hip.h: mt_scatter_rows_f32)."""
    check(_lib.load().mt_scatter_rows_f32(_p(src), _p(src_idx), _p(idx), _p(dst), M, D, int(accumulate), _s()), "scatter_rows")


def row_absmax(x, out, M, D):
    check(_lib.load().mt_row_absmax_f32(_p(x), _p(out), M, D, _s()), "row_absmax")


def coords_to_grid(coords, L, tile, ngrids, prow, pcol, err=None):
    check(_lib.load().mt_coords_to_grid(_p(coords), L, float(tile), ngrids, _p(prow), _p(pcol), _p(err), _s()), "coords_to_grid")


def mfma_probe(sink, workgroups, iters):
    check(_lib.load().mt_mfma_probe(_p(sink), workgroups, iters, _s()), "mfma_probe")


def measured_mfma_peak_tflops(workgroups: int = 1024, iters: int = 20000, reps: int = 3) -> float:
    """The chip's own fp16 MFMA rate on a bare register-operand loop (one wave per SIMD slot, 4 workgroups per CU's worth of
    waves): best of `reps`, TFLOP/s.  ~60 ms of GPU time."""
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    mfma_probe(sink, workgroups, 200)
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        mfma_probe(sink, workgroups, iters)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, workgroups * 4.0 * iters * 4 * 32768 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    return best


def scaler_update(scale, tracker, found_inf, step_dev=None, growth=2.0, backoff=0.5, interval=2000):
    check(_lib.load().mt_scaler_update(_p(scale), _p(tracker), _p(found_inf), _p(step_dev), growth, backoff, interval,
                                       _s()), "scaler_update")


def check_finite(g, n, found_inf):
    check(_lib.load().mt_check_finite(_p(g), n, _p(found_inf), _s()), "check_finite")


# ------------------------------------------------------------------------------------------------
# Optional per-kernel timing with HIP events on the launch stream (bench.py --> roofline.achieved).
# TIMER maps "<op>[shape]" -> list of (start_event, end_event); None = disabled (zero overhead path).
# ------------------------------------------------------------------------------------------------
TIMER = None
RECORD = None          # bench.py: list that receives (fn, args, kwargs) of every token-side launch of a step (replayed in isolation as a graph)
RECORD_KEEP = []       # tensors the recorded launches point to (Tape.new appends while RECORD is set)
TIMELINE = None        # bench.py: with TIMER set, also (key, start event, end event, stream handle) of every launch, in launch order


def _timed(name_fn):
    def deco(fn):
        def wrapper(*a, **k):
            if TIMER is None and RECORD is None:
                return fn(*a, **k)
            key = name_fn(*a, **k)
            if RECORD is not None and key == "token_side":
                RECORD.append((fn, a, k))
            if TIMER is None:
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            TIMER.setdefault(key, []).append((e0, e1))
            if TIMELINE is not None:
                TIMELINE.append((key, e0, e1, torch.cuda.current_stream().cuda_stream))
            return r
        wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
        return wrapper
    return deco


gemm_nt = _timed(lambda A, W, out, M, N, K, **k: f"gemm_nt[{M}x{N}x{K}]")(gemm_nt)
gemm_tn = _timed(lambda A, B, out, M, N1, N2, **k: f"gemm_tn[{N1}x{N2}]")(gemm_tn)
dilated_attn_fwd = _timed(lambda *a, **k: "dilated_attn_fwd")(dilated_attn_fwd)
dense_attn_fwd = _timed(lambda *a, **k: "dense_attn_fwd")(dense_attn_fwd)
gelu_f16_fwd = _timed(lambda *a, **k: "gelu_f16_fwd")(gelu_f16_fwd)
gelu_f16_bwd = _timed(lambda *a, **k: "gelu_f16_bwd")(gelu_f16_bwd)
dilated_mix_ln_fwd = _timed(lambda *a, **k: "dilated_mix_ln_fwd")(dilated_mix_ln_fwd)
dilated_mix_ln_bwd = _timed(lambda *a, **k: "dilated_mix_ln_bwd")(dilated_mix_ln_bwd)
layernorm_fwd = _timed(lambda x, w, b, y, stats, M, D, **k: f"layernorm_fwd[{D}]" if M > 1024 else "token_side")(layernorm_fwd)
layernorm_bwd = _timed(lambda dy, x, w, stats, dx, M, D, **k: f"layernorm_bwd[{D}]" if M > 1024 else "token_side")(layernorm_bwd)
add_layernorm_fwd = _timed(lambda x, branch, w, b, h, y, stats, M, D, **k: f"add_layernorm_fwd[{D}]")(add_layernorm_fwd)
cast_f32_to_f16 = _timed(lambda *a, **k: "cast")(cast_f32_to_f16)
inject_attn_fwd = _timed(lambda *a, **k: "inject_attn_fwd")(inject_attn_fwd)
inject_attn_bwd = _timed(lambda *a, **k: "inject_attn_bwd")(inject_attn_bwd)
extract_attn_fwd = _timed(lambda *a, **k: "extract_attn_fwd")(extract_attn_fwd)
extract_attn_bwd = _timed(lambda *a, **k: "extract_attn_bwd")(extract_attn_bwd)
inject_resid_bwd = _timed(lambda *a, **k: "inject_resid_bwd")(inject_resid_bwd)
colsum = _timed(lambda *a, **k: "colsum")(colsum)
_DETAIL = bool(os.environ.get("MT_TIMER_DETAIL"))
sgemm_multi = _timed(lambda problems: ("sgemm[" + "+".join(f"{q.M}x{q.N}x{q.K}b{q.batch}" for q in problems) + "]" if _DETAIL else "token_side"))(sgemm_multi)
adamw_step = _timed(lambda *a, **k: "adamw")(adamw_step)
# the rest of the token side (T <= ~200 rows: adds, copies, DropPath rows, T x T attention, pathway networks, loss head) -- until round 5
# these launches were outside the table (it listed 184 of the ~220 token-side launches of a step)
_small = lambda n: "token_side" if n <= (1 << 20) else "elementwise"
axpy = _timed(lambda a, b, alpha, y, n=None: _small(n if n is not None else b.numel()))(axpy)
axpy_bcast = _timed(lambda a, b, alpha, y, period, n=None: _small(n if n is not None else a.numel()))(axpy_bcast)
fold_rows = _timed(lambda x, reps, period, out: _small(reps * period))(fold_rows)
copy_rows = _timed(lambda src, dst, M, D, **k: "token_side" if M <= 1024 else "copy_rows")(copy_rows)
droppath_rows = _timed(lambda x, M, D, drop: "token_side" if M <= 1024 else "droppath_rows")(droppath_rows)
dropout_f32 = _timed(lambda x, y, M, D, drop, **k: "token_side" if M <= 1024 else "dropout_f32")(dropout_f32)
token_mha_fwd = _timed(lambda *a, **k: "token_side")(token_mha_fwd)
token_mha_bwd = _timed(lambda *a, **k: "token_side")(token_mha_bwd)
act_fwd = _timed(lambda x, y, act, n=None: _small(n if n is not None else x.numel()))(act_fwd)
act_bwd = _timed(lambda x, dy, dx, act, n=None: _small(n if n is not None else x.numel()))(act_bwd)
gene_snn_fwd = _timed(lambda *a, **k: "token_side")(gene_snn_fwd)
gene_snn_bwd = _timed(lambda *a, **k: "token_side")(gene_snn_bwd)
l2norm_row = _timed(lambda *a, **k: "token_side")(l2norm_row)
distill_loss = _timed(lambda *a, **k: "token_side")(distill_loss)


def timer_summary(timer):
    """{key: (launches, total_ms)} after a device sync."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in timer.items()}
