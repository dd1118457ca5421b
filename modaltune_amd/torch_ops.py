"""`torch.ops.modaltune_hip.*`: the fused launchers of include/modaltune_hip.h registered with the PyTorch dispatcher
(SURVEY §8b "thin TORCH_LIBRARY(modaltune_hip, ...) shim validates dtype / contiguity / shape and forwards"; BASELINE north_star
"drops them in via PyTorch-ROCm custom ops").

Registered through `torch.library` (the Python face of TORCH_LIBRARY): every op has a schema, a CUDA (= HIP on ROCm)
implementation that allocates its outputs with torch, validates, and forwards to the C ABI on the current stream, and a
Meta ("fake") implementation, so the ops can be traced / shape-propagated.  They are functional (inputs are not mutated) and
carry no autograd formula: the reference-facing autograd boundary is the nn.Module bridge (aggregators._ModelFn), whose tape
pairs each forward launcher with its backward launcher; what is registered here is the op-level surface a maintainer can call
or compose directly.  There is no CPU implementation (the product path has no fallback).

    import modaltune_amd.torch_ops        # registers the namespace
    y = torch.ops.modaltune_hip.gemm_nt(a16, w16, bias, True)
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .config import Branch

H16, F32 = torch.float16, torch.float32
NS = "modaltune_hip"

_lib_def = torch.library.Library(NS, "DEF")
_impl_cuda = torch.library.Library(NS, "IMPL", "CUDA")
_impl_meta = torch.library.Library(NS, "IMPL", "Meta")
SCHEMAS = {}


def _op(schema: str):
    name = schema.split("(")[0]
    _lib_def.define(schema)
    SCHEMAS[name] = schema

    def deco(fns):
        cuda_fn, meta_fn = fns
        _impl_cuda.impl(name, cuda_fn)
        _impl_meta.impl(name, meta_fn)
        return fns
    return deco


def _need(t: Tensor, dtype, name: str, dims: Optional[int] = None):
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"modaltune_hip: {name} must be contiguous {dtype} (got {t.dtype}, contiguous={t.is_contiguous()})")
    if dims is not None and t.dim() != dims:
        raise RuntimeError(f"modaltune_hip: {name} must have {dims} dims (got {tuple(t.shape)})")


def _plan(N: int, B: int, seg: List[int], ratio: List[int]):
    br = [Branch(seg=min(int(s), N), ratio=int(r), nseg=-(-N // min(int(s), N)), n=-(-min(int(s), N) // int(r))) for s, r in zip(seg, ratio)]
    return ops.make_plan(br, N, B), len(br)


# ---- C = A @ W^T + bias (nn.Linear on fp16 operands, fp32 accumulate)
def _gemm_nt(a: Tensor, w: Tensor, bias: Optional[Tensor], out_f16: bool) -> Tensor:
    _need(a, H16, "a", 2); _need(w, H16, "w", 2)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or K % 64:
        raise RuntimeError(f"modaltune_hip::gemm_nt: a [M, K] @ w [N, K]^T with K % 64 == 0 (got {tuple(a.shape)}, {tuple(w.shape)})")
    if bias is not None:
        _need(bias, F32, "bias", 1)
    out = torch.empty(M, N, dtype=H16 if out_f16 else F32, device=a.device)
    ops.gemm_nt(a, w, out, M, N, K, bias=bias)
    return out


_op("gemm_nt(Tensor a, Tensor w, Tensor? bias, bool out_f16) -> Tensor")(
    (_gemm_nt, lambda a, w, bias, out_f16: a.new_empty((a.shape[0], w.shape[0]), dtype=H16 if out_f16 else F32)))


# ---- LayerNorm forward: (y fp16, stats fp32 [M, 2])
def _layernorm_fwd(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tuple[Tensor, Tensor]:
    _need(x, F32, "x", 2); _need(w, F32, "w", 1); _need(b, F32, "b", 1)
    M, D = x.shape
    y, st = torch.empty(M, D, dtype=H16, device=x.device), torch.empty(M, 2, dtype=F32, device=x.device)
    ops.layernorm_fwd(x, w, b, y, st, M, D, eps=eps)
    return y, st


_op("layernorm_fwd(Tensor x, Tensor w, Tensor b, float eps) -> (Tensor, Tensor)")(
    (_layernorm_fwd, lambda x, w, b, eps: (x.new_empty(x.shape, dtype=H16), x.new_empty((x.shape[0], 2)))))


def _layernorm_bwd(dy: Tensor, x: Tensor, w: Tensor, stats: Tensor) -> Tensor:
    _need(dy, H16, "dy", 2); _need(x, F32, "x", 2); _need(stats, F32, "stats", 2)
    M, D = x.shape
    dx = torch.empty(M, D, dtype=F32, device=x.device)
    ops.layernorm_bwd(dy, x, w, stats, dx, M, D)
    return dx


_op("layernorm_bwd(Tensor dy, Tensor x, Tensor w, Tensor stats) -> Tensor")((_layernorm_bwd, lambda dy, x, w, stats: x.new_empty(x.shape)))


# ---- LongNet dilated attention: all branches, branch mix + inner LayerNorm (DA:146-262)
def _dilated_attention_fwd(qkv_hm: Tensor, N: int, B: int, seg: List[int], ratio: List[int], ln_w: Tensor, ln_b: Tensor):
    """qkv_hm: fp16 head-major [3][16][B*N][48] with q pre-scaled by MT_QK_SCALE_LOG2.  Returns (y fp16 [B*N, 768], o_br, lse_br,
    lse_tot, stats): everything the backward needs."""
    _need(qkv_hm, H16, "qkv_hm")
    M = B * N
    if qkv_hm.numel() != 3 * 768 * M:
        raise RuntimeError("modaltune_hip::dilated_attention_fwd: qkv_hm must hold 3 x 16 x B*N x 48 halves")
    plan, nb = _plan(N, B, seg, ratio)
    dev = qkv_hm.device
    o_br, lse_br = torch.empty(nb, M, 768, dtype=H16, device=dev), torch.empty(nb, M, 16, dtype=F32, device=dev)
    ops.dilated_attn_fwd(qkv_hm, plan, o_br, lse_br)
    y, st, tot = torch.empty(M, 768, dtype=H16, device=dev), torch.empty(M, 2, dtype=F32, device=dev), torch.empty(M, 16, dtype=F32, device=dev)
    ops.dilated_mix_ln_fwd(o_br, lse_br, plan, ln_w, ln_b, y, st, tot)
    return y, o_br, lse_br, tot, st


def _dilated_attention_fwd_meta(qkv_hm, N, B, seg, ratio, ln_w, ln_b):
    M, nb = B * N, len(seg)
    return (qkv_hm.new_empty((M, 768)), qkv_hm.new_empty((nb, M, 768)), qkv_hm.new_empty((nb, M, 16), dtype=F32),
            qkv_hm.new_empty((M, 16), dtype=F32), qkv_hm.new_empty((M, 2), dtype=F32))


_op("dilated_attention_fwd(Tensor qkv_hm, int N, int B, int[] seg, int[] ratio, Tensor ln_w, Tensor ln_b) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")(
    (_dilated_attention_fwd, _dilated_attention_fwd_meta))


def _dilated_attention_bwd(dy: Tensor, qkv_hm: Tensor, o_br: Tensor, lse_br: Tensor, lse_tot: Tensor, stats: Tensor, N: int, B: int,
                           seg: List[int], ratio: List[int], ln_w: Tensor) -> Tensor:
    """-> dqkv fp16 [B*N, 2304] token-major (q columns: gradient of the pre-scaled q)."""
    _need(dy, H16, "dy", 2)
    M = B * N
    plan, nb = _plan(N, B, seg, ratio)
    dev = dy.device
    dmixed, delta = torch.empty(16, M, 48, dtype=H16, device=dev), torch.empty(nb, M, 16, dtype=F32, device=dev)
    ops.dilated_mix_ln_bwd(dy, o_br, lse_br, lse_tot, plan, ln_w, stats, dmixed, delta)
    ws = torch.empty(ops.dilated_attn_bwd_workspace_bytes(plan) // 2, dtype=H16, device=dev)
    dqkv = torch.empty(M, 2304, dtype=H16, device=dev)
    ops.dilated_attn_bwd(qkv_hm, dmixed, lse_tot, delta, plan, ws, dqkv)
    return dqkv


_op("dilated_attention_bwd(Tensor dy, Tensor qkv_hm, Tensor o_br, Tensor lse_br, Tensor lse_tot, Tensor stats, int N, int B, int[] seg, "
    "int[] ratio, Tensor ln_w) -> Tensor")((_dilated_attention_bwd, lambda dy, *a: dy.new_empty((dy.shape[0], 2304))))


# ---- dense attention with the 2-D ALiBi bias (TITAN blocks)
def _dense_alibi_plan(qkv: Tensor, cells: Optional[Tensor], dims: Optional[Tensor], slopes: Optional[Tensor], N: int, B: int, H: int):
    dev = qkv.device
    if cells is None:
        return ops.make_dense_plan(N, B, H), ()
    dist = torch.empty(ops.alibi_dist_halves(N), dtype=H16, device=dev)
    ops.alibi_dist(cells.to(torch.int32).contiguous(), N, dist)
    nslope = (-slopes.to(F32) * 1.4426950408889634).contiguous()
    return ops.make_dense_plan(N, B, H, dist, nslope), (dist, nslope)


def _dense_attention_fwd(qkv: Tensor, N: int, B: int, H: int, cells: Optional[Tensor], dims: Optional[Tensor], slopes: Optional[Tensor]):
    """qkv fp16 token-major [B*N, 3*H*64] (q pre-scaled by MT_DENSE_QK_SCALE_LOG2); cells int [N-1, 2] grid (row, col) of the
    tokens after cls, dims int [2] = (H, W), slopes fp32 [H] (all three or none).  -> (o fp16 [B*N, H*64], lse fp32 [B*N, H])."""
    _need(qkv, H16, "qkv", 2)
    if qkv.shape != (B * N, 3 * H * 64):
        raise RuntimeError(f"modaltune_hip::dense_attention_fwd: qkv must be [B*N, 3*H*64] (got {tuple(qkv.shape)})")
    plan, keep = _dense_alibi_plan(qkv, cells, dims, slopes, N, B, H)
    o, lse = torch.empty(B * N, H * 64, dtype=H16, device=qkv.device), torch.empty(B * N, H, dtype=F32, device=qkv.device)
    ops.dense_attn_fwd(qkv, plan, o, lse)
    return o, lse


_op("dense_attention_fwd(Tensor qkv, int N, int B, int H, Tensor? cells, Tensor? dims, Tensor? slopes) -> (Tensor, Tensor)")(
    (_dense_attention_fwd, lambda qkv, N, B, H, cells, dims, slopes: (qkv.new_empty((B * N, H * 64)), qkv.new_empty((B * N, H), dtype=F32))))


def _dense_attention_bwd(d_o: Tensor, qkv: Tensor, o: Tensor, lse: Tensor, N: int, B: int, H: int, cells: Optional[Tensor],
                         dims: Optional[Tensor], slopes: Optional[Tensor]) -> Tensor:
    _need(d_o, H16, "d_o", 2)
    plan, keep = _dense_alibi_plan(qkv, cells, dims, slopes, N, B, H)
    delta, dqkv = torch.empty(B * N, H, dtype=F32, device=qkv.device), torch.empty_like(qkv)
    ops.dense_attn_bwd(qkv, o, d_o, lse, plan, delta, dqkv)
    return dqkv


_op("dense_attention_bwd(Tensor d_o, Tensor qkv, Tensor o, Tensor lse, int N, int B, int H, Tensor? cells, Tensor? dims, Tensor? slopes) -> Tensor")(
    (_dense_attention_bwd, lambda d_o, qkv, *a: qkv.new_empty(qkv.shape)))


# ---- Injector / Extractor attention cores (AM:225-229)
def _inject_attention_fwd(q: Tensor, k: Tensor, v: Tensor, rows_per_pass: int) -> Tuple[Tensor, Tensor]:
    _need(q, H16, "q", 2); _need(k, F32, "k", 3); _need(v, F32, "v", 3)
    M, T = q.shape[0], k.shape[1]
    a, lse = torch.empty(M, 192, dtype=H16, device=q.device), torch.empty(M, 12, dtype=F32, device=q.device)
    ops.inject_attn_fwd(q, k, v, a, M, rows_per_pass, T, lse=lse)
    return a, lse


_op("inject_attention_fwd(Tensor q, Tensor k, Tensor v, int rows_per_pass) -> (Tensor, Tensor)")(
    (_inject_attention_fwd, lambda q, k, v, r: (q.new_empty(q.shape), q.new_empty((q.shape[0], 12), dtype=F32))))


def _extract_attention_fwd(q: Tensor, kv: Tensor, L: int) -> Tuple[Tensor, Tensor]:
    _need(q, F32, "q", 3); _need(kv, H16, "kv", 2)
    B, T = q.shape[0], q.shape[1]
    dev = q.device
    kps = -(-(-(-L // max(1, min(64, L // 256)))) // 64) * 64
    nsplit = -(-L // kps)
    out, lse = torch.empty(B, T, 192, dtype=F32, device=dev), torch.empty(B, T, 12, dtype=F32, device=dev)
    pa, pml = torch.empty(B * 12 * nsplit * T * 16, dtype=F32, device=dev), torch.empty(B * 12 * nsplit * T * 2, dtype=F32, device=dev)
    ops.extract_attn_fwd(q, kv, out, lse, pa, pml, B, T, L, nsplit)
    return out, lse


_op("extract_attention_fwd(Tensor q, Tensor kv, int L) -> (Tensor, Tensor)")(
    (_extract_attention_fwd, lambda q, kv, L: (q.new_empty(q.shape), q.new_empty((q.shape[0], q.shape[1], 12)))))


# ---- fused multi-tensor AdamW over a flat buffer (functional: returns the updated p, m, v)
def _adamw(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int):
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _need(t, F32, nm, 1)
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    ops.adamw_step(p2, g, m2, v2, p.numel(), lr, beta1, beta2, eps, weight_decay, step)
    return p2, m2, v2


_op("adamw(Tensor p, Tensor g, Tensor m, Tensor v, float lr, float beta1, float beta2, float eps, float weight_decay, int step) -> (Tensor, Tensor, Tensor)")(
    (_adamw, lambda p, g, m, v, *a: (p.new_empty(p.shape), p.new_empty(p.shape), p.new_empty(p.shape))))


# ---- the whole model forward as ONE op (inference / embedding extraction; training goes through the nn.Module bridge)
_MODELS = {}


def register_model(model) -> int:
    """Handle for `torch.ops.modaltune_hip.model_forward` (an engine cannot travel through an op schema)."""
    h = len(_MODELS) + 1
    _MODELS[h] = model
    return h


def _model_forward(handle: int, x: Tensor, coords: Tensor, genes: Tensor, task_onehots: Tensor) -> Tensor:
    """x [L, in_chans], coords [L, 2], genes: the pathway vectors concatenated [sum n_i], task_onehots [B, num_tasks] ->
    logits [B, output_dim] (eval-mode forward of the registered LongNet adapter, B task passes batched)."""
    model = _MODELS[handle]
    with torch.no_grad():
        return model.engine.forward(x, coords, genes, task_onehots, need_grad=False).clone()


_op("model_forward(int handle, Tensor x, Tensor coords, Tensor genes, Tensor task_onehots) -> Tensor")(
    (_model_forward, lambda h, x, coords, genes, oh: x.new_empty((oh.shape[0], _MODELS[h].cfg.output_dim), dtype=F32)))
