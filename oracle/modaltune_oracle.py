"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path `modaltune_amd/`).

A from-scratch torch-CPU restatement of the reference's Modal-Adapter train step, written against the
reference semantics with each function citing the reference file:line it
follows (paths relative to the reference repo root).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

Parity status: PINNED — tests/test_oracle_golden.py checks every function here against golden vectors
produced by importing and running the reference itself in the build container
(tests/golden/make_golden.py; fixtures under tests/golden/*.npz).

Everything is functional: weights come in as a {state_dict key -> tensor} mapping using the
reference's key names, so the same tensors drive the reference, this oracle and the HIP path.
Dropout / DropPath / AlphaDropout are off (parity is only defined at p = 0).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

LN_EPS = 1e-5

# Diagnostic switch (tests/test_grad_rounding_cpu.py only; parity is defined with it OFF): emulate the storage precision of the
# reference's GPU run -- `torch.cuda.amp.autocast` (train_modaltune.py:216) rounds the operands and the result of every nn.Linear /
# attention product over the patch rows to fp16 -- in an otherwise fp64 computation.  Rounding is straight-through (the backward
# sees the identity), so the gradients are the exact gradients of the network evaluated at the ROUNDED forward activations: what
# remains is the forward rounding's effect on the gradients, with no fp16 gradient stream at all (F16_GRAD_SCALE adds that stream).
F16_PATCH_OPERANDS = False
F16_GRAD_SCALE = 0.0          # > 0: the gradients flowing back through those products are rounded to fp16 too, at this loss scale


class _Round16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.float16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        if F16_GRAD_SCALE > 0.0:
            return (g * F16_GRAD_SCALE).to(torch.float16).to(g.dtype) / F16_GRAD_SCALE
        return g


def _r16(t):
    return _Round16.apply(t) if F16_PATCH_OPERANDS else t


def _ln(x, sd, prefix, eps=LN_EPS):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _linear(x, sd, prefix, patch_rows=False):
    """nn.Linear; patch_rows: x holds the slide's patch rows (the products the GPU runs with fp16 operands, see F16_PATCH_OPERANDS)."""
    if patch_rows and F16_PATCH_OPERANDS:
        return _r16(F.linear(_r16(x), _r16(sd[prefix + ".weight"]), sd[prefix + ".bias"]))
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


# ------------------------------------------------------------------------------------------------
# torch.nn.MultiheadAttention(embed_dim=E, heads, batch_first, kdim=vdim=D) as used by the adapter
# (models/vitadapter/adapter_modules.py:42-49,157-164): separate q/k/v projection weights, a packed
# in_proj_bias, per-head softmax(q k^T / sqrt(E/heads)) v, concat, out_proj.
# ------------------------------------------------------------------------------------------------
def mha(q_in, k_in, v_in, sd, prefix, heads, patch_q=False, patch_kv=False):
    E = sd[prefix + ".q_proj_weight"].shape[0]
    b = sd[prefix + ".in_proj_bias"]
    rq = _r16 if patch_q else (lambda t: t)
    rk = _r16 if patch_kv else (lambda t: t)
    q = rq(F.linear(rq(q_in), rq(sd[prefix + ".q_proj_weight"]), b[:E]))
    k = rk(F.linear(rk(k_in), rk(sd[prefix + ".k_proj_weight"]), b[E:2 * E]))
    v = rk(F.linear(rk(v_in), rk(sd[prefix + ".v_proj_weight"]), b[2 * E:]))
    if patch_q or patch_kv:          # (both sides of the cross attention sit in fp16 images next to the patch rows)
        q, k, v = _r16(q), _r16(k), _r16(v)
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    hd = E // heads
    q = q.view(B, Lq, heads, hd).transpose(1, 2)
    k = k.view(B, Lk, heads, hd).transpose(1, 2)
    v = v.view(B, Lk, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, Lq, E)
    if patch_q:
        return _r16(F.linear(_r16(a), _r16(sd[prefix + ".out_proj.weight"]), sd[prefix + ".out_proj.bias"]))
    return F.linear(a, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


def cross_attention_pre(tgt, memory, sd, prefix, heads, pos=None, query_pos=None, patch_q=False, patch_kv=False):
    """CrossAttentionLayer.forward_pre (adapter_modules.py:210-234), normalize_before=True, with_cffn=True.
    patch_q / patch_kv: which side holds the slide's patch rows (diagnostic fp16 emulation only, see F16_PATCH_OPERANDS)."""
    t2 = _ln(tgt, sd, prefix + ".norm")
    mem = _ln(memory, sd, prefix + ".norm_kq")
    q = _linear(t2 if query_pos is None else t2 + query_pos, sd, prefix + ".q_proj", patch_rows=patch_q)
    kv = mem if pos is None else mem + pos
    a = mha(q, kv, kv, sd, prefix + ".multihead_attn", heads, patch_q=patch_q, patch_kv=patch_kv)
    if patch_q and F16_PATCH_OPERANDS:       # (the Injector's output_proj epilogue writes the fp32 residual stream directly)
        return tgt + F.linear(_r16(a), _r16(sd[prefix + ".output_proj.weight"]), sd[prefix + ".output_proj.bias"])
    return tgt + _linear(a, sd, prefix + ".output_proj")


def injector(x, c, pe, sd, prefix, heads):
    """Injector.forward (adapter_modules.py:359-369): attn(query=x, feat=c, pos=pe, query_pos=None);
    y = x + gamma * attn where attn already contains the inner residual."""
    attn = cross_attention_pre(x, c, sd, prefix + ".attn", heads, pos=pe, query_pos=None, patch_q=True)
    return x + sd[prefix + ".gamma"] * attn


def extractor(c, x, pe, sd, prefix, heads):
    """Extractor.forward (adapter_modules.py:321-335) + FFNLayer.forward_pre (284-287); SURVEY A.2."""
    attn = cross_attention_pre(c, x, sd, prefix + ".attn", heads, pos=None, query_pos=pe, patch_kv=True)
    c1 = c + attn
    t = _ln(c1, sd, prefix + ".ffn.norm")
    t = _linear(F.relu(_linear(t, sd, prefix + ".ffn.linear1")), sd, prefix + ".ffn.linear2")
    return c1 + t


def prompt_self_attention(c, pe, sd, prefix, heads):
    """SelfAttentionLayer.forward_pre (adapter_modules.py:81-94); SURVEY A.3."""
    t = _ln(c, sd, prefix + ".norm")
    k_in = t + pe
    q = _linear(k_in, sd, prefix + ".q_proj")
    a = mha(q, k_in, t, sd, prefix + ".self_attn", heads)
    return c + _linear(a, sd, prefix + ".output_proj")


# ------------------------------------------------------------------------------------------------
# Dilated attention (torchscale/component/dilated_attention.py:22-59,82-144,212-255)
# ------------------------------------------------------------------------------------------------
def _softmax_attention(qs, ks, vs, scale, impl):
    """qs, ks, vs: [B, nseg, n, g, d, r] sparse sequences -> (o [B,nseg,n,g,d,r], lse [B,nseg,r,g,n]).
    impl "explicit": materialised softmax (any dtype, differentiable; small n).  impl "flash": ATen's CPU flash kernel
    (fp32, returns the logsumexp; used for the long sequences of the cpu_baseline timing).  Same arithmetic."""
    if impl == "flash":
        B, nseg, n, g, d, r = qs.shape
        f = lambda t: t.permute(0, 1, 5, 3, 2, 4).reshape(B * nseg, r * g, n, d)       # [batch, heads, n, d]
        o, lse = torch.ops.aten._scaled_dot_product_flash_attention_for_cpu(f(qs), f(ks), f(vs), 0.0, False, scale=scale)
        o = o.reshape(B, nseg, r, g, n, d).permute(0, 1, 4, 3, 5, 2)
        return o, lse.reshape(B, nseg, r, g, n)
    sc = torch.einsum("bjqgdr,bjkgdr->bjrgqk", qs, ks) * scale
    lse = torch.logsumexp(sc.float() if sc.dtype != torch.float64 else sc, dim=-1)        # [B,nseg,r,g,n]
    o = torch.einsum("bjrgqk,bjkgdr->bjqgdr", torch.softmax(sc, dim=-1), vs)                # [B,nseg,n,g,d,r]
    return o, lse


def dilated_attention_core(q, k, v, seg_lengths: Sequence[int], ratios: Sequence[int],
                           return_branches: bool = False, impl: str = "auto"):
    """q,k,v: [B, N, H, d] -> [B, N, H*d].

    Independent restatement: per branch, every head group r walks positions r, r+dr, ... inside each
    segment; entries beyond the segment or beyond N are all-zero rows that still act as keys
    (logit 0, value 0: dilated_attention.py:98-101, 24-28).  Branch outputs are mixed with
    softmax-over-branches of the per-(position, head) LSE, computed without gradient (132-141);
    (position, head) pairs a branch does not visit carry lse = -1e8 (44, 52).
    """
    B, N, H, d = q.shape
    scale = d ** -0.5
    outs, lses = [], []
    for sl, dr in zip(seg_lengths, ratios):
        s = min(int(sl), N)
        nseg = -(-N // s)
        n = -(-s // dr)
        g = H // dr
        padN = nseg * s - N

        def sparse(t):
            t = F.pad(t, (0, 0, 0, 0, 0, padN))                      # zero rows beyond N
            t = t.view(B, nseg, s, H, d)
            t = F.pad(t, (0, 0, 0, 0, 0, n * dr - s))                # zero rows beyond the segment
            t = t.view(B, nseg, n, dr, dr, g, d)                     # [.., i, r1(pos), r2(head group), hh, d]
            return torch.diagonal(t, dim1=3, dim2=4)                 # [B, nseg, n, g, d, r]: pos offset == head group

        qs, ks, vs = sparse(q), sparse(k), sparse(v)
        use = impl if impl != "auto" else ("flash" if (n > 512 and q.dtype == torch.float32) else "explicit")
        o, lse = _softmax_attention(qs, ks, vs, scale, use)
        # scatter back to dense [B, N, H, d]; unvisited (pos, head) -> O = 0, lse = -1e8
        od = q.new_zeros(B, nseg, n, dr, dr, g, d)
        ld = torch.full((B, nseg, n, dr, dr, g), -1e8, dtype=lse.dtype)
        idx = torch.arange(dr)
        od[:, :, :, idx, idx] = o.permute(0, 1, 2, 5, 3, 4)          # [B, nseg, n, r, g, d]
        ld[:, :, :, idx, idx] = lse.permute(0, 1, 4, 2, 3)           # [B, nseg, n, r, g]
        od = od.reshape(B, nseg, n * dr, H, d)[:, :, :s].reshape(B, nseg * s, H, d)[:, :N]
        ld = ld.reshape(B, nseg, n * dr, H)[:, :, :s].reshape(B, nseg * s, H)[:, :N]
        outs.append(od)
        lses.append(ld)
    with torch.no_grad():
        L = torch.stack(lses, 0)
        w = torch.softmax(L, dim=0)
    out = sum(o * w[i].unsqueeze(-1).to(o.dtype) for i, o in enumerate(outs))
    out = out.reshape(B, N, H * d)
    if return_branches:
        return out, outs, lses
    return out


def dilated_attention_core_sp(qs, ks, vs, seg_lengths: Sequence[int], ratios: Sequence[int], return_branches: bool = False):
    """Sequence-parallel DilatedAttention (dilated_attention.py:61-111,212-255 with args.seq_parallel = True).

    qs / ks / vs: one [B, Lloc, H, d] tensor per rank (rank r holds chunk r of the sequence).  Per branch (sl, dr):
      * sl <= Lloc: the rank works on its chunk alone -- segments of sl tokens counted from the chunk's first row
        (gathering() sees only the local tensor, DA:96-104);
      * sl > Lloc (requires sl % Lloc == 0, DA:63): the chunk is ONE segment (sl = min(sl, Lloc), DA:97); every rank
        sparsifies its chunk on its own (zero rows up to a multiple of dr included), the sparse K / V of the
        num_rank_per_segment = sl // Lloc ranks of its group [rank // nrps * nrps, +nrps) are concatenated along the
        length (all-gather over all ranks, then the slice of DA:76-79), and the rank's own sparse queries attend over them
        (non-causal).  The all-gather's backward is a reduce-scatter (TS/component/utils.py:43-82): a rank's dK / dV is the
        sum over the ranks of its group -- here simply autograd through the concatenation.
    Returns the mixed output [B, Lloc, H*d] per rank (and, optionally, the per-branch outputs / LSEs per rank)."""
    W = len(qs)
    B, L, H, d = qs[0].shape
    scale = d ** -0.5
    outs = [[] for _ in range(W)]
    lses = [[] for _ in range(W)]
    for sl, dr in zip(seg_lengths, ratios):
        sl, dr = int(sl), int(dr)
        gathered = W > 1 and sl > L
        s = min(sl, L)
        nseg = -(-L // s)
        n = -(-s // dr)
        g = H // dr
        padN = nseg * s - L

        def sparse(t):
            t = F.pad(t, (0, 0, 0, 0, 0, padN))
            t = t.view(B, nseg, s, H, d)
            t = F.pad(t, (0, 0, 0, 0, 0, n * dr - s))
            t = t.view(B, nseg, n, dr, dr, g, d)
            return torch.diagonal(t, dim1=3, dim2=4)                 # [B, nseg, n, g, d, r]

        sq = [sparse(t) for t in qs]
        sk = [sparse(t) for t in ks]
        sv = [sparse(t) for t in vs]
        if gathered:
            assert sl % L == 0, "segment length must be a multiple of the local sequence length (DA:63)"
            nrps = sl // L
        for r in range(W):
            if gathered:
                first = r // nrps * nrps
                grp = range(first, min(W, first + nrps))
                kk, vv = torch.cat([sk[i] for i in grp], dim=2), torch.cat([sv[i] for i in grp], dim=2)
            else:
                kk, vv = sk[r], sv[r]
            o, lse = _softmax_attention(sq[r], kk, vv, scale, "explicit")
            od = qs[r].new_zeros(B, nseg, n, dr, dr, g, d)
            ld = torch.full((B, nseg, n, dr, dr, g), -1e8, dtype=lse.dtype)
            idx = torch.arange(dr)
            od[:, :, :, idx, idx] = o.permute(0, 1, 2, 5, 3, 4)
            ld[:, :, :, idx, idx] = lse.permute(0, 1, 4, 2, 3)
            outs[r].append(od.reshape(B, nseg, n * dr, H, d)[:, :, :s].reshape(B, nseg * s, H, d)[:, :L])
            lses[r].append(ld.reshape(B, nseg, n * dr, H)[:, :, :s].reshape(B, nseg * s, H)[:, :L])
    mixed = []
    for r in range(W):
        with torch.no_grad():
            w = torch.softmax(torch.stack(lses[r], 0), dim=0)
        mixed.append(sum(o * w[i].unsqueeze(-1).to(o.dtype) for i, o in enumerate(outs[r])).reshape(B, L, H * d))
    if return_branches:
        return mixed, outs, lses
    return mixed


def encoder_layer(x, sd, prefix, seg_lengths, ratios, heads=16, attn_impl="auto"):
    """EncoderLayer.forward (torchscale/architecture/encoder.py:121-175) with DilatedAttention.forward
    (dilated_attention.py:146-262) and FeedForwardNetwork.forward (feedforward_network.py:132-143);
    pre-LN (subln), alpha = 1, dropout/droppath off.  SURVEY A.4."""
    B, N, D = x.shape
    h = _ln(x, sd, prefix + ".self_attn_layer_norm")
    q = _linear(h, sd, prefix + ".self_attn.q_proj", patch_rows=True).view(B, N, heads, D // heads)
    k = _linear(h, sd, prefix + ".self_attn.k_proj", patch_rows=True).view(B, N, heads, D // heads)
    v = _linear(h, sd, prefix + ".self_attn.v_proj", patch_rows=True).view(B, N, heads, D // heads)
    a = _r16(dilated_attention_core(q, k, v, seg_lengths, ratios, impl=attn_impl))
    a = _ln(a, sd, prefix + ".self_attn.inner_attn_ln")
    x = x + _linear(a, sd, prefix + ".self_attn.out_proj", patch_rows=True)
    h = _ln(x, sd, prefix + ".final_layer_norm")
    h = _linear(h, sd, prefix + ".ffn.fc1", patch_rows=True)
    h = F.gelu(h.float()).type_as(h) if not F16_PATCH_OPERANDS else F.gelu(h)     # GELU is forced to fp32 (feedforward_network.py:136)
    h = _ln(h, sd, prefix + ".ffn.ffn_layernorm")
    return x + _linear(h, sd, prefix + ".ffn.fc2", patch_rows=True)


# ------------------------------------------------------------------------------------------------
# Gene encoder (models/genomic_utils/gene_encoder.py:97-223)
# ------------------------------------------------------------------------------------------------
def gene_encoder(genes: Sequence[torch.Tensor], sd, prefix="gene_encoder", depth=3):
    rows = []
    for i, g in enumerate(genes):
        h = F.elu(_linear(g, sd, f"{prefix}.gene_networks.{i}.0.0"))
        rows.append(F.elu(_linear(h, sd, f"{prefix}.gene_networks.{i}.1.0")))
    z = torch.cat(rows).unsqueeze(0)                                   # [1, G, 256]
    for k in range(depth):
        p = f"{prefix}.mlp_mixer.{k}"
        t = _ln(z, sd, p + ".0.norm")                                  # token mixing: Conv1d(k=1) over the group axis
        t = F.conv1d(t, sd[p + ".0.fn.0.weight"], sd[p + ".0.fn.0.bias"])
        t = F.conv1d(F.gelu(t), sd[p + ".0.fn.3.weight"], sd[p + ".0.fn.3.bias"])
        z = z + t
        t = _ln(z, sd, p + ".1.norm")                                  # channel mixing
        t = _linear(F.gelu(_linear(t, sd, p + ".1.fn.0")), sd, p + ".1.fn.3")
        z = z + t
    z = _ln(z, sd, f"{prefix}.mlp_mixer.{depth}")
    z = _linear(z, sd, f"{prefix}.mlp_mixer.{depth + 1}")              # [1, G, 768]
    z = _linear(z.permute(0, 2, 1), sd, f"{prefix}.pathway_compression").permute(0, 2, 1)
    return z                                                            # [1, final_groups, 768]


# ------------------------------------------------------------------------------------------------
# Positional table (gigapath/pos_embed.py:34-81, slide_encoder.py:198-211)
# ------------------------------------------------------------------------------------------------
def pos_embed_rows(coords, embed_dim, ngrids, dtype):
    import numpy as np
    half = embed_dim // 2
    omega = np.arange(half // 2, dtype=float)
    omega /= half / 2.0
    omega = 1.0 / 10000 ** omega
    grid = torch.floor(coords / 256.0)
    row, col = grid[..., 0].long(), grid[..., 1].long()
    assert int(row.max()) < ngrids and int(col.max()) < ngrids

    def enc(p):
        out = np.einsum("m,d->md", p.reshape(-1).numpy().astype(np.float32), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([enc(col), enc(row)], axis=1)                 # w (col) half first
    emb = torch.from_numpy(emb).float().to(dtype)                      # table is stored as fp32 (slide_encoder.py:150)
    return emb.view(*coords.shape[:-1], embed_dim)


# ------------------------------------------------------------------------------------------------
# Full model forward (models/aggregators/longvit_adapter.py:205-347) and train-step loss
# ------------------------------------------------------------------------------------------------
def model_forward(sd: Dict[str, torch.Tensor], cfg, x, coords, genes, task_token, seg_lengths,
                  taps: Optional[dict] = None, clinical=None):
    """x [1,L,in], coords [1,L,2], genes list of [1,n_i], task_token [num_tasks] -> [1, output_dim]."""
    heads = cfg.num_heads
    ratios = (1, 2, 4, 8, 16)
    x = F.linear(_r16(x), _r16(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"])          # LVA:232
    x = x + pos_embed_rows(coords, cfg.embed_dim, cfg.slide_ngrids, x.dtype)             # LVA:235-237
    cls = sd["cls_token"] + 0.0                                                          # + pos_embed[0] == zeros
    c = gene_encoder(genes, sd, depth=cfg.gene.depth)                                    # LVA:257
    ngc = int(getattr(cfg, "prompt_agg", "avg") == "cls")
    if ngc:                                                                              # LVA:259-261: learned token in front of the gene tokens
        c = torch.cat((sd["gene_cls"], c), dim=1)
    if cfg.is_multi:                                                                     # LVA:263-266
        t = _ln(_linear(task_token.unsqueeze(0), sd, "task_weight.0"), sd, "task_weight.1")
        c = torch.cat((t.unsqueeze(0), c), dim=1)
    ncl = int(getattr(cfg, "clinical", False))
    if ncl:                                                                              # LVA:578-580 (clinical variant)
        ce = _linear(F.relu(_linear(clinical, sd, "clinical_mlp.0")), sd, "clinical_mlp.2")
        c = torch.cat((_ln(ce, sd, "clinical_mlp.3").unsqueeze(0), c), dim=1)
    pe = sd["gene_pe"]
    if taps is not None:
        taps["x0"] = x.detach().clone(); taps["c0"] = c.detach().clone()
    a0 = int(cfg.interaction_indexes[0][0])
    if a0 != 0:                                                                          # LVA:269-281: plain backbone layers first
        h = torch.cat((cls, x), dim=1)
        for l in range(a0):
            h = encoder_layer(h, sd, f"encoder.layers.{l}", seg_lengths, ratios)
        cls, x = h[:, :1], h[:, 1:]
    for i, (a, b) in enumerate(cfg.interaction_indexes):                                 # LVA:294-307
        if i > 0 and cfg.use_prompt_sa:
            c = prompt_self_attention(c, pe, sd, f"prompt_selfattention.{i}", heads)
        x = injector(x, c, pe, sd, f"interactions.{i}.injector", heads)                  # AM:487-491
        h = torch.cat((cls, x), dim=1)
        for l in range(a, b + 1):
            h = encoder_layer(h, sd, f"encoder.layers.{l}", seg_lengths, ratios)
        cls, x = h[:, :1], h[:, 1:]
        c = extractor(c, x, pe, sd, f"interactions.{i}.extractor", heads)                # AM:511-515
        if i == len(cfg.interaction_indexes) - 1 and cfg.use_extra_extractor:
            for j in range(2):
                c = extractor(c, x, pe, sd, f"interactions.{i}.extra_extractors.{j}", heads)
        if taps is not None:
            taps[f"cls{i}"] = cls.detach().clone(); taps[f"c{i}"] = c.detach().clone()
            taps[f"x{i}_head"] = x[:, :8].detach().clone()
    nt = int(cfg.is_multi)
    img = x.mean(dim=1).unsqueeze(0) if getattr(cfg, "global_pool", False) else cls      # LVA:309-312 (x: the patch rows, cls excluded)
    clin_out, task_out = c[:, :ncl], c[:, ncl:ncl + nt]                                  # LVA:315-325 / 620-641
    gene_out = c[:, ncl + nt:ncl + nt + 1] if ngc else c[:, ncl + nt:].mean(dim=1, keepdim=True)
    if cfg.token_agg == "sum":
        out = img + gene_out + (task_out if nt else 0) + (clin_out if ncl else 0)
    else:
        parts = [img] + ([task_out] if nt else []) + [gene_out] + ([clin_out] if ncl else [])
        out = torch.cat(parts, dim=-1)
    out = _ln(out, sd, "final_norm")
    return _linear(out.squeeze(1), sd, "final_project")


# ------------------------------------------------------------------------------------------------
# TITAN configuration (models/aggregators/titan_adapter.py:249-438; adapter_modules.py:526-558).  The slide encoder itself
# is not in the reference tree (parity unpinned): `vit` is any object with the surface the reference uses.
# ------------------------------------------------------------------------------------------------
def titan_gridding(features, coords, patch_size_lv0):
    """preprocess_features (TA:295-327): features [L, C], integer coords [L, 2] -> (grid [1, C, H, W], coords grid
    [1, 2, H, W], bg mask [1, H, W])."""
    features, coords = features.reshape(-1, features.shape[-1]), coords.reshape(-1, 2)
    g = torch.div(coords - coords.min(dim=0).values, patch_size_lv0, rounding_mode="floor")
    g = g - g.min(dim=0).values
    H, W = (int(v) + 1 for v in g.max(dim=0).values)
    idx = g[:, 0] * W + g[:, 1]
    fg = torch.zeros(H * W, features.shape[-1], dtype=features.dtype).index_add_(0, idx, features)
    cg = torch.zeros(H * W, 2, dtype=torch.int64).index_add_(0, idx, coords.to(torch.int64))
    return (fg.view(H, W, -1).permute(2, 0, 1).unsqueeze(0), cg.view(H, W, 2).permute(2, 0, 1).unsqueeze(0),
            (fg != 0).any(dim=1).view(1, H, W))


def alibi_bias_2d(cells, slopes):
    """The additive attention bias the TITAN blocks receive (TA:253-269 `get_alibi(w, h, bg_mask)`; the snapshot's source is absent,
    the structure is the one tests/golden/titan_standin.py implements and modaltune_amd.titan verifies against the user's module):
    bias[h, i, j] = -slopes[h] * euclidean distance of the grid cells of tokens i, j; token 0 = cls: zero row and column.
    cells [Lv, 2] (row, col) of the kept tokens in order; returns [H, 1 + Lv, 1 + Lv]."""
    pos = cells.to(torch.float64)
    T = pos.shape[0] + 1
    bias = torch.zeros(slopes.numel(), T, T, dtype=torch.float64)
    bias[:, 1:, 1:] = -slopes.to(torch.float64).view(-1, 1, 1) * torch.cdist(pos, pos)
    return bias


def dense_alibi_attention(q, k, v, bias=None):
    """softmax(q k^T / sqrt(d) + bias) v per head: the attention of a TITAN ViT block as the reference drives it
    (`blocks.modules_list[i](x, attn_bias, bg_mask)`, TA:359-361; adapter_modules.py:535).  q, k, v [B, N, H, d]; bias [H, N, N]
    or None; returns [B, N, H, d]."""
    d = q.shape[-1]
    s = torch.einsum("bihd,bjhd->bhij", q, k) / math.sqrt(d)
    if bias is not None:
        s = s + bias.to(s.dtype)
    return torch.einsum("bhij,bjhd->bihd", torch.softmax(s, dim=-1), v)


def titan_model_forward(sd, cfg, vit, x, coords, genes, task_token, patch_size_lv0=1024, clinical=None):
    """TITANGeneAdapter.forward (TA:329-438) / the clinical variant: x [1, L, C], coords [1, L, 2] -> [1, output_dim]."""
    heads = cfg.num_heads
    fg, cg, bgm = titan_gridding(x, coords, patch_size_lv0)
    B, nc, w, h = fg.shape
    t = fg.flatten(2, 3).transpose(1, 2)                                                  # TA:253-293, B == 1
    bias = vit.get_alibi(w, h, bgm).to(t.dtype)
    t = vit.norm_pre(vit._pos_embed(vit.patch_embed(t), cg, w, h))
    mask = torch.cat((torch.ones((1, 1), dtype=torch.bool), bgm.view(1, -1)), dim=1)
    t = t[mask].unsqueeze(0)
    c = gene_encoder(genes, sd, depth=cfg.gene.depth)
    if cfg.is_multi:
        tk = _ln(_linear(task_token.unsqueeze(0), sd, "task_weight.0"), sd, "task_weight.1")
        c = torch.cat((tk.unsqueeze(0), c), dim=1)
    ncl = int(getattr(cfg, "clinical", False))
    if ncl:
        ce = _linear(F.relu(_linear(clinical, sd, "clinical_mlp.0")), sd, "clinical_mlp.2")
        c = torch.cat((_ln(ce, sd, "clinical_mlp.3").unsqueeze(0), c), dim=1)
    pe = sd["gene_pe"]
    cls, xx = t[:, :1], t[:, 1:]
    for i, (a, b) in enumerate(cfg.interaction_indexes):                                  # TA:376-391, AM:526-558
        if i > 0 and cfg.use_prompt_sa:
            c = prompt_self_attention(c, pe, sd, f"prompt_selfattention.{i}", heads)
        xx = injector(xx, c, pe, sd, f"interactions.{i}.injector", heads)
        hcat = torch.cat((cls, xx), dim=1)
        for l in range(a, b + 1):
            hcat = vit.blocks.modules_list[l](hcat, bias, mask)
        cls, xx = hcat[:, :1], hcat[:, 1:]
        c = extractor(c, xx, pe, sd, f"interactions.{i}.extractor", heads)
        if i == len(cfg.interaction_indexes) - 1 and cfg.use_extra_extractor:
            for j in range(2):
                c = extractor(c, xx, pe, sd, f"interactions.{i}.extra_extractors.{j}", heads)
    img, _ = vit.forward_attn_pool(vit.norm(torch.cat((cls, xx), dim=1)), bg_mask=mask)   # TA:399-402
    img = img.unsqueeze(0)
    nt = int(cfg.is_multi)
    clin_out, task_out = c[:, :ncl], c[:, ncl:ncl + nt]
    gene_out = c[:, ncl + nt:].mean(dim=1, keepdim=True)
    if cfg.token_agg == "sum":
        out = img + gene_out + (task_out if nt else 0) + (clin_out if ncl else 0)
    else:
        out = torch.cat([img] + ([task_out] if nt else []) + [gene_out] + ([clin_out] if ncl else []), dim=-1)
    return _linear(_ln(out, sd, "final_norm").squeeze(1), sd, "final_project")


def projector_forward(text, psd):
    """Projection_layer (train_modaltune.py:44-59) on [4,512] + row L2 normalisation (TM:211-213)."""
    h = F.linear(text, psd["conv1.0.weight"].flatten(1), psd["conv1.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), psd["conv1.1.weight"].flatten(), psd["conv1.1.bias"].flatten(), LN_EPS)
    h = F.linear(F.relu(h), psd["conv1.3.weight"].flatten(1), psd["conv1.3.bias"])
    return h / h.norm(dim=-1, keepdim=True)


def distill_loss(logits, text_proj, temperature=1.0):
    """train_modaltune.py:225-233: KLDiv(sum) of log_softmax(logit/|logit|) vs softmax(text[[0,1,3]]) * T^2 * 10."""
    logit = logits / logits.norm(dim=-1, keepdim=True)
    logp = F.log_softmax(logit / temperature, dim=1)
    p = F.softmax(text_proj[[0, 1, 3], :] / temperature, dim=1)
    if logp.shape[0] != p.shape[0]:        # single-task model: [1, O] logits against 3 targets (nn.KLDivLoss broadcasts)
        logp = logp.expand_as(p)
    return F.kl_div(logp, p, reduction="sum") * (temperature ** 2) * 10


def multitask_logits(sd, cfg, x, coords, genes, seg_lengths, task_ids=(0, 1, 2), taps=None, clinical=None):
    """multitask_forward (train_modaltune.py:156-179): one model call per task id, or a single call without a task token
    when the model is single-task (is_multi False; the [1, O] logits then broadcast against the 3 text rows in the loss)."""
    if not cfg.is_multi:
        tp = {} if taps is not None else None
        out = model_forward(sd, cfg, x, coords, genes, None, seg_lengths, taps=tp, clinical=clinical)
        if taps is not None:
            taps[0] = tp
        return out
    eye = torch.eye(cfg.multi_task, dtype=x.dtype)
    outs = []
    for t in task_ids:
        tp = {} if taps is not None else None
        outs.append(model_forward(sd, cfg, x, coords, genes, eye[t], seg_lengths, taps=tp, clinical=clinical))
        if taps is not None:
            taps[t] = tp
    return torch.cat(outs, dim=0)


def train_step_loss_and_grads(sd, cfg, trainable: Sequence[str], x, coords, genes, text, psd, seg_lengths, clinical=None):
    """Forward x3 + loss + backward (train_modaltune.py:211-235). Returns logits, loss, {key: grad}."""
    sd = {k: (v.clone().requires_grad_(True) if k in set(trainable) else v) for k, v in sd.items()}
    logits = multitask_logits(sd, cfg, x, coords, genes, seg_lengths, clinical=clinical)
    loss = distill_loss(logits, projector_forward(text, psd))
    loss.backward()
    grads = {k: sd[k].grad for k in trainable}
    return logits.detach(), loss.detach(), grads


def adamw_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01):
    """torch.optim.AdamW semantics (decoupled decay; TM:145-149 uses lr/20, wd 0.01, betas (0.9, 0.999))."""
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p = p - (lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps)
    return p, m, v
