"""Kernel-level parity of the TITAN-side HIP kernels (csrc/dense_attn.hip, csrc/titan.hip) against the CPU oracle
(oracle/modaltune_oracle.py: dense_alibi_attention, alibi_bias_2d, titan_gridding), through the C ABI.

Tolerances (fp16 operands, fp32 accumulation, written per assertion): attention outputs 3e-3, its gradients 2e-2 -- the bars of
the dilated kernels' tests; integer / index work (gridding, token order) bit-exact.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H16, F32, I32 = torch.float16, torch.float32, torch.int32
QK = 0.125 * 1.4426950408889634      # MT_DENSE_QK_SCALE_LOG2


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _cells(Lv, span, seed):
    """Lv distinct cells on a span x span grid (row-major sorted, as the token order is), int32 [Lv, 2]."""
    r = np.random.Generator(np.random.PCG64([seed, Lv, span]))
    flat = np.sort(r.choice(span * span, size=Lv, replace=False))
    return torch.from_numpy(np.stack([flat // span, flat % span], 1).astype(np.int32))


def _slopes(H):
    return torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], dtype=torch.float64)


@pytest.mark.parametrize("N,B,span,bias", [(70, 2, 12, True), (129, 1, 40, True), (577, 3, 30, True), (1089, 2, 2000, True),
                                           (64, 1, 9, True), (200, 2, 20, False),
                                           # edge sizes: cls alone, cls + one cell, one row short of / exactly one and two key tiles
                                           (1, 1, 1, True), (2, 3, 3, True), (63, 2, 8, True), (128, 1, 12, True), (256, 2, 16, False)])
def test_dense_alibi_attention_fwd_bwd_vs_oracle(N, B, span, bias):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd import ops
    from oracle import modaltune_oracle as O
    H, d = 12, 64
    D, M, Lv = H * d, B * N, N - 1
    g = torch.Generator().manual_seed(1000 + N)
    q, k, v = (torch.randn(B, N, H, d, generator=g) * s for s in (1.2, 1.2, 1.0))
    do = torch.randn(B, N, H, d, generator=g)
    # the kernels see fp16 operands: q' = QK * q rounded once; the oracle gets exactly those values back
    q16, k16, v16, do16 = (QK * q).half(), k.half(), v.half(), do.half()
    qkv = torch.cat([q16.reshape(M, D), k16.reshape(M, D), v16.reshape(M, D)], dim=1).contiguous().cuda()
    cells = _cells(Lv, span, N) if Lv > 0 else torch.zeros(0, 2, dtype=torch.int32)
    slopes = _slopes(H)
    dev = "cuda"
    dist = nslope = None
    if bias:
        dist = torch.full((ops.alibi_dist_halves(N),), float("nan"), dtype=H16, device=dev)
        ops.alibi_dist(cells.cuda() if Lv > 0 else None, N, dist)
        nslope = (-slopes * math.log2(math.e)).float().cuda()
    plan = ops.make_dense_plan(N, B, H, dist, nslope)
    o = torch.empty(M, D, dtype=H16, device=dev)
    lse = torch.empty(M, H, dtype=F32, device=dev)
    ops.dense_attn_fwd(qkv, plan, o, lse)
    delta = torch.empty(M, H, dtype=F32, device=dev)
    dqkv = torch.full((M, 3 * D), float("nan"), dtype=H16, device=dev)
    ops.dense_attn_bwd(qkv, o, do16.reshape(M, D).cuda(), lse, plan, delta, dqkv)
    torch.cuda.synchronize()

    qo = (q16.double() / QK).requires_grad_(True)
    ko, vo = k16.double().requires_grad_(True), v16.double().requires_grad_(True)
    bt = O.alibi_bias_2d(cells, slopes) if bias else None
    ref = O.dense_alibi_attention(qo, ko, vo, bt)
    (ref * do16.double()).sum().backward()
    assert rel(o.view(B, N, H, d), ref.detach()) < 3e-3
    # lse = natural-log LSE of the biased, scaled logits
    s = torch.einsum("bihd,bjhd->bhij", qo.detach(), ko.detach()) / 8.0 + (bt if bias else 0.0)
    assert float((lse.view(B, N, H).double().cpu() - torch.logsumexp(s, dim=-1).transpose(1, 2)).abs().max()) < 2e-3
    dg = dqkv.view(B, N, 3, H, d).double().cpu()
    assert torch.isfinite(dg).all()
    # (relative to max(|reference|, 1e-2): with one or two tokens dq / dk are exact zeros in the reference)
    relf = lambda a, b: float((a - b).abs().max() / max(float(b.abs().max()), 1e-2))
    assert relf(dg[:, :, 0] * QK, qo.grad) < 2e-2         # q columns: gradient of the pre-scaled q'
    assert relf(dg[:, :, 1], ko.grad) < 2e-2
    assert relf(dg[:, :, 2], vo.grad) < 2e-2


def test_alibi_distance_table_layout_and_values():
    """mt_alibi_dist: every (lane-side token a, tile-side token b) pair sits where the kernels' accumulator register i of
    sub-block sub in lane (l31, hh) looks for it, holds the fp16-rounded euclidean cell distance, and is zero to / from cls and
    past the end; also far outside the old +-1024 window of the side-table scheme."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd import ops
    g = torch.Generator().manual_seed(5)
    for N, span in ((7, 2049), (130, 300), (333, 30000)):
        cells = torch.randint(0, span, (N - 1, 2), generator=g, dtype=I32)
        tab = torch.full((ops.alibi_dist_halves(N),), float("nan"), dtype=H16, device="cuda")
        ops.alibi_dist(cells.cuda(), N, tab)
        nA, nt = 4 * ((N + 127) // 128), (N + 63) // 64
        assert tab.numel() == nA * nt * 2048
        t5 = tab.cpu().view(nA, nt, 4, 64, 8)                     # [A, t, piece j, lane, e]
        p = torch.zeros(N, 2, dtype=torch.float64)
        p[1:] = cells.double()
        want = (p[:, None] - p[None]).pow(2).sum(-1).sqrt()
        want[0, :] = 0.0
        want[:, 0] = 0.0
        full = torch.zeros(nA * 32, nt * 64, dtype=torch.float64)
        full[:N, :N] = want
        j, lane, e = torch.meshgrid(torch.arange(4), torch.arange(64), torch.arange(8), indexing="ij")
        sub, half, l31, hh = j >> 1, j & 1, lane & 31, lane >> 5
        i = 8 * half + e
        b_in = sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh          # tile-side token of (piece, lane, element)
        for A in range(nA):
            for t in range(nt):
                got = t5[A, t].double()
                exact = full[(A * 32 + l31), (t * 64 + b_in)]
                exp = exact.half().double()
                if span <= 2049:                                   # dx^2 + dy^2 < 2^24: the fp32 arithmetic is exact up to the root
                    assert torch.equal(got, exp), (N, A, t)
                else:                                              # (fp32 sum rounded before the root: a tie may fall the other way)
                    assert float(((got - exact).abs() - exact * 2.0 ** -11).max()) <= 0.0, (N, A, t)
                assert torch.equal(got == 0, exact == 0)


def test_gelu_f16_fwd_bwd():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (3.0 * torch.randn(1000 * 64, generator=g)).half().cuda()
    dy = torch.randn(1000 * 64, generator=g).half().cuda()
    y, dx = torch.empty_like(x), torch.empty_like(x)
    ops.gelu_f16_fwd(x, y)
    ops.gelu_f16_bwd(x, dy, dx)
    xr = x.double().cpu().requires_grad_(True)
    yr = torch.nn.functional.gelu(xr)
    yr.backward(dy.double().cpu())
    assert float((y.double().cpu() - yr.detach()).abs().max()) < 2e-3
    assert float((dx.double().cpu() - xr.grad).abs().max()) < 4e-3


@pytest.mark.parametrize("N,B,heads", [(257, 2, 12), (1000, 3, 8)])
def test_pool_attention_core(N, B, heads):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd import ops
    E = 768
    hd = E // heads
    g = torch.Generator().manual_seed(5 + N)
    q = torch.randn(1, E, generator=g)
    kv = (0.7 * torch.randn(B * N, 2 * E, generator=g)).half()
    dout = torch.randn(B, 1, E, generator=g)
    out = torch.empty(B, 1, E, dtype=F32, device="cuda")
    scores = torch.empty(B * heads * N, dtype=F32, device="cuda")
    lse = torch.empty(B * heads, dtype=F32, device="cuda")
    part = torch.empty(ops.pool_attn_workspace_floats(B, N, heads, 1), dtype=F32, device="cuda")
    dkv = torch.empty(B * N, 2 * E, dtype=H16, device="cuda")
    ops.pool_attn_fwd(q.cuda(), kv.cuda(), out, scores, lse, part, B, N, E, heads, 1)
    ops.pool_attn_bwd(q.cuda(), kv.cuda(), scores, lse, out, dout.cuda(), dkv, B, N, E, heads, 1)
    kvr = kv.double().requires_grad_(True)
    k, v = kvr[:, :E].view(B, N, heads, hd), kvr[:, E:].view(B, N, heads, hd)
    s = torch.einsum("hd,bnhd->bhn", q.double().view(heads, hd), k) / math.sqrt(hd)
    ref = torch.einsum("bhn,bnhd->bhd", torch.softmax(s, -1), v).reshape(B, 1, E)
    (ref * dout.double()).sum().backward()
    assert rel(out, ref.detach()) < 1e-4
    assert rel(dkv, kvr.grad) < 5e-3


@pytest.mark.parametrize("L,grid,psz", [(300, 24, 1024), (1000, 40, 1000), (77, 9, 256)])
def test_device_gridding_matches_oracle_exactly(L, grid, psz):
    """csrc/titan.hip's grid-free token construction == the rows `x[bg_mask]` keeps of the reference's gridded tensor
    (oracle.titan_gridding restates TA:295-327): same tokens in the same order, sums bit-exact (patch-order adds), also with
    patches sharing a cell and a cell whose features sum to exactly zero (background, TA:326)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.titan import device_tokens
    from oracle import modaltune_oracle as O
    r = np.random.Generator(np.random.PCG64([L, grid]))
    C = 768
    x = r.standard_normal((L, C)).astype(np.float32)
    cell = r.choice(grid * grid, size=L - L // 10, replace=False)
    cell = np.concatenate([cell, r.choice(cell, size=L // 10)])           # duplicates (some cells hold 2-3 patches)
    x[5] = 0.0                                                            # an occupied cell with all-zero features ...
    dup_of_5 = np.nonzero(cell == cell[5])[0]
    for j in dup_of_5:
        x[j] = 0.0                                                        # (every patch of that cell)
    coords = (np.stack([cell // grid, cell % grid], 1) * psz + r.integers(0, psz, size=(L, 2)) + 7 * psz + 3).astype(np.int64)
    fg, cg, bgm = O.titan_gridding(torch.from_numpy(x), torch.from_numpy(coords), psz)
    want = fg[0].flatten(1).T[bgm.view(-1)]                               # [Lv, C] row-major foreground cells
    x16, cells, dims, Lv = device_tokens(torch.from_numpy(x).cuda(), torch.from_numpy(coords).cuda(), psz)
    torch.cuda.synchronize()
    assert Lv == want.shape[0] and tuple(dims.cpu().tolist()) == tuple(fg.shape[-2:])
    assert torch.equal(cells.cpu().long(), torch.nonzero(bgm[0]))
    assert torch.equal(x16.cpu(), want.half())                            # fp32 sums in patch order, rounded once
    x16b, cells_b, _, _ = device_tokens(torch.from_numpy(x).cuda(), torch.from_numpy(coords).cuda(), psz)
    assert torch.equal(x16, x16b) and torch.equal(cells, cells_b)


def test_device_gridding_flags_bad_coords():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd.titan import device_tokens
    x = torch.randn(20, 768).cuda()
    coords = (torch.arange(40, dtype=torch.float32).view(20, 2) * 1024).cuda()
    coords[3, 1] = float("nan")
    err = torch.zeros(1, dtype=I32, device="cuda")
    device_tokens(x, coords, 1024, err)
    assert int(err) & 1


def test_dense_attention_full_size_properties():
    """BASELINE config 4's largest bag (6144 cells + cls, 3 passes): properties that need no O(N^2) oracle on the host --
    (i) a torch fp32 softmax reference on the GPU for two heads of one pass; (ii) rows of P sum to one: with v == 1 the output is
    1 and lse is finite; (iii) the three passes are independent: permuting the pass order permutes the outputs."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd import ops
    N, B, H, d = 6145, 3, 12, 64
    D, M = H * d, B * N
    g = torch.Generator(device="cuda").manual_seed(9)
    qkv = (torch.randn(M, 3 * D, device="cuda", generator=g) * 0.6).half()
    side = 80
    flat = torch.randperm(side * side, device="cuda", generator=g)[:N - 1].sort().values
    cells = torch.stack([flat // side, flat % side], 1).int()
    dtab = torch.empty(ops.alibi_dist_halves(N), dtype=H16, device="cuda")      # (the plan points into it: keep it alive)
    ops.alibi_dist(cells, N, dtab)
    slopes = _slopes(H).cuda()
    nslope = (-slopes * math.log2(math.e)).float()
    plan = ops.make_dense_plan(N, B, H, dtab, nslope)
    o, lse = torch.empty(M, D, dtype=H16, device="cuda"), torch.empty(M, H, dtype=F32, device="cuda")
    ops.dense_attn_fwd(qkv, plan, o, lse)
    dist = torch.cdist(cells.float(), cells.float())
    for b, h in ((0, 0), (2, 11)):
        q, k, v = (qkv[b * N:(b + 1) * N, w * D + h * d:w * D + (h + 1) * d].float() for w in range(3))
        s = (q @ k.T) * math.log(2.0)                       # q carries 64^-1/2 log2(e)
        s[1:, 1:] -= slopes[h].float() * dist
        ref = torch.softmax(s, -1) @ v
        got = o[b * N:(b + 1) * N, h * d:(h + 1) * d].float()
        assert float((got - ref).abs().max() / ref.abs().max()) < 3e-3
        assert float((lse[b * N:(b + 1) * N, h] - torch.logsumexp(s, -1)).abs().max()) < 2e-3
    ones = qkv.clone()
    ones[:, 2 * D:] = 1.0
    o1 = torch.empty_like(o)
    ops.dense_attn_fwd(ones, plan, o1, lse)
    assert float((o1.float() - 1.0).abs().max()) < 2e-3 and torch.isfinite(lse).all()
    perm = qkv.view(B, N, 3 * D)[[2, 0, 1]].reshape(M, 3 * D).contiguous()
    o2 = torch.empty_like(o)
    ops.dense_attn_fwd(perm, plan, o2, lse)
    assert torch.equal(o2.view(B, N, D), o.view(B, N, D)[[2, 0, 1]])
