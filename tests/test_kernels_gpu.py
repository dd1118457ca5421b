"""GPU parity tests of the individual C-ABI kernels against fp64 torch-CPU restatements / the oracle.
Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import unit_inputs  # noqa: E402
from modaltune_amd import synth  # noqa: E402
from modaltune_amd.config import ModelConfig, branch_table, segment_lengths  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from modaltune_amd import ops as _ops
    return _ops


DEV = "cuda"


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rng(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (1000, 192, 768), (257, 3072, 768), (130, 768, 3072), (515, 384, 768),
                                   (4100, 768, 768), (2303, 3072, 768), (2049, 768, 3072), (5000, 2304, 768)])   # >= 2048 rows: 8-wave kernel
def test_gemm_nt_bias(ops, M, N, K):
    g = rng(M + N)
    A = torch.randn(M, K, generator=g).half()
    W = (torch.randn(N, K, generator=g) * 0.05).half()
    bias = torch.randn(N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    for dt in (torch.float16, torch.float32):
        out = torch.zeros(M, N, dtype=dt, device=DEV)
        ops.gemm_nt(A.to(DEV), W.to(DEV), out, M, N, K, bias=bias.to(DEV))
        torch.cuda.synchronize()
        assert rel(out, ref) < (2e-3 if dt == torch.float16 else 2e-5)


def test_gemm_nt_rowmaps_resid_inject_posemb(ops):
    from modaltune_amd._lib import rowmap
    g = rng(7)
    B, L, N_, K, Nc = 3, 70, 71, 768, 768
    # A is a [B*N_, K] token buffer; logical rows = the patch rows (skip row 0 of every pass)
    Abuf = torch.randn(B * N_, K, generator=g).half()
    W = (torch.randn(Nc, K, generator=g) * 0.05).half()
    bias = torch.randn(Nc, generator=g)
    resid = torch.randn(L, Nc, generator=g)            # one slide broadcast to the B passes
    gamma = torch.randn(Nc, generator=g) * 0.1
    M = B * L
    amap = rowmap(L, N_, 1)
    cmap = rowmap(L, N_, 1)
    rmap = rowmap(L, 0, 0)
    out = torch.zeros(B * N_, Nc, device=DEV)
    ops.gemm_nt(Abuf.to(DEV), W.to(DEV), out, M, Nc, K, amap=amap, cmap=cmap, epilogue=ops.EPI_INJECT, bias=bias.to(DEV),
                resid=resid.to(DEV), ldr=Nc, rmap=rmap, colscale=gamma.to(DEV))
    torch.cuda.synchronize()
    A3 = Abuf.view(B, N_, K)[:, 1:].double()
    ref = (1 + gamma.double()) * resid.double() + gamma.double() * (A3 @ W.double().t() + bias.double())
    got = out.view(B, N_, Nc)[:, 1:]
    assert rel(got, ref) < 2e-5
    assert float(out.view(B, N_, Nc)[:, 0].abs().max()) == 0.0      # cls rows untouched
    # BIAS_RESID on the 8-wave kernel (M >= 2048), fp32 residual stream
    Mb = 2300
    Ab = torch.randn(Mb, K, generator=g).half()
    rb = torch.randn(Mb, Nc, generator=g)
    ob = torch.zeros(Mb, Nc, device=DEV)
    ops.gemm_nt(Ab.to(DEV), W.to(DEV), ob, Mb, Nc, K, epilogue=ops.EPI_BIAS_RESID, bias=bias.to(DEV), resid=rb.to(DEV), ldr=Nc)
    torch.cuda.synchronize()
    assert rel(ob, rb.double() + Ab.double() @ W.double().t() + bias.double()) < 2e-5
    # BIAS_RESID accumulate in place (dx += dy @ W)
    acc = torch.randn(M, Nc, generator=g)
    acc_d = acc.to(DEV).clone()
    A2 = torch.randn(M, K, generator=g).half()
    ops.gemm_nt(A2.to(DEV), W.to(DEV), acc_d, M, Nc, K, epilogue=ops.EPI_BIAS_RESID, resid=acc_d, ldr=Nc)
    torch.cuda.synchronize()
    assert rel(acc_d, acc.double() + A2.double() @ W.double().t()) < 2e-5
    # POSEMB
    from modaltune_amd.config import sincos_1d_table
    tab = torch.from_numpy(sincos_1d_table(64, Nc // 2))
    prow = torch.randint(0, 64, (M,), generator=g, dtype=torch.int32)
    pcol = torch.randint(0, 64, (M,), generator=g, dtype=torch.int32)
    out2 = torch.zeros(M, Nc, device=DEV)
    ops.gemm_nt(A2.to(DEV), W.to(DEV), out2, M, Nc, K, epilogue=ops.EPI_POSEMB, bias=bias.to(DEV), pos_table=tab.to(DEV),
                pos_row=prow.to(DEV), pos_col=pcol.to(DEV))
    torch.cuda.synchronize()
    ref2 = A2.double() @ W.double().t() + bias.double() + torch.cat([tab[pcol.long()], tab[prow.long()]], 1).double()
    assert rel(out2, ref2) < 2e-5


@pytest.mark.parametrize("N,K", [(3072, 768), (2304, 768)])
def test_gemm_nt_256x256_tile(ops, N, K):
    """M >= 8192 and N >= 2304 take the 8-wave 256 x 256 tile (ragged last row tile)."""
    g = rng(N)
    M = 8300
    A = torch.randn(M, K, generator=g).half()
    W = (torch.randn(N, K, generator=g) * 0.05).half()
    bias = torch.randn(N, generator=g)
    out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    ops.gemm_nt(A.to(DEV), W.to(DEV), out, M, N, K, bias=bias.to(DEV))
    torch.cuda.synchronize()
    assert rel(out, A.double() @ W.double().t() + bias.double()) < 2e-3


@pytest.mark.parametrize("M,N,K,kind", [(30003, 3072, 768, "bias"), (30003, 768, 3072, "bias"), (30003, 2304, 768, "qkv"), (30003, 768, 768, "bias"),
                                        (30003, 768, 2304, "none"), (29953, 768, 768, "none"), (12291, 3072, 768, "bias"), (24579, 2304, 1536, "qkv")])
def test_gemm_nt_persistent_kernel_backbone_shapes(ops, M, N, K, kind, monkeypatch):
    """The backbone's big-M shapes take the persistent one-wave-per-SIMD kernel (csrc/gemm_ps.hip: 192 x 256 tiles walked by one
    workgroup per CU, the finished tile drained under the next one): ragged last row tile, bias as the first slice's C operand,
    head-major q|k|v stores, the dX form without bias.  Checked against fp64 and against the ping-pong kernel on the same operands
    (MT_GEMM_PS=0), which it must reproduce to fp16 rounding of the same fp32 sums."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    A = (torch.randn(M, K, device=DEV, generator=g) * 0.5).half()
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
    bias = torch.randn(N, device=DEV, generator=g) if kind != "none" else None
    epi = ops.EPI_QKV_HM if kind == "qkv" else ops.EPI_BIAS
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("MT_GEMM_PS", mode)
        out = torch.full((M * N,), float("nan"), dtype=torch.float16, device=DEV)
        ops.gemm_nt(A, W, out, M, N, K, bias=bias, epilogue=epi)
        torch.cuda.synchronize()
        outs.append(out)
    ref = A.double() @ W.double().t()
    if bias is not None:
        ref += bias.double()
    if kind == "qkv":
        ref = ref.view(M, N // 48, 48).permute(1, 0, 2)
        got = [o.view(N // 48, M, 48) for o in outs]
    else:
        got = [o.view(M, N) for o in outs]
    assert int(torch.isnan(outs[0]).sum()) == 0
    assert rel(got[0], ref) < 2e-3 and rel(got[1], ref) < 2e-3
    # same products, same fp32 accumulation width, one fp16 rounding: the two kernels differ by summation order only
    assert float((got[0].float() - got[1].float()).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K,resid", [(9219, 768, 3072, True), (9219, 768, 2304, False), (8243, 768, 3072, True), (22503, 768, 3072, True),
                                           (6147, 3072, 768, False), (21753, 768, 2304, False)])
def test_gemm_nt_pingpong_tile_height_choice(ops, M, N, K, resid):
    """Round 6: the ping-pong kernel picks its tile heights by cost (gemm.hip launch_nt<256>): all 256-row tiles, whole rounds of them
    plus ONE round of 128-row tiles for the rest, or 128-row tiles only where the big ones would leave more than half the chip idle
    (M = 9 219, N = 768: 108 big tiles or 219 small ones).  Every split writes every row exactly once: fp64 reference, ragged last tile,
    the fc2 form with the fp32 residual epilogue and the plain fp16 one."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    A = (torch.randn(M, K, device=DEV, generator=g) * 0.5).half()
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
    bias = torch.randn(N, device=DEV, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    if resid:
        r = torch.randn(M, N, device=DEV, generator=g)
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        ops.gemm_nt(A, W, out, M, N, K, epilogue=ops.EPI_BIAS_RESID, bias=bias, resid=r, ldr=N)
        ref += r.double()
    else:
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
        ops.gemm_nt(A, W, out, M, N, K, bias=bias)
    torch.cuda.synchronize()
    assert int(torch.isnan(out).sum()) == 0
    assert rel(out, ref) < 2e-3


@pytest.mark.parametrize("M", [333, 2600, 8300])
def test_gemm_nt_head_major_qkv_epilogue(ops, M):
    g = rng(19)
    N, K = 2304, 768
    A = torch.randn(M, K, generator=g).half()
    W = (torch.randn(N, K, generator=g) * 0.05).half()
    bias = torch.randn(N, generator=g)
    out = torch.zeros(3 * 16 * M * 48, dtype=torch.float16, device=DEV)
    ops.gemm_nt(A.to(DEV), W.to(DEV), out, M, N, K, bias=bias.to(DEV), epilogue=ops.EPI_QKV_HM)
    torch.cuda.synchronize()
    ref = (A.double() @ W.double().t() + bias.double()).view(M, 3, 16, 48).permute(1, 2, 0, 3)
    assert rel(out.view(3, 16, M, 48), ref) < 2e-3


@pytest.mark.parametrize("M,N1,N2", [(1000, 192, 768), (4097, 64, 64), (333, 384, 768), (9000, 768, 192), (30003, 192, 192),
                                     (2050, 384, 192), (700, 64, 192), (515, 320, 448), (31, 128, 64)])
def test_gemm_tn_and_colsum(ops, M, N1, N2):
    """Weight gradient A^T B over every tile form (64 / 128 / 192 on each side), accumulating into C, with the bias gradient
    (column sums of A) riding on the same pass, and the standalone column-sum kernel."""
    g = rng(M)
    A = torch.randn(M, N1, generator=g).half()
    B = torch.randn(M, N2, generator=g).half()
    C0 = torch.randn(N1, N2, generator=g)
    b0 = torch.randn(N1, generator=g)
    out, bsum = C0.to(DEV).clone(), b0.to(DEV).clone()
    ops.gemm_tn(A.to(DEV), B.to(DEV), out, M, N1, N2, colsum=bsum)
    out2 = torch.zeros(N1, N2, device=DEV)
    ops.gemm_tn(A.to(DEV), B.to(DEV), out2, M, N1, N2)
    cs = torch.zeros(N1, device=DEV)
    ops.colsum(A.to(DEV), cs, M, N1)
    torch.cuda.synchronize()
    ref = A.double().t() @ B.double()
    assert rel(out, C0.double() + ref) < 1e-4 and rel(out2, ref) < 1e-4
    assert rel(bsum, b0.double() + A.double().sum(0)) < 1e-4
    assert rel(cs, A.double().sum(0)) < 1e-4


def test_sgemm_small(ops):
    g = rng(3)
    M, N, K = 65, 50, 77
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    out = torch.zeros(M, N, device=DEV)
    ops.sgemm(A.to(DEV), (K, 1), Bm.to(DEV), (K, 1), out, (N, 1), M, N, K, bias=bias.to(DEV), act=ops.ACT_GELU)
    torch.cuda.synchronize()
    assert rel(out, torch.nn.functional.gelu(A.double() @ Bm.double().t() + bias.double())) < 1e-5
    # transposed operands + accumulate + batch + bias on rows:  C[b] (n-major) += A[b]^T-strided
    Bt = 3
    A2 = torch.randn(Bt, K, M, generator=g)       # element (m,k) at k*M + m
    C0 = torch.randn(Bt, N, M, generator=g)       # element (m,n) at n*M + m
    bm = torch.randn(M, generator=g)
    Cd = C0.to(DEV).clone()
    ops.sgemm(A2.to(DEV), (1, M), Bm.to(DEV), (K, 1), Cd, (1, M), M, N, K, bias=bm.to(DEV), bias_on_m=True, accumulate=True,
              batch=Bt, a_bs=K * M, b_bs=0, c_bs=N * M)
    torch.cuda.synchronize()
    ref = C0.double() + (A2.double().transpose(1, 2) @ Bm.double().t() + bm.double()[:, None]).transpose(1, 2)
    assert rel(Cd, ref) < 1e-5
    # rowsum rider: dW = dy^T x with db = column sums of dy on the same launch (nn.Linear backward)
    R, Nout, Kin = 195, 50, 77
    dy = torch.randn(R, Nout, generator=g)
    x = torch.randn(R, Kin, generator=g)
    dW = torch.zeros(Nout, Kin, device=DEV)
    db0 = torch.randn(Nout, generator=g)
    db = db0.to(DEV).clone()
    ops.sgemm(dy.to(DEV), (1, Nout), x.to(DEV), (1, Kin), dW, (Kin, 1), Nout, Kin, R, accumulate=True, rowsum=db)
    torch.cuda.synchronize()
    assert rel(dW, dy.double().t() @ x.double()) < 1e-5
    assert rel(db, db0.double() + dy.double().sum(0)) < 1e-5


def _mask(ops, spec, shape):
    """The element-dropout scale factors of a dense tensor, read back from the standalone dropout kernel."""
    ones = torch.ones(shape, device=DEV)
    m = torch.empty_like(ones)
    R, D = int(np.prod(shape[:-1])), shape[-1]
    ops.dropout_f32(ones, m, R, D, spec)
    return m


@pytest.mark.parametrize("act", ["none", "gelu", "relu", "elu"])
@pytest.mark.parametrize("R,N,K", [(195, 192, 768), (66, 768, 192), (6, 256, 128), (33, 52, 76)])
def test_sgemm_fused_linear_forward_and_backward(ops, act, R, N, K):
    """nn.Linear + activation + Dropout + residual on one launch, and its backward (mask and act' applied to dy at the operand
    load, dX and dW (+ db) products in ONE mt_sgemm_multi launch) against fp64 torch autograd with the same mask."""
    code = {"none": ops.ACT_NONE, "gelu": ops.ACT_GELU, "relu": ops.ACT_RELU, "elu": ops.ACT_ELU}[act]
    g = rng(R * 7 + N + K + code)
    x, W, b = torch.randn(R, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    res, dy = torch.randn(R, N, generator=g), torch.randn(R, N, generator=g)
    rngbuf = torch.tensor([11, 0, 5, 0], dtype=torch.int32, device=DEV)
    spec = ops.dropout_spec(rngbuf, site=321, p=0.25) if N % 4 == 0 else None     # (the standalone mask kernel wants D % 4 == 0)
    mask = _mask(ops, spec, (R, N)).double().cpu() if spec is not None else torch.ones(R, N, dtype=torch.float64)
    xd, Wd, bd, resd, dyd = (t.to(DEV) for t in (x, W, b, res, dy))
    y = torch.full((R, N), float("nan"), device=DEV)
    pre = torch.full((R, N), float("nan"), device=DEV)
    ops.sgemm(xd, (K, 1), Wd, (K, 1), y, (N, 1), R, N, K, bias=bd, act=code, pre_out=pre, c_drop=spec, resid=resd)
    x64, W64, b64 = (t.double().requires_grad_(True) for t in (x, W, b))
    pre64 = x64 @ W64.t() + b64
    f = {"none": lambda t: t, "gelu": torch.nn.functional.gelu, "relu": torch.relu, "elu": torch.nn.functional.elu}[act]
    y64 = res.double() + f(pre64) * mask
    torch.cuda.synchronize()
    assert rel(pre, pre64) < 1e-5 and rel(y, y64) < 1e-5
    y64.backward(dy.double())
    dx = torch.randn(R, K, generator=g)
    dW0, db0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    dxd, dWd, dbd = dx.to(DEV).clone(), dW0.to(DEV).clone(), db0.to(DEV).clone()
    fuse = dict(a_aux=pre if code != ops.ACT_NONE else None, a_act=code, a_drop=spec)
    ops.sgemm_multi([ops.sgemm_problem(dyd, (N, 1), Wd, (1, K), dxd, (K, 1), R, K, N, accumulate=True, **fuse),
                     ops.sgemm_problem(dyd, (1, N), xd, (1, K), dWd, (K, 1), N, K, R, accumulate=True, rowsum=dbd, **fuse)])
    torch.cuda.synchronize()
    assert rel(dxd, dx.double() + x64.grad) < 1e-5
    assert rel(dWd, dW0.double() + W64.grad) < 1e-5
    assert rel(dbd, db0.double() + b64.grad) < 1e-5


def test_sgemm_multi_three_products_and_group_axis_form(ops):
    """Three unrelated products (different shapes, layouts, batch) in one launch land where three launches put them; the
    group-axis (Conv1d kernel 1) backward with dy as the strided A operand keeps the mask / act' indexing of the dense dy."""
    g = rng(77)
    shapes = [(65, 50, 77, 1), (16, 256, 6, 1), (20, 33, 40, 3)]
    probs, refs, outs = [], [], []
    keep = []
    for M, N, K, Bt in shapes:
        A, Bm = torch.randn(Bt, M, K, generator=g), torch.randn(Bt, N, K, generator=g)
        out = torch.full((Bt, M, N), float("nan"), device=DEV)
        Ad, Bd = A.to(DEV), Bm.to(DEV)
        keep += [Ad, Bd]
        probs.append(ops.sgemm_problem(Ad, (K, 1), Bd, (K, 1), out, (N, 1), M, N, K, batch=Bt, a_bs=M * K, b_bs=N * K, c_bs=M * N))
        refs.append(A.double() @ Bm.double().transpose(1, 2)); outs.append(out)
    ops.sgemm_multi(probs)
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert rel(o, r) < 1e-5
    # y[go, c] = drop(gelu(sum_g W[go, g] x[g, c] + b[go]));  dx[g, c] = sum_go dpre[go, c] W[go, g] with A = dy^T (strides (1, Cc))
    Go, G, Cc = 12, 6, 256
    W, x, b, dy = (torch.randn(*sh, generator=g) for sh in ((Go, G), (G, Cc), (Go,), (Go, Cc)))
    rngbuf = torch.tensor([3, 0, 9, 0], dtype=torch.int32, device=DEV)
    spec = ops.dropout_spec(rngbuf, site=77, p=0.3)
    mask = _mask(ops, spec, (Go, Cc)).double().cpu()
    Wd, xd, bd, dyd = (t.to(DEV) for t in (W, x, b, dy))
    y, pre = torch.empty(Go, Cc, device=DEV), torch.empty(Go, Cc, device=DEV)
    ops.sgemm(Wd, (G, 1), xd, (1, Cc), y, (Cc, 1), Go, Cc, G, bias=bd, bias_on_m=True, act=ops.ACT_GELU, pre_out=pre, c_drop=spec)
    W64, x64, b64 = (t.double().requires_grad_(True) for t in (W, x, b))
    y64 = torch.nn.functional.gelu(W64 @ x64 + b64[:, None]) * mask
    y64.backward(dy.double())
    dxd, dWd, dbd = torch.zeros(G, Cc, device=DEV), torch.zeros(Go, G, device=DEV), torch.zeros(Go, device=DEV)
    fuse = dict(a_aux=pre, a_act=ops.ACT_GELU, a_drop=spec)
    ops.sgemm_multi([ops.sgemm_problem(dyd, (1, Cc), Wd, (1, G), dxd, (1, Cc), Cc, G, Go, accumulate=True, **fuse),
                     ops.sgemm_problem(dyd, (Cc, 1), xd, (Cc, 1), dWd, (G, 1), Go, G, Cc, accumulate=True, rowsum=dbd, **fuse)])
    torch.cuda.synchronize()
    assert rel(y, y64) < 1e-5
    assert rel(dxd, x64.grad) < 1e-5 and rel(dWd, W64.grad) < 1e-5 and rel(dbd, b64.grad) < 1e-5


@pytest.mark.parametrize("M,N,K", [(195, 192, 768), (195, 768, 192), (192, 768, 195), (6, 3, 256), (3, 256, 6), (1, 4992, 3),
                                   (17, 33, 1), (64, 64, 64), (16, 16, 1024), (195, 192, 200)])
@pytest.mark.parametrize("layout", ["nt", "tn", "nn"])
def test_sgemm_small_shapes_and_strides(ops, M, N, K, layout):
    """The fp32-MFMA token-side GEMM over the shapes the step launches: k-contiguous operands (16-byte loads),
    transposed operands (the dW products), ragged M / N / K (the k range is split over four waves in multiples of 16),
    K below one MFMA k-block, rowsum rider on every layout."""
    g = rng(M * 1000 + N * 10 + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g)
    Ad = (A if layout != "tn" else A.t().contiguous()).to(DEV)            # tn: A stored [K, M]
    Bd = (Bm if layout == "nt" else Bm.t().contiguous()).to(DEV)          # tn / nn: B stored [K, N]
    a_str = (K, 1) if layout != "tn" else (1, M)
    b_str = (K, 1) if layout == "nt" else (1, N)
    out = torch.full((M, N), float("nan"), device=DEV)
    rs0 = torch.randn(M, generator=g)
    rs = rs0.to(DEV).clone()
    ops.sgemm(Ad, a_str, Bd, b_str, out, (N, 1), M, N, K, rowsum=rs)
    torch.cuda.synchronize()
    ref = A.double() @ Bm.double().t()
    assert float((out.double().cpu() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    assert float((rs.double().cpu() - (rs0.double() + A.double().sum(1))).abs().max()) < 1e-4


def test_pack_weights_grouped_and_broadcast_add(ops):
    """mt_pack_weights_f16 (all trainable fp16 caches in one launch: stored + transposed copies, row offsets of a
    concatenated cache, ragged 32x32 tiles) and mt_axpy_bcast."""
    g = rng(11)
    srcs = [torch.randn(192, 768, generator=g), torch.randn(192, 768, generator=g), torch.randn(768, 192, generator=g),
            torch.randn(50, 70, generator=g)]
    dev = [t.to(DEV) for t in srcs]
    kv_w = torch.zeros(384, 768, dtype=torch.float16, device=DEV)
    kv_t = torch.zeros(768, 384, dtype=torch.float16, device=DEV)
    o_w = torch.zeros(768, 192, dtype=torch.float16, device=DEV)
    o_t = torch.zeros(192, 768, dtype=torch.float16, device=DEV)
    r_w = torch.zeros(50, 70, dtype=torch.float16, device=DEV)
    recs = [[dev[0].data_ptr(), kv_w.data_ptr(), kv_t.data_ptr(), 192, 768, 0, 768, 384],
            [dev[1].data_ptr(), kv_w.data_ptr(), kv_t.data_ptr(), 192, 768, 192, 768, 384],
            [dev[2].data_ptr(), o_w.data_ptr(), o_t.data_ptr(), 768, 192, 0, 192, 768],
            [dev[3].data_ptr(), r_w.data_ptr(), 0, 50, 70, 0, 70, 0]]
    table = torch.tensor(recs, dtype=torch.int64, device=DEV)
    ops.pack_weights(table, len(recs))
    torch.cuda.synchronize()
    kv = torch.cat([srcs[0], srcs[1]]).half()
    assert torch.equal(kv_w.cpu(), kv) and torch.equal(kv_t.cpu(), kv.t())
    assert torch.equal(o_w.cpu(), srcs[2].half()) and torch.equal(o_t.cpu(), srcs[2].half().t())
    assert torch.equal(r_w.cpu(), srcs[3].half())
    a = torch.randn(3, 65, 768, generator=g)
    pe = torch.randn(65, 768, generator=g)
    y = torch.zeros(3, 65, 768, device=DEV)
    ops.axpy_bcast(a.to(DEV), pe.to(DEV), 1.0, y, 65 * 768)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), a + pe)
    # its adjoint: d pe += sum over the passes (mt_fold_rows, passes in ascending order)
    dpe0 = torch.randn(65, 768, generator=g)
    dpe = dpe0.to(DEV)
    ops.fold_rows(a.to(DEV), 3, 65 * 768, dpe)
    one = torch.randn(1, 4, generator=g)
    d1 = torch.zeros(4, device=DEV)
    ops.fold_rows(one.to(DEV), 1, 4, d1)
    torch.cuda.synchronize()
    assert torch.equal(dpe.cpu(), dpe0 + ((a[0] + a[1]) + a[2]))
    assert torch.equal(d1.cpu(), one[0])


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("D", [256, 768, 2304, 3072])
def test_layernorm_fwd_bwd_f32(ops, D):
    g = rng(D)
    M = 67
    x = torch.randn(M, D, generator=g) * 2 + 0.5
    w = 1 + 0.1 * torch.randn(D, generator=g)
    b = 0.1 * torch.randn(D, generator=g)
    add = torch.randn(5, D, generator=g)
    dy = torch.randn(M, D, generator=g)
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (D,), wd, bd, 1e-5) + add.double()[torch.arange(M) % 5]
    ref.backward(dy.double())
    y = torch.zeros(M, D, device=DEV)
    stats = torch.zeros(M, 2, device=DEV)
    ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), y, stats, M, D, add_rows=add.to(DEV), add_period=5)
    dx = torch.zeros(M, D, device=DEV)
    dw = torch.zeros(D, device=DEV)
    db = torch.zeros(D, device=DEV)
    if D <= 768:
        ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), stats, dx, M, D, dw=dw, db=db)
    else:
        ops.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), stats, dx, M, D)
    torch.cuda.synchronize()
    assert rel(y, ref) < 1e-5
    assert rel(dx, xd.grad) < 1e-4
    if D <= 768:
        assert rel(dw, wd.grad) < 1e-4 and rel(db, bd.grad) < 1e-4


def test_layernorm_gelu_f16_and_accumulate(ops):
    g = rng(11)
    M, D = 130, 3072
    a = (torch.randn(M, D, generator=g) * 1.5).half()
    w = 1 + 0.1 * torch.randn(D, generator=g)
    b = 0.1 * torch.randn(D, generator=g)
    dy = torch.randn(M, D, generator=g).half()
    ad = a.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(torch.nn.functional.gelu(ad), (D,), w.double(), b.double(), 1e-5)
    ref.backward(dy.double())
    y = torch.zeros(M, D, device=DEV, dtype=torch.float16)
    stats = torch.zeros(M, 2, device=DEV)
    ops.layernorm_fwd(a.to(DEV), w.to(DEV), b.to(DEV), y, stats, M, D, gelu_in=True)
    da = torch.zeros(M, D, device=DEV, dtype=torch.float16)
    ops.layernorm_bwd(dy.to(DEV), a.to(DEV), w.to(DEV), stats, da, M, D, gelu_in=True)
    torch.cuda.synchronize()
    assert rel(y, ref) < 2e-3
    assert rel(da, ad.grad) < 3e-3
    # fp32 residual stream: dy fp16, x fp32, dx fp32 accumulated, through a row map
    from modaltune_amd._lib import rowmap
    D2, B, L, N_ = 768, 2, 33, 34
    x = torch.randn(B * N_, D2, generator=g)
    w2 = 1 + 0.1 * torch.randn(D2, generator=g)
    dyh = torch.randn(B * L, D2, generator=g).half()
    acc0 = torch.randn(B * N_, D2, generator=g)
    xd = x.view(B, N_, D2)[:, 1:].double().requires_grad_(True)
    r2 = torch.nn.functional.layer_norm(xd, (D2,), w2.double(), torch.zeros(D2).double(), 1e-5)
    r2.backward(dyh.double().view(B, L, D2))
    yy = torch.zeros(B * L, D2, device=DEV, dtype=torch.float16)
    st = torch.zeros(B * L, 2, device=DEV)
    xm = rowmap(L, N_, 1)
    ops.layernorm_fwd(x.to(DEV), w2.to(DEV), torch.zeros(D2, device=DEV), yy, st, B * L, D2, xmap=xm)
    accd = acc0.to(DEV).clone()
    ops.layernorm_bwd(dyh.to(DEV), x.to(DEV), w2.to(DEV), st, accd, B * L, D2, xmap=xm, dxmap=xm, accumulate=True)
    # second output: dense fp16 copy of the accumulated dx (operand of the next dX GEMM)
    acc2, d16 = acc0.to(DEV)[:B * L].clone(), torch.zeros(B * L, D2, device=DEV, dtype=torch.float16)
    ops.layernorm_bwd(dyh.to(DEV), x.to(DEV), w2.to(DEV), st, acc2, B * L, D2, xmap=xm, accumulate=True, dx16=d16)
    torch.cuda.synchronize()
    assert torch.equal(d16, acc2.half())
    assert rel(yy.view(B, L, D2), r2) < 2e-3
    want = acc0.double().view(B, N_, D2).clone()
    want[:, 1:] += xd.grad
    assert rel(accd, want.view(B * N_, D2)) < 1e-4


# ------------------------------------------------------------------------------------------ dilated attention
def _hm(qkv_tok):
    """token-major [M, 2304] (q|k|v, head h at 48h) -> head-major [3][16][M][48] as the kernels read it."""
    M = qkv_tok.shape[0]
    return qkv_tok.view(M, 3, 16, 48).permute(1, 2, 0, 3).contiguous()


QK = 0.14433756729740643 * 1.4426950408889634      # MT_QK_SCALE_LOG2: the kernels take q pre-multiplied by it


def _prescale(qkv16):
    """fp16 q|k|v [.., 2304] -> (kernel input with q' = fp16(QK q), the exact fp64 q|k|v the kernels then see: q'/QK | k | v)."""
    k16 = qkv16.clone()
    k16[..., :768] = (qkv16[..., :768].float() * QK).half()
    eff = k16.double()
    eff[..., :768] /= QK
    return k16, eff


def _hm_inv(dm_hm, M):
    """head-major [16][M][48] -> token-major [M, 768]."""
    return dm_hm.view(16, M, 48).permute(1, 0, 2).reshape(M, 768)


def _attn_case(case, B=2):
    N, segs, ratios = unit_inputs.LAYER_CASES[case]
    return N, segs, ratios


def _qkv_for_case(golden_dir, case):
    g = np.load(os.path.join(golden_dir, "unit_layer.npz"))
    seed = int(g["seed"])
    N, segs, ratios = unit_inputs.LAYER_CASES[case]
    cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)))
    sd = {k: torch.from_numpy(v).double() for k, v in synth.synth_state_dict(cfg, synth.toy_group_sizes(), seed).items()
          if k.startswith("encoder.layers.0.self_attn.")}
    x = torch.from_numpy(unit_inputs.layer_inputs(seed, N)) * unit_inputs.ATTN_INPUT_SCALE
    p = "encoder.layers.0.self_attn."
    q, k, v = (torch.nn.functional.linear(x, sd[p + f"{n}_proj.weight"], sd[p + f"{n}_proj.bias"]) for n in "qkv")
    return g, N, segs, ratios, torch.cat([q, k, v], dim=-1)      # [B, N, 2304] float64


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_dilated_attention_fwd_mix_vs_reference_golden(ops, golden_dir, case):
    from oracle import modaltune_oracle as O
    g, N, segs, ratios, qkv = _qkv_for_case(golden_dir, case)
    B = qkv.shape[0]
    M = B * N
    qkv16, qkv_eff = _prescale(qkv.half())
    bt = branch_table(N, segs, ratios)
    plan = ops.make_plan(bt, N, B)
    nb = len(bt)
    o_br = torch.zeros(nb, M, 768, dtype=torch.float16, device=DEV)
    lse_br = torch.zeros(nb, M, 16, device=DEV)
    ops.dilated_attn_fwd(_hm(qkv16.to(DEV).view(M, 2304)), plan, o_br, lse_br)
    ln_w = torch.ones(768, device=DEV)
    ln_b = torch.zeros(768, device=DEV)
    y = torch.zeros(M, 768, dtype=torch.float16, device=DEV)
    stats = torch.zeros(M, 2, device=DEV)
    lse_tot = torch.zeros(M, 16, device=DEV)
    ops.dilated_mix_ln_fwd(o_br, lse_br, plan, ln_w, ln_b, y, stats, lse_tot)
    torch.cuda.synchronize()
    # reference on the SAME fp16-rounded q,k,v (isolates kernel error from input rounding)
    q, k, v = (t.view(B, N, 16, 48) for t in qkv_eff.split(768, dim=-1))
    mixed, outs, lses = O.dilated_attention_core(q, k, v, segs, ratios, return_branches=True)
    for i, b in enumerate(bt):
        cov = lses[i] > -1e7                                  # [B, N, H]
        got_o = o_br[i].view(B, N, 16, 48).double().cpu()
        got_l = lse_br[i].view(B, N, 16).double().cpu()
        assert float(((got_o - outs[i]).abs() * cov.unsqueeze(-1)).max()) < 3e-3 * float(outs[i].abs().max())
        assert float(((got_l - lses[i]).abs() * cov).max()) < 2e-3
    ref_ln = torch.nn.functional.layer_norm(mixed, (768,), None, None, 1e-5)
    assert rel(y.view(B, N, 768), ref_ln) < 4e-3
    # and against the golden produced by the reference's own DilatedAttention on fp64 inputs
    gold = torch.from_numpy(g[f"{case}_attn"]).double()
    gold_ln = torch.nn.functional.layer_norm(gold, (768,), None, None, 1e-5)
    assert rel(y.view(B, N, 768), gold_ln) < 6e-3
    tot = torch.logsumexp(torch.stack(lses, 0), dim=0)
    assert float((lse_tot.view(B, N, 16).double().cpu() - tot).abs().max()) < 2e-3


def test_dilated_attention_deferred_rescale_branch(ops):
    """Force the rarely-taken online-softmax rescale (guide rule 26): logits jump by >> 2^8 at late key tiles,
    for some rows only, so the deferred-rescale branch fires mid-sequence; compare all rows with the fp64 oracle."""
    from oracle import modaltune_oracle as O
    g = rng(77)
    B, N = 1, 1500
    segs, ratios = [1024, 5792, 32768, 185363, 1048576], [1, 2, 4, 8, 16]
    q = torch.randn(B, N, 16, 48, generator=g) * 0.5
    k = torch.randn(B, N, 16, 48, generator=g) * 0.5
    v = torch.randn(B, N, 16, 48, generator=g)
    for pos, key, gain in ((5, 700, 40.0), (1100, 1400, 60.0), (300, 1023, 25.0), (1499, 1300, 80.0)):
        k[0, key] = q[0, pos] * gain          # a huge logit for query `pos` at a late key
    qkv, qkv_eff = _prescale(torch.cat([q.reshape(B, N, 768), k.reshape(B, N, 768), v.reshape(B, N, 768)], -1).half())
    bt = branch_table(N, segs, ratios)
    plan = ops.make_plan(bt, N, B)
    M = B * N
    o_br = torch.zeros(5, M, 768, dtype=torch.float16, device=DEV)
    lse_br = torch.zeros(5, M, 16, device=DEV)
    ops.dilated_attn_fwd(_hm(qkv.to(DEV).view(M, 2304)), plan, o_br, lse_br)
    torch.cuda.synchronize()
    qd, kd, vd = (t.view(B, N, 16, 48) for t in qkv_eff.split(768, dim=-1))
    _, outs, lses = O.dilated_attention_core(qd, kd, vd, segs, ratios, return_branches=True)
    for i in range(5):
        cov = lses[i] > -1e7
        got_o = o_br[i].view(B, N, 16, 48).double().cpu()
        got_l = lse_br[i].view(B, N, 16).double().cpu()
        assert torch.isfinite(got_o).all()
        assert float(((got_o - outs[i]).abs() * cov.unsqueeze(-1)).max()) < 4e-3 * float(outs[i].abs().max())
        assert float((((got_l - lses[i]) / lses[i].abs().clamp(min=1.0)).abs() * cov).max()) < 2e-3


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_dilated_attention_bwd_vs_oracle_autograd(ops, golden_dir, case):
    from oracle import modaltune_oracle as O
    g, N, segs, ratios, qkv = _qkv_for_case(golden_dir, case)
    B = qkv.shape[0]
    M = B * N
    qkv16, qkv_eff = _prescale(qkv.half())
    bt = branch_table(N, segs, ratios)
    plan = ops.make_plan(bt, N, B)
    nb = len(bt)
    gen = rng(5)
    ln_w = (1 + 0.1 * torch.randn(768, generator=gen))
    ln_b = 0.1 * torch.randn(768, generator=gen)
    dy = (torch.randn(B, N, 768, generator=gen) * 0.1).half()
    # oracle: fp64 autograd through dilated attention + inner LN on the fp16-rounded inputs
    qd = qkv_eff.requires_grad_(True)
    q, k, v = (t.view(B, N, 16, 48) for t in qd.split(768, dim=-1))
    mixed = O.dilated_attention_core(q, k, v, segs, ratios)
    yref = torch.nn.functional.layer_norm(mixed, (768,), ln_w.double(), ln_b.double(), 1e-5)
    yref.backward(dy.double())
    # HIP
    qkv_d = _hm(qkv16.to(DEV).view(M, 2304))
    o_br = torch.zeros(nb, M, 768, dtype=torch.float16, device=DEV)
    lse_br = torch.zeros(nb, M, 16, device=DEV)
    ops.dilated_attn_fwd(qkv_d, plan, o_br, lse_br)
    y = torch.zeros(M, 768, dtype=torch.float16, device=DEV)
    stats = torch.zeros(M, 2, device=DEV)
    lse_tot = torch.zeros(M, 16, device=DEV)
    ops.dilated_mix_ln_fwd(o_br, lse_br, plan, ln_w.to(DEV), ln_b.to(DEV), y, stats, lse_tot)
    dmixed = torch.zeros(M, 768, dtype=torch.float16, device=DEV)
    delta = torch.zeros(nb, M, 16, device=DEV)
    ops.dilated_mix_ln_bwd(dy.to(DEV).view(M, 768), o_br, lse_br, lse_tot, plan, ln_w.to(DEV), stats, dmixed, delta)
    dqkv = torch.full((M, 2304), float("nan"), device=DEV, dtype=torch.float16)
    wsb = torch.full((ops.dilated_attn_bwd_workspace_bytes(plan) // 2,), float("nan"), device=DEV, dtype=torch.float16)
    ops.dilated_attn_bwd(qkv_d, dmixed, lse_tot, delta, plan, wsb, dqkv)
    torch.cuda.synchronize()
    assert rel(y.view(B, N, 768), yref) < 4e-3
    got = dqkv.view(B, N, 2304).double().cpu()
    assert torch.isfinite(got).all()
    got[..., :768] *= QK          # the q columns are the gradient w.r.t. q' = QK q
    for name, sl in (("dq", slice(0, 768)), ("dk", slice(768, 1536)), ("dv", slice(1536, 2304))):
        r = rel(got[..., sl], qd.grad[..., sl])
        assert r < 2e-2, (name, r)


@pytest.mark.parametrize("N", [1030, 1100, 1500, 2049])
def test_dilated_attention_padded_segment_tail(ops, N):
    """A short last segment: most of its sparse sequence is zero padding, which the kernels do not compute -- padded
    key tiles enter sum(P) in closed form (forward) and are skipped outright (backward: K = 0 adds nothing to dQ,
    padded keys / queries get no gradient).  Forward per-branch outputs + LSE and backward dq/dk/dv against the fp64
    oracle with the reference's real segment lengths."""
    from oracle import modaltune_oracle as O
    gen = rng(N)
    B = 1
    segs, ratios = [1024, 5792, 32768, 185363, 1048576], [1, 2, 4, 8, 16]
    qkv16, qkv_eff = _prescale((torch.randn(B, N, 2304, generator=gen) * 0.7).half())
    ln_w = (1 + 0.1 * torch.randn(768, generator=gen))
    ln_b = 0.1 * torch.randn(768, generator=gen)
    dy = (torch.randn(B, N, 768, generator=gen) * 0.1).half()
    qd = qkv_eff.requires_grad_(True)
    q, k, v = (t.view(B, N, 16, 48) for t in qd.split(768, dim=-1))
    mixed, outs, lses = O.dilated_attention_core(q, k, v, segs, ratios, return_branches=True)
    yref = torch.nn.functional.layer_norm(mixed, (768,), ln_w.double(), ln_b.double(), 1e-5)
    yref.backward(dy.double())
    bt = branch_table(N, segs, ratios)
    plan = ops.make_plan(bt, N, B)
    M, nb = B * N, len(bt)
    qkv_d = _hm(qkv16.to(DEV).view(M, 2304))
    o_br = torch.zeros(nb, M, 768, dtype=torch.float16, device=DEV)
    lse_br = torch.zeros(nb, M, 16, device=DEV)
    ops.dilated_attn_fwd(qkv_d, plan, o_br, lse_br)
    y = torch.zeros(M, 768, dtype=torch.float16, device=DEV)
    stats = torch.zeros(M, 2, device=DEV)
    lse_tot = torch.zeros(M, 16, device=DEV)
    ops.dilated_mix_ln_fwd(o_br, lse_br, plan, ln_w.to(DEV), ln_b.to(DEV), y, stats, lse_tot)
    dmixed = torch.zeros(M, 768, dtype=torch.float16, device=DEV)
    delta = torch.zeros(nb, M, 16, device=DEV)
    ops.dilated_mix_ln_bwd(dy.to(DEV).view(M, 768), o_br, lse_br, lse_tot, plan, ln_w.to(DEV), stats, dmixed, delta)
    dqkv = torch.full((M, 2304), float("nan"), device=DEV, dtype=torch.float16)
    wsb = torch.full((ops.dilated_attn_bwd_workspace_bytes(plan) // 2,), float("nan"), device=DEV, dtype=torch.float16)
    ops.dilated_attn_bwd(qkv_d, dmixed, lse_tot, delta, plan, wsb, dqkv)
    torch.cuda.synchronize()
    for i in range(nb):
        cov = lses[i].detach() > -1e7
        got_o = o_br[i].view(B, N, 16, 48).double().cpu()
        got_l = lse_br[i].view(B, N, 16).double().cpu()
        assert float(((got_o - outs[i].detach()).abs() * cov.unsqueeze(-1)).max()) < 3e-3 * float(outs[i].detach().abs().max())
        assert float(((got_l - lses[i].detach()).abs() * cov).max()) < 2e-3
    assert rel(y.view(B, N, 768), yref.detach()) < 4e-3
    got = dqkv.view(B, N, 2304).double().cpu()
    assert torch.isfinite(got).all()
    got[..., :768] *= QK          # the q columns are the gradient w.r.t. q' = QK q
    for name, sl in (("dq", slice(0, 768)), ("dk", slice(768, 1536)), ("dv", slice(1536, 2304))):
        r = rel(got[..., sl], qd.grad[..., sl])
        assert r < 2e-2, (name, r)


# ------------------------------------------------------------------------------------------ adapter attention
def _mha_ref(q, k, v, heads):
    B, Lq, E = q.shape
    hd = E // heads
    qh = q.view(B, Lq, heads, hd).transpose(1, 2)
    kh = k.view(B, -1, heads, hd).transpose(1, 2)
    vh = v.view(B, -1, heads, hd).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Lq, E)


@pytest.mark.parametrize("T,L", [(65, 301), (7, 301), (66, 1100), (33, 513)])
def test_inject_attention(ops, T, L):
    g = rng(T)
    B = 3
    q = torch.randn(B, L, 192, generator=g).half()
    k = torch.randn(B, T, 192, generator=g)
    v = torch.randn(B, T, 192, generator=g)
    da = torch.randn(B, L, 192, generator=g).half()
    qd, kd, vd = q.double().requires_grad_(True), k.double().requires_grad_(True), v.double().requires_grad_(True)
    ref = _mha_ref(qd, kd, vd, 12)
    ref.backward(da.double())
    a = torch.zeros(B * L, 192, dtype=torch.float16, device=DEV)
    alse = torch.zeros(B * L, 12, device=DEV)
    ops.inject_attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), a, B * L, L, T, lse=alse)
    dq = torch.zeros(B * L, 192, dtype=torch.float16, device=DEV)
    dk = torch.zeros(B, T, 192, device=DEV)
    dv = torch.zeros(B, T, 192, device=DEV)
    ops.inject_attn_bwd(q.to(DEV), a, alse, da.to(DEV), k.to(DEV), v.to(DEV), dq, dk, dv, B * L, L, T)
    torch.cuda.synchronize()
    assert rel(a.view(B, L, 192), ref) < 2e-3
    assert rel(dq.view(B, L, 192), qd.grad) < 3e-3
    assert rel(dk, kd.grad) < 1e-3 and rel(dv, vd.grad) < 1e-3      # p / ds are staged as fp16 for the token-side reduction


@pytest.mark.parametrize("T,L,nsplit", [(65, 700, 4), (7, 129, 1), (66, 1000, 16)])
def test_extract_attention(ops, T, L, nsplit):
    g = rng(T + L)
    B = 2
    q = torch.randn(B, T, 192, generator=g)
    kv = torch.randn(B, L, 384, generator=g).half()
    dout = torch.randn(B, T, 192, generator=g)
    qd = q.double().requires_grad_(True)
    kvd = kv.double().requires_grad_(True)
    ref = _mha_ref(qd, kvd[..., :192], kvd[..., 192:], 12)
    ref.backward(dout.double())
    out = torch.zeros(B, T, 192, device=DEV)
    lse = torch.zeros(B, T, 12, device=DEV)
    pa = torch.zeros(B * 12 * nsplit * T * 16, device=DEV)
    pm = torch.zeros(B * 12 * nsplit * T * 2, device=DEV)
    ops.extract_attn_fwd(q.to(DEV), kv.to(DEV), out, lse, pa, pm, B, T, L, nsplit)
    dq = torch.zeros(B, T, 192, device=DEV)
    dkv = torch.zeros(B * L, 384, dtype=torch.float16, device=DEV)
    ops.extract_attn_bwd(q.to(DEV), kv.to(DEV), out, lse, dout.to(DEV), dq, dkv, B, T, L)
    torch.cuda.synchronize()
    assert rel(out, ref) < 1.5e-3           # q (scaled) is rounded to fp16 for the MFMA score products, P to fp16 for P.V
    assert rel(dq, qd.grad) < 3e-3          # ds is staged as fp16 for the MFMA reduction over the keys
    assert rel(dkv.view(B, L, 384), kvd.grad) < 3e-3


@pytest.mark.parametrize("T", [65, 7, 64, 127, 128, 1])
def test_token_mha(ops, T):
    g = rng(T + 100)
    B, E = 3, 192
    q, k, v, do = (torch.randn(B, T, E, generator=g) for _ in range(4))
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = _mha_ref(qd, kd, vd, 12)
    ref.backward(do.double())
    out = torch.zeros(B, T, E, device=DEV)
    probs = torch.zeros(B, 12, T, T, device=DEV)
    ops.token_mha_fwd(q.to(DEV), k.to(DEV), v.to(DEV), out, probs, B, T, E, 12)
    dq, dk, dv = (torch.zeros(B, T, E, device=DEV) for _ in range(3))
    ops.token_mha_bwd(q.to(DEV), k.to(DEV), v.to(DEV), probs, do.to(DEV), dq, dk, dv, B, T, E, 12)
    torch.cuda.synchronize()
    # the T x T prompt self-attention runs in fp32 throughout
    assert rel(out, ref) < 1e-4
    assert rel(dq, qd.grad) < 1e-4
    assert rel(dk, kd.grad) < 1e-4
    assert rel(dv, vd.grad) < 1e-4


# ------------------------------------------------------------------------------------------ loss / optimiser / misc
def test_distill_loss_and_adamw(ops):
    from oracle import modaltune_oracle as O
    g = rng(9)
    logits = torch.randn(3, 256, generator=g)
    text = torch.randn(4, 256, generator=g)
    text = text / text.norm(dim=-1, keepdim=True)
    ld = logits.double().requires_grad_(True)
    ref = O.distill_loss(ld, text.double())
    ref.backward()
    loss = torch.zeros(1, device=DEV)
    dl = torch.zeros(3, 256, device=DEV)
    ops.distill_loss(logits.to(DEV), text[[0, 1, 3]].contiguous().to(DEV), loss, dl, 3, 256, loss_scale=4.0)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref.detach())) < 1e-5 * abs(float(ref.detach()))
    assert rel(dl / 4.0, ld.grad) < 1e-4
    # AdamW, two steps, with a loss scale
    n = 10007
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    p, m, v = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    scale = torch.tensor([8.0], device=DEV)
    found = torch.zeros(1, dtype=torch.int32, device=DEV)
    pr, mr, vr = p0.double(), torch.zeros(n).double(), torch.zeros(n).double()
    for step in (1, 2):
        ops.check_finite(gr.to(DEV) * 8.0, n, found)
        ops.adamw_step(p, (gr * 8.0).to(DEV), m, v, n, 5e-6, 0.9, 0.999, 1e-8, 0.01, step, scale, found)
        pr, mr, vr = O.adamw_update(pr, gr.double(), mr, vr, step, 5e-6)
    torch.cuda.synchronize()
    assert int(found) == 0
    assert rel(p, pr) < 1e-6
    # non-finite gradient -> update skipped
    bad = (gr * 8.0).to(DEV).clone()
    bad[5] = float("inf")
    before = p.clone()
    ops.check_finite(bad, n, found)
    ops.adamw_step(p, bad, m, v, n, 5e-6, 0.9, 0.999, 1e-8, 0.01, 3, scale, found)
    torch.cuda.synchronize()
    assert int(found) == 1 and torch.equal(p, before)


def test_elementwise(ops):
    g = rng(21)
    n = 4099
    x = torch.randn(n, generator=g)
    dy = torch.randn(n, generator=g)
    for act, f in ((ops.ACT_RELU, torch.relu), (ops.ACT_GELU, torch.nn.functional.gelu), (ops.ACT_ELU, torch.nn.functional.elu)):
        xd = x.double().requires_grad_(True)
        r = f(xd)
        r.backward(dy.double())
        y = torch.zeros(n, device=DEV)
        dx = torch.zeros(n, device=DEV)
        ops.act_fwd(x.to(DEV), y, act)
        ops.act_bwd(x.to(DEV), dy.to(DEV), dx, act)
        torch.cuda.synchronize()
        assert rel(y, r) < 1e-6 and rel(dx, xd.grad) < 1e-5
    h = torch.zeros(n, dtype=torch.float16, device=DEV)
    ops.cast_f32_to_f16(x.to(DEV), h)
    back = torch.zeros(n, device=DEV)
    ops.cast_f16_to_f32(h, back)
    torch.cuda.synchronize()
    assert torch.equal(h.cpu(), x.half()) and torch.equal(back.cpu(), x.half().float())


# ------------------------------------------------------------------------------------------ grouped pathway networks
@pytest.mark.parametrize("sizes", [[1, 5, 199, 33, 64, 300, 7], [9] * 40, [5] * 70 + [130, 1]])
def test_gene_snn_grouped(ops, sizes):
    """All pathway networks in one launch vs per-pathway torch (gene_encoder.py:97-131): forward and weight grads.
    Fewer than 64 pathways run the kernels that share a pathway among 8 workgroups, more the workgroup-per-pathway ones."""
    g = rng(len(sizes))
    G, Lt = len(sizes), 256
    offs, pieces, cur = [], [], 0

    def slot(t):
        nonlocal cur
        o = cur
        pieces.append(t.reshape(-1))
        pad = (-t.numel()) % 4
        if pad:
            pieces.append(torch.zeros(pad))
        cur += t.numel() + pad
        return o
    W1 = [torch.randn(Lt, n, generator=g) * 0.3 for n in sizes]
    b1 = [torch.randn(Lt, generator=g) * 0.1 for _ in sizes]
    W2 = [torch.randn(Lt, Lt, generator=g) * 0.06 for _ in sizes]
    b2 = [torch.randn(Lt, generator=g) * 0.1 for _ in sizes]
    for i in range(G):
        offs.append([slot(W1[i]), slot(b1[i]), slot(W2[i]), slot(b2[i])])
    flat = torch.cat(pieces)
    genes = [torch.randn(n, generator=g) for n in sizes]
    dz = torch.randn(G, Lt, generator=g)
    goff = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    # reference
    ref_z, params = [], []
    for i in range(G):
        ps = [t.double().requires_grad_(True) for t in (W1[i], b1[i], W2[i], b2[i])]
        params.append(ps)
        h = torch.nn.functional.elu(ps[0] @ genes[i].double() + ps[1])
        ref_z.append(torch.nn.functional.elu(ps[2] @ h + ps[3]))
    ref_z = torch.stack(ref_z)
    ref_z.backward(dz.double())
    d = lambda t: t.to(DEV)
    a1, a2, z = (torch.zeros(G, Lt, device=DEV) for _ in range(3))
    grads = torch.zeros_like(flat).to(DEV)
    t_offs = torch.tensor(offs, dtype=torch.int64, device=DEV)
    t_sizes = torch.tensor(sizes, dtype=torch.int32, device=DEV)
    t_goff = torch.from_numpy(goff).to(DEV)
    gflat = d(torch.cat(genes))
    ops.gene_snn_fwd(d(flat), t_offs, t_sizes, t_goff, gflat, G, Lt, a1, a2, z)
    ops.gene_snn_bwd(d(flat), grads, t_offs, t_sizes, t_goff, gflat, G, Lt, a1, a2, d(dz))
    torch.cuda.synchronize()
    assert rel(z, ref_z) < 1e-5
    gc = grads.cpu()
    for i in range(G):
        for j, t in enumerate(params[i]):
            got = gc[offs[i][j]:offs[i][j] + t.numel()].view(t.shape)
            assert rel(got, t.grad) < 1e-5, (i, j)


def test_gene_encoder_331_pathways_vs_reference_golden(ops, golden_dir):
    """Engine gene encoder with the reference-default grouping against the reference's output."""
    import json
    from modaltune_amd import synth
    from modaltune_amd.config import ModelConfig
    from modaltune_amd.engine import Engine
    sizes = json.load(open(os.path.join(golden_dir, "pathway_sizes_331.json")))
    g = np.load(os.path.join(golden_dir, "unit_gene331.npz"))
    seed = int(g["seed"])
    cfg = ModelConfig(depth=3, interaction_indexes=((0, 0), (1, 1), (2, 2)))
    eng = Engine(cfg, sizes, DEV)
    eng.load_state_dict(synth.synth_state_dict(cfg, sizes, seed))
    genes = [torch.from_numpy(a).to(DEV) for a in synth.synth_inputs(8, sizes, seed)["genes"]]
    eng.tape.reset()
    eng.tape.grad_enabled = False
    y = eng._gene_encoder(genes)             # [1, G64, P = 1, D]: dropout is off, one pass
    torch.cuda.synchronize()
    assert tuple(y.data.shape) == (1, 64, 1, 768)
    assert rel(y.data.view(1, 64, 768), torch.from_numpy(g["y"])) < 1e-4


# ------------------------------------------------------------------------------------------ train-mode stochastic ops
def _rng(seed):
    return torch.Generator().manual_seed(seed)


def test_dropout_masks_statistics_determinism_and_consistency(ops):
    """Counter-based Dropout / DropPath masks: rates and scales, determinism in (seed, step, site), and the SAME mask
    from every kernel that applies a site (GEMM epilogue forward; LayerNorm-backward fp16 copy and cast in backward)."""
    M, D, rpp = 3000, 768, 1000
    rng = torch.tensor([123, 456, 7, 0], dtype=torch.int32, device=DEV)
    p, pp = 0.25, 0.3
    spec = ops.dropout_spec(rng, 40, p, 41, pp, rpp)
    ones = torch.ones(M, D, device=DEV)
    mask = torch.zeros(M, D, device=DEV)
    ops.dropout_f32(ones, mask, M, D, spec)
    torch.cuda.synchronize()
    m = mask.cpu()
    for b in range(M // rpp):
        blk = m[b * rpp:(b + 1) * rpp]
        vals = torch.unique(blk)
        assert len(vals) <= 2 and float(vals.min()) == 0.0
        if float(vals.max()) > 0:       # kept pass: scale 1 / ((1-p)(1-pp)), element keep rate 1 - p
            assert abs(float(vals.max()) - 1.0 / ((1 - p) * (1 - pp))) < 1e-5
            assert abs(float((blk > 0).float().mean()) - (1 - p)) < 5e-3
    mask2 = torch.zeros_like(mask)
    ops.dropout_f32(ones, mask2, M, D, spec)
    assert torch.equal(mask, mask2)                          # same (seed, step, site) -> same mask
    ops.rng_advance(rng)
    ops.dropout_f32(ones, mask2, M, D, spec)
    torch.cuda.synchronize()
    assert int(rng[2]) == 8 and not torch.equal(mask, mask2)     # next step -> new mask
    rng[2] = 7
    other = torch.zeros_like(mask)
    ops.dropout_f32(ones, other, M, D, ops.dropout_spec(rng, 44, p, 45, pp, rpp))
    assert not torch.equal(mask, other)                      # another site -> another mask
    # GEMM epilogue: resid + drop(A W^T + bias)
    gen = _rng(5)
    K = 64
    A = torch.randn(M, K, generator=gen).half().to(DEV)
    W = (torch.randn(D, K, generator=gen) * 0.1).half().to(DEV)
    bias = torch.randn(D, generator=gen).to(DEV)
    resid = torch.randn(M, D, generator=gen).to(DEV)
    plain = torch.zeros(M, D, device=DEV)
    ops.gemm_nt(A, W, plain, M, D, K, bias=bias)
    dropped = torch.zeros(M, D, device=DEV)
    ops.gemm_nt(A, W, dropped, M, D, K, epilogue=ops.EPI_BIAS_RESID, bias=bias, resid=resid, ldr=D, drop=spec)
    torch.cuda.synchronize()
    assert rel(dropped, resid + plain * mask) < 1e-6
    # backward producers of the masked fp16 gradient: cast and the LayerNorm-backward second output
    gsrc = torch.randn(M, D, generator=gen).to(DEV)
    c16 = torch.zeros(M, D, dtype=torch.float16, device=DEV)
    ops.cast_f32_to_f16(gsrc, c16, drop=spec, D=D)
    torch.cuda.synchronize()
    assert torch.equal(c16, (gsrc * mask).half())
    x = torch.randn(M, D, generator=gen).to(DEV)
    w = (1 + 0.1 * torch.randn(D, generator=gen)).to(DEV)
    y = torch.zeros(M, D, dtype=torch.float16, device=DEV)
    st = torch.zeros(M, 2, device=DEV)
    ops.layernorm_fwd(x, w, torch.zeros(D, device=DEV), y, st, M, D)
    dy = torch.randn(M, D, generator=gen).half().to(DEV)
    acc_a, acc_b = torch.zeros(M, D, device=DEV), torch.zeros(M, D, device=DEV)
    d16_plain = torch.zeros(M, D, dtype=torch.float16, device=DEV)
    d16_drop = torch.zeros(M, D, dtype=torch.float16, device=DEV)
    ops.layernorm_bwd(dy, x, w, st, acc_a, M, D, accumulate=True, dx16=d16_plain)
    ops.layernorm_bwd(dy, x, w, st, acc_b, M, D, accumulate=True, dx16=d16_drop, dx16_drop=spec)
    torch.cuda.synchronize()
    assert torch.equal(acc_a, acc_b)                         # the fp32 stream is not masked
    assert torch.equal(d16_drop, (acc_b * mask).half())
    # the residual add riding on the next LayerNorm: h = resid + drop(branch), y = LN(h) -- same mask again
    br16 = torch.randn(M, D, generator=gen).half().to(DEV)
    lb = (0.1 * torch.randn(D, generator=gen)).to(DEV)
    for sp, mk in ((None, torch.ones_like(mask)), (spec, mask)):
        h = torch.zeros(M, D, device=DEV)
        y2 = torch.zeros(M, D, dtype=torch.float16, device=DEV)
        st2 = torch.zeros(M, 2, device=DEV)
        ops.add_layernorm_fwd(resid, br16, w, lb, h, y2, st2, M, D, drop=sp)
        torch.cuda.synchronize()
        href = resid + br16.float() * mk
        assert rel(h, href) < 1e-6
        ref = torch.nn.functional.layer_norm(href.double(), (D,), w.double(), lb.double(), 1e-5)
        assert rel(y2, ref) < 2e-3
        assert rel(st2[:, 0], href.double().mean(-1)) < 1e-5
        assert rel(st2[:, 1], 1.0 / torch.sqrt(href.double().var(-1, unbiased=False) + 1e-5)) < 1e-5


@pytest.mark.parametrize("G", [40, 72])
def test_gene_snn_alpha_dropout(ops, G):
    """nn.AlphaDropout(p) after the pathway ELUs: dropped units take the constant a * alpha' + b, kept ones a * elu + b;
    the backward regenerates the same masks (gradient of a kept unit = a * upstream, of a dropped one = 0).
    (G = 40: the shared-pathway kernels; 72: a workgroup per pathway.)"""
    Lt, n = 256, 9
    gen = _rng(11)
    sizes = [n] * G
    W1 = torch.randn(G, Lt, n, generator=gen) * 0.3
    b1 = torch.randn(G, Lt, generator=gen) * 0.1
    W2 = torch.randn(G, Lt, Lt, generator=gen) * 0.06
    b2 = torch.randn(G, Lt, generator=gen) * 0.1
    pieces, offs, cur = [], [], 0
    for i in range(G):
        row = []
        for tns in (W1[i], b1[i], W2[i], b2[i]):
            row.append(cur)
            pieces.append(tns.reshape(-1))
            pad = (-tns.numel()) % 4
            if pad:
                pieces.append(torch.zeros(pad))
            cur += tns.numel() + pad
        offs.append(row)
    flat = torch.cat(pieces).to(DEV)
    genes = torch.randn(G * n, generator=gen).to(DEV)
    goff = torch.arange(G, dtype=torch.int64, device=DEV) * n
    t_offs = torch.tensor(offs, dtype=torch.int64, device=DEV)
    t_sizes = torch.tensor(sizes, dtype=torch.int32, device=DEV)
    rs = torch.tensor([9, 8, 3, 0], dtype=torch.int32, device=DEV)
    p = 0.25
    spec = ops.dropout_spec(rs, 300, p)
    a1, a2, z, z0 = (torch.zeros(G, Lt, device=DEV) for _ in range(4))
    ops.gene_snn_fwd(flat, t_offs, t_sizes, goff, genes, G, Lt, a1, a2, z0)
    ops.gene_snn_fwd(flat, t_offs, t_sizes, goff, genes, G, Lt, a1, a2, z, alpha_drop=spec)
    torch.cuda.synchronize()
    alpha_p = -1.7580993408473766
    a = ((1 - p) * (1 + p * alpha_p ** 2)) ** -0.5
    b = -a * alpha_p * p
    dropped = (z - (a * alpha_p + b)).abs() < 1e-6
    assert abs(float(dropped.float().mean()) - p) < 0.02
    # kept units of the second layer: a * elu(a2) + b with a2 computed from the (dropped) first layer
    kept = ~dropped
    assert rel(z[kept], a * torch.nn.functional.elu(a2[kept]) + b) < 1e-6
    assert not torch.allclose(z, z0)
    dz = torch.randn(G, Lt, generator=gen).to(DEV)
    grads = torch.zeros_like(flat)
    ops.gene_snn_bwd(flat, grads, t_offs, t_sizes, goff, genes, G, Lt, a1, a2, dz, alpha_drop=spec)
    torch.cuda.synchronize()
    # db2 = dz * a * keep2 * elu'(a2)
    for i in (0, 17, 39):
        db2 = grads[offs[i][3]:offs[i][3] + Lt]
        pre = a2[i]
        want = dz[i] * a * kept[i].float() * torch.where(pre > 0, torch.ones_like(pre), torch.exp(pre))
        assert rel(db2, want) < 1e-5
    # three task passes in one launch (weights streamed once): own masks per pass -- pass 0 draws the single-pass masks of
    # pathway-major index (i * 3 + 0) -- and the weight gradients are the sum over the passes of the single-pass formula
    P3 = 3
    a1p, a2p, zp = torch.zeros(G, Lt, device=DEV), torch.zeros(G, P3, Lt, device=DEV), torch.zeros(G, P3, Lt, device=DEV)
    ops.gene_snn_fwd(flat, t_offs, t_sizes, goff, genes, G, Lt, a1p, a2p, zp, alpha_drop=spec, passes=P3)
    torch.cuda.synchronize()
    assert torch.equal(a1p, a1)
    dropped3 = (zp - (a * alpha_p + b)).abs() < 1e-6
    assert abs(float(dropped3.float().mean()) - p) < 0.02
    assert not torch.equal(dropped3[:, 0], dropped3[:, 1]) and not torch.equal(dropped3[:, 1], dropped3[:, 2])
    kept3 = ~dropped3
    assert rel(zp[kept3], a * torch.nn.functional.elu(a2p[kept3]) + b) < 1e-6
    dzp = torch.randn(G, P3, Lt, generator=gen).to(DEV)
    grads3 = torch.zeros_like(flat)
    ops.gene_snn_bwd(flat, grads3, t_offs, t_sizes, goff, genes, G, Lt, a1p, a2p, dzp, alpha_drop=spec, passes=P3)
    torch.cuda.synchronize()
    for i in (0, 17, 39):
        pre = a2p[i]                                                       # [P, Lt]
        da2 = dzp[i] * a * kept3[i].float() * torch.where(pre > 0, torch.ones_like(pre), torch.exp(pre))
        assert rel(grads3[offs[i][3]:offs[i][3] + Lt], da2.sum(0)) < 1e-5      # db2
        # dW2 = sum_p da2_p (x) h1_p with h1_p = W2^-1-free reconstruction: h1_p solves a2_p = W2 h1_p + b2 -> use the identity
        # dW2 h = sum_p da2_p (h1_p . h) for h = a fixed probe; h1_p . probe is recovered from a2 via W2: skip the inverse and
        # check the contraction with da1 instead: db1 = sum_p (W2^T da2_p) * mask1_p * elu'(a1)
        W2i = flat[offs[i][2]:offs[i][2] + Lt * Lt].view(Lt, Lt)
        dh1 = da2 @ W2i                                                    # [P, Lt]
        db1 = grads3[offs[i][1]:offs[i][1] + Lt]
        # mask1_p * a is the derivative of the first AlphaDropout; recover it from the gradient itself on units where only
        # one pass can contribute is fragile -- compare against a finite-difference of the forward instead
        eps = 1e-2
        fl2 = flat.clone()
        fl2[offs[i][1]:offs[i][1] + Lt] += eps
        zq, a2q, a1q = torch.zeros_like(zp), torch.zeros_like(a2p), torch.zeros_like(a1p)
        ops.gene_snn_fwd(fl2, t_offs, t_sizes, goff, genes, G, Lt, a1q, a2q, zq, alpha_drop=spec, passes=P3)
        fl2[offs[i][1]:offs[i][1] + Lt] -= 2 * eps
        zm = torch.zeros_like(zp)
        ops.gene_snn_fwd(fl2, t_offs, t_sizes, goff, genes, G, Lt, a1q, a2q, zm, alpha_drop=spec, passes=P3)
        torch.cuda.synchronize()
        fd = float(((zq[i] - zm[i]) * dzp[i]).sum()) / (2 * eps)          # d/d(b1 + t) of <z_i, dz_i> along the all-ones direction
        assert abs(fd - float(db1.sum())) < 2e-2 * max(1.0, abs(fd))
