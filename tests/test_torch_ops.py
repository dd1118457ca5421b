"""`torch.ops.modaltune_hip.*` (modaltune_amd/torch_ops.py; SURVEY §8b custom-op registration): the namespace registers with
schemas and fake (Meta) kernels without a GPU; on a GPU the ops run the HIP launchers and agree with the C-ABI front end."""
import math

import pytest
import torch


def test_namespace_registers_with_schemas_and_meta_kernels():
    import modaltune_amd.torch_ops as TO
    want = {"gemm_nt", "layernorm_fwd", "layernorm_bwd", "dilated_attention_fwd", "dilated_attention_bwd", "dense_attention_fwd",
            "dense_attention_bwd", "inject_attention_fwd", "extract_attention_fwd", "adamw", "model_forward"}
    assert want <= set(TO.SCHEMAS)
    for name in want:
        assert hasattr(torch.ops.modaltune_hip, name)
    # shape propagation through the Meta kernels (what tracing uses): no device, no launch
    a, w = torch.empty(100, 768, dtype=torch.float16, device="meta"), torch.empty(2304, 768, dtype=torch.float16, device="meta")
    y = torch.ops.modaltune_hip.gemm_nt(a, w, None, True)
    assert y.shape == (100, 2304) and y.dtype == torch.float16
    qkv = torch.empty(3 * 70, 3 * 12 * 64, dtype=torch.float16, device="meta")
    o, lse = torch.ops.modaltune_hip.dense_attention_fwd(qkv, 70, 3, 12, None, None, None)
    assert o.shape == (210, 768) and lse.shape == (210, 12)
    # no CPU kernel: the hot path has no fallback
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.modaltune_hip.gemm_nt(torch.zeros(4, 64, dtype=torch.float16), torch.zeros(8, 64, dtype=torch.float16), None, True)


@pytest.mark.gpu
def test_ops_run_the_hip_launchers():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import modaltune_amd.torch_ops  # noqa: F401
    g = torch.Generator().manual_seed(0)
    a = torch.randn(300, 768, generator=g).half().cuda()
    w = (torch.randn(256, 768, generator=g) * 0.05).half().cuda()
    b = torch.randn(256, generator=g).cuda()
    y = torch.ops.modaltune_hip.gemm_nt(a, w, b, False)
    ref = a.float() @ w.float().T + b
    assert float((y - ref).abs().max() / ref.abs().max()) < 2e-3
    x = torch.randn(300, 768, generator=g).cuda()
    lw, lb = torch.rand(768, generator=g).cuda() + 0.5, torch.randn(768, generator=g).cuda()
    yn, st = torch.ops.modaltune_hip.layernorm_fwd(x, lw, lb, 1e-6)
    refn = torch.nn.functional.layer_norm(x, (768,), lw, lb, 1e-6)
    assert float((yn.float() - refn).abs().max()) < 1e-2
    dx = torch.ops.modaltune_hip.layernorm_bwd(torch.ones_like(yn), x, lw, st)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (768,), lw, lb, 1e-6).sum().backward()
    assert float((dx - xr.grad).abs().max()) < 1e-3 * float(xr.grad.abs().max()) + 1e-5
    N, B, H = 130, 2, 12
    qkv = (torch.randn(B * N, 3 * H * 64, generator=g) * 0.5).half().cuda()
    cells = torch.stack([torch.arange(N - 1) // 12, torch.arange(N - 1) % 12], 1).cuda()
    slopes = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)]).cuda()
    dims = torch.tensor([11, 12]).cuda()
    o, lse = torch.ops.modaltune_hip.dense_attention_fwd(qkv, N, B, H, cells, dims, slopes)
    q, k, v = (t.float().view(B, N, H, 64) for t in qkv.split(H * 64, dim=1))
    s = torch.einsum("bihd,bjhd->bhij", q, k) * math.log(2.0)                      # q is pre-scaled by 64^-1/2 log2(e)
    bias = torch.zeros(H, N, N, device="cuda")
    bias[:, 1:, 1:] = -slopes.view(-1, 1, 1) * torch.cdist(cells.float(), cells.float())
    ref_o = torch.einsum("bhij,bjhd->bihd", torch.softmax(s + bias, -1), v).reshape(B * N, H * 64)
    assert float((o.float() - ref_o).abs().max() / ref_o.abs().max()) < 3e-3
    dqkv = torch.ops.modaltune_hip.dense_attention_bwd(torch.ones_like(o), qkv, o, lse, N, B, H, cells, dims, slopes)
    assert dqkv.shape == qkv.shape and torch.isfinite(dqkv.float()).all()
    p = torch.randn(1000, generator=g).cuda()
    gr = torch.randn(1000, generator=g).cuda()
    p2, m2, v2 = torch.ops.modaltune_hip.adamw(p, gr, torch.zeros_like(p), torch.zeros_like(p), 1e-3, 0.9, 0.999, 1e-8, 0.01, 1)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=0.01)
    pr.grad = gr.clone()
    opt.step()
    assert float((p2 - pr.detach()).abs().max()) < 1e-6
