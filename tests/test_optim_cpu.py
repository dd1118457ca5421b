"""modaltune_amd.optim.AdamW off the GPU: with parameters that are not views of one device buffer it IS torch.optim.AdamW (same
state, same results, GradScaler's hand-over of grad_scale / found_inf honoured by an explicit unscale / skip).  The fused path is
covered on the GPU (tests/test_model_gpu.py::test_fused_adamw_is_torch_adamw_on_the_models_flat_buffers)."""
import torch

from modaltune_amd.optim import AdamW


def _pair(seed=0):
    g = torch.Generator().manual_seed(seed)
    ws = [torch.randn(5, 3, generator=g), torch.randn(7, generator=g)]
    return [torch.nn.Parameter(w.clone()) for w in ws], [torch.nn.Parameter(w.clone()) for w in ws]


def test_foreign_parameters_take_torchs_own_step_on_the_same_state():
    a, b = _pair()
    oa = AdamW([{"params": a, "lr": 1e-2}], weight_decay=0.01, betas=(0.9, 0.99))
    ob = torch.optim.AdamW([{"params": b, "lr": 1e-2}], weight_decay=0.01, betas=(0.9, 0.99), foreach=False)
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        for x, y in zip(a, b):
            gr = torch.randn(x.shape, generator=g)
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        oa.zero_grad(); ob.zero_grad()
    assert oa.last_step_fused is False
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0]["lr"] == sb["param_groups"][0]["lr"] and float(sa["state"][0]["step"]) == 3.0
    ob2 = torch.optim.AdamW([{"params": b, "lr": 1e-2}], weight_decay=0.01, betas=(0.9, 0.99))
    ob2.load_state_dict(sa)                                   # a checkpoint of ours loads into torch's optimiser


def test_grad_scaler_attributes_are_honoured_by_the_fallback():
    """What GradScaler.step leaves on an optimiser that declares _step_supports_amp_scaling: grad_scale (gradients are still scaled)
    and found_inf (skip the step)."""
    a, b = _pair(3)
    oa, ob = AdamW(a, lr=1e-2), torch.optim.AdamW(b, lr=1e-2, foreach=False)
    assert AdamW._step_supports_amp_scaling
    for x, y in zip(a, b):
        gr = torch.randn(x.shape, generator=torch.Generator().manual_seed(5))
        x.grad, y.grad = 8.0 * gr, gr.clone()
    oa.grad_scale, oa.found_inf = torch.tensor(8.0), torch.tensor(0.0)
    oa.step(); ob.step()
    assert all(torch.allclose(x, y, rtol=1e-6, atol=1e-7) for x, y in zip(a, b))
    before = [x.detach().clone() for x in a]
    oa.found_inf = torch.tensor(1.0)
    oa.step()
    assert all(torch.equal(x, y) for x, y in zip(a, before))  # skipped


def test_add_param_group_and_edited_groups_drop_the_cached_flat_binding():
    """the flat views (moments, gathered-gradient views) are built for ONE parameter set; add_param_group or an in-place edit
    of param_groups must make the next step rebind instead of copying into stale views."""
    a, _ = _pair(7)
    oa = AdamW([{"params": a[:1], "lr": 1e-2}])
    oa._flat = {"ids": (id(a[0]),), "step_dev": torch.zeros(1, dtype=torch.int32)}      # as a bound optimiser would hold
    oa._flat_failed = True
    oa.add_param_group({"params": a[1:]})
    assert oa._flat is None and oa._flat_failed is False
    for x in a:
        x.grad = torch.ones_like(x)
    oa.step()                                                   # CPU parameters: torch's own step, on both groups
    assert oa.last_step_fused is False and all(float(oa.state[x]["step"]) == 1.0 for x in a)
