"""Pins of oracle functions that the reference fixtures do not reach directly."""
import numpy as np
import torch

from oracle import modaltune_oracle as O


def test_adamw_update_is_torch_optim_adamw():
    """oracle.adamw_update restates torch.optim.AdamW (the optimiser the reference builds, train_modaltune.py:145-149:
    lr/20, weight_decay 0.01, default betas / eps): 3 steps on random fp64 tensors, compared with the real thing."""
    g = torch.Generator().manual_seed(3)
    for lr, wd in ((1e-4 / 20, 0.01), (3e-3, 0.1)):
        p0 = torch.randn(257, dtype=torch.float64, generator=g)
        grads = [torch.randn(257, dtype=torch.float64, generator=g) * s for s in (1.0, 1e-3, 10.0)]
        p_t = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([p_t], lr=lr, weight_decay=wd)
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        for step, gk in enumerate(grads, 1):
            p_t.grad = gk.clone()
            opt.step()
            p, m, v = O.adamw_update(p, gk, m, v, step, lr, weight_decay=wd)
            assert float((p - p_t.detach()).abs().max()) < 1e-13 * max(1.0, float(p.abs().max()))
            st = opt.state[p_t]
            assert torch.allclose(m, st["exp_avg"], rtol=1e-13, atol=0) and torch.allclose(v, st["exp_avg_sq"], rtol=1e-13, atol=0)
