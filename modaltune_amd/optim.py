"""`torch.optim.AdamW`-shaped optimiser for the drop-in modules: the second import a trainer may swap.

The reference builds `torch.optim.AdamW(filter(requires_grad, model.parameters()), lr, weight_decay, betas)` and steps it through
`GradScaler.step` (train_modaltune.py:139-149, 235-238): 244 tensors (1 544 with the real pathways) walked per step by torch's
multi-tensor AdamW after GradScaler's unscale pass and a host read-back of `found_inf`.  The drop-in models keep every trainable
tensor as a view of ONE flat fp32 buffer (engine.ParamStore) and hand autograd gradients that are views of ONE flat buffer in the
same layout, so the whole update is one `mt_adamw_step` launch (csrc/optim.hip) -- unscale, inf check and skip included:

    from modaltune_amd.optim import AdamW          # instead of torch.optim.AdamW; same constructor, same state_dict
    opt = AdamW(params, lr=..., weight_decay=..., betas=...)
    scaler.step(opt)                                # GradScaler hands over `grad_scale` / `found_inf`: no unscale (division) pass and
                                                    # no read-back; its inf check over the gradients (_check_inf_per_device) still runs

It IS a `torch.optim.AdamW` (schedulers, `state_dict()`, `zero_grad()`, param groups behave as before).  Gradients that are not
views of one flat buffer (autograd cloned them, a DDP reducer owns them) are gathered by one multi-tensor copy first; whenever the
fused precondition does not hold at all -- parameters of some other module, a parameter without a gradient, amsgrad / maximize /
per-group hyper-parameters that differ -- the step is torch's own, on the same state.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops


class AdamW(torch.optim.AdamW):
    _step_supports_amp_scaling = True          # GradScaler.step: leave unscaling and the skip decision to step() (grad_scale / found_inf)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False, **kw):
        # torch's own implementation must stay usable on this object's state: no foreach / fused / capturable flavours of it
        kw.pop("foreach", None); kw.pop("fused", None); kw.pop("capturable", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize, foreach=False, **kw)
        self._step_supports_amp_scaling = True
        self._flat = None             # (base pointer, numel, exp_avg, exp_avg_sq, step_dev) once the parameters proved to be one flat buffer
        self._flat_failed = False
        self._host_steps = 0          # fused steps taken so far (the per-parameter `step` entries are brought up to date lazily)
        self._steps_dirty = False
        self.last_step_fused: Optional[bool] = None

    def add_param_group(self, param_group):
        """A new group changes the parameter set the flat views (moments, gathered-gradient views) were built for: rebind at the next step
        (moments already accumulated are carried over through the per-parameter state, see _bind_flat)."""
        if getattr(self, "_flat", None) is not None:
            self._sync_steps()                # (the step counts of the parameters bound so far, before the set changes)
        super().add_param_group(param_group)
        self._flat, self._flat_failed = None, False

    # ------------------------------------------------------------------ the flat view of the parameters
    def _all_params(self) -> List[torch.Tensor]:
        return [p for g in self.param_groups for p in g["params"]]

    def _uniform_groups(self) -> bool:
        g0 = self.param_groups[0]
        keys = ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize")
        return all(all(g[k] == g0[k] for k in keys) for g in self.param_groups) and not g0["amsgrad"] and not g0["maximize"] \
            and not torch.is_tensor(g0["lr"])

    def _bind_flat(self):
        """Do all parameters live in one contiguous fp32 buffer (gaps allowed: ParamStore pads slots to 16 bytes with zeros that
        AdamW leaves at zero)?  If so, build the flat moment buffers and make every parameter's optimiser state a VIEW of them, so
        `state_dict()` / `load_state_dict()` / a later torch step see ordinary per-parameter AdamW state."""
        ps = self._all_params()
        if not ps or any(p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() for p in ps):
            return None
        st = ps[0].untyped_storage()
        base = st.data_ptr()
        if any(p.untyped_storage().data_ptr() != base for p in ps):
            return None
        lo = min(p.data_ptr() for p in ps)
        hi = max(p.data_ptr() + p.numel() * 4 for p in ps)
        n = (hi - lo) // 4
        if (lo - base) % 16 or n <= 0:
            return None
        # everything inside [lo, hi) that is not a parameter must be padding the update may touch: true for ParamStore.flat (its
        # gaps are zero-filled slot padding); refuse anything that is mostly holes (a few tensors of a much larger storage)
        if sum(p.numel() for p in ps) < 0.95 * n:
            return None
        dev = ps[0].device
        m = torch.zeros(n, dtype=torch.float32, device=dev)
        v = torch.zeros(n, dtype=torch.float32, device=dev)
        for p in ps:
            o = (p.data_ptr() - lo) // 4
            s = self.state[p]
            if "exp_avg" in s:          # state loaded / produced by torch steps before the first fused one: carry it over
                m[o:o + p.numel()].copy_(s["exp_avg"].reshape(-1))
                v[o:o + p.numel()].copy_(s["exp_avg_sq"].reshape(-1))
                self._host_steps = max(self._host_steps, int(float(s["step"])))
            s["exp_avg"] = m[o:o + p.numel()].view(p.shape)
            s["exp_avg_sq"] = v[o:o + p.numel()].view(p.shape)
            s.setdefault("step", torch.tensor(0.0, dtype=torch.float32))
        flat_p = torch.empty(0, dtype=torch.float32, device=dev).set_(st, (lo - base) // 4, (n,), (1,))
        step_dev = torch.full((1,), self._host_steps, dtype=torch.int32, device=dev)
        return {"lo": lo, "n": n, "p": flat_p, "m": m, "v": v, "step_dev": step_dev, "gbuf": None, "gviews": None,
                "ids": tuple(id(p) for p in ps)}

    def _gathered_grad(self, fl) -> Optional[torch.Tensor]:
        """Gradients that exist for every parameter but live in their own allocations (autograd clones them when several nodes feed
        one parameter -- the per-task calls of a slide run one by one --, a DDP reducer owns them, ...): ONE multi-tensor copy into a
        flat buffer laid out like the parameters, then the fused step as usual.  None when some parameter has no gradient (torch's
        AdamW leaves such a parameter alone; the flat kernel would still decay it)."""
        ps = self._all_params()
        grads = [p.grad for p in ps]
        if any(g is None or g.dtype != torch.float32 or not g.is_cuda or g.shape != p.shape for g, p in zip(grads, ps)):
            return None
        if fl["gbuf"] is None:
            fl["gbuf"] = torch.zeros(fl["n"], dtype=torch.float32, device=ps[0].device)      # (the padding between slots stays zero)
            fl["gviews"] = [fl["gbuf"][(p.data_ptr() - fl["lo"]) // 4:(p.data_ptr() - fl["lo"]) // 4 + p.numel()].view(p.shape) for p in ps]
        torch._foreach_copy_(fl["gviews"], grads)
        return fl["gbuf"]

    def _flat_grad(self, fl) -> Optional[torch.Tensor]:
        """The gradients as ONE flat tensor laid out like the parameters (what the module bridge hands autograd), else None."""
        ps = self._all_params()
        g0 = ps[0].grad
        if g0 is None or g0.dtype != torch.float32 or not g0.is_cuda:
            return None
        gst = g0.untyped_storage()
        gbase = gst.data_ptr()
        delta = g0.data_ptr() - ps[0].data_ptr()
        for p in ps:
            g = p.grad
            if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() - p.data_ptr() != delta \
                    or g.untyped_storage().data_ptr() != gbase:
                return None
        glo = fl["lo"] + delta
        if glo < gbase or glo + fl["n"] * 4 > gbase + gst.nbytes() or (glo - gbase) % 4:
            return None
        return torch.empty(0, dtype=torch.float32, device=g0.device).set_(gst, (glo - gbase) // 4, (fl["n"],), (1,))

    def _sync_steps(self):
        if self._steps_dirty and self._flat is not None:
            # (skipped steps -- GradScaler found an inf -- do not count: the device counter is the truth)
            k = float(int(self._flat["step_dev"].item()))
            self._host_steps = int(k)
            for p in self._all_params():
                self.state[p]["step"] = torch.tensor(k, dtype=torch.float32)
            self._steps_dirty = False

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat, self._flat_failed, self._steps_dirty = None, False, False      # rebind (the loaded tensors are fresh allocations)
        self._host_steps = 0

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        fl = None
        if not self._flat_failed and self._uniform_groups():
            if self._flat is not None and self._flat["ids"] != tuple(id(p) for p in self._all_params()):
                self._sync_steps()            # param_groups were edited in place: the cached views describe another parameter set
                self._flat = None
            if self._flat is None:
                self._flat = self._bind_flat()
                self._flat_failed = self._flat is None
            fl = self._flat
        g = self._flat_grad(fl) if fl is not None else None
        if g is None and fl is not None:
            g = self._gathered_grad(fl)
        if g is not None:
            grp = self.param_groups[0]
            b1, b2 = grp["betas"]
            # found_inf: GradScaler's 0.0 / 1.0 float, read by the kernel as a word that is zero or not; grad_scale: its fp32 scale
            ops.adamw_step(fl["p"], g, fl["m"], fl["v"], fl["n"], float(grp["lr"]), float(b1), float(b2), float(grp["eps"]),
                           float(grp["weight_decay"]), 0, scale=grad_scale, found_inf=found_inf, step_dev=fl["step_dev"])
            if found_inf is not None:
                fl["step_dev"].add_((found_inf == 0).to(torch.int32).reshape(1))
            else:
                fl["step_dev"].add_(1)
            self._steps_dirty = True
            self.last_step_fused = True
            self._bump_versions()             # (derived fp16 weight caches key on the parameters' version counters)
            return loss
        # torch's own step on the same state (per-tensor): unscale first when GradScaler left that to us
        self.last_step_fused = False
        self._sync_steps()
        if grad_scale is not None or found_inf is not None:
            grads = [p.grad for p in self._all_params() if p.grad is not None]
            if found_inf is not None and bool(found_inf.item()):
                return loss
            if grad_scale is not None and grads:
                torch._foreach_div_(grads, grad_scale.to(grads[0].device))
            try:                              # (the base class must not see the scaler's attributes a second time)
                gs, fi = self.__dict__.pop("grad_scale", None), self.__dict__.pop("found_inf", None)
                super().step()
            finally:
                if gs is not None:
                    self.grad_scale = gs
                if fi is not None:
                    self.found_inf = fi
        else:
            super().step()
        if self._flat is not None:            # (a later fused step continues from the same count)
            self._flat["step_dev"].add_(1)
            self._host_steps += 1
        return loss

    def _bump_versions(self):
        """The update went through a raw pointer: tell autograd / the modules' weight caches that the parameters changed."""
        ps = self._all_params()
        torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])
