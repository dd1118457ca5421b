"""One slide train step exactly as train_modaltune.py:195-240 does it, on the HIP engine:
frozen text projector -> 3 task passes (batched) -> KL distillation loss -> backward -> GradScaler-style
unscale/skip -> data-parallel mean of the flat gradient buffer -> fused AdamW.

Data parallelism follows the reference's intent (base_trainer.py:192-211: one process per GPU, DDP mean of the
trainable gradients): slides shard over ranks, the only collective is one all-reduce of the flat fp32 gradient
buffer per step (RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import dp, ops
from .engine import Engine, F32
from .tape import Param, Var


class TrainStep:
    def __init__(self, engine: Engine, lr: float = 1e-4 / 20, weight_decay: float = 0.01, betas=(0.9, 0.999),
                 eps: float = 1e-8, init_scale: float = 2.0 ** 15, growth_interval: int = 2000,
                 process_group=None, task_ids: Sequence[int] = (0, 1, 2), text_rows: Sequence[int] = (0, 1, 3)):
        self.engine, self.dev = engine, engine.device
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        n = engine.store.n_flat
        self.m = torch.zeros(n, dtype=F32, device=self.dev)
        self.v = torch.zeros(n, dtype=F32, device=self.dev)
        self.scale = torch.full((1,), float(init_scale), dtype=F32, device=self.dev)
        self.tracker = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.found_inf = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.growth_interval = growth_interval
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)   # completed optimiser steps (skips excluded)
        self.pg = process_group
        self.task_ids, self.text_rows = list(task_ids), list(text_rows)
        nt = engine.cfg.multi_task
        self.onehots = torch.eye(nt, dtype=F32, device=self.dev)[self.task_ids].contiguous()
        self.loss = torch.zeros(1, dtype=F32, device=self.dev)
        self.proj: Optional[Dict[str, torch.Tensor]] = None
        self.last_logits: Optional[torch.Tensor] = None

    # frozen random text projector (train_modaltune.py:44-59,114-116)
    def set_projector(self, state: Dict[str, "np.ndarray | torch.Tensor"]):
        self.proj = {k: (torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v).to(self.dev, F32).contiguous()
                     for k, v in state.items()}

    def project_text(self, text: torch.Tensor) -> torch.Tensor:
        """Projection_layer + row L2 normalisation (TM:211-213); returns the 3 target rows [3, O] (unnormalised
        softmax inputs are the L2-normalised projections, as in the reference)."""
        p, tape = self.proj, self.engine.tape
        was = tape.grad_enabled
        tape.grad_enabled = False
        fr = lambda t: Param(t, None)
        x = Var(text.to(self.dev, F32).reshape(-1, text.shape[-1]).contiguous(), needs_grad=False)
        O = p["conv1.0.bias"].numel()
        h = tape.linear(x, fr(p["conv1.0.weight"].view(O, -1)), fr(p["conv1.0.bias"]))
        h = tape.layernorm(h, fr(p["conv1.1.weight"].view(-1)), fr(p["conv1.1.bias"].view(-1)))
        r = Var(tape.new(*h.data.shape))
        ops.act_fwd(h.data, r.data, ops.ACT_RELU)
        h = tape.linear(r, fr(p["conv1.3.weight"].view(O, -1)), fr(p["conv1.3.bias"]))
        # row L2 normalisation: y = h / ||h||  -> LayerNorm-free: use sgemm for the norms is overkill; R = 4 rows
        out = tape.new(len(self.text_rows), O)
        ops_l2norm_rows(h.data, out, self.text_rows)
        tape.grad_enabled = was
        return out

    def step(self, x, coords, genes, text, update: bool = True, clinical=None) -> torch.Tensor:
        """One train step on one slide.  Returns the (device) loss scalar; no host sync happens here."""
        eng = self.engine
        if eng.stochastic:
            ops.rng_advance(eng.rng)          # a fresh set of dropout / DropPath masks per step
        target = self.project_text(text)
        eng.store.flat_grad.zero_()
        logits = eng.forward(x, coords, genes, self.onehots, need_grad=True, clinical=clinical)
        self.last_logits = logits
        R, O = logits.shape
        dlogits = torch.empty_like(logits)
        ops.distill_loss(logits, target, self.loss, dlogits, R, O, 1.0, self.scale)
        eng.backward(dlogits)
        if update:
            self.optimizer_step()
        return self.loss

    # ------------------------------------------------------------------ hipGraph replay of the whole step
    def step_graphed(self, x, coords, genes, text, clinical=None) -> torch.Tensor:
        """Same arithmetic as step(), replayed from a captured hipGraph: the ~900 kernel launches of a step are
        recorded once (after two eager warm-up steps) and replayed with one host call; inputs are uploaded into static
        buffers first.  With world_size > 1 the forward+backward graph and the optimiser graph are separate and the
        gradient all-reduce runs between them."""
        eng = self.engine
        x = x.reshape(-1, x.shape[-1])
        L = x.shape[0]
        B = self.onehots.shape[0]
        gflat = genes.reshape(-1) if torch.is_tensor(genes) else torch.cat([g.reshape(-1) for g in genes])
        key = (L, int(gflat.numel()))
        if getattr(self, "_gkey", None) != key:
            self._gkey, self._graphs, self._gwarm = key, None, 0
            self._sgenes = torch.empty(int(gflat.numel()), dtype=F32, device=self.dev)     # one flat static buffer
            self._stext = torch.empty(tuple(text.shape), dtype=F32, device=self.dev)
            self._sclin = torch.empty(1, eng.cfg.clinfeat_dim, dtype=F32, device=self.dev) if eng.cfg.clinical else None
        eng.stage_inputs(x, coords, B=B)
        self._sgenes.copy_(gflat)
        self._stext.copy_(text)
        if self._sclin is not None:
            self._sclin.copy_(clinical.reshape(1, -1))
        world = torch.distributed.get_world_size(self.pg) if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1

        def fwd_bwd():
            if eng.stochastic:
                ops.rng_advance(eng.rng)
            target = self.project_text(self._stext)
            eng.store.flat_grad.zero_()
            logits = eng.forward(None, None, self._sgenes, self.onehots, need_grad=True, staged=True, geometry=(B, L),
                                 clinical=self._sclin)
            self.last_logits = logits
            R, O = logits.shape
            dlogits = torch.empty_like(logits)
            ops.distill_loss(logits, target, self.loss, dlogits, R, O, 1.0, self.scale)
            eng.backward(dlogits)

        if self._graphs is None and self._gwarm < 2:          # eager warm-up (allocator, lazy kernel attributes)
            fwd_bwd()
            self.optimizer_step()
            self._gwarm += 1
            return self.loss
        if self._graphs is None:
            torch.cuda.synchronize()
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            if world == 1:
                with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                    fwd_bwd()
                    self.optimizer_step()
                self._graphs = (g1, None)
            else:       # (thread_local: RCCL's watchdog thread must not invalidate the capture)
                with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                    fwd_bwd()
                with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                    self._adam_and_refresh(world)
                self._graphs = (g1, g2)
        g1, g2 = self._graphs
        g1.replay()
        if g2 is not None:
            dp.allreduce_sum_(eng.store.flat_grad, self.pg)
            g2.replay()
        return self.loss

    def _adam_and_refresh(self, world: int):
        eng = self.engine
        n = eng.store.n_flat
        ops.check_finite(eng.store.flat_grad, n, self.found_inf)
        ops.adamw_step(eng.store.flat, eng.store.flat_grad, self.m, self.v, n, self.lr, self.betas[0], self.betas[1], self.eps,
                       self.wd, 0, scale=self.scale, found_inf=self.found_inf, grad_mult=1.0 / world, step_dev=self.step_dev)
        ops.scaler_update(self.scale, self.tracker, self.found_inf, self.step_dev, 2.0, 0.5, self.growth_interval)
        eng.refresh_trainable_caches()

    def optimizer_step(self):
        world = dp.allreduce_sum_(self.engine.store.flat_grad, self.pg)      # sum over ranks; mean folded into AdamW
        self._adam_and_refresh(world)

    def unscaled_grads(self) -> Dict[str, torch.Tensor]:
        s = float(self.scale)
        return {k: g / s for k, g in self.engine.store.grads.items()}


def ops_l2norm_rows(h: torch.Tensor, out: torch.Tensor, rows: Sequence[int]):
    """out[i] = h[rows[i]] / ||h[rows[i]]|| via the LayerNorm-free path: sgemm computes the squared norm, then axpy scales.
    R <= 4 rows of O = 256: done with two tiny kernels per row (host loop is 3 iterations)."""
    O = h.shape[-1]
    for i, r in enumerate(rows):
        ops.l2norm_row(h[r], out[i], O)
