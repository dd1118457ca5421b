"""A minimal reverse-mode tape over the C-ABI kernels for the token-side (T <= 66 rows) fp32 ops.

Every op launches hand-written HIP kernels (modaltune_amd.ops) for both directions and records a
closure that the engine replays in reverse; torch is only the allocator.  Gradients accumulate
(+=) into `Var.grad`; parameter gradients accumulate straight into the flat gradient buffer views.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

from . import ops


_ACTIVE: Optional["Tape"] = None      # the tape whose forward is being recorded (set by Tape.reset)


class _Marker:
    __slots__ = ("tag", "fn")

    def __init__(self, tag, fn):
        self.tag, self.fn = tag, fn

    def __call__(self):
        if self.fn is not None:
            self.fn()


class Var:
    """A contiguous fp32 activation [..., C] with an optional (lazily zero-allocated) gradient."""
    __slots__ = ("data", "grad", "needs_grad", "_tape")

    def __init__(self, data: torch.Tensor, needs_grad: bool = True):
        self.data = data
        self.grad: Optional[torch.Tensor] = None
        self.needs_grad = needs_grad
        self._tape = _ACTIVE

    def g(self) -> torch.Tensor:
        if self.grad is None:
            self.grad = self._tape.zeros_like(self.data) if self._tape is not None else torch.zeros_like(self.data)
        return self.grad

    def acc(self, d: torch.Tensor):
        """grad += d, where d is a freshly produced buffer nobody else holds: the first contribution simply BECOMES
        the gradient (no zero fill, no add launch)."""
        if not self.needs_grad:
            return
        if self.grad is None:
            self.grad = d
        else:
            ops.axpy(self.grad, d, 1.0, self.grad)

    @property
    def rows(self) -> int:
        return self.data.numel() // self.data.shape[-1]

    @property
    def cols(self) -> int:
        return self.data.shape[-1]


class Param:
    """A trainable fp32 tensor + the view of the flat gradient buffer that belongs to it."""
    __slots__ = ("data", "grad")

    def __init__(self, data: torch.Tensor, grad: Optional[torch.Tensor]):
        self.data = data
        self.grad = grad          # None for frozen tensors (selective backward: no dW is ever computed)


class Tape:
    def __init__(self, device, shared: Optional[dict] = None):
        """shared (Engine: the per-call tapes of the module API): {"need": elements the last such tape asked for, "pool": [free
        arenas]} -- a short-lived tape takes its zero-initialised gradient arena from there (sized by its predecessors' demand)
        and hands it back when it dies, so the ~110 gradient buffers of a call cost ONE fill instead of one each."""
        self.device = device
        self.shared = shared
        self.back: List[Callable[[], None]] = []
        if shared is not None:
            if "ones" not in shared:
                shared["ones"] = torch.ones(4096, device=device)
            self._ones = shared["ones"]
        else:
            self._ones = torch.ones(4096, device=device)
        self.grad_enabled = True
        self.lease = None         # (Engine: keeps a recycled per-call workspace alive as long as this tape's closures may read it)
        # Zero-initialised gradient storage: one flat buffer cleared with ONE fill per step instead of a fill launch per
        # gradient (~150 per step).  It is sized from the demand of the previous step (the first step falls back to
        # individual fills); a captured step therefore must be preceded by two eager ones, which the trainer does.
        self._arena: Optional[torch.Tensor] = None
        self._arena_used = 0
        self._arena_miss = 0
        self.on_realloc: Optional[Callable[[], None]] = None     # owner hook: captured graphs point into the arena

    def reset(self):
        global _ACTIVE
        _ACTIVE = self
        self.back.clear()
        need = self._arena_used + self._arena_miss
        if self.shared is not None and self._arena is None and self.grad_enabled:
            need = max(need, int(self.shared.get("need", 0)))
            pool = self.shared["pool"]
            fit = [i for i, a in enumerate(pool) if a.numel() >= need]
            if need > 0 and fit:
                self._arena = pool.pop(min(fit, key=lambda i: pool[i].numel()))
        if need > 0 and (self._arena is None or need > self._arena.numel()):
            self._arena = torch.empty(need + need // 4 + 1024, device=self.device, dtype=torch.float32)
            if self.on_realloc is not None:
                self.on_realloc()
        self._arena_used = self._arena_miss = 0
        if self._arena is not None and self.grad_enabled:      # forward-only passes allocate no gradients
            self._arena.zero_()

    def __del__(self):
        sh = getattr(self, "shared", None)
        try:
            if sh is not None:
                sh["need"] = max(int(0.9 * sh.get("need", 0)), self._arena_used + self._arena_miss)
                if self._arena is not None and len(sh["pool"]) < 6:
                    sh["pool"].append(self._arena)
        except Exception:       # interpreter shutdown
            pass

    def zeros_like(self, t: torch.Tensor) -> torch.Tensor:
        n = t.numel()
        n_al = (n + 3) // 4 * 4                      # keep every gradient 16-byte aligned
        if self._arena is not None and self._arena_used + n_al <= self._arena.numel():
            v = self._arena[self._arena_used:self._arena_used + n].view(t.shape)
            self._arena_used += n_al
            return v
        self._arena_miss += n_al
        return torch.zeros_like(t)

    def record(self, fn: Callable[[], None]):
        if self.grad_enabled:
            self.back.append(fn)

    def record_marker(self, tag, fn: Optional[Callable[[], None]] = None):
        """A stage boundary of the backward: run_backward() treats it as an ordinary closure (calls fn), run_backward_stage() stops
        there and hands `tag` back (the engine marks its interaction blocks: everything recorded after the marker belongs to block
        `tag` and the blocks above, so when the backward reaches it their parameter gradients are final)."""
        if self.grad_enabled:
            self.back.append(_Marker(tag, fn))

    def run_backward(self):
        for fn in reversed(self.back):
            fn()
        self.back.clear()

    def run_backward_stage(self):
        """Runs the recorded closures in reverse up to (and consuming) the next marker; returns its tag, or None once the tape is
        exhausted.  The marker's own fn is NOT called: the caller owns what happens at the boundary."""
        back = self.back
        while back:
            fn = back.pop()
            if isinstance(fn, _Marker):
                return fn.tag
            fn()
        return None

    def new(self, *shape) -> torch.Tensor:
        t = torch.empty(*shape, device=self.device, dtype=torch.float32)
        if ops.RECORD is not None:      # (bench.py replays the recorded token-side launches later: their buffers must stay)
            ops.RECORD_KEEP.append(t)
        return t

    # ------------------------------------------------------------------ ops
    def _take_residual_grad(self, resid: Optional[Var], y: Var, scale: float = 1.0):
        """d resid += scale * dy of y = scale * resid + f(.): dy is dead after the closure that calls this, so with scale 1 the
        first contribution simply becomes resid's gradient (the launches that read dy are already queued ahead of any later
        writer); any other scale costs one launch either way."""
        if resid is None or not resid.needs_grad:
            return
        if scale == 1.0:
            resid.acc(y.grad)
        elif resid.grad is None:
            resid.grad = self.new(*y.grad.shape)
            ops.axpy(y.grad, y.grad, scale - 1.0, resid.grad)
        else:
            ops.axpy(resid.grad, y.grad, scale, resid.grad)

    def linear(self, x: Var, W: Param, b: Optional[Param], act: int = ops.ACT_NONE, drop=None, resid: Optional[Var] = None,
               resid_scale: float = 1.0) -> Var:
        """y = [resid_scale * resid +] Dropout(act(x @ W^T + b)) over the last dim: nn.Linear with the activation, the nn.Dropout
        (and / or the DropPath of the branch: `drop` may carry path_p) that follows it and the residual add on the GEMM's epilogue
        (one launch); the backward applies the mask and act' to dy as the dX and dW products load it, and both products share a
        launch (mt_sgemm_multi)."""
        return self.linear_group([(x, W, b)], act=act, drop=drop, resid=resid, resid_scale=resid_scale)[0]

    def linear_group(self, items, act: int = ops.ACT_NONE, drop=None, resid: Optional[Var] = None, resid_scale: float = 1.0) -> List[Var]:
        """Independent nn.Linear modules [(x, W, b), ...] (sibling projections: q | k | v of one normed input) whose forward
        products share launches; each keeps its own backward closure."""
        probs, outs = [], []
        for x, W, b in items:
            K, N, R = x.cols, W.data.shape[0], x.rows
            y = Var(self.new(*x.data.shape[:-1], N))
            pre = self.new(*y.data.shape) if act != ops.ACT_NONE else None
            probs.append(ops.sgemm_problem(x.data, (K, 1), W.data, (K, 1), y.data, (N, 1), R, N, K, bias=None if b is None else b.data,
                                           act=act, pre_out=pre, c_drop=drop, resid=None if resid is None else resid.data,
                                           resid_scale=resid_scale))
            self.record(self._linear_bwd(x, W, b, y, pre, act, drop, resid, resid_scale))
            outs.append(y)
        ops.sgemm_multi(probs)
        return outs

    def _linear_bwd(self, x: Var, W: Param, b: Optional[Param], y: Var, pre, act: int, drop, resid: Optional[Var], resid_scale: float = 1.0):
        K, N, R = x.cols, W.data.shape[0], x.rows

        def bwd():
            if y.grad is None:
                return
            dy = y.grad
            fuse = dict(a_aux=pre, a_act=act, a_drop=drop, a_ld=N)       # dpre = mask(dy) * act'(pre), formed at the operand load
            probs = []
            if x.needs_grad:   # dx += dpre @ W
                probs.append(ops.sgemm_problem(dy, (N, 1), W.data, (1, K), x.g(), (K, 1), R, K, N, accumulate=True, **fuse))
            want_db = b is not None and b.grad is not None
            if W.grad is not None:   # dW += dpre^T @ x ; the bias gradient (row sums of dpre^T) rides on the same product
                probs.append(ops.sgemm_problem(dy, (1, N), x.data, (1, K), W.grad, (K, 1), N, K, R, accumulate=True,
                                               rowsum=b.grad if want_db else None, **fuse))
            elif want_db:
                assert R <= self._ones.numel()
                probs.append(ops.sgemm_problem(dy, (1, N), self._ones, (0, 1), b.grad, (1, 1), N, 1, R, accumulate=True, **fuse))
            if probs:
                ops.sgemm_multi(probs)
            self._take_residual_grad(resid, y, resid_scale)
        return bwd

    def dropout(self, x: Var, spec) -> Var:
        """y = Dropout(x) with a counter-based mask (ops.dropout_spec); spec None = identity.  x dense [..., D], D % 4 == 0."""
        if spec is None:
            return x
        D, R = x.cols, x.rows
        y = Var(self.new(*x.data.shape))
        ops.dropout_f32(x.data, y.data, R, D, spec)

        def bwd():
            if y.grad is None or not x.needs_grad:
                return
            tmp = self.new(*x.data.shape)
            ops.dropout_f32(y.grad, tmp, R, D, spec)          # the same mask, regenerated
            x.acc(tmp)
        self.record(bwd)
        return y

    def axis_linear(self, x: Var, W: Param, b: Param, act: int = ops.ACT_NONE, drop=None, resid: Optional[Var] = None) -> Var:
        """y[b, go, c] = [resid +] Dropout(act(sum_g W[go, g] x[b, g, c] + bias[go])): Conv1d(kernel 1) over the group axis
        (gene_encoder.py:140-158) and pathway_compression (gene_encoder.py:212); fused like `linear`."""
        Bb, G = x.data.shape[0], x.data.shape[1]
        assert Bb == 1, "one slide per call; task passes with their own dropout masks ride in the trailing dims [1, G, P, C]"
        Cc = x.data.numel() // G          # everything behind the group axis is 'channels' to the kernel-1 convolution
        Go = W.data.shape[0]
        Wm = W.data.view(Go, G)
        y = Var(self.new(Bb, Go, *x.data.shape[2:]))
        pre = self.new(*y.data.shape) if act != ops.ACT_NONE else None
        ops.sgemm(Wm, (G, 1), x.data, (1, Cc), y.data, (Cc, 1), Go, Cc, G, bias=b.data, bias_on_m=True, act=act, pre_out=pre,
                  c_drop=drop, resid=None if resid is None else resid.data)

        def bwd():
            if y.grad is None:
                return
            dy = y.grad.view(Go, Cc)
            fuse = dict(a_aux=pre, a_act=act, a_drop=drop)
            probs = []
            if x.needs_grad:   # dx[g, c] += sum_go dpre[go, c] W[go, g], written as (dpre^T W)[c, g] so that dy is the A operand
                probs.append(ops.sgemm_problem(dy, (1, Cc), Wm, (1, G), x.g(), (1, Cc), Cc, G, Go, accumulate=True, **fuse))
            if W.grad is not None:   # dW[go, g] += sum_c dpre[go, c] x[g, c];  db[go] += sum_c dpre[go, c] on the same product
                probs.append(ops.sgemm_problem(dy, (Cc, 1), x.data.view(G, Cc), (Cc, 1), W.grad.view(Go, G), (G, 1), Go, G, Cc,
                                               accumulate=True, rowsum=b.grad, **fuse))
            elif b.grad is not None:
                probs.append(ops.sgemm_problem(dy, (Cc, 1), self._ones, (0, 1), b.grad, (1, 1), Go, 1, Cc, accumulate=True, **fuse))
            if probs:
                ops.sgemm_multi(probs)
            self._take_residual_grad(resid, y)
        self.record(bwd)
        return y

    def layernorm(self, x: Var, w: Param, b: Param, add_rows: Optional[Param] = None) -> Var:
        """y = LN(x) * w + b (+ add_rows[row % period])."""
        D, R = x.cols, x.rows
        y = Var(self.new(*x.data.shape))
        stats = self.new(R, 2)
        period = add_rows.data.shape[0] if add_rows is not None else 0
        ops.layernorm_fwd(x.data, w.data, b.data, y.data, stats, R, D,
                          add_rows=None if add_rows is None else add_rows.data, add_period=period)

        def bwd():
            if y.grad is None:
                return
            if add_rows is not None and add_rows.grad is not None:   # d pe[t] += sum over the batch of dy
                reps, per = R // period, period * D
                ops.fold_rows(y.grad, reps, per, add_rows.grad)
            train = w.grad is not None
            dx = x.g() if x.needs_grad else self.new(*x.data.shape)
            ops.layernorm_bwd(y.grad, x.data, w.data, stats, dx, R, D, accumulate=x.needs_grad,
                              dw=w.grad if train else None, db=b.grad if train else None)
        self.record(bwd)
        return y

    def add(self, a: Var, b: Var) -> Var:
        y = Var(self.new(*a.data.shape))
        ops.axpy(a.data, b.data, 1.0, y.data)

        def bwd():
            if y.grad is None:
                return
            donated = a is b     # y.grad is dead after this closure: the first operand without a gradient takes it over
            for v in (a, b):
                if not v.needs_grad:
                    continue
                if v.grad is None and not donated:
                    v.grad = y.grad
                    donated = True
                else:
                    g = v.g()
                    ops.axpy(g, y.grad, 1.0, g)
        self.record(bwd)
        return y

    def add_rows_param(self, a: Var, p: Param) -> Var:
        """y[b, t, :] = a[b, t, :] + p[t, :]  (with_pos_embed, adapter_modules.py:64-65)."""
        Bb = a.data.shape[0]
        per = p.data.numel()
        y = Var(self.new(*a.data.shape))
        ops.axpy_bcast(a.data, p.data, 1.0, y.data, per)

        def bwd():
            if y.grad is None:
                return
            if p.grad is not None:       # dp[t, :] += sum_b dy[b, t, :]
                ops.fold_rows(y.grad, Bb, per, p.grad)
            if a.needs_grad:
                if a.grad is None:
                    a.grad = y.grad      # y.grad is dead after this closure
                else:
                    ops.axpy(a.grad, y.grad, 1.0, a.grad)
        self.record(bwd)
        return y

    def token_mha(self, q: Var, k: Var, v: Var, heads: int) -> Var:
        Bb, T, E = q.data.shape
        out = Var(self.new(Bb, T, E))
        probs = self.new(Bb, heads, T, T)
        ops.token_mha_fwd(q.data, k.data, v.data, out.data, probs, Bb, T, E, heads)

        def bwd():
            if out.grad is None:
                return
            dq, dk, dv = self.new(Bb, T, E), self.new(Bb, T, E), self.new(Bb, T, E)
            ops.token_mha_bwd(q.data, k.data, v.data, probs, out.grad, dq, dk, dv, Bb, T, E, heads)
            for var, d in ((q, dq), (k, dk), (v, dv)):
                var.acc(d)
        self.record(bwd)
        return out
