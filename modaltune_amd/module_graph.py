"""hipGraph replay under the nn.Module bridge (modaltune_amd.aggregators): the reference trainer's loop -- three model(...) calls, torch
loss, GradScaler.scale(loss).backward() (train_modaltune.py:172-177,225-238) -- with the engine's ~1 000 launches per step issued as TWO
graph replays, one inside the first forward call of a slide (all task passes: the speculative batching of `_forward_one_task`), one inside
the backward node.  What it buys: the trainer's own loop synchronises the host in front of the backward (its `text[[0, 1, 3], :]` index
goes through a pageable host-to-device copy), so an eager backward starts on an idle GPU and its first hundred small launches run at the
host's pace; a replay does not (bench.py `module_api`).

A bag geometry (patch count, pass count, gene count) that has been seen `capture_after` times with gradients enabled is run twice more
through this class WITHOUT capture (same launches on the long-lived workspaces and tapes: the tapes size their zero-initialised gradient
arenas from those two visits, as TrainStep's eager visits do) and captured on the next visit; other geometries, chained calls (a slide's later calls that were not part of the batched pass) and every no-grad call keep the eager
path.  The captured pair works on long-lived workspaces of its own (the engine's slots 8 ..: a TrainStep, an EmbeddingExtractor or a no-grad
forward on the same engine use slots 0 ..), so only ONE replayed forward may be waiting for its backward at a time: a second forward before
that backward runs eagerly on a private workspace.  Dropout / DropPath:
the forward graph advances the engine's device-side Philox state first, so every replay draws fresh masks (the backward graph
regenerates them from the same state).
"""
from __future__ import annotations

import os
import weakref
from collections import OrderedDict
from typing import Optional

import torch

from . import ops
from .tape import Tape

F32 = torch.float32


class _Lease:
    """Held by the autograd node of a forward this class served until its backward has run (or the node has died)."""
    __slots__ = ("done", "ent", "calls", "groups", "split", "__weakref__")

    def __init__(self, ent, calls, groups, split):
        self.done = False
        self.ent, self.calls, self.groups, self.split = ent, calls, groups, split      # calls None: captured (the backward is a replay)


class _Entry:
    __slots__ = ("gf", "gb", "logits", "dl", "split")

    def __init__(self):
        self.gf = self.gb = self.logits = self.dl = None
        self.split = False


class ModuleReplay:
    def __init__(self, module, capture_after: int = 2, cache_size: int = 4):
        self.module = module
        self.capture_after, self.cache_size = int(capture_after), int(cache_size)
        self.enabled = os.environ.get("MT_MODULE_GRAPH", "1") not in ("0", "off")
        self.cache: "OrderedDict[tuple, _Entry]" = OrderedDict()
        self.visits = {}
        self.gen = -1
        self.pool = None
        self.stream = None
        self.static_key = None
        self.sgenes = self.sonehots = self.sclin = None
        self.lease = None                 # weakref to the _Lease of the replayed forward whose backward is still to come
        self.res = None                   # streams / gradient sets / tapes of the two pass groups
        self.replays = self.captures = self.eager_fallbacks = self.primed = 0
        self.ncap = {}                    # captures per geometry: one that had to be captured AGAIN was evicted in between -> the cache grows

    # ---------------------------------------------------------------- resources
    def _busy(self) -> bool:
        l = self.lease() if self.lease is not None else None
        return l is not None and not l.done

    def _resources(self, split: bool = False):
        eng = self.module.engine
        if self.res is None:
            tapes = []
            for _ in range(5):                         # [0]: the batched pass, [1] ..: the pass groups
                t = Tape(eng.device)
                t.on_realloc = eng._bump_generation    # captured graphs point into the tapes' gradient arenas
                tapes.append(t)
            self.res = {"tapes": tapes}
        if split and "streams" not in self.res:
            sp = self.module._split_state(3 if os.environ.get("MT_MODULE_GROUPS") == "singles" else 2)            # the module's own two streams and gradient sets (shared with its eager split path)
            self.res.update(streams=sp["streams"], sets=sp["sets"])
        return self.res

    @staticmethod
    def _groups(B: int):
        if os.environ.get("MT_MODULE_GROUPS") == "singles":       # experiments: every task pass a group of its own (balanced: they meet at the loss)
            return [(i, i + 1) for i in range(B)]
        return [(0, B - B // 3), (B - B // 3, B)]

    SLOT0 = 8         # workspace slots of this class: its own, so that a TrainStep / EmbeddingExtractor / no-grad forward on the same engine
                      # (slots 0 ..) between a replayed forward and its backward cannot overwrite the saved activations

    def _slots(self, groups):
        return [self.SLOT0 + sum(1 for (a2, b2) in groups[:gi] if b2 - a2 == b - a) for gi, (a, b) in enumerate(groups)]

    # ---------------------------------------------------------------- forward
    def forward(self, x, coords, genes, onehots, clinical, token) -> Optional[tuple]:
        """Returns (logits [B, O] -- a fresh tensor --, entry, lease) when this call was served by a replay (or by the capture that
        makes the next ones replays), None when the eager path has to run it."""
        m = self.module
        eng = m.engine
        if not self.enabled or token is not None or eng.collect_taps or hasattr(eng, "forward_slide"):
            return None
        x2 = x.reshape(-1, x.shape[-1])
        L, B = int(x2.shape[0]), int(onehots.shape[0])
        if m.is_multi and B != m.cfg.multi_task:
            return None                    # (a single task's call: its siblings chain behind it -- eager)
        if torch.cuda.is_current_stream_capturing():
            return None
        gl = [genes] if torch.is_tensor(genes) else list(genes)
        ngenes = int(sum(g.numel() for g in gl))
        split = bool(B >= 3 and L >= m.split_min_patches and m.split_passes)
        key = (L, B, ngenes, split, bool(eng.stochastic), None if clinical is None else tuple(clinical.shape))
        ent = self.cache.get(key)
        seen = self.visits.get(key, 0)
        if (ent is None or ent.gf is None) and seen < self.capture_after:
            if len(self.visits) > 4096:
                self.visits.clear()
            self.visits[key] = seen + 1
            return None
        if self._busy():
            self.eager_fallbacks += 1
            return None
        if not eng._caches_ready:
            eng._build_caches()
        skey = (ngenes, B, int(onehots.shape[1]), None if clinical is None else int(clinical.numel()))
        if self.static_key != skey:
            self.static_key = skey
            dev = eng.device
            self.sgenes = torch.empty(ngenes, dtype=F32, device=dev)
            self.sonehots = torch.empty(B, int(onehots.shape[1]), dtype=F32, device=dev)
            self.sclin = torch.empty(1, int(clinical.numel()), dtype=F32, device=dev) if clinical is not None else None
            self.cache.clear()
        groups = self._groups(B) if split else [(0, B)]
        slots = self._slots(groups)
        for (a, b), sl in list(zip(groups, slots))[1:]:
            eng._workspace(b - a, L, slot=sl)                       # (all workspaces exist -- and have grown -- before anything is captured)
        eng.stage_inputs(x2, coords, eng._workspace(groups[0][1] - groups[0][0], L, slot=slots[0]))      # (may grow the workspace: bumps eng.generation)
        self.sgenes.copy_(torch.cat([g.reshape(-1) for g in gl]).to(F32) if len(gl) > 1 else gl[0].reshape(-1), non_blocking=True)
        self.sonehots.copy_(onehots, non_blocking=True)
        if self.sclin is not None:
            self.sclin.copy_(clinical.reshape(1, -1), non_blocking=True)
        if self.gen != eng.generation:            # buffers the old captures point to are gone
            for e in self.cache.values():
                e.gf = e.gb = None
            self.gen = eng.generation
        ent = self.cache.get(key)
        if ent is None:
            ent = _Entry()
            self.cache[key] = ent
            while len(self.cache) > max(1, self.cache_size):
                self.cache.popitem(last=False)
        else:
            self.cache.move_to_end(key)
        self._buffers(ent, B)
        calls = None
        if ent.gf is None and seen < self.capture_after + 2:
            # priming visit: the same launches, eagerly (the tapes' gradient arenas are sized from these two visits)
            self.visits[key] = seen + 1
            hook, eng.grad_ready_hook = eng.grad_ready_hook, None
            try:
                calls = self._forward_body(ent, L, B, groups, slots, split)
            finally:
                eng.grad_ready_hook = hook
            self.primed += 1
        else:
            if ent.gf is None:
                self.ncap[key] = self.ncap.get(key, 0) + 1
                if self.ncap[key] >= 2 and self.cache_size < 16 and len(self.cache) >= self.cache_size:
                    self.cache_size *= 2      # more recurring geometries than entries (a rotation would recapture on every visit)
                self._capture(ent, L, B, groups, slots, split)
                if self.gen != eng.generation:    # the capture itself moved a buffer: these graphs are stale -- prime and capture again
                    ent.gf = ent.gb = None
                    self.gen = eng.generation
                    self.visits[key] = self.capture_after
                    return None
            ent.gf.replay()
            self.replays += 1
        lease = _Lease(ent, calls, groups, split)
        self.lease = weakref.ref(lease)
        return ent.logits.clone(), lease

    def _forward_body(self, ent, L, B, groups, slots, split):
        m = self.module
        eng = m.engine
        res = self._resources(split)
        if eng.stochastic:
            ops.rng_advance(eng.rng)
        if not split:
            logits = eng.forward(None, None, self.sgenes, self.sonehots, need_grad=True, staged=True, geometry=(B, L), clinical=self.sclin,
                                 tape=res["tapes"][0], site_group=1, ws_slot=slots[0])
            ent.logits.copy_(logits)
            return [eng.last_call]
        ws0 = eng._workspace(groups[0][1] - groups[0][0], L, slot=slots[0])
        eng._embed_patches(None, None, ws0, True, L)            # task-independent: once, in front of the fork
        share = {"x0": ws0["x0"]}
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        calls = []
        for gi, ((a, b), st, gset) in enumerate(zip(groups, res["streams"], res["sets"])):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                old = eng.store.use_grad_set(*gset)
                try:
                    logits = eng.forward(None, None, self.sgenes, self.sonehots[a:b], need_grad=True, staged=True, geometry=(b - a, L),
                                         clinical=self.sclin, share=share, tape=res["tapes"][1 + gi], site_group=gi + 1, ws_slot=slots[gi])
                    calls.append(eng.last_call)
                    ent.logits[a:b].copy_(logits)
                finally:
                    eng.store.use_grad_set(*old)
        for st in res["streams"]:
            main.wait_stream(st)
        return calls

    def _backward_body(self, ent, calls, groups, split):
        eng = self.module.engine
        store = eng.store
        if not split:
            eng.backward(ent.dl, call=calls[0])
            return
        res = self._resources(True)
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        for (a, b), call, st, (gflat, _) in zip(groups, calls, res["streams"], res["sets"]):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                if gflat is not store.flat_grad:
                    gflat.zero_()
                eng.backward(ent.dl[a:b], call=call)
        for st in res["streams"]:
            main.wait_stream(st)
        for gflat, _ in res["sets"]:
            if gflat is not store.flat_grad:
                ops.axpy(store.flat_grad, gflat, 1.0, store.flat_grad)

    def _buffers(self, ent, B):
        eng = self.module.engine
        if ent.logits is None or ent.logits.shape[0] != B:
            O = int(eng.store.tensors["final_project.bias"].numel())
            ent.logits = torch.zeros(B, O, dtype=F32, device=eng.device)
            ent.dl = torch.zeros(B, O, dtype=F32, device=eng.device)

    def _capture(self, ent, L, B, groups, slots, split):
        eng = self.module.engine
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=eng.device)
        ent.split = split
        self._resources(split)
        main, side = torch.cuda.current_stream(), self.stream
        side.wait_stream(main)
        if self.pool is None or not any(e.gf is not None for e in self.cache.values()):
            self.pool = torch.cuda.graph_pool_handle()      # (a pool dies with the last graph captured into it)
        hook, eng.grad_ready_hook = eng.grad_ready_hook, None
        try:
            with torch.cuda.stream(side):
                graphs = []
                for body in (lambda: self._forward_body(ent, L, B, groups, slots, split),
                             lambda: self._backward_body(ent, graphs[0][1], groups, split)):
                    g = torch.cuda.CUDAGraph()
                    g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
                    try:
                        out = body()
                    except BaseException:
                        try:
                            g.capture_end()
                        except Exception:
                            pass
                        raise
                    g.capture_end()
                    graphs.append((g, out))
                gf, gb = graphs[0][0], graphs[1][0]
        finally:
            eng.grad_ready_hook = hook
        main.wait_stream(side)
        ent.gf, ent.gb = gf, gb
        self.captures += 1

    # ---------------------------------------------------------------- backward
    def backward(self, lease, scaled_into):
        """`scaled_into(dst)`: writes the scaled fp32 gradient of the logits into the entry's static buffer; then the backward graph (or,
        on a priming visit, the same launches eagerly)."""
        ent = lease.ent
        scaled_into(ent.dl)
        if lease.calls is None:
            ent.gb.replay()
        else:
            self._backward_body(ent, lease.calls, lease.groups, lease.split)
            lease.calls = None
        lease.done = True
