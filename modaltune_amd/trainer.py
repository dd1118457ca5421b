"""One slide train step exactly as train_modaltune.py:195-240 does it, on the HIP engine:
frozen text projector -> 3 task passes (batched) -> KL distillation loss -> backward -> GradScaler-style
unscale/skip -> data-parallel mean of the flat gradient buffer -> fused AdamW.

Data parallelism follows the reference's intent (base_trainer.py:192-211: one process per GPU, DDP mean of the
trainable gradients): slides shard over ranks; the collective is a bucketed SUM all-reduce of the flat fp32 gradient
buffer (RCCL on GPUs, gloo in the rehearsals) whose buckets are launched from inside the backward, as soon as the
stage that owns them has run (dp.grad_buckets: head + interaction block 2, block 1, block 0, gene encoder), and waited
for just before AdamW -- the collectives of the upper blocks run under the backward of the lower ones.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import dp, ops
from ._lib import rowmap
from .engine import Engine, F32
from .tape import Param, Var


class _Captured:
    """The hipGraphs of one bag geometry: `segs` = [(graph, bucket or None)] replayed in order; after a segment with a
    bucket index the reducer starts that bucket (world_size > 1 cuts the backward at the bucket boundaries; on one GPU
    the whole step, optimiser included, is a single segment)."""
    __slots__ = ("segs", "visits", "logits")

    def __init__(self):
        self.segs, self.visits, self.logits = None, 0, None


class TrainStep:
    def __init__(self, engine: Engine, lr: float = 1e-4 / 20, weight_decay: float = 0.01, betas=(0.9, 0.999),
                 eps: float = 1e-8, init_scale: float = 2.0 ** 15, growth_interval: int = 2000,
                 process_group=None, task_ids: Sequence[int] = (0, 1, 2), text_rows: Sequence[int] = (0, 1, 3),
                 graph_cache_size: int = 8, capture_after: int = 2, split_passes="auto"):
        self.engine, self.dev = engine, engine.device
        self.wd, self.betas, self.eps = weight_decay, betas, eps
        n = engine.store.n_flat
        self.m = torch.zeros(n, dtype=F32, device=self.dev)
        self.v = torch.zeros(n, dtype=F32, device=self.dev)
        self.scale = torch.full((1,), float(init_scale), dtype=F32, device=self.dev)
        self.tracker = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.found_inf = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.growth_interval = growth_interval
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.dev)   # completed optimiser steps (skips excluded)
        # the learning rate lives on the device: AdamW reads it there, so a scheduler (the reference steps
        # GradualWarmupScheduler + CosineAnnealingLR every epoch, TM:151-154,242) reaches captured graphs as well
        self.lr_dev = torch.full((1,), float(lr), dtype=F32, device=self.dev)
        self._lr = float(lr)
        self.pg = process_group
        self.task_ids, self.text_rows = list(task_ids), list(text_rows)
        nt = engine.cfg.multi_task
        if engine.cfg.is_multi:
            self.onehots = torch.eye(nt, dtype=F32, device=self.dev)[self.task_ids].contiguous()
        else:       # single-task model (is_multi False, TM:172-179): one pass, no task token; its [1, O] logits meet all 3 targets
            self.onehots = torch.zeros(1, 1, dtype=F32, device=self.dev)
        self.loss = torch.zeros(1, dtype=F32, device=self.dev)
        self.proj: Optional[Dict[str, torch.Tensor]] = None
        self.last_logits: Optional[torch.Tensor] = None
        # data-parallel reducer (buckets in the order the backward finalises them)
        nint = len(engine.cfg.interaction_indexes)
        self._nint = nint
        self.reducer = dp.GradReducer(engine.store.flat_grad, dp.grad_buckets(engine.store.slots, nint, n), process_group,
                                      flat_param=engine.store.flat)
        engine.store.sync = self.reducer.wait_params      # state_dict() / a forward outside the step see complete parameters
        # captured graphs: LRU over bag geometries (real data has a new length almost every slide: a geometry is captured
        # only once it has been seen `capture_after` times, everything else runs the eager schedule)
        self.graph_cache_size, self.capture_after = int(graph_cache_size), int(capture_after)
        self._gcache: "OrderedDict[tuple, _Captured]" = OrderedDict()      # LRU over geometries that HOLD captured graphs
        self._visits: Dict[tuple, int] = {}        # eager visits of not-yet-captured geometries (never evicts a capture)
        self._ggen = -1
        self._opt_graph = None
        self._pool = None                   # graph memory pool shared by all captures (see _capture)
        self._cap_stream = None
        self._cap = None                    # state of a segmented capture in progress
        self._trial_keep: list = []         # captures a schedule trial holds outside the LRU (they keep the shared graph pool alive)
        self._static_key = None
        self.graph_replays = 0
        self.eager_steps = 0
        self.patch_size_lv0 = 1024          # TITAN configuration only (titan_adapter.py:335)
        # bench.py --gpus N: set to a list to collect, per step, HIP events around the two places a collective can be EXPOSED on the
        # compute stream -- ("grad", e0, e1): last backward kernel -> the optimiser may start (bucket all-reduces / the last bucket's
        # reduce-scatter not hidden under the backward); ("param", e0, e1): the next step's wait for the sharded parameter all-gather
        self.comm_events: Optional[list] = None
        # The task passes of a step as TWO concurrent groups (B = 2 and B = 1 on two HIP streams) instead of one batched B = 3 pass: one
        # group's HBM-bound kernels (LayerNorm family, branch mix, combine) run under the other's MFMA-bound ones (GEMMs, attention) and
        # the tails of one group's launches fill with the other's workgroups -- same-box, eager: 42.0 -> 40.4-40.8 ms at L = 10 000
        # (tools/experiments/pass_overlap.py; the same two groups one after the other on ONE stream: 45.0).  The groups share the staged
        # input and the patch embedding (computed once before the fork), own their workspace / tape / dropout masks, and accumulate
        # into their own flat gradient buffers (summed before the optimiser: the token-side dW products are read-modify-write).
        # The loss is a sum over the task rows (TM:225-233), so each group runs loss + backward behind its own forward; the streams
        # meet once, in front of the optimiser.  MT_SPLIT_PASSES=0 / split_passes=False: the batched pass.
        B = int(self.onehots.shape[0])
        self.split_passes = bool(split_passes) and B >= 2 and os.environ.get("MT_SPLIT_PASSES", "1") not in ("0", "off")
        self._groups = [(0, B - B // 3 if B >= 3 else 1), (B - B // 3 if B >= 3 else 1, B)] if B >= 2 else [(0, B)]
        if os.environ.get("MT_PASS_GROUPS") == "singles" and B >= 3:      # experiments: every task pass a group of its own (B streams)
            self._groups = [(i, i + 1) for i in range(B)]
        # workspace slot per group: the engine keys its workspace storage on the pass count, so groups of EQUAL size (B = 2: one pass
        # each) must not share it -- they run concurrently on two streams
        self._group_slots = [sum(1 for (a2, b2) in self._groups[:gi] if b2 - a2 == b - a) for gi, (a, b) in enumerate(self._groups)]
        self._pass_streams = self._grad_sets = self._group_tapes = self._loss_parts = None
        self._group_hook = None             # tests: called on the host after each group has been enqueued (throttles the interleaving)
        # same-box, hipGraph replay, ms per step batched -> split: L = 16 000: 69.5 -> 67.6; 12 000: 51.4 -> 50.6; 10 000: 41.8 -> 40.5;
        # 9 000: 37.7 -> 36.5; 8 000: 33.7 -> 32.9; 6 500: 26.6 -> 27.4 (!); 4 096: 18.2 -> 18.1; 2 500: 12.1 -> 12.2; 1 024: 8.2 -> 8.1
        self.split_min_patches = 7500
        # ... but between ~4 000 and ~7 500 patches the winner changes with the tile rounding of every GEMM and attention launch (round 6,
        # same box, ms batched -> groups: 4 096: 18.37 -> 17.68; 5 500: 23.47 -> 23.11; 6 500: 26.44 -> 27.09; 7 000: 28.65 -> 28.61), so a
        # geometry that is about to be CAPTURED (it keeps coming back) is decided by measurement: both schedules run eagerly a few times
        # with update=False semantics (nothing but the dropout counter moves, and that is restored) and the faster one is captured.
        # Geometries that never repeat keep the threshold.  split_passes="auto" (the default) / True (threshold only) / False.
        self.auto_split = split_passes == "auto" and os.environ.get("MT_SPLIT_PASSES", "auto") == "auto"
        self.split_decisions: Dict[int, bool] = {}
        self.split_trials: Dict[int, dict] = {}
        # Data-parallel schedule of a long bag (world > 1).  "groups_joined": the two pass groups with PER-BUCKET joins -- their backwards
        # run stage by stage (a stage = one interaction block); when both groups have left a block the main stream sums that bucket's
        # ranges of the two gradient sets and starts its all-reduce, the groups keep running below (DDP's overlap of the reduction with
        # the backward, base_trainer.py:205-211, under the two-stream schedule).  "groups_exposed": round 5's form -- every bucket starts
        # behind the groups' final join, the whole reduction is exposed in front of AdamW.  "batched": one B = 3 pass, buckets started
        # from inside its backward.  bench.py --gpus N times all three at start-up and keeps the fastest (`comm.schedule_chosen`).
        # MT_DP_SCHEDULE overrides; on one GPU "groups_joined" only matters when forced (what the extra joins cost: DESIGN section 6).
        self.dp_schedule = os.environ.get("MT_DP_SCHEDULE", "groups_joined")
        self.force_bucket_joins = os.environ.get("MT_BUCKET_JOINS") == "force"
        self.buckets_started_early = 0      # per step: buckets whose all-reduce was started before the last backward kernel was enqueued

    # ------------------------------------------------------------------ learning-rate schedule hook
    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float):
        self.set_lr(value)

    def set_lr(self, lr: float):
        """Scheduler hook: call once per epoch / step with the schedule's value (fill kernel, no host sync); eager steps
        and replays of already captured graphs both train with it from the next step on."""
        self._lr = float(lr)
        self.lr_dev.fill_(float(lr))

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
        out = tape.new(len(self.text_rows), O)
        ops_l2norm_rows(h.data, out, self.text_rows)
        tape.grad_enabled = was
        return out

    # ------------------------------------------------------------------ the step's two halves
    def _world(self) -> int:
        return self.reducer.world

    def _dp(self) -> bool:
        """Collectives are part of the step (world > 1, or dp.single_rank_rehearsal())."""
        return self.reducer.active

    def _on_grad_ready(self, block: int):
        """Engine callback from inside the backward: the gradients of interaction block `block` (and above) are final."""
        b = self._nint - 1 - block
        if self._cap is not None:
            self._segment_break(b)
        else:
            self.reducer.start(b)

    def _can_split(self) -> bool:
        """The pass groups are possible at all for this engine (multi-task model, no per-block taps; TITAN: the native backbone with its
        native embedding)."""
        eng = self.engine
        if not (self.split_passes and eng.cfg.is_multi and not eng.collect_taps):
            return False
        if hasattr(eng, "forward_slide"):
            # TITAN configuration: possible (native backbone with its native embedding) but never a win -- round 6, every geometry of the
            # bench rotation tried both ways, ms per replayed step batched / groups: 2 516 tokens 8.03 / 9.05, 3 027: 9.40 / 10.20,
            # 4 001: 12.42 / 13.27, 4 589: 14.50 / 14.78, 5 491: 18.10 / 18.56, 6 061: 19.96 / 20.83 (twice the token-side launches for
            # half-sized big kernels) -- so no trial is spent on it; MT_SPLIT_PASSES=force still runs it
            bb = getattr(eng, "backbone", None)
            return bool(os.environ.get("MT_SPLIT_PASSES") == "force" and getattr(eng, "native", False) and bb is not None
                        and getattr(bb, "embed_w", None) is not None)
        return True

    def _split_now(self, L: Optional[int] = None) -> bool:
        eng = self.engine
        if not (self.split_passes and eng.cfg.is_multi and not eng.collect_taps):
            return False
        if self._dp() and self.dp_schedule == "batched":
            return False
        if L is not None and L in self.split_decisions and os.environ.get("MT_SPLIT_PASSES") != "force":
            return self.split_decisions[L] and self._can_split()
        if L is not None and L < self.split_min_patches and os.environ.get("MT_SPLIT_PASSES") != "force":
            return False
        if hasattr(eng, "forward_slide"):
            # TITAN configuration: built (native backbone with its native embedding), parity-green -- and 0.2-0.3 ms SLOWER at ~4k
            # tokens (12.88 -> 13.06-13.19 ms same-box: twice the token-side launches for half-sized big kernels): only on request
            bb = getattr(eng, "backbone", None)
            return bool(os.environ.get("MT_SPLIT_PASSES") == "force" and getattr(eng, "native", False) and bb is not None
                        and getattr(bb, "embed_w", None) is not None)
        return True

    def _split_setup(self):
        eng = self.engine
        self._pass_streams = [torch.cuda.Stream(device=self.dev) for _ in self._groups]
        self._grad_sets = [(eng.store.flat_grad, eng.store.grads)] + [eng.store.new_grad_set() for _ in self._groups[1:]]
        from .tape import Tape
        self._group_tapes = []
        for _ in self._groups:
            t = Tape(self.dev)
            t.on_realloc = eng._bump_generation          # captured graphs point into the tapes' gradient arenas
            self._group_tapes.append(t)
        self._loss_parts = [torch.zeros(1, dtype=F32, device=self.dev) for _ in self._groups]

    def _fwd_bwd_split(self, x, coords, genes, text, clinical, staged_geometry=None, reduce: bool = True):
        """_fwd_bwd with the task passes in two concurrent groups (see __init__)."""
        eng = self.engine
        if self._pass_streams is None:
            self._split_setup()
        self._wait_params()
        eng.grad_ready_hook = None            # (world > 1: the buckets start behind the join, optimizer_step -> start_rest)
        if eng.stochastic:
            ops.rng_advance(eng.rng)
        target = self.project_text(text)
        if eng.store.sync is not None:
            eng.store.sync()
        if not eng._caches_ready:
            eng._build_caches()
        gB = [b - a for a, b in self._groups]
        titan = hasattr(eng, "forward_slide")
        if titan:
            # TITAN configuration: gridding, token gather, patch-embedding MLP and the ALiBi distance table once, in front of the fork
            share = {}
            eng.forward_slide(x, coords, genes, self.onehots[:gB[0]], patch_size_lv0=self.patch_size_lv0, need_grad=True, clinical=clinical,
                              staged=staged_geometry is not None, share=share, prologue_only=True)
            share["x0"] = share["tok"][1:]
            L = int(share["tok"].shape[0]) - 1
            for nb, sl in zip(gB, self._group_slots):
                eng._workspace(nb, L, slot=sl)
        else:
            if staged_geometry is None:
                x = x.reshape(-1, x.shape[-1])
                L = x.shape[0]
                ws0 = eng._workspace(gB[0], L)
                eng.stage_inputs(x, coords, ws0)
            else:
                L = staged_geometry[1]
                ws0 = eng._workspace(gB[0], L)         # (step_graphed staged the slide into the first group's workspace)
            for nb, sl in list(zip(gB, self._group_slots))[1:]:
                eng._workspace(nb, L, slot=sl)          # (grown before the fork: a growth bumps the generation)
            eng._embed_patches(None, None, ws0, True, L)        # task-independent: once, in front of the fork
            share = {"x0": ws0["x0"]}
        R, O = target.shape
        logits_all = torch.empty(R, O, dtype=F32, device=self.dev)
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        # per-bucket joins: the groups' backwards advance stage by stage so that a bucket's all-reduce starts as soon as BOTH groups
        # have left its interaction block (see __init__: dp_schedule)
        joined = (self._dp() and self.dp_schedule == "groups_joined" and reduce) or self.force_bucket_joins
        eng.record_markers = bool(joined)
        calls = []
        self.buckets_started_early = 0
        try:
            for gi, (a, b) in enumerate(self._groups):
                st = self._pass_streams[gi]
                st.wait_event(fork)
                with torch.cuda.stream(st):
                    old = eng.store.use_grad_set(*self._grad_sets[gi])
                    try:
                        self._grad_sets[gi][0].zero_()
                        if titan:
                            logits = eng.forward_slide(None, None, genes, self.onehots[a:b], patch_size_lv0=self.patch_size_lv0, need_grad=True,
                                                       clinical=clinical, share=share, staged=True, tape=self._group_tapes[gi], site_group=gi + 1,
                                                       ws_slot=self._group_slots[gi])
                        else:
                            logits = eng.forward(None, None, genes, self.onehots[a:b], need_grad=True, staged=True, geometry=(b - a, L),
                                                 clinical=clinical, share=share, tape=self._group_tapes[gi], site_group=gi + 1,
                                                 ws_slot=self._group_slots[gi])
                        call = eng.last_call
                        dlogits = torch.empty_like(logits)
                        ops.distill_loss(logits, target[a:b], self._loss_parts[gi], dlogits, b - a, O, 1.0, self.scale)
                        if joined:
                            eng.backward_begin(dlogits, call)
                            calls.append(call)
                        else:
                            eng.backward(dlogits, call=call)
                        logits_all[a:b].copy_(logits)
                    finally:
                        eng.store.use_grad_set(*old)
                if self._group_hook is not None:
                    self._group_hook(gi)
            fg = eng.store.flat_grad
            summed = set()
            if joined:
                nint = self._nint
                for stage in range(nint + 1):
                    evs = []
                    for gi, call in enumerate(calls):
                        st = self._pass_streams[gi]
                        with torch.cuda.stream(st):
                            blk = eng.backward_stage(call)
                            if stage < nint:
                                if blk != nint - 1 - stage:
                                    raise RuntimeError(f"pass group {gi}: backward stage {stage} ended at block marker {blk}")
                                ev = torch.cuda.Event()
                                ev.record(st)
                                evs.append(ev)
                            elif blk is not None:
                                raise RuntimeError(f"pass group {gi}: a block marker ({blk}) below the last stage")
                    if stage == nint:
                        break
                    # bucket `stage` (= the parameters of block nint - 1 - stage, and the head for stage 0) is final in every group's
                    # set: sum its ranges on the main stream and start its all-reduce; the groups keep running their lower blocks
                    for ev in evs:
                        main.wait_event(ev)
                    for o, n in self.reducer.buckets[stage]:
                        for flat, _ in self._grad_sets[1:]:
                            ops.axpy(fg[o:o + n], flat[o:o + n], 1.0, fg[o:o + n])
                    summed.add(stage)
                    self.buckets_started_early += 1
                    if self._cap is not None:
                        # under capture the graph is cut here (the collective is launched eagerly between two replays): a cut needs
                        # every forked stream back on the capture stream, so the groups re-fork behind it
                        self._segment_break(stage)
                        fork = torch.cuda.Event()
                        fork.record(main)
                        for st in self._pass_streams:
                            st.wait_event(fork)
                    else:
                        self.reducer.start(stage)
        finally:
            eng.record_markers = False
            for st in self._pass_streams:
                main.wait_stream(st)
        # the streams have met: one gradient buffer, one loss (buckets summed at their own joins are left alone: their all-reduce
        # may be in flight)
        if summed:
            for b, bk in enumerate(self.reducer.buckets):
                if b not in summed:
                    for o, n in bk:
                        for flat, _ in self._grad_sets[1:]:
                            ops.axpy(fg[o:o + n], flat[o:o + n], 1.0, fg[o:o + n])
        else:
            for flat, _ in self._grad_sets[1:]:
                ops.axpy(fg, flat, 1.0, fg)
        ops.axpy(self._loss_parts[0], self._loss_parts[1], 1.0, self.loss)
        for part in self._loss_parts[2:]:
            ops.axpy(self.loss, part, 1.0, self.loss)
        self.last_logits = logits_all

    def _fwd_bwd(self, x, coords, genes, text, clinical, staged_geometry=None, reduce: bool = True):
        eng = self.engine
        Lnow = staged_geometry[1] if staged_geometry is not None else (x.reshape(-1, x.shape[-1]).shape[0] if torch.is_tensor(x) else None)
        if self._split_now(Lnow):
            return self._fwd_bwd_split(x, coords, genes, text, clinical, staged_geometry, reduce=reduce)
        self._wait_params()               # the all-gather of the last step's sharded parameter update (no-op otherwise)
        eng.grad_ready_hook = self._on_grad_ready if (reduce and self._dp()) else None
        if eng.stochastic:
            ops.rng_advance(eng.rng)          # a fresh set of dropout / DropPath masks per step
        target = self.project_text(text)
        eng.store.flat_grad.zero_()
        if hasattr(eng, "forward_slide"):      # TITAN configuration: gridding + backbone embed first (staged: gridding already done)
            logits = eng.forward_slide(x, coords, genes, self.onehots, patch_size_lv0=self.patch_size_lv0, need_grad=True,
                                       clinical=clinical, staged=staged_geometry is not None)
        elif staged_geometry is None:
            logits = eng.forward(x, coords, genes, self.onehots, need_grad=True, clinical=clinical)
        else:
            logits = eng.forward(None, None, genes, self.onehots, need_grad=True, staged=True, geometry=staged_geometry,
                                 clinical=clinical)
        self.last_logits = logits
        R, O = target.shape
        if logits.shape[0] == R:
            dlogits = torch.empty_like(logits)
            ops.distill_loss(logits, target, self.loss, dlogits, R, O, 1.0, self.scale)
        else:       # single-task: nn.KLDivLoss broadcasts the one logits row over the R text rows (TM:225-233)
            rows = torch.empty(R, O, dtype=F32, device=self.dev)
            ops.copy_rows(logits, rows, R, O, smap=rowmap(1, 0, 0))
            drows = torch.empty_like(rows)
            ops.distill_loss(rows, target, self.loss, drows, R, O, 1.0, self.scale)
            dlogits = torch.empty(1, O, dtype=F32, device=self.dev)
            ops.copy_rows(drows[0:1], dlogits, 1, O)
            for r in range(1, R):
                ops.copy_rows(drows[r:r + 1], dlogits, 1, O, accumulate=True)
        try:
            eng.backward(dlogits)
        finally:
            eng.grad_ready_hook = None        # a later direct eng.forward / backward must not start stray collectives

    def _check_finite(self):
        """GradScaler's inf check over what this rank holds; with a sharded last bucket the flag is MAX-reduced so that every
        rank takes the same skip decision (a collective: never inside a captured graph)."""
        eng = self.engine
        ops.check_finite(eng.store.flat_grad, eng.store.n_flat, self.found_inf)
        self.reducer.sync_flag_(self.found_inf)

    def _adam_and_refresh(self, world: int, check: bool = True):
        eng = self.engine
        n = eng.store.n_flat
        if check:
            ops.check_finite(eng.store.flat_grad, n, self.found_inf)
        p, g = eng.store.flat, eng.store.flat_grad
        for o, k in self.reducer.adam_pieces(n):      # one launch, or: everything but the sharded bucket + this rank's shards
            ops.adamw_step(p[o:o + k], g[o:o + k], self.m[o:o + k], self.v[o:o + k], k, self._lr, self.betas[0], self.betas[1], self.eps,
                           self.wd, 0, scale=self.scale, found_inf=self.found_inf, grad_mult=1.0 / world, step_dev=self.step_dev,
                           lr_dev=self.lr_dev)
        ops.scaler_update(self.scale, self.tracker, self.found_inf, self.step_dev, 2.0, 0.5, self.growth_interval)
        eng.refresh_trainable_caches()

    def _mark(self):
        if self.comm_events is None or self._cap is not None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _wait_params(self):
        e0 = self._mark() if self.reducer._param_pending else None
        self.reducer.wait_params()
        if e0 is not None:
            self.comm_events.append(("param", e0, self._mark()))

    def optimizer_step(self):
        """Launch whatever buckets the backward has not started, wait for the collectives, AdamW + weight-cache refresh; with
        a sharded last bucket (world > 1) the updated shards are then all-gathered asynchronously (waited for at the top of the
        next step)."""
        e0 = self._mark()
        self.reducer.start_rest()
        world = self.reducer.wait()
        if self.reducer.sharded:
            self._check_finite()
        if e0 is not None:
            self.comm_events.append(("grad", e0, self._mark()))
        if self.reducer.sharded:
            self._adam_and_refresh(world, check=False)
            self.reducer.start_param_gather()
        else:
            self._adam_and_refresh(world)

    def step(self, x, coords, genes, text, update: bool = True, clinical=None) -> torch.Tensor:
        """One train step on one slide, eager launches.  Returns the (device) loss scalar; no host sync happens here."""
        self._fwd_bwd(x, coords, genes, text, clinical, reduce=update)
        self.eager_steps += 1
        if update:
            self.optimizer_step()
        return self.loss

    # ------------------------------------------------------------------ hipGraph replay of the whole step
    def step_graphed(self, x, coords, genes, text, clinical=None) -> torch.Tensor:
        """Same arithmetic as step().  A bag geometry (patch count, gene count) that keeps coming back is captured into
        hipGraphs after `capture_after` eager visits and replayed from then on (the ~900 launches of a step become a few
        host calls); up to `graph_cache_size` geometries stay captured (LRU).  Inputs are uploaded into static buffers
        first.  With world_size > 1 the capture is cut where a gradient bucket becomes final: the reducer starts that
        bucket's all-reduce between two replays, and the optimiser graph runs after the wait."""
        eng = self.engine
        titan = hasattr(eng, "forward_slide")
        if not eng._caches_ready:
            eng._build_caches()
        x = x.reshape(-1, x.shape[-1])
        L = x.shape[0]
        B = self.onehots.shape[0]
        gflat = genes.reshape(-1) if torch.is_tensor(genes) else torch.cat([g.reshape(-1) for g in genes])
        skey = (int(gflat.numel()), tuple(text.shape))
        if self._static_key != skey:
            self._static_key = skey
            self._sgenes = torch.empty(int(gflat.numel()), dtype=F32, device=self.dev)     # one flat static buffer
            self._stext = torch.empty(tuple(text.shape), dtype=F32, device=self.dev)
            self._sclin = torch.empty(1, eng.cfg.clinfeat_dim, dtype=F32, device=self.dev) if eng.cfg.clinical else None
            self._gcache.clear()
        if titan:
            # TITAN configuration: the gridding kernels and the one host read-back (the token count: every shape downstream depends
            # on it) run eagerly; the captured part starts at the token gather and is keyed on (patches, TOKENS).
            Lv = eng.stage_slide(x, coords, self.patch_size_lv0)
            for nb, sl in (list(zip([b - a for a, b in self._groups], self._group_slots)) if self._split_now(Lv) else [(B, 0)]):
                eng._workspace(nb, Lv, slot=sl)   # (may grow the workspace: bumps eng.generation)
        else:
            Lv = L
            gB = [b - a for a, b in self._groups] if self._split_now(L) else [B]
            for nb, sl in list(zip(gB, self._group_slots))[1:]:
                eng._workspace(nb, L, slot=sl)    # (all groups' workspaces exist -- and have grown -- before anything is captured)
            eng.stage_inputs(x, coords, B=gB[0])  # (may grow the workspace: bumps eng.generation)
        if self.auto_split and Lv not in self.split_decisions and B >= 3 and Lv >= 2048 and not self._dp() and self._can_split():
            # this geometry's schedule is still to be decided by a trial (below, on the visit that captures it): the workspaces of BOTH
            # schedules exist and have grown NOW, on an eager visit -- a growth bumps eng.generation and retires every capture
            for nb, sl in [(B, 0)] + list(zip([b - a for a, b in self._groups], self._group_slots)):
                eng._workspace(nb, Lv, slot=sl)
        self._wait_params()                       # last step's sharded parameter all-gather ran under the staging above
        self._sgenes.copy_(gflat, non_blocking=True)
        self._stext.copy_(text, non_blocking=True)
        if self._sclin is not None:
            self._sclin.copy_(clinical.reshape(1, -1), non_blocking=True)
        world, dp_on = self._world(), self._dp()
        if self._ggen != eng.generation:          # buffers the old captures point to are gone (visit counts stay)
            for e in self._gcache.values():
                e.segs = e.logits = None
            self._opt_graph = None
            self._ggen = eng.generation
        # (the data-parallel schedule is part of the key: bench.py --gpus N times the three schedules one after the other)
        key = (L, Lv, world, bool(eng.stochastic), self.dp_schedule if dp_on else None, self.force_bucket_joins)
        ent = self._gcache.get(key)
        if (self.auto_split and not dp_on and Lv not in self.split_decisions and B >= 3 and Lv >= 2048 and self._can_split()
                and (ent is None or ent.segs is None) and self._visits.get(key, 0) >= self.capture_after
                and not torch.cuda.is_current_stream_capturing()):
            self._trial_split(x, coords, genes, text, clinical, Lv, key)     # (sets split_decisions[Lv]; training state untouched)
            return self.step_graphed(x, coords, genes, text, clinical=clinical)

        def fwd_bwd():
            self._fwd_bwd(None, None, self._sgenes, self._stext, self._sclin, staged_geometry=(B, Lv))

        # Visit counts live OUTSIDE the LRU: on ragged data almost every slide has a new length, and a stream of one-off
        # lengths must neither evict the captured graphs of a hot geometry nor reset its count.
        if ent is None or ent.segs is None:
            seen = self._visits.get(key, 0)
            if seen < self.capture_after:         # eager visits (allocator, lazy kernel attributes, the tape's gradient arena
                if len(self._visits) > 4096:      # sized from the last step)
                    self._visits.clear()
                self._visits[key] = seen + 1
                fwd_bwd()
                self.eager_steps += 1
                self.optimizer_step()
                return self.loss
            if ent is None:
                ent = _Captured()
                self._gcache[key] = ent
                while len(self._gcache) > max(1, self.graph_cache_size):
                    self._gcache.popitem(last=False)
            self._capture(ent, fwd_bwd, world)
        else:
            self._gcache.move_to_end(key)
        for g, bucket in ent.segs:
            g.replay()
            if bucket is not None:
                self.reducer.start(bucket)
        self.last_logits = ent.logits
        self.graph_replays += 1
        if dp_on:
            e0 = self._mark()
            self.reducer.start_rest()
            self.reducer.wait()
            if self.reducer.sharded:
                self._check_finite()
            if e0 is not None:
                self.comm_events.append(("grad", e0, self._mark()))
            self._opt_graph.replay()
            self.reducer.start_param_gather()
        return self.loss

    def _trial_split(self, x, coords, genes, text, clinical, L: int, key: tuple, reps: int = 3):
        """Decide the schedule of bag length L by timing what will actually run: the step captured both ways (batched / two pass groups)
        and each capture replayed `reps` times behind one untimed replay.  Eager timings do not rank the two (the groups' ~1 300 eager
        launches are partly host-bound: 18.8 / 19.0 ms eager against 18.4 / 17.7 replayed at L = 4 096).  The replays are real steps, so
        every piece of training state they touch -- weights, moments, step count, loss scale, dropout counter -- is saved in front and
        restored behind them (lr is irrelevant then); the winner's capture is kept for the step that follows."""
        import time
        eng = self.engine
        saved = [(t, t.clone()) for t in (eng.store.flat, self.m, self.v, self.step_dev, self.scale, self.tracker, self.found_inf, eng.rng)]
        was_auto, self.auto_split = self.auto_split, False
        counters = (self.graph_replays, self.eager_steps)
        res, caps = {}, {}
        try:
            for split in (False, True):
                self.split_decisions[L] = split
                self._gcache.pop(key, None)       # (same key for both schedules: capture afresh)
                for i in range(reps + 2):         # capture (visit counts are already past capture_after), one untimed replay, reps timed
                    if i == 2:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                    self.step_graphed(x, coords, genes, text, clinical=clinical)
                torch.cuda.synchronize()
                res["groups" if split else "batched"] = round(1e3 * (time.perf_counter() - t0) / reps, 3)
                caps[split] = self._gcache.get(key)
                if caps[split] is not None:
                    self._trial_keep.append(caps[split])
        finally:
            self._trial_keep = []
            self.auto_split = was_auto
            self.graph_replays, self.eager_steps = counters
            for t, c in saved:
                t.copy_(c)
            eng.refresh_trainable_caches()
        win = res["groups"] < res["batched"]
        self.split_decisions[L] = win
        self.split_trials[L] = res
        if caps.get(win) is not None:
            self._gcache[key] = caps[win]

    @property
    def _graphs(self):
        """Captured segment lists of the cached geometries (None when nothing is captured yet)."""
        segs = [e.segs for e in self._gcache.values() if e.segs is not None]
        return segs or None

    def _capture(self, ent: _Captured, fwd_bwd, world: int):
        torch.cuda.synchronize()
        main = torch.cuda.current_stream()
        if self._cap_stream is None:          # ONE capture stream: the allocator hands a freed block only to the stream it was
            self._cap_stream = torch.cuda.Stream()      # allocated on, and torch.cuda.Stream() walks a ring of 32 streams
        side = self._cap_stream
        side.wait_stream(main)
        # ONE private pool for every capture of this TrainStep: the graphs never run concurrently, so a later capture may reuse the
        # blocks an evicted (or still cached) geometry's temporaries occupied.  A fresh pool per capture left every evicted graph's
        # segments reserved-but-unusable until the allocator's out-of-memory sweep: +0.24 GiB per recapture at L ~ 4 000 when more
        # lengths rotate than the LRU holds (tools/soak.py).
        if self._pool is None or not (self._opt_graph is not None or any(e.segs for e in list(self._gcache.values()) + self._trial_keep)):
            self._pool = torch.cuda.graph_pool_handle()      # (a pool dies with the last graph captured into it: take a fresh handle)
        pool = self._pool
        segs = []
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()          # (thread_local: RCCL's watchdog thread must not invalidate the capture)
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
            self._cap = {"segs": segs, "cur": g, "pool": pool}
            try:
                fwd_bwd()
                if not self._dp():
                    self.optimizer_step()
            except BaseException:
                try:
                    self._cap["cur"].capture_end()
                except Exception:
                    pass
                self._cap = None
                raise
            self._cap["cur"].capture_end()
            segs.append((self._cap["cur"], None))
            self._cap = None
            if self._dp() and self._opt_graph is None:
                og = torch.cuda.CUDAGraph()
                og.capture_begin(pool=pool, capture_error_mode="thread_local")
                try:
                    self._adam_and_refresh(world, check=not self.reducer.sharded)
                finally:
                    og.capture_end()
                self._opt_graph = og
        main.wait_stream(side)
        ent.segs, ent.logits = segs, self.last_logits

    def _segment_break(self, bucket: int):
        """Inside a capture with world_size > 1: close the current graph at a gradient-bucket boundary, open the next."""
        cap = self._cap
        cap["cur"].capture_end()
        cap["segs"].append((cap["cur"], bucket))
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=cap["pool"], capture_error_mode="thread_local")
        cap["cur"] = g

    def loss_value(self) -> float:
        """The last step's loss on the host.  This is where the trainer synchronises anyway (`loss.item()`, TM:239), so the
        deferred input check (bad / non-finite coords binned on the device) is raised here too."""
        v = float(self.loss)
        self.engine.check_inputs()
        return v

    def unscaled_grads(self) -> Dict[str, torch.Tensor]:
        s = float(self.scale)
        return {k: g / s for k, g in self.engine.store.grads.items()}


def ops_l2norm_rows(h: torch.Tensor, out: torch.Tensor, rows: Sequence[int]):
    """out[i] = h[rows[i]] / ||h[rows[i]]||: R <= 4 rows of O = 256, one tiny launch per row."""
    O = h.shape[-1]
    for i, r in enumerate(rows):
        ops.l2norm_row(h[r], out[i], O)
