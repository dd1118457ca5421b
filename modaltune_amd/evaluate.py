"""Eval / embedding-extraction pass: the forward kernels of the train step without the backward.

Reference: train_modaltune.py:156-179 (`multitask_forward`: one model call per task id, concatenated) and
train_modaltune.py:252-327 (`get_features`: eval mode, no_grad, logits [3, 256] per case for train / val / test,
stacked on the host for the CPU-side probes).  Here the task passes of a slide are ONE batched engine call (B = len
(task_ids)) with need_grad = False (no activations are saved), optionally replayed from a captured hipGraph.
The probes themselves (sklearn LogisticRegression / lifelines Cox, TM:329-458) are host code and out of scope.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .engine import Engine, F32


def multitask_forward(model, task_ids: Optional[Sequence[int]] = None, num_tasks: Optional[int] = None, **kwargs) -> torch.Tensor:
    """Drop-in for the trainer's `multitask_forward` (TM:156-179): logits [len(task_ids), output_dim].
    kwargs as the reference passes them: x, coords, genes, clinical."""
    if not model.is_multi:
        return model(**kwargs)
    num_tasks = num_tasks or model.cfg.multi_task
    if task_ids is None:
        task_ids = range(num_tasks)
    onehots = torch.eye(num_tasks, dtype=F32)[list(task_ids)]
    return model.forward_tasks(kwargs["x"], kwargs["coords"], kwargs["genes"], onehots, clinical=kwargs.get("clinical"))


class EmbeddingExtractor:
    """Forward-only pass over slides with static buffers + hipGraph replay per bag geometry: an LRU of captured geometries, a
    geometry is captured once it has come back `capture_after` times (real data has a new bag length almost every slide: those run
    the eager schedule).  TITAN configuration: the gridding and its one host read-back (the token count) run eagerly into static
    buffers, the capture starts at the token gather and is keyed on (patches, TOKENS) -- as TrainStep.step_graphed does."""

    def __init__(self, engine: Engine, task_ids: Sequence[int] = (0, 1, 2), graphed: bool = True, graph_cache_size: int = 8,
                 capture_after: int = 1):
        self.engine, self.dev, self.graphed = engine, engine.device, graphed
        nt = max(1, engine.cfg.multi_task)
        self.onehots = torch.eye(nt, dtype=F32, device=self.dev)[list(task_ids)].contiguous()
        self.graph_cache_size, self.capture_after = int(graph_cache_size), int(capture_after)
        self._cache: "OrderedDict[tuple, dict]" = OrderedDict()
        self._visits: Dict[tuple, int] = {}
        self._static_key = None
        self._pool = None
        self.graph_replays = 0
        self.patch_size_lv0 = 1024          # TITAN configuration only (titan_adapter.py:335)
        self.split_passes = os.environ.get("MT_SPLIT_PASSES", "1") not in ("0", "off")
        self.split_min_patches = 7500
        self._streams = self._tapes = None

    def _forward_groups(self, B: int, L: int) -> torch.Tensor:
        """The forward of a long bag as two concurrent pass groups (trainer.TrainStep._fwd_bwd_split, forward half): the patch
        embedding once in front of the fork, own workspace and tape per group, logits joined behind it."""
        eng = self.engine
        if self._streams is None:
            from .tape import Tape
            self._streams = [torch.cuda.Stream(device=self.dev) for _ in range(2)]
            self._tapes = [Tape(self.dev) for _ in range(2)]
        a = B - B // 3
        ws0 = eng._workspace(a, L)
        eng._embed_patches(None, None, ws0, True, L)
        share = {"x0": ws0["x0"]}
        out = torch.empty(B, eng.cfg.output_dim, dtype=F32, device=self.dev)
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        for gi, (lo, hi) in enumerate(((0, a), (a, B))):
            st = self._streams[gi]
            st.wait_event(fork)
            with torch.cuda.stream(st):
                lg = eng.forward(None, None, self._sgenes, self.onehots[lo:hi], need_grad=False, staged=True, geometry=(hi - lo, L),
                                 clinical=self._sclin, share=share, tape=self._tapes[gi], site_group=gi + 1)
                out[lo:hi].copy_(lg)
        for st in self._streams:
            main.wait_stream(st)
        return out

    @property
    def _graph(self):
        """The most recently used captured graph (None while nothing is captured)."""
        live = [e["graph"] for e in self._cache.values() if e.get("graph") is not None]
        return live[-1] if live else None

    @torch.no_grad()
    def __call__(self, x, coords, genes: Sequence[torch.Tensor], clinical=None) -> torch.Tensor:
        """Logits (= the slide embeddings the probes consume) [len(task_ids), output_dim], on the device (a fresh
        tensor per call: replays write a static buffer that is copied out)."""
        eng = self.engine
        titan = hasattr(eng, "forward_slide")
        if not eng._caches_ready:
            eng._build_caches()
        x = x.reshape(-1, x.shape[-1])
        L, B = x.shape[0], self.onehots.shape[0]
        if isinstance(genes, dict):
            genes = [genes[k] for k in sorted(genes.keys())]
        split = False
        replayable = self.graphed and (not titan or (getattr(eng, "native", False) and getattr(eng.backbone, "embed_w", None) is not None))
        if not replayable:      # (TITAN on the module's own torch blocks: nothing to capture)
            if titan:
                return eng.forward_slide(x, coords, list(genes), self.onehots, patch_size_lv0=self.patch_size_lv0, need_grad=False,
                                         clinical=clinical)
            return eng.forward(x, coords, list(genes), self.onehots, need_grad=False, clinical=clinical)
        gflat = genes.reshape(-1) if torch.is_tensor(genes) else torch.cat([g.reshape(-1) for g in genes])
        if titan:
            Lv = eng.stage_slide(x, coords, self.patch_size_lv0)      # eager gridding + the one read-back -> token count
            eng._workspace(B, Lv)                                     # (may grow the workspace: bumps eng.generation)
        else:
            Lv = L
            # long bags: the task passes as two concurrent groups (B - B // 3 and B // 3 passes on two HIP streams), as in the train step
            split = self.split_passes and B >= 3 and L >= self.split_min_patches and eng.cfg.is_multi and not eng.collect_taps
            gB = [B - B // 3, B // 3] if split else [B]
            for nb in gB[1:]:
                eng._workspace(nb, L)
            eng.stage_inputs(x, coords, B=gB[0])                      # (may grow the workspace: bumps eng.generation)
        skey = int(gflat.numel())
        if self._static_key != skey:
            self._static_key = skey
            self._sgenes = torch.empty(skey, dtype=F32, device=self.dev)     # one flat static buffer
            self._sclin = torch.empty(1, eng.cfg.clinfeat_dim, dtype=F32, device=self.dev) if eng.cfg.clinical else None
            self._cache.clear()
        self._sgenes.copy_(gflat, non_blocking=True)
        if self._sclin is not None:
            self._sclin.copy_(clinical.reshape(1, -1), non_blocking=True)
        # the engine's generation is part of the key: a workspace that grew under another user of the engine (the trainer
        # shares the B = 3 storage), rebuilt weight caches (load_state_dict) or a stochastic toggle retire the captures
        key = (L, Lv, eng.generation)
        for k in [k for k in self._cache if k[2] != eng.generation]:
            del self._cache[k]
        if titan:
            run = lambda: eng.forward_slide(None, None, self._sgenes, self.onehots, patch_size_lv0=self.patch_size_lv0, need_grad=False,
                                            clinical=self._sclin, staged=True)
        elif split:
            run = lambda: self._forward_groups(B, L)
        else:
            run = lambda: eng.forward(None, None, self._sgenes, self.onehots, need_grad=False, staged=True, geometry=(B, L),
                                      clinical=self._sclin)
        ent = self._cache.get(key)
        if ent is None:
            seen = self._visits.get(key, 0)
            if seen < self.capture_after:
                if len(self._visits) > 4096:
                    self._visits.clear()
                self._visits[key] = seen + 1
                return run()
            torch.cuda.synchronize()
            if self._pool is None or not self._cache:
                self._pool = torch.cuda.graph_pool_handle()           # one pool for all captures (they never run concurrently)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                out = run()
            ent = self._cache[key] = {"graph": g, "out": out}
            while len(self._cache) > max(1, self.graph_cache_size):
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(key)
        ent["graph"].replay()
        self.graph_replays += 1
        return ent["out"].clone()


def get_features(extractor: EmbeddingExtractor, slides: Iterable[Dict]) -> Tuple[np.ndarray, List]:
    """Host-side collection as TM:262-327 does per split: returns (features [n, tasks, O], case ids)."""
    feats, ids = [], []
    for s in slides:
        feats.append(extractor(s["x"], s["coords"], s["genes"], s.get("clinical")).float().cpu().numpy())
        ids.append(s.get("case_id"))
    extractor.engine.check_inputs()       # the host has just synchronised on every slide's logits: raise for bad coords now
    return np.stack(feats) if feats else np.zeros((0,)), ids
