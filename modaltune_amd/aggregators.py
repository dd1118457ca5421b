"""Drop-in nn.Module surface of the reference for the hot path.

Same registry names, constructor kwargs, forward signature, `is_multi` attribute and state_dict keys as
models/aggregators/aggregators.py:6-58 and models/aggregators/longvit_adapter.py:30-347 — backed by the HIP
engine instead of PyTorch modules, so train_modaltune.py:123-126,172-177 can use it unchanged:

    model = Aggregator.create("longnetvit_gene_adapter", gene_group_defination=groups, **json_cfg, multi_task=3)
    logits = model(x=images, coords=coords, genes=gene_data, clinical=[], task_token=torch.eye(3)[t].cuda())

The module-level forward is the compatibility path (one pass per call, any number of forwards before a backward);
`forward_tasks` batches the task passes like the fused TrainStep does.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Any, Dict, Iterator, Sequence, Tuple

import torch
import torch.nn as nn

from . import init, ops
from .config import ModelConfig
from .engine import Engine, F32


class Aggregator(nn.Module):
    """Registry base (models/aggregators/aggregators.py:6-41)."""
    subclasses: Dict[str, Any] = {}

    @classmethod
    def register(cls, subclass_name: str):
        def decorator(subclass):
            cls.subclasses[subclass_name] = subclass
            return subclass
        return decorator

    @classmethod
    def create(cls, subclass_name: str, **params):
        if subclass_name not in cls.subclasses and subclass_name.startswith("titan"):
            from . import titan  # noqa: F401  (registers titan_gene_adapter / titan_gene_clinical_adapter)
        if subclass_name not in cls.subclasses:
            raise ValueError("Unknown subclass name {}".format(subclass_name))
        return cls.subclasses[subclass_name](**params)


class _StepGroup:
    """The forward calls of ONE slide that a single backward pass will meet: the reference calls the model once per task id and
    then runs loss.backward() (TM:172-177,235).  Calls of a group are chained in the autograd graph through a scalar token, so
    the engine runs their backward nodes last-call-first and the FIRST call's node runs last; every node replays its tape into
    the same flat gradient buffer and only that last node hands the (unscaled) sum to autograd -- one zero / unscale / hand-over
    per step instead of one per call, and every parameter's AccumulateGrad (and DDP reducer hook) fires exactly once."""
    __slots__ = ("key", "count", "limit", "scale", "started", "share")

    def __init__(self, key, limit: int):
        # (the chain token itself lives on the module, `_token`: a group is reachable from its calls' autograd nodes, and a token
        # kept here would close a reference cycle node -> group -> token -> node that only the cyclic collector frees -- with it
        # the calls' tapes and leased workspaces of several steps stayed alive at once)
        self.key, self.count, self.limit = key, 0, max(1, int(limit))
        self.scale, self.started, self.share = None, False, {}


def _slide_key(x, coords):
    def k(t):
        return (t.data_ptr(), t._version, tuple(t.shape), str(t.device)) if torch.is_tensor(t) else id(t)
    return (k(x), k(coords))


def _bridge_backward(ctx, dlogits):
    """Shared by the LongNet and TITAN bridges: returns (gradient of the chain token input or None, tuple of parameter
    gradients or Nones).  A node reached WITHOUT a gradient for its logits (only the chain token's) passes the chain on and keeps
    its tape: a trainer that backpropagates the task losses one by one (`loss_t.backward()` per task instead of the reference's
    single `loss.backward()`) meets the node again later with its own gradient, and every such pass hands over exactly what it
    replayed (param.grad accumulates across the passes as with any module)."""
    module, eng, grp = ctx.module, ctx.module.engine, ctx.group
    store = eng.store
    nparams = len(module._slots)
    split = getattr(ctx, "split", None)          # a batched pass that ran as two concurrent pass groups (_ModelFn.forward)
    if dlogits is not None:
        if ctx.call is None:
            raise RuntimeError("backward through a forward that ran with gradients disabled (or a second backward through the same call"
                               + ("; the task passes of this slide were batched into ONE call -- set model.speculate = False to "
                                  "backpropagate the task losses one by one" if ctx.batched else "") + ")")
        # The tape runs an fp16 activation-gradient stream: rescale the incoming gradients to max |.| = 2^10 on the DEVICE (no
        # read-back; the factor of a pass's first node serves the whole pass), run the tapes, undo the scale once when
        # the flat gradient is handed over.
        dl = dlogits.to(F32).contiguous()
        if not grp.started:
            grp.scale = torch.empty(2, dtype=F32, device=dl.device)
            ops.absmax_scale(dl, grp.scale, 1024.0)
            store.flat_grad.zero_()
            grp.started = True
        replay = getattr(ctx, "replay", None)
        scaled = torch.empty_like(dl) if replay is None else None
        if replay is None:
            ops.axpy_dev(None, dl, grp.scale[0:1], scaled)
        hook, eng.grad_ready_hook = eng.grad_ready_hook, None      # (a TrainStep sharing the engine must not see this pass)
        try:
            if replay is not None:
                module._replay.backward(replay, lambda dst: ops.axpy_dev(None, dl, grp.scale[0:1], dst))
                ctx.replay = None
            elif split is None:
                eng.backward(scaled, call=ctx.call)
            else:        # every group replays its tape on its own stream into its own gradient set; the sets are summed behind the join
                main = torch.cuda.current_stream()
                fork = torch.cuda.Event()
                fork.record(main)
                for (a, b), call, st, (gflat, _) in zip(split["groups"], ctx.call, split["streams"], split["sets"]):
                    st.wait_event(fork)
                    with torch.cuda.stream(st):
                        if gflat is not store.flat_grad:
                            gflat.zero_()
                        eng.backward(scaled[a:b], call=call)
                for st in split["streams"]:
                    main.wait_stream(st)
                for gflat, _ in split["sets"]:
                    if gflat is not store.flat_grad:
                        ops.axpy(store.flat_grad, gflat, 1.0, store.flat_grad)
        finally:
            eng.grad_ready_hook = hook
        ctx.call = None                                            # the tape and its private workspace are dead from here
    if ctx.has_pred:       # an earlier call of the group runs after this node and delivers
        return torch.zeros((), dtype=F32, device=store.flat_grad.device), (None,) * nparams
    if module._group is grp:
        module._group = module._token = None
    grp.share.clear()                                              # (the slide's shared patch embedding and workspace leases)
    if not grp.started:
        return None, (None,) * nparams
    out = torch.empty_like(store.flat_grad)
    ops.axpy_dev(None, store.flat_grad, grp.scale[1:2], out)
    grp.started = False
    return None, tuple(out[o:o + n].view(shape) for o, n, shape in module._slots)


class _ModelFn(torch.autograd.Function):
    """autograd bridge: forward runs the HIP engine (own tape + workspace per call); backward replays that call's tape.
    The trainable parameters are REAL inputs of the Function and their gradients its outputs, so autograd's
    AccumulateGrad nodes run as for any module: torch optimisers / GradScaler see ordinary gradients, and
    DistributedDataParallel's reducer hooks fire (utils/base_trainer.py:205-211 wraps the model in DDP).  The calls of one step
    are chained (`_StepGroup`)."""

    @staticmethod
    def forward(ctx, module, x, coords, genes, onehots, need, clinical, token, *params):
        eng = module.engine
        ctx.set_materialize_grads(False)
        grp = module._group if need else None
        B = int(onehots.shape[0])
        L = int(x.reshape(-1, x.shape[-1]).shape[0])
        ctx.split = ctx.replay = None
        served = module._replay.forward(x, coords, genes, onehots, clinical, token) if (need and grp is not None and module._replay is not None) else None
        if served is not None:
            # steady state of a training loop: this geometry's forward / backward are hipGraph replays (module_graph.ModuleReplay)
            logits, ctx.replay = served
            ctx.module, ctx.call, ctx.group, ctx.has_pred = module, "replay", grp, False
            ctx.batched = module.is_multi and logits.shape[0] > 1
            return logits, torch.zeros((), dtype=F32, device=logits.device)
        if need and grp is not None and B >= 3 and L >= module.split_min_patches and module.split_passes and not eng.collect_taps:
            # A batched pass over a long bag runs as TWO concurrent pass groups (B - B // 3 and B // 3 task passes on two HIP streams:
            # trainer.TrainStep._fwd_bwd_split has the measurements) -- own workspace, tape, dropout masks and gradient set per group,
            # the task-independent patch embedding once in front of the fork.
            sp = module._split_state()
            groups = [(0, B - B // 3), (B - B // 3, B)]
            eng.prepare_shared(x, coords, grp.share)
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            calls, parts = [], []
            for (a, b), st, gset in zip(groups, sp["streams"], sp["sets"]):
                st.wait_event(fork)
                with torch.cuda.stream(st):
                    old = eng.store.use_grad_set(*gset)
                    try:
                        parts.append(eng.forward(x, coords, genes, onehots[a:b], need_grad=True, fresh=True, clinical=clinical, share=grp.share))
                        calls.append(eng.last_call)
                    finally:
                        eng.store.use_grad_set(*old)
            for st in sp["streams"]:
                main.wait_stream(st)
            logits = torch.cat(parts, dim=0)
            ctx.split = {"groups": groups, "streams": sp["streams"], "sets": sp["sets"]}
            ctx.module, ctx.call, ctx.group, ctx.has_pred = module, calls, grp, token is not None
            ctx.batched = True
            return logits, torch.zeros((), dtype=F32, device=logits.device)
        logits = eng.forward(x, coords, genes, onehots, need_grad=need, fresh=need, clinical=clinical,
                             share=grp.share if grp is not None else None)
        ctx.module, ctx.call, ctx.group, ctx.has_pred = module, (eng.last_call if need else None), grp, token is not None
        ctx.batched = module.is_multi and logits.shape[0] > 1
        return logits.clone(), torch.zeros((), dtype=F32, device=logits.device)

    @staticmethod
    def backward(ctx, dlogits, dtoken):
        tok, grads = _bridge_backward(ctx, dlogits)
        return (None,) * 7 + (tok,) + grads


@Aggregator.register("longnetvit_gene_adapter")
class LongNetGeneAdapter(Aggregator):
    """LongNet-ViT + Modal Adapter (reference LongNetGeneAdapter, longvit_adapter.py:30-347) on the HIP engine."""
    CLINICAL = False

    def __init__(self, gene_group_defination: Dict[Any, Sequence[str]] = None, multi_task: int = 1, device="cuda",
                 weights_location: str = None, init_seed: int = None, **kwargs):
        super().__init__()
        gene_group_defination = gene_group_defination or {}
        cfg = ModelConfig.from_longnet_ctor(kwargs, multi_task=multi_task, clinical=self.CLINICAL)
        self.cfg = cfg
        self.is_multi = multi_task > 1                       # longvit_adapter.py:88 (read at TM:174)
        sizes = [len(v) for v in gene_group_defination.values()]
        self.engine = Engine(cfg, sizes, device)
        # The reference's constructor leaves a TRAINABLE model behind (longvit_adapter.py:162,176-203: every adapter / gene / head
        # module initialised, gamma = init_values) on a backbone loaded from {GIGAPATH_WEIGHT_LOC}/slide_encoder.pth when
        # `pretrained` (longvit_adapter.py:75-77; a missing file warns and keeps the random init, slide_encoder.py:317-322).
        # `weights_location` / $GIGAPATH_WEIGHT_LOC override the reference's constant; `init_seed` None = one draw from torch's
        # global RNG, i.e. torch.manual_seed() in front of the constructor fixes the model as it does for the reference.
        state = init.init_state_dict(cfg, sizes, init_seed)
        frozen = [k for k, _, _, train in self.engine.store.specs if not train]
        self.pretrained_report = init.load_slide_encoder(state, frozen, cfg.pretrained, weights_location)
        self.engine.load_state_dict(state)
        self._params: "OrderedDict[str, nn.Parameter]" = OrderedDict()
        for k, shape, kind, train in self.engine.store.specs:
            self._params[k] = nn.Parameter(self.engine.store.tensors[k], requires_grad=bool(train))
        self._trainable = OrderedDict((k, p) for k, p in self._params.items() if p.requires_grad)
        self._slots = [self.engine.store.slots[k] for k in self._trainable]      # (offset, numel, shape) in the flat buffers
        self._versions = None
        self.training_grad = True
        self._group = self._token = None
        self._spec = self._hist = self._spec_rows = None      # speculative batching of the per-task calls (_forward_one_task)
        self.speculate = True                                 # (False: every call runs on its own; the calls of a step still share one hand-over)
        from .module_graph import ModuleReplay
        self._replay = ModuleReplay(self)                     # hipGraph replay of a recurring geometry's forward / backward
        self._init_nosync()
        self.train(True)

    def _split_state(self, n: int = 2):
        """Streams and gradient sets of the pass groups a long batched pass runs as (created on first use; n > 2: experiments)."""
        if getattr(self, "_split", None) is None:
            eng = self.engine
            self._split = {"streams": [torch.cuda.Stream(device=eng.device) for _ in range(2)],
                           "sets": [(eng.store.flat_grad, eng.store.grads), eng.store.new_grad_set()]}
        while len(self._split["streams"]) < n:
            self._split["streams"].append(torch.cuda.Stream(device=self.engine.device))
            self._split["sets"].append(self.engine.store.new_grad_set())
        return self._split

    def _init_nosync(self):
        self.split_passes = os.environ.get("MT_SPLIT_PASSES", "1") not in ("0", "off")      # (see _ModelFn.forward)
        self.split_min_patches = 7500
        self._split = None
        self.nosync_after = 2              # slides served in full by the same prediction before task tokens stop being read back (0: never)
        self._nosync_rows = self._ns_eye = self._ns_stream = self._ns_seen = None
        self._ns_pending, self._ns_pins, self._ns_slide, self._streak = [], [], 0, 0

    def _open_group(self, x, coords):
        """The chain token for this call (None = first call of a new group) -- see _StepGroup.  A group is the consecutive
        grad-mode calls on the same slide tensors, at most `multi_task` of them (what one step of the reference makes)."""
        key = _slide_key(x, coords)
        grp = self._group
        if grp is not None and grp.key == key and grp.count < grp.limit and self._token is not None:
            return self._token
        self._group, self._token = _StepGroup(key, self.cfg.multi_task), None
        return None

    def _close_call(self, need, out):
        logits, tok = out
        if need and self._group is not None:
            self._token, self._group.count = tok, self._group.count + 1
        return logits

    def train(self, mode: bool = True):
        """model.train() leaves Dropout / DropPath active in the reference (frozen != eval); eval() and
        no_grad forwards run without them.  Set the config's dropout / drop_path_rate to 0 for parity comparisons."""
        super().train(mode)
        if getattr(self, "_ns_pending", None):
            self._drain_decodes(block=True)      # a mode switch ends a loop: every deferred task-token check has to have been seen
        if hasattr(self, "engine"):
            self.engine.stochastic = bool(mode) and (self.cfg.dropout > 0 or self.cfg.drop_path_rate > 0)
        return self

    # ---- parameter / state_dict surface under the reference's key names
    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True) -> Iterator[Tuple[str, nn.Parameter]]:
        for k, p in self._params.items():
            yield (prefix + ("." if prefix else "") + k, p)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else OrderedDict()
        if getattr(self, "_ns_pending", None):
            self._drain_decodes(block=True)      # (checkpointing: raise for a mis-served task token before the weights are written)
        for k, p in self._params.items():
            out[prefix + k] = p if keep_vars else p.detach()
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """nn.Module.load_state_dict's contract: strict raises on any missing / unexpected key, a non-strict load copies what
        matches and RETURNS the two lists (train_modaltune.py:546-547 prints them)."""
        have = self.engine.store.tensors
        missing = [k for k in have if k not in state_dict]
        unexpected = [k for k in state_dict if k not in have]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for {}: missing keys {}, unexpected keys {}".format(
                type(self).__name__, missing, unexpected))
        self.engine.load_state_dict(state_dict, strict=False)
        self._versions = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def _apply(self, fn, recurse=True):
        """.to() / .cuda() / .float() (train_modaltune.py:126 `.to(device)`): the tensors already live on the engine's device in their
        final dtypes (fp32 masters, derived fp16 caches), so a request for exactly that is a no-op -- anything else is REFUSED rather
        than silently ignored (the engine's buffers, caches and captured graphs cannot move)."""
        probe = fn(torch.empty(0, dtype=F32, device=self.engine.device))
        want, have = torch.device(probe.device), self.engine.device

        def index(d):
            return d.index if d.index is not None else (torch.cuda.current_device() if d.type == "cuda" and torch.cuda.is_available() else 0)
        if want.type != have.type or (want.type == "cuda" and index(want) != index(have)):
            raise RuntimeError(f"this model lives on {have} (the `device=` argument of its constructor); it cannot be moved to {want}: "
                               "construct it on the target device instead")
        if probe.dtype != F32:
            raise RuntimeError(f"the parameters are fp32 masters (fp16 operand caches are derived from them); conversion to {probe.dtype} is not supported")
        return self

    def _sync_weight_caches(self):
        v = tuple(p._version for p in self._trainable.values())
        if v != self._versions:
            self.engine.refresh_trainable_caches()
            self._versions = v

    # ---- forward (longvit_adapter.py:205-215 signature)
    def forward(self, x, coords, genes, task_token=None, attn_mask=None, multiway_split_position=None,
                incremental_state=None, clinical=None, **kwargs):
        if self.is_multi:
            if task_token is None:
                raise ValueError("task_token is required when multi_task > 1")
            return self._forward_one_task(x, coords, genes, task_token.reshape(1, -1), clinical)
        onehots = torch.zeros(1, 1, device=self.engine.device)
        return self.forward_tasks(x, coords, genes, onehots, clinical=clinical)

    @staticmethod
    def _gene_list(genes):
        if isinstance(genes, dict):
            return [genes[k] for k in sorted(genes.keys())] if all(isinstance(k, int) for k in genes) else list(genes.values())
        return genes

    def _extra_key(self):
        return None

    def _call_key(self, x, coords, genes, clinical, need):
        """Identity of "the same slide under the same weights and mode": what makes a cached task pass reusable."""
        def k(t):
            return (t.data_ptr(), t._version, tuple(t.shape)) if torch.is_tensor(t) else id(t)
        gk = k(genes) if torch.is_tensor(genes) else tuple(k(g) for g in genes)
        return (k(x), k(coords), gk, k(clinical) if self.CLINICAL else None, bool(need), bool(self.training), self._versions,
                self.engine.stochastic, self._extra_key())

    # -- speculative batching of the per-task calls (TM:156-179 `multitask_forward`: one model call per task id)
    def _forward_one_task(self, x, coords, genes, onehot, clinical):
        """The reference trainer calls the model once per task id with `torch.eye(num_tasks)[t]` (TM:174-176) and concatenates.
        Everything but the task token is shared by those calls, so once the module has SEEN a slide served with task ids
        r0, r1, ... it answers the next slide's first call (task r0) with ONE batched engine pass over all of them and hands the
        later calls their rows of that result: one B = len(rows) forward, and -- the rows being slices of one autograd output --
        one backward, exactly what the fused TrainStep runs.  A call that does not fit the prediction (other task id, repeated
        id, different tensors / weights / mode) simply runs on its own and the observed pattern is learnt afresh.

        LEARNING a pattern reads each one-hot back (one small host sync per call).  Once the same pattern has served
        `nosync_after` slides in a row the module stops reading: every slide's first call runs the batched pass over the learnt
        rows and each call takes its row by a DEVICE-side index (`_serve_nosync`) -- a host sync in front of every call would
        otherwise drain the queue three times per step and leave the GPU waiting for the host at the start of the forward and
        of the backward (round 5: 2 ms per step at L = 10 000).  The one-hots are still copied back, asynchronously, and checked
        when they have arrived (`_drain_decodes`; `train()` / `eval()` / `state_dict()` wait for the outstanding ones): a task id
        outside the learnt rows raises there (one or two calls late) and the call itself returns NaN logits (a device-side validity
        flag: never silently wrong, not even for the last call of a loop); a pattern that merely shrank or changed order sends the
        module back to learning."""
        genes = self._gene_list(genes)
        if not self.speculate:
            return self.forward_tasks(x, coords, genes, onehot, clinical=clinical)
        self._sync_weight_caches()
        need = torch.is_grad_enabled() and self.training_grad
        key = self._call_key(x, coords, genes, clinical, need)
        self._drain_decodes()
        if self._nosync_rows is not None and onehot.is_cuda:
            return self._serve_nosync(x, coords, genes, onehot, clinical, key)
        vals = onehot.detach().reshape(-1).tolist()
        row = vals.index(1.0) if (vals.count(1.0) == 1 and vals.count(0.0) == len(vals) - 1) else None
        if row is None:                       # not a one-hot: no prediction possible
            self._spec = self._hist = self._spec_rows = None
            self._streak = 0
            return self.forward_tasks(x, coords, genes, onehot, clinical=clinical)
        sp = self._spec
        if sp is not None and sp["key"] == key and row in sp["rows"] and row not in sp["used"]:
            sp["used"].add(row)
            self._hist["rows"].append(row)
            i = sp["rows"].index(row)
            return sp["logits"][i:i + 1]
        hist = self._hist
        if hist is not None and hist["key"] == key and sp is None and row not in hist["rows"]:
            hist["rows"].append(row)          # another task of the slide being learnt
        else:                                 # a new slide (or a break of the pattern): what the last one showed is the prediction
            if hist is not None:
                r = hist["rows"]
                new_rows = list(r) if (len(r) >= 2 and len(set(r)) == len(r)) else None
                # a prediction that was served in full counts towards the switch to the read-back-free mode
                full = sp is not None and new_rows is not None and new_rows == self._spec_rows and sp["used"] == set(new_rows)
                self._streak = self._streak + 1 if full else 0
                self._spec_rows = new_rows
                if self.nosync_after and self._streak >= self.nosync_after and onehot.is_cuda:
                    self._nosync_rows, self._streak = list(new_rows), 0
                    self._spec = self._hist = None
                    return self._serve_nosync(x, coords, genes, onehot, clinical, key)
            # (`hold`: the keyed objects stay alive as long as the key is compared against, so neither the caching allocator nor
            # CPython can hand their addresses / ids to another slide's tensors)
            hold = (x, coords, genes, clinical)
            self._hist = {"key": key, "rows": [row], "hold": hold}
            self._spec = None
            rows = self._spec_rows
            if rows and row == rows[0]:
                eye = torch.eye(self.cfg.multi_task, dtype=F32, device=self.engine.device)[rows]
                logits = self.forward_tasks(x, coords, genes, eye, clinical=clinical)
                self._spec = {"key": key, "rows": rows, "used": {row}, "logits": logits, "hold": hold}
                return logits[0:1]
        return self.forward_tasks(x, coords, genes, onehot, clinical=clinical)

    def _serve_nosync(self, x, coords, genes, onehot, clinical, key):
        """One call in the read-back-free mode: the slide's batched pass over the learnt rows (first call of the slide), then this
        call's row of it picked by an index computed on the device from the one-hot.  Exact (an index_select, no arithmetic on the
        logits) and differentiable; the one-hot goes to the host asynchronously for the deferred check."""
        rows = self._nosync_rows
        dev = self.engine.device
        sp = self._spec
        if sp is None or sp["key"] != key or sp.get("rows") is not rows:
            if self._ns_eye is None or self._ns_eye[0] is not rows:
                self._ns_eye = (rows, torch.eye(self.cfg.multi_task, dtype=F32, device=dev)[rows].contiguous(),
                                torch.tensor(rows, dtype=torch.long, device=dev))
            logits = self.forward_tasks(x, coords, genes, self._ns_eye[1], clinical=clinical)
            self._ns_slide += 1
            sp = self._spec = {"key": key, "rows": rows, "logits": logits, "hold": (x, coords, genes, clinical), "slide": self._ns_slide}
        oh = onehot.detach().reshape(-1)
        # --- deferred check: one-hot -> pinned host memory on a side stream (behind everything queued so far, never waited for here)
        if self._ns_stream is None:
            self._ns_stream = torch.cuda.Stream(device=dev)
        pin = self._ns_pins.pop() if self._ns_pins else torch.empty(self.cfg.multi_task, dtype=F32).pin_memory()
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self._ns_stream):
            self._ns_stream.wait_event(ready)
            pin.copy_(oh.to(F32), non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        self._ns_pending.append((done, pin, oh, rows, sp["slide"]))
        # position of this call's task id among `rows` -- and a device-side validity flag: a token that is not a one-hot of a learnt
        # row would otherwise be answered with rows[0]'s logits until the deferred check lands (which the LAST call of a loop never
        # sees): such a call returns NaN logits at once (0.0 added to the picked row otherwise: exact)
        m, idx = oh[self._ns_eye[2]].to(F32).max(dim=0)
        valid = (m == 1) & (torch.linalg.vector_norm(oh.to(F32), 1) == 1)
        poison = torch.where(valid, 0.0, float("nan"))
        return sp["logits"].index_select(0, idx.reshape(1)) + poison

    def _drain_decodes(self, block: bool = False):
        """Consume the one-hots whose asynchronous read-back has completed (never blocks unless asked to): validate the calls they
        belonged to and follow the pattern of task ids per slide."""
        pend = self._ns_pending
        while pend and (block or pend[0][0].query()):
            done, pin, _, rows, slide = pend.pop(0)
            if block:
                done.synchronize()
            vals = pin.tolist()
            self._ns_pins.append(pin)
            row = vals.index(1.0) if (vals.count(1.0) == 1 and vals.count(0.0) == len(vals) - 1) else None
            if row is None or row not in rows:
                self._nosync_rows = self._spec = self._hist = self._spec_rows = None
                self._ns_seen, self._streak = None, 0
                pend.clear()
                raise RuntimeError(
                    f"task token {vals} (passed to model(...) a few calls ago) is not a one-hot of one of the task ids {rows} this slide's "
                    "batched pass was speculated on: the logits that call returned were NOT this task's.  The module had stopped reading "
                    "task tokens back after the same ids had been served for several slides in a row; it is learning the pattern afresh "
                    "now.  Set model.nosync_after = 0 (read every task token back) or model.speculate = False (one engine pass per call) "
                    "for a loop whose task ids change between slides.")
            seen = self._ns_seen
            if seen is None or seen[0] != slide:
                if seen is not None and sorted(seen[1]) != sorted(rows):
                    # the previous slide was served with fewer / other ids than the batched pass computed: correct, but wasteful
                    self._nosync_rows = self._spec = self._hist = self._spec_rows = None
                    self._ns_seen, self._streak = None, 0
                    return
                self._ns_seen = (slide, [row])
            else:
                seen[1].append(row)

    def forward_tasks(self, x, coords, genes, task_onehots, clinical=None):
        """All task passes of one slide in one batched engine call: logits [B, output_dim]."""
        self._sync_weight_caches()
        genes = self._gene_list(genes)
        need = torch.is_grad_enabled() and self.training_grad     # (grad mode is off inside Function.forward)
        if not self.CLINICAL:
            clinical = None                      # the base adapter ignores `clinical` like the reference's **kwargs
        token = self._open_group(x, coords) if need else None
        return self._close_call(need, self._apply_bridge(x, coords, genes, task_onehots.to(self.engine.device, F32), need, clinical, token))

    def _apply_bridge(self, x, coords, genes, onehots, need, clinical, token):
        return _ModelFn.apply(self, x, coords, genes, onehots, need, clinical, token, *self._trainable.values())


@Aggregator.register("longnetvit_gene_clinical_adapter")
class LongNetGeneSimpleClinicalAdapter(LongNetGeneAdapter):
    """Clinical-prior variant (reference longvit_adapter.py:350-672): one extra token clinical_mlp(clinical[1, 5])
    in front of the task / gene tokens (T = 66), and its outcome added (sum) / concatenated (cat) in the head."""
    CLINICAL = True
