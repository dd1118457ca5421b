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

from collections import OrderedDict
from typing import Any, Dict, Iterator, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
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


class _ModelFn(torch.autograd.Function):
    """autograd bridge: forward runs the HIP engine (own tape + workspace per call); backward replays that call's tape.
    The trainable parameters are REAL inputs of the Function and their gradients its outputs, so autograd's
    AccumulateGrad nodes run as for any module: `param.grad` accumulates over the three forward calls of a step
    (TM:175-177), torch optimisers / GradScaler see ordinary gradients, and DistributedDataParallel's reducer hooks fire
    (utils/base_trainer.py:205-211 wraps the model in DDP)."""

    @staticmethod
    def forward(ctx, module, x, coords, genes, onehots, need, clinical, *params):
        eng = module.engine
        logits = eng.forward(x, coords, genes, onehots, need_grad=need, fresh=need, clinical=clinical)
        ctx.module, ctx.call, ctx.nparams = module, (eng.last_call if need else None), len(params)
        return logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        module, eng = ctx.module, ctx.module.engine
        if ctx.call is None:
            raise RuntimeError("backward through a forward that ran with gradients disabled")
        store = eng.store
        # The tape runs an fp16 activation-gradient stream: rescale the incoming gradient to max |.| = 2^10 on the DEVICE
        # (no read-back), run the tape, and undo the scale while copying the flat gradient out.
        dl = dlogits.to(F32).contiguous()
        s = torch.empty(2, dtype=F32, device=dl.device)
        ops.absmax_scale(dl, s, 1024.0)
        scaled = torch.empty_like(dl)
        ops.axpy_dev(None, dl, s[0:1], scaled)
        hook, eng.grad_ready_hook = eng.grad_ready_hook, None      # (a TrainStep sharing the engine must not see this pass)
        store.flat_grad.zero_()
        eng.backward(scaled, call=ctx.call)
        eng.grad_ready_hook = hook
        out = torch.empty_like(store.flat_grad)
        ops.axpy_dev(None, store.flat_grad, s[1:2], out)
        grads = tuple(out[o:o + n].view(shape) for o, n, shape in module._slots)
        return (None,) * 7 + grads


@Aggregator.register("longnetvit_gene_adapter")
class LongNetGeneAdapter(Aggregator):
    """LongNet-ViT + Modal Adapter (reference LongNetGeneAdapter, longvit_adapter.py:30-347) on the HIP engine."""
    CLINICAL = False

    def __init__(self, gene_group_defination: Dict[Any, Sequence[str]] = None, multi_task: int = 1, device="cuda", **kwargs):
        super().__init__()
        gene_group_defination = gene_group_defination or {}
        cfg = ModelConfig.from_json(kwargs, multi_task=multi_task, clinical=self.CLINICAL)
        self.cfg = cfg
        self.is_multi = multi_task > 1                       # longvit_adapter.py:88 (read at TM:174)
        self.engine = Engine(cfg, [len(v) for v in gene_group_defination.values()], device)
        self._params: "OrderedDict[str, nn.Parameter]" = OrderedDict()
        for k, shape, kind, train in self.engine.store.specs:
            self._params[k] = nn.Parameter(self.engine.store.tensors[k], requires_grad=bool(train))
        self._trainable = OrderedDict((k, p) for k, p in self._params.items() if p.requires_grad)
        self._slots = [self.engine.store.slots[k] for k in self._trainable]      # (offset, numel, shape) in the flat buffers
        self._versions = None
        self.training_grad = True
        self.train(True)

    def train(self, mode: bool = True):
        """model.train() leaves Dropout / DropPath active in the reference (frozen != eval); eval() and
        no_grad forwards run without them.  Set the config's dropout / drop_path_rate to 0 for parity comparisons."""
        super().train(mode)
        if hasattr(self, "engine"):
            self.engine.stochastic = bool(mode) and (self.cfg.dropout > 0 or self.cfg.drop_path_rate > 0)
        return self

    # ---- parameter / state_dict surface under the reference's key names
    def named_parameters(self, prefix: str = "", recurse: bool = True, remove_duplicate: bool = True) -> Iterator[Tuple[str, nn.Parameter]]:
        for k, p in self._params.items():
            yield (prefix + ("." if prefix else "") + k, p)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else OrderedDict()
        for k, p in self._params.items():
            out[prefix + k] = p if keep_vars else p.detach()
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self.engine.load_state_dict(state_dict, strict=strict)
        self._versions = None
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def _apply(self, fn, recurse=True):     # .to()/.cuda()/.float(): tensors already live on the GPU in their final dtypes
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
            onehots = task_token.reshape(1, -1)
        else:
            onehots = torch.zeros(1, 1, device=self.engine.device)
        return self.forward_tasks(x, coords, genes, onehots, clinical=clinical)

    def forward_tasks(self, x, coords, genes, task_onehots, clinical=None):
        """All task passes of one slide in one batched engine call: logits [B, output_dim]."""
        self._sync_weight_caches()
        if isinstance(genes, dict):
            genes = [genes[k] for k in sorted(genes.keys())] if all(isinstance(k, int) for k in genes) else list(genes.values())
        need = torch.is_grad_enabled() and self.training_grad     # (grad mode is off inside Function.forward)
        if not self.CLINICAL:
            clinical = None                      # the base adapter ignores `clinical` like the reference's **kwargs
        return _ModelFn.apply(self, x, coords, genes, task_onehots.to(self.engine.device, F32), need, clinical,
                              *self._trainable.values())


@Aggregator.register("longnetvit_gene_clinical_adapter")
class LongNetGeneSimpleClinicalAdapter(LongNetGeneAdapter):
    """Clinical-prior variant (reference longvit_adapter.py:350-672): one extra token clinical_mlp(clinical[1, 5])
    in front of the task / gene tokens (T = 66), and its outcome added (sum) / concatenated (cat) in the head."""
    CLINICAL = True
