"""Import shims so the (read-only) reference at /root/reference can be imported on a CPU-only box.

Used ONLY by tests/golden/make_golden.py in the build container; nothing here travels into the
product path and nothing from /root/reference is copied.  Shims follow SURVEY.md §8c:
  * stub `timm` (register_model decorator, drop_path) and `fairscale.nn` (identity wrappers)
  * fake TITAN snapshot package (titan_adapter imports it unconditionally)
  * numpy legacy print options (str(list(np.int64)) is eval'ed by the reference config)
  * a CPU stand-in for flash_attn_func returning (out[b,l,h,d], lse[b,h,l])
"""
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def _flash_standin(q, k, v, dropout=0.0, bias=None, softmax_scale=None, is_causal=False):
    # q,k,v: [b, l, h, d]; explicit softmax attention in the input dtype, lse in fp32
    assert bias is None and not is_causal
    d = q.shape[-1]
    scale = softmax_scale if softmax_scale is not None else d ** -0.5
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    lse = torch.logsumexp(s if s.dtype == torch.float64 else s.float(), dim=-1)
    p = torch.softmax(s, dim=-1)
    out = torch.einsum("bhqk,bkhd->bqhd", p, v)
    return out, lse


def install():
    sys.dont_write_bytecode = True
    np.set_printoptions(legacy="1.25")
    if REF not in sys.path:
        sys.path.insert(0, REF)

    # --- timm -----------------------------------------------------------------------------
    timm = types.ModuleType("timm")
    timm.__path__ = []
    models = types.ModuleType("timm.models")
    models.__path__ = []
    registry = types.ModuleType("timm.models.registry")
    registry.register_model = lambda f: f
    layers = types.ModuleType("timm.models.layers")

    def drop_path(x, drop_prob: float = 0.0, training: bool = False, scale_by_keep: bool = True):
        if drop_prob == 0.0 or not training:
            return x
        keep = 1 - drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and scale_by_keep:
            mask.div_(keep)
        return x * mask

    layers.drop_path = drop_path
    timm.models = models
    timm.create_model = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("timm stub"))
    models.registry = registry
    models.layers = layers
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.registry": registry,
                        "timm.models.layers": layers})

    # --- fairscale --------------------------------------------------------------------------
    fs = types.ModuleType("fairscale")
    fs.__path__ = []
    fsnn = types.ModuleType("fairscale.nn")
    fsnn.checkpoint_wrapper = lambda m, *a, **k: m
    fsnn.wrap = lambda m, *a, **k: m
    fs.nn = fsnn
    sys.modules.update({"fairscale": fs, "fairscale.nn": fsnn})

    # --- import-only deps of the trainer module (train_modaltune.py:22,26,32): never executed here ------
    for name in ("lifelines", "lifelines.utils", "wandb", "warmup_scheduler"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["warmup_scheduler"].GradualWarmupScheduler = object
    sys.modules["lifelines"].utils = sys.modules["lifelines.utils"]

    # --- fake TITAN snapshot package ------------------------------------------------------------
    import importlib
    consts = importlib.import_module("utils.constants")
    snap = consts.TITAN_SNAPSHOT_ID
    pkg = types.ModuleType(snap)
    pkg.__path__ = []
    vt = types.ModuleType(snap + ".vision_transformer")

    # (our own stand-in with the surface titan_adapter.py uses -- tests/golden/titan_standin.py -- so that the
    # reference's TITAN adapter code can be run end to end; the real snapshot is not in the reference tree)
    import titan_standin
    vt.VisionTransformer = titan_standin.VisionTransformer
    ct = types.ModuleType(snap + ".configuration_titan")
    ct.TitanConfig = titan_standin.TitanConfig
    pkg.vision_transformer = vt
    pkg.configuration_titan = ct
    sys.modules.update({snap: pkg, snap + ".vision_transformer": vt, snap + ".configuration_titan": ct})

    # --- import the reference and patch the flash stand-in on both module identities ---------
    import models.aggregators  # noqa: F401  (reference package)
    for name, mod in list(sys.modules.items()):
        if name.endswith("component.multihead_attention") or name.endswith("component.flash_attention"):
            if hasattr(mod, "flash_attn_func"):
                mod.flash_attn_func = _flash_standin


def zero_dropout(model):
    """Parity is only defined with every stochastic op off."""
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.AlphaDropout)):
            m.p = 0.0
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
