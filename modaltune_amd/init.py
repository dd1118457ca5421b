"""Constructor-time weights of the drop-in modules: the reference's init families and its pretrained-backbone loading.

`Aggregator.create(...)` must hand back a model the reference trainer can train as it is (train_modaltune.py:123-149 never
loads a state_dict): the reference's constructor initialises every trainable module (longvit_adapter.py:162,176-203) and
then loads the frozen slide encoder from `{GIGAPATH_WEIGHT_LOC}/slide_encoder.pth` when `pretrained` is true
(longvit_adapter.py:75-77, prov_gigapath/gigapath/slide_encoder.py:292-322), falling back to the random init with a
warning when the file is absent.  Host-side, once per model: torch-CPU draws from one seeded generator, copied to the
device by `Engine.load_state_dict`.

Families (the reference's effective init, not its RNG order -- checkpoints are what interchange):
  * LayerNorm affine (1, 0); Injector gamma = `init_values` (adapter_modules.py:357);
  * backbone + patch_embed Linears: trunc-normal(0.02), bias 0 (LongNetGeneAdapter._init_weights re-draws them through
    `self.apply` inside the base constructor, slide_encoder.py:142,160; longvit_adapter.py:184-197); cls_token normal(0.02)
    (slide_encoder.py:157);
  * Injector / Extractor attention and the prompt self-attention: xavier-uniform for every matrix
    (longvit_adapter.py:176-177,199-203 -> adapter_modules.py:59-62,177-180); their biases are 0 inside `interactions`
    (`interactions.apply(_init_weights)` zeroes the nn.Linear ones, nn.MultiheadAttention zeroes its own), while the two
    nn.Linear biases of a prompt self-attention layer keep nn.Linear's default U(+-1/sqrt(fan_in));
  * Extractor FFN Linears, gene encoder Linears, task_weight, clinical_mlp, final_project: trunc-normal(0.02), bias 0;
    the gene mixer's kernel-1 Conv1d pair is not covered by `_init_weights` and keeps torch's default
    (kaiming-uniform(a = sqrt 5) = U(+-1/sqrt(fan_in)) for weight and bias); gene_pe trunc-normal(0.02) (longvit_adapter.py:182).
"""
from __future__ import annotations

import math
import os
import warnings
from collections import OrderedDict
from typing import Dict, Optional, Sequence, Tuple

import torch

from .config import ModelConfig
from .synth import param_specs

GIGAPATH_WEIGHT_LOC = "/huggingface/hub/models--prov-gigapath--prov-gigapath/"      # utils/constants.py:15
TITAN_CODE_PATH = "/huggingface/models/models--MahmoodLab--TITAN/snapshots/"         # utils/constants.py:22
TITAN_SNAPSHOT_ID = "b2fb4f475256eb67c6e9ccbf2d6c9c3f25f20791"                       # utils/constants.py:23


def _seed_from_global_rng() -> int:
    """The reference draws its init from torch's global RNG (so `torch.manual_seed(s)` in front of the constructor fixes the
    model, utils/base_trainer.py seeds it once per run): take ONE draw from that stream as this model's seed."""
    return int(torch.randint(0, 2 ** 62, (), dtype=torch.int64))


def _uniform(shape, bound: float, g: torch.Generator) -> torch.Tensor:
    return (torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0) * bound


def _trunc_normal(shape, std: float, g: torch.Generator) -> torch.Tensor:
    """torch.nn.init.trunc_normal_(std=std) -- mean 0, cut at the ABSOLUTE bounds +-2 (its defaults; the reference never passes
    a / b), i.e. at +-100 sigma for std = 0.02: a plain normal draw, clamped.  (torch's inverse-CDF sampler is not used: its fp32
    uniform draw is inclusive at -1, so about one element in 2^24 comes out as exactly -2.0 = a 100-sigma weight.)"""
    return (torch.randn(shape, generator=g, dtype=torch.float32) * std).clamp_(-2.0, 2.0)


def _xavier_uniform(shape, g: torch.Generator) -> torch.Tensor:
    fan_out, fan_in = int(shape[0]), int(math.prod(shape[1:]))
    return _uniform(shape, math.sqrt(6.0 / (fan_in + fan_out)), g)


def _family(key: str, kind: str) -> str:
    """Which recipe a state_dict key belongs to (see the module docstring)."""
    if kind in ("lnw", "lnb", "gamma"):
        return kind
    if key == "cls_token":
        return "normal"
    if key in ("gene_pe", "gene_cls"):       # nn.init.trunc_normal_(std=0.02) (longvit_adapter.py:149,182)
        return "trunc"
    if key.startswith("interactions.") and ".attn." in key:
        return "xavier" if kind == "w" else "zero"
    if key.startswith("prompt_selfattention."):
        if kind == "w":
            return "xavier"
        return "linear_default_bias" if key.endswith(("q_proj.bias", "output_proj.bias")) and ".self_attn." not in key else "zero"
    if key.startswith("gene_encoder.mlp_mixer.") and (".0.fn.0." in key or ".0.fn.3." in key):
        return "conv_default"
    return "trunc" if kind == "w" else "zero"


def init_state_dict(cfg: ModelConfig, group_sizes: Sequence[int], seed: Optional[int] = None,
                    trainable_only: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """A freshly initialised model under the reference's state_dict names (CPU fp32 tensors).  trainable_only: the adapter-side
    keys alone (the TITAN configuration's frozen side is the backbone module's own business)."""
    g = torch.Generator().manual_seed(_seed_from_global_rng() if seed is None else int(seed))
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    fan_in_of: Dict[str, int] = {}
    specs = param_specs(cfg, group_sizes)
    for k, shape, kind, _ in specs:
        if kind == "w" and k.endswith("weight"):
            fan_in_of[k[:-len("weight")]] = int(math.prod(shape[1:]))
    for k, shape, kind, train in specs:
        if trainable_only and not train:
            continue
        fam = _family(k, kind)
        if fam == "lnw":
            t = torch.ones(shape)
        elif fam in ("lnb", "zero"):
            t = torch.zeros(shape)
        elif fam == "gamma":
            t = torch.full(shape, float(cfg.init_values))
        elif fam == "normal":
            t = torch.randn(shape, generator=g) * 0.02
        elif fam == "trunc":
            t = _trunc_normal(shape, 0.02, g)
        elif fam == "xavier":
            t = _xavier_uniform(shape, g)
        elif fam in ("conv_default", "linear_default_bias"):
            stem = k[:-len("weight")] if k.endswith("weight") else k[:-len("bias")]
            t = _uniform(shape, 1.0 / math.sqrt(max(1, fan_in_of[stem])), g)
        else:
            raise AssertionError(fam)
        out[k] = t.to(torch.float32)
    return out


def gigapath_weight_file(weights_location: Optional[str] = None) -> str:
    """`os.path.join(weights_location, "slide_encoder.pth")` (slide_encoder.py:296) with the reference's constant as the default
    location; the environment variable GIGAPATH_WEIGHT_LOC or the `weights_location` kwarg override it."""
    loc = weights_location or os.environ.get("GIGAPATH_WEIGHT_LOC") or GIGAPATH_WEIGHT_LOC
    return os.path.join(loc, "slide_encoder.pth")


def load_slide_encoder(state: "OrderedDict[str, torch.Tensor]", frozen_keys: Sequence[str], pretrained: bool,
                       weights_location: Optional[str] = None, verbose: bool = True) -> Tuple[list, list]:
    """LongNetViT.load_slide_encoder (slide_encoder.py:292-322): overwrite the frozen backbone entries of `state` with
    `torch.load(path)["model"]`, non-strictly; a missing file leaves the random init and says so.  Returns
    (missing, unexpected) as the reference prints them."""
    if not pretrained:
        return [], []
    path = gigapath_weight_file(weights_location)
    if not os.path.exists(path):
        msg = "Pretrained weights not found at {}. Randomly initialized the model!".format(path)
        warnings.warn(msg)
        if verbose:
            print("\033[93m " + msg + " \033[00m")
        return list(frozen_keys), []
    ckpt = torch.load(path, map_location="cpu")
    ckpt = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    frozen = set(frozen_keys)
    missing = [k for k in frozen_keys if k not in ckpt]
    unexpected = [k for k in ckpt if k not in frozen and k != "pos_embed"]      # (pos_embed: derived here, never stored)
    for k in frozen_keys:
        if k in ckpt:
            v = ckpt[k].detach().to(torch.float32)
            if tuple(v.shape) != tuple(state[k].shape):
                raise ValueError(f"{path}: {k} has shape {tuple(v.shape)}, the model expects {tuple(state[k].shape)}")
            state[k] = v
    if verbose:
        for k in missing:
            print("Missing ", k)
        for k in unexpected:
            print("Unexpected ", k)
        print("\033[92m Successfully Loaded Pretrained GigaPath model from {} \033[00m".format(path))
    return missing, unexpected


def build_titan_backbone(pretrained: bool, device, code_path: Optional[str] = None, snapshot_id: Optional[str] = None):
    """The reference's own way to the TITAN slide encoder (titan_adapter.py:16-37,88-107,233-247): import
    `{TITAN_SNAPSHOT_ID}.vision_transformer.VisionTransformer` and `.configuration_titan.TitanConfig` from
    `TITAN_CODE_PATH`, build it from `TitanConfig().vision_config`, and (pretrained) load the `vision_encoder.*` tensors of
    `model.safetensors`.  Returns the module, or raises ImportError / FileNotFoundError saying what is absent."""
    import importlib
    import sys
    code_path = code_path or os.environ.get("TITAN_CODE_PATH") or TITAN_CODE_PATH
    snapshot_id = snapshot_id or os.environ.get("TITAN_SNAPSHOT_ID") or TITAN_SNAPSHOT_ID
    if code_path not in sys.path:
        sys.path.append(code_path)
    try:
        vt = importlib.import_module(f"{snapshot_id}.vision_transformer")
        ct = importlib.import_module(f"{snapshot_id}.configuration_titan")
    except ImportError as e:
        raise ImportError(f"the TITAN snapshot package {snapshot_id!r} is not importable from {code_path!r} ({e}); set TITAN_CODE_PATH / "
                          f"TITAN_SNAPSHOT_ID or pass backbone=<VisionTransformer instance>") from e
    vc = ct.TitanConfig().vision_config
    vit = vt.VisionTransformer(grid_size=vc.grid_size, global_pool=vc.global_pool, embed_dim=vc.embed_dim, depth=vc.depth,
                               num_heads=vc.num_heads, mlp_ratio=vc.mlp_ratio, qkv_bias=vc.qkv_bias,
                               mlp_patch_embed_dim=vc.mlp_patch_embed_dim, pos_encode_type=vc.pos_encode_type,
                               attentional_pool=vc.attentional_pool, attn_pooler_queries=vc.attn_pooler_queries,
                               attn_pooler_heads=vc.attn_pooler_heads)
    if pretrained:
        from safetensors import safe_open
        path = os.path.join(code_path, snapshot_id, "model.safetensors")
        if not os.path.exists(path):
            raise FileNotFoundError(f"pretrained=True but {path} does not exist (titan_adapter.py:233-247 loads it unconditionally)")
        tensors = {}
        with safe_open(path, framework="pt", device="cpu") as f:
            for k in f.keys():
                if "vision_encoder" in k:
                    tensors[k.split("vision_encoder.")[1]] = f.get_tensor(k)
        res = vit.load_state_dict(tensors)            # strict, as titan_adapter.py:244 (the module is the bare backbone at this point)
        print(f"Missing keys: {list(res.missing_keys)} ")
        print(f"Unexpected keys: {list(res.unexpected_keys)} ")
    for p in vit.parameters():
        p.requires_grad = False
    return vit.to(device)
