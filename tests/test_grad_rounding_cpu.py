"""Where the gradient exceptions of tests/test_model_gpu.py come from -- the CPU half of the evidence.

The oracle is run in fp64 with ONLY the storage precision of the reference's GPU run emulated: operands and results of the patch-row
products rounded to fp16 as under `torch.cuda.amp.autocast` (train_modaltune.py:216; oracle.F16_PATCH_OPERANDS, straight-through, so
the gradients are the EXACT gradients of the network evaluated at the rounded forward activations; no HIP kernel, no fp16 gradient
stream is involved).  Against the reference's fp64 golden this shifts the logits by the 2-3e-4 the HIP path measures, leaves every
gradient outside the gene encoder within 0.5 %, and moves exactly the tensors the GPU tests name as exceptions -- the bias of the
mixer's first token-mixing convolution (`mlp_mixer.*.0.fn.0.bias`) and `pathway_compression.weight`, gradients of norm ~1e-4 against
a largest norm of ~0.1: sums over the 64 gene tokens that cancel almost completely (gene_encoder.py:140-158,212) -- by 0.8-2.6 %.
The GPU half (`test_gradient_exceptions_follow_the_forward_fp16_rounding`) shows the HIP gradients of those tensors lying 4 x closer
to this emulation than to the fp64 golden.  Rounding the gradient stream to fp16 as well (F16_GRAD_SCALE) changes none of the digits."""
import os

import numpy as np
import pytest
import torch

import test_oracle_golden as TOG
from oracle import modaltune_oracle as O

NAMED = ("gene_encoder.mlp_mixer.0.0.fn.0.bias", "gene_encoder.mlp_mixer.1.0.fn.0.bias", "gene_encoder.mlp_mixer.2.0.fn.0.bias",
         "gene_encoder.pathway_compression.weight")


def _emulated(path, grad_scale=0.0):
    O.F16_PATCH_OPERANDS, O.F16_GRAD_SCALE = True, grad_scale
    try:
        return TOG._run_model_case(path, torch.float64)
    finally:
        O.F16_PATCH_OPERANDS, O.F16_GRAD_SCALE = False, 0.0


@pytest.mark.parametrize("name", ["L37_d3", "L37_d3_cat", "L37_d6_pre_gp"])
def test_fp16_operand_rounding_alone_moves_the_named_gene_encoder_gradients(golden_dir, name):
    path = os.path.join(golden_dir, f"model_{name}.npz")
    g, cfg, logits, loss, grads = _emulated(path)
    shift = TOG._maxrel(logits, g["f64_logits"])
    assert 1e-4 < shift < 6e-4, shift                       # what the HIP path measures against the same golden (2.3e-4 ... 3.9e-4)
    names = [str(n) for n in g["f64_grad_names"]]
    ref = g["f64_grad_norms"]
    dev = {n: abs(float(grads[n].norm()) - r) / r for n, r in zip(names, ref) if r > 1e-9 * ref.max()}
    full = {k[len("f64_grad/"):]: float(np.linalg.norm(grads[k[len("f64_grad/"):]].numpy() - g[k]) / np.linalg.norm(g[k]))
            for k in g.files if k.startswith("f64_grad/")}
    outside = {n: d for n, d in dev.items() if not n.startswith("gene_encoder.")}
    assert max(outside.values()) < 5e-3, sorted(outside.items(), key=lambda kv: -kv[1])[:3]
    named = {n: dev[n] for n in NAMED if n in dev}
    named["gene_encoder.pathway_compression.weight"] = max(named.get("gene_encoder.pathway_compression.weight", 0.0),
                                                            full.get("gene_encoder.pathway_compression.weight", 0.0))
    worst_named = max(named.values())
    assert 8e-3 < worst_named < 3.5e-2, named                # the size of the GPU tests' named tolerances (2.5 % / 3.5 %)
    rest = {n: d for n, d in dev.items() if n.startswith("gene_encoder.") and n not in NAMED}
    assert max(rest.values()) < 1.2e-2 and max(rest.values()) < worst_named, sorted(rest.items(), key=lambda kv: -kv[1])[:3]
    for k, e in full.items():
        if k not in NAMED:
            assert e < 5e-3, (k, e)


def test_an_fp16_gradient_stream_on_top_changes_nothing(golden_dir):
    """The deviation is the FORWARD rounding seen through cancelling sums: rounding every gradient that flows back through the
    patch-row products to fp16 too (loss scale 2^15, the trainer's initial GradScaler scale) leaves it where it was."""
    path = os.path.join(golden_dir, "model_L37_d3.npz")
    g, _, _, _, a = _emulated(path)
    _, _, _, _, b = _emulated(path, grad_scale=32768.0)
    for n in NAMED:
        ra, rb = float(a[n].norm()), float(b[n].norm())
        assert abs(ra - rb) < 2e-3 * ra, (n, ra, rb)
