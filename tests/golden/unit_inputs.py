"""Seeded inputs of the function-level golden cases (shared by make_golden.py and the tests, so the
fixtures only need to hold the reference's OUTPUTS)."""
import numpy as np


def adapter_inputs(seed, T, L=37, D=768):
    r = np.random.Generator(np.random.PCG64([seed, T]))
    x = r.standard_normal((1, L, D))
    c = r.standard_normal((1, T, D))
    pe = 0.5 * r.standard_normal((T, D))
    return x, c, pe


def layer_inputs(seed, N, B=2, D=768):
    r = np.random.Generator(np.random.PCG64([seed, N]))
    return r.standard_normal((B, N, D))


LAYER_CASES = {"a": (37, [8, 20, 64, 256], [1, 2, 4, 8]),
               "b": (150, [16, 40, 64, 128, 256], [1, 2, 4, 8, 16]),
               "c": (131, [1024, 5792, 32768, 185363, 1048576], [1, 2, 4, 8, 16])}
ATTN_INPUT_SCALE = 3.0   # the raw dilated-attention case projects 3*x so the softmax is far from uniform
