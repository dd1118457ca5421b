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


# Sequence-parallel DilatedAttention (args.seq_parallel, dilated_attention.py:61-111): name -> (ranks W, batch B, local length,
# segment lengths, ratios).  "w2": the two long branches gather over both ranks; "w4": (32, 2) gathers inside the rank pairs
# {0,1} / {2,3}, (256, 4) over all four, (8, 1) and (16, 1) stay local (16 = the chunk itself: sl > Lloc is false).
SEQPAR_CASES = {"w2": (2, 1, 48, [16, 48, 96, 384], [1, 2, 4, 8]),
                "w4": (4, 1, 16, [8, 16, 32, 256], [1, 1, 2, 4])}


def seqpar_inputs(seed, name, H=16, d=48):
    """q, k, v [W, B, Lloc, H, d] and the output cotangent dy [W, B, Lloc, H * d] (fp64)."""
    W, B, L, _, _ = SEQPAR_CASES[name]
    r = np.random.Generator(np.random.PCG64([seed, W, L]))
    q = 0.6 * r.standard_normal((W, B, L, H, d))
    k = 0.6 * r.standard_normal((W, B, L, H, d))
    v = r.standard_normal((W, B, L, H, d))
    dy = r.standard_normal((W, B, L, H * d))
    return q, k, v, dy
