"""ctypes binding of the C-ABI library (include/modaltune_hip.h).

The product path has NO fallback: if libmodaltune_hip.so is missing or a launcher returns a negative
MtStatus, a RuntimeError is raised (the reference's error convention is Python exceptions).
torch is imported first so that the HIP runtime already mapped by PyTorch-ROCm (soname libamdhip64.so.7)
is the one the library binds to -- streams and device pointers are then shared with torch.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MODALTUNE_HIP_LIB", os.path.join(_HERE, "_C", "libmodaltune_hip.so"))   # override: diagnostic builds

P, I, L, F, D = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double


class MtRowMap(C.Structure):
    _fields_ = [("seg_rows", I), ("seg_stride", I), ("row0", I)]


class MtDropout(C.Structure):
    _fields_ = [("rng", P), ("site", C.c_uint), ("p", F), ("path_site", C.c_uint), ("path_p", F), ("rows_per_pass", I)]


class MtGemmEpilogue(C.Structure):
    _fields_ = [("bias", P), ("resid", P), ("ldr", L), ("rmap", MtRowMap), ("colscale", P),
                ("pos_table", P), ("pos_row", P), ("pos_col", P), ("drop", MtDropout)]


MT_SGEMM_MAX = 3


class MtSgemm(C.Structure):
    _fields_ = [("A", P), ("as0", L), ("as1", L), ("a_bs", L), ("B", P), ("bs0", L), ("bs1", L), ("b_bs", L),
                ("bias", P), ("bias_on_m", I), ("C", P), ("cs0", L), ("cs1", L), ("c_bs", L),
                ("M", I), ("N", I), ("K", I), ("batch", I), ("act", I), ("accumulate", I),
                ("rowsum", P), ("pre_out", P), ("resid", P), ("c_drop", MtDropout),
                ("a_aux", P), ("a_act", I), ("a_drop", MtDropout), ("resid_scale", F), ("a_ld", I)]


MT_MAX_BRANCHES = 8


class MtDilatedPlan(C.Structure):
    _fields_ = [("nbranch", I), ("N", I), ("B", I), ("seg", I * MT_MAX_BRANCHES), ("ratio", I * MT_MAX_BRANCHES),
                ("nseg", I * MT_MAX_BRANCHES), ("n", I * MT_MAX_BRANCHES), ("qlimit", I * MT_MAX_BRANCHES)]


class MtDensePlan(C.Structure):
    _fields_ = [("N", I), ("B", I), ("H", I), ("dist", P), ("nslope", P)]


class MtLongNetLayerWeights(C.Structure):
    _fields_ = [(n, P) for n in ("ln1_w", "ln1_b", "inner_ln_w", "inner_ln_b", "ln2_w", "ln2_b", "ffn_ln_w", "ffn_ln_b", "b_qkv", "b_out",
                                 "b_fc1", "b_fc2", "w_qkv", "w_out", "w_fc1", "w_fc2", "wt_qkv", "wt_out", "wt_fc1", "wt_fc2")]


class MtLongNetLayerBuffers(C.Structure):
    _fields_ = [(n, P) for n in ("hin", "hmid", "qkv", "o_br", "lse_br", "lse_tot", "a1", "st1", "stin", "st2", "stf", "u16", "br16", "t16",
                                 "dh", "dy16", "dh16", "dt16", "da1", "dmixed", "dqkv16", "delta", "attn_ws")]


class MtVitBlockWeights(C.Structure):
    _fields_ = [(n, P) for n in ("n1_w", "n1_b", "n2_w", "n2_b")] + [("n1_eps", F), ("n2_eps", F)] + \
               [(n, P) for n in ("b_qkv", "b_proj", "b_fc1", "b_fc2", "w_qkv", "w_proj", "w_fc1", "w_fc2", "wt_qkv", "wt_proj", "wt_fc1", "wt_fc2")]


class MtVitBlockBuffers(C.Structure):
    _fields_ = [(n, P) for n in ("hin", "hmid", "qkv", "o16", "lse", "a1", "st1", "st2", "u16", "br16", "t16", "dh", "dy16", "dh16", "dt16",
                                 "da1", "dqkv16", "delta")]


RM = C.POINTER(MtRowMap)
DR = C.POINTER(MtDropout)
EP = C.POINTER(MtGemmEpilogue)
PL = C.POINTER(MtDilatedPlan)
DP = C.POINTER(MtDensePlan)

# name -> argtypes (restype int unless listed in _RESTYPE)
SIGNATURES = {
    "mt_version": [],
    "mt_status_string": [I],
    "mt_build_id": [],
    "mt_gemm_nt_f16": [P, L, RM, P, I, I, I, I, EP, P, L, RM, I, P],
    "mt_gemm_tn_f16": [P, L, RM, P, L, RM, I, I, I, P, L, P, P],
    "mt_colsum_f16": [P, L, RM, I, I, P, P],
    "mt_sgemm_small": [P, L, L, L, P, L, L, L, P, I, P, L, L, L, I, I, I, I, I, I, P, P],
    "mt_sgemm_multi": [C.POINTER(MtSgemm), I, P],
    "mt_layernorm_fwd": [P, L, RM, I, I, P, P, P, I, P, L, RM, I, P, I, I, P],
    "mt_add_layernorm_fwd": [P, P, DR, P, P, P, P, P, I, I, P],
    "mt_layernorm_fwd_eps": [P, L, RM, I, I, P, P, P, I, P, L, RM, I, P, I, I, F, P],
    "mt_add_layernorm_fwd_eps": [P, P, DR, P, P, P, P, P, I, I, F, P],
    "mt_layernorm_bwd": [P, L, RM, I, P, L, RM, I, I, P, P, P, L, RM, I, I, P, P, P, DR, I, I, P],
    "mt_dilated_attn_fwd": [P, PL, P, P, P],
    "mt_dilated_mix_ln_fwd": [P, P, PL, P, P, P, P, P, P],
    "mt_dilated_mix_ln_bwd": [P, P, P, P, PL, P, P, P, P, P],
    "mt_dilated_attn_bwd_workspace_bytes": [PL],
    "mt_dilated_attn_bwd": [P, P, P, P, PL, P, P, I, P],
    "mt_gene_snn_fwd": [P, P, P, P, P, I, I, I, P, P, P, DR, P],
    "mt_gene_snn_bwd": [P, P, P, P, P, P, I, I, I, P, P, P, DR, P],
    "mt_inject_attn_fwd": [P, I, I, P, P, I, P, P, P],
    "mt_inject_attn_bwd": [P, P, P, P, I, I, P, P, I, P, P, P, P],
    "mt_extract_attn_fwd": [P, P, I, I, I, P, P, P, P, I, P],
    "mt_extract_attn_bwd": [P, P, P, P, P, I, I, I, P, P, P],
    "mt_token_mha_fwd": [P, P, P, I, I, I, I, P, P, P],
    "mt_token_mha_bwd": [P, P, P, P, P, I, I, I, I, P, P, P, P],
    "mt_cast_f32_to_f16": [P, P, L, DR, I, P],
    "mt_rng_advance": [P, P],
    "mt_dropout_f32": [P, L, RM, P, I, I, DR, P],
    "mt_droppath_rows_f32": [P, I, I, DR, P],
    "mt_cast_f16_to_f32": [P, P, L, P],
    "mt_pack_weight_f16": [P, I, I, P, I, P],
    "mt_pack_weights_f16": [P, I, P],
    "mt_act_fwd": [P, P, L, I, P],
    "mt_act_bwd": [P, P, P, L, I, P],
    "mt_axpy": [P, P, F, P, L, P],
    "mt_axpy_bcast": [P, P, F, P, L, L, P],
    "mt_fold_rows": [P, I, L, P, P],
    "mt_copy_rows_f32": [P, L, RM, P, L, RM, I, I, I, P],
    "mt_inject_resid_bwd": [P, L, RM, P, L, RM, P, P, P, L, RM, I, P, P, I, I, P],
    "mt_l2norm_rows": [P, P, I, I, P],
    "mt_distill_loss": [P, P, I, I, F, P, P, P, P],
    "mt_adamw_step": [P, P, P, P, L, D, D, D, D, D, I, P, F, P, P, P, P],
    "mt_scaler_update": [P, P, P, P, F, F, I, P],
    "mt_check_finite": [P, L, P, P],
    "mt_absmax_scale": [P, L, F, P, P],
    "mt_axpy_dev": [P, P, P, P, L, P],
    "mt_coords_to_grid": [P, I, F, I, P, P, P, P],
    "mt_mfma_probe": [P, I, I, P],
    "mt_longnet_layer_fwd": [P, P, PL, I, I, I, P, P, DR, I, P, DR, DR, P],
    "mt_longnet_layer_bwd": [P, P, PL, I, I, I, I, I, DR, DR, DR, P],
    "mt_vit_block_fwd": [P, P, DP, I, I, I, P, P, I, P, P],
    "mt_vit_block_bwd": [P, P, DP, I, I, I, I, I, P],
    "mt_alibi_dist": [P, I, P, P],
    "mt_alibi_dist_halves": [I],
    "mt_dense_attn_fwd": [P, DP, P, P, P],
    "mt_dense_attn_bwd": [P, P, P, P, DP, P, P, I, P],
    "mt_gelu_f16_fwd": [P, P, L, P],
    "mt_gelu_f16_bwd": [P, P, P, L, P],
    "mt_pool_attn_workspace_floats": [I, I, I, I],
    "mt_pool_attn_fwd": [P, P, I, I, I, I, I, P, P, P, P, P],
    "mt_pool_attn_bwd": [P, P, P, P, P, P, I, I, I, I, I, P, P],
    "mt_titan_grid": [P, I, F, P, P, P, P],
    "mt_titan_cell_sums": [P, L, P, I, I, P, P, P, P, P],
    "mt_titan_token_order": [P, P, P, I, P, P, P, P],
    "mt_titan_gather_tokens": [P, P, I, I, P, P],
    "mt_scatter_rows_f32": [P, P, P, P, I, I, I, P],
    "mt_row_absmax_f32": [P, P, I, I, P],
}
_RESTYPE = {"mt_status_string": C.c_char_p, "mt_build_id": C.c_char_p, "mt_dilated_attn_bwd_workspace_bytes": C.c_long, "mt_pool_attn_workspace_floats": C.c_long,
             "mt_alibi_dist_halves": C.c_long}

_lib = None


def load():
    """Load the library and bind every declared entry point; raises if the build is missing."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps PyTorch-ROCm's HIP runtime first)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"modaltune_amd: {LIB_PATH} is missing -- run `python __graft_entry__.py` (hipcc, gfx950) first; "
            "there is no CPU/PyTorch fallback for the hot path")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, I)
    _lib = lib
    return lib


def build_info() -> dict:
    """{"build_id": what the LOADED binary was built from, "build_id_matches_tree": whether that is the tree next to it}."""
    from ._build_id import tree_build_id
    bid = load().mt_build_id().decode()
    try:
        ok = bid == tree_build_id()
    except Exception:      # (sources not shipped next to the binary)
        ok = None
    return {"build_id": bid, "build_id_matches_tree": ok, "lib": os.path.relpath(LIB_PATH, os.path.dirname(_HERE))}


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().mt_status_string(status).decode()
        raise RuntimeError(f"modaltune_hip {what} failed: {msg} ({status})")


def rowmap(seg_rows=0, seg_stride=0, row0=0):
    return MtRowMap(seg_rows, seg_stride, row0)
