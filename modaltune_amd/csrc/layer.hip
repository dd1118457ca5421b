// Composite launchers: the launch list of one frozen backbone layer / block, forward and backward, behind ONE C-ABI call each
// (SURVEY §8b lists `mt_lnqkv_fwd`, `mt_dilated_attn_fwd/bwd`, `mt_mix_ln_outproj_fwd/bwd`, `mt_ffn_fwd/bwd` as the per-layer
// surface; these entries enqueue that whole sequence).  Host code only: every kernel is launched through the same extern "C"
// entry points the Python schedule calls one by one (modaltune_amd/engine.py `_layer`, titan.py `block_fwd/bwd`), in the same
// order with the same arguments, so results are bit-identical; what goes away is the Python -> ctypes hop per kernel (9 + 11
// hops per LongNet layer and step, 36 layers' worth per step: the eager schedule of short bags is bound by exactly that).
#include "common.h"

namespace {

MtGemmEpilogue epi_bias(const float* bias) {
  MtGemmEpilogue e{};
  e.bias = bias;
  return e;
}
#define MT_TRY(call)              \
  do {                            \
    const int st__ = (call);      \
    if (st__ != MT_OK) return st__; \
  } while (0)

}  // namespace

// ---------------------------------------------------------------- LongNet EncoderLayer (ENC:121-175, DA:146-262, FFN:132-143)
extern "C" int mt_longnet_layer_fwd(const MtLongNetLayerWeights* w, const MtLongNetLayerBuffers* b, const MtDilatedPlan* plan, int M,
                                    int D, int F, const float* pend_x, const mt_half* pend_branch, const MtDropout* pend_drop, int defer,
                                    float* out, const MtDropout* drop_attn, const MtDropout* drop_ffn, mt_stream_t s) {
  if (!w || !b || !plan || M < 1 || D != 768 || (!defer && !out)) return MT_ERR_BAD_ARG;
  // self_attn_layer_norm (pre-norm); with an outstanding fc2 add of the layer below: hin = x + drop(branch) on the same pass
  if (!pend_x)
    MT_TRY(mt_layernorm_fwd(b->hin, D, nullptr, MT_OUT_F32, 0, w->ln1_w, w->ln1_b, nullptr, 0, b->u16, D, nullptr, MT_OUT_F16, b->st1, M, D, s));
  else
    MT_TRY(mt_add_layernorm_fwd(pend_x, pend_branch, pend_drop, w->ln1_w, w->ln1_b, b->hin, b->u16, b->st1, M, D, s));
  MtGemmEpilogue e = epi_bias(w->b_qkv);
  MT_TRY(mt_gemm_nt_f16(b->u16, D, nullptr, w->w_qkv, M, 3 * D, D, MT_EPI_QKV_HM, &e, b->qkv, 3 * D, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_dilated_attn_fwd(b->qkv, plan, b->o_br, b->lse_br, s));
  MT_TRY(mt_dilated_mix_ln_fwd(b->o_br, b->lse_br, plan, w->inner_ln_w, w->inner_ln_b, b->u16, b->stin, b->lse_tot, s));
  e = epi_bias(w->b_out);
  MT_TRY(mt_gemm_nt_f16(b->u16, D, nullptr, w->w_out, M, D, D, MT_EPI_BIAS, &e, b->br16, D, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_add_layernorm_fwd(b->hin, b->br16, drop_attn, w->ln2_w, w->ln2_b, b->hmid, b->u16, b->st2, M, D, s));
  e = epi_bias(w->b_fc1);
  MT_TRY(mt_gemm_nt_f16(b->u16, D, nullptr, w->w_fc1, M, F, D, MT_EPI_BIAS, &e, b->a1, F, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_layernorm_fwd(b->a1, F, nullptr, MT_OUT_F16, 1, w->ffn_ln_w, w->ffn_ln_b, nullptr, 0, b->t16, F, nullptr, MT_OUT_F16, b->stf, M, F, s));
  e = epi_bias(w->b_fc2);
  if (defer) return mt_gemm_nt_f16(b->t16, F, nullptr, w->w_fc2, M, D, F, MT_EPI_BIAS, &e, b->br16, D, nullptr, MT_OUT_F16, s);
  e.resid = b->hmid; e.ldr = D;
  if (drop_ffn) e.drop = *drop_ffn;
  return mt_gemm_nt_f16(b->t16, F, nullptr, w->w_fc2, M, D, F, MT_EPI_BIAS_RESID, &e, out, D, nullptr, MT_OUT_F32, s);
}

extern "C" int mt_longnet_layer_bwd(const MtLongNetLayerWeights* w, const MtLongNetLayerBuffers* b, const MtDilatedPlan* plan, int M,
                                    int D, int F, int dh16_valid, int feeds_lower, const MtDropout* drop_attn, const MtDropout* drop_ffn,
                                    const MtDropout* drop_lower_ffn, mt_stream_t s) {
  if (!w || !b || !plan || M < 1 || D != 768) return MT_ERR_BAD_ARG;
  const MtGemmEpilogue none{};
  const mt_half* src16 = b->dh16;
  if (!dh16_valid) {      // gradient of the (dropped) FFN branch output
    MT_TRY(mt_cast_f32_to_f16(b->dh, b->dy16, (long)M * D, drop_ffn, D, s));
    src16 = b->dy16;
  }
  MT_TRY(mt_gemm_nt_f16(src16, D, nullptr, w->wt_fc2, M, F, D, MT_EPI_BIAS, &none, b->dt16, F, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_layernorm_bwd(b->dt16, F, nullptr, MT_OUT_F16, b->a1, F, nullptr, MT_OUT_F16, 1, w->ffn_ln_w, b->stf, b->da1, F, nullptr,
                          MT_OUT_F16, 0, nullptr, nullptr, nullptr, nullptr, M, F, s));
  MT_TRY(mt_gemm_nt_f16(b->da1, F, nullptr, w->wt_fc1, M, D, F, MT_EPI_BIAS, &none, b->dy16, D, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_layernorm_bwd(b->dy16, D, nullptr, MT_OUT_F16, b->hmid, D, nullptr, MT_OUT_F32, 0, w->ln2_w, b->st2, b->dh, D, nullptr, MT_OUT_F32,
                          1, nullptr, nullptr, b->dh16, drop_attn, M, D, s));
  MT_TRY(mt_gemm_nt_f16(b->dh16, D, nullptr, w->wt_out, M, D, D, MT_EPI_BIAS, &none, b->u16, D, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_dilated_mix_ln_bwd(b->u16, b->o_br, b->lse_br, b->lse_tot, plan, w->inner_ln_w, b->stin, b->dmixed, b->delta, s));
  MT_TRY(mt_dilated_attn_bwd(b->qkv, b->dmixed, b->lse_tot, b->delta, plan, b->attn_ws, b->dqkv16, MT_ATTN_BWD_ALL, s));
  MT_TRY(mt_gemm_nt_f16(b->dqkv16, 3 * D, nullptr, w->wt_qkv, M, D, 3 * D, MT_EPI_BIAS, &none, b->dy16, D, nullptr, MT_OUT_F16, s));
  return mt_layernorm_bwd(b->dy16, D, nullptr, MT_OUT_F16, b->hin, D, nullptr, MT_OUT_F32, 0, w->ln1_w, b->st1, b->dh, D, nullptr, MT_OUT_F32, 1,
                          nullptr, nullptr, feeds_lower ? b->dh16 : nullptr, feeds_lower ? drop_lower_ffn : nullptr, M, D, s);
}

// ---------------------------------------------------------------- dense pre-norm ViT block (TITAN configuration, TA:359-361)
extern "C" int mt_vit_block_fwd(const MtVitBlockWeights* w, const MtVitBlockBuffers* b, const MtDensePlan* plan, int M, int D, int F,
                                const float* pend_x, const mt_half* pend_branch, int defer, float* out, mt_stream_t s) {
  if (!w || !b || !plan || M < 1 || D != 768 || (!defer && !out)) return MT_ERR_BAD_ARG;
  if (!pend_x)
    MT_TRY(mt_layernorm_fwd_eps(b->hin, D, nullptr, MT_OUT_F32, 0, w->n1_w, w->n1_b, nullptr, 0, b->u16, D, nullptr, MT_OUT_F16, b->st1, M, D,
                                w->n1_eps, s));
  else
    MT_TRY(mt_add_layernorm_fwd_eps(pend_x, pend_branch, nullptr, w->n1_w, w->n1_b, b->hin, b->u16, b->st1, M, D, w->n1_eps, s));
  MtGemmEpilogue e = epi_bias(w->b_qkv);
  MT_TRY(mt_gemm_nt_f16(b->u16, D, nullptr, w->w_qkv, M, 3 * D, D, MT_EPI_BIAS, &e, b->qkv, 3 * D, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_dense_attn_fwd(b->qkv, plan, b->o16, b->lse, s));
  e = epi_bias(w->b_proj);
  MT_TRY(mt_gemm_nt_f16(b->o16, D, nullptr, w->w_proj, M, D, D, MT_EPI_BIAS, &e, b->br16, D, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_add_layernorm_fwd_eps(b->hin, b->br16, nullptr, w->n2_w, w->n2_b, b->hmid, b->u16, b->st2, M, D, w->n2_eps, s));
  e = epi_bias(w->b_fc1);
  MT_TRY(mt_gemm_nt_f16(b->u16, D, nullptr, w->w_fc1, M, F, D, MT_EPI_BIAS, &e, b->a1, F, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_gelu_f16_fwd(b->a1, b->t16, (long)M * F, s));
  e = epi_bias(w->b_fc2);
  if (defer) return mt_gemm_nt_f16(b->t16, F, nullptr, w->w_fc2, M, D, F, MT_EPI_BIAS, &e, b->br16, D, nullptr, MT_OUT_F16, s);
  e.resid = b->hmid; e.ldr = D;
  return mt_gemm_nt_f16(b->t16, F, nullptr, w->w_fc2, M, D, F, MT_EPI_BIAS_RESID, &e, out, D, nullptr, MT_OUT_F32, s);
}

extern "C" int mt_vit_block_bwd(const MtVitBlockWeights* w, const MtVitBlockBuffers* b, const MtDensePlan* plan, int M, int D, int F,
                                int dh16_valid, int feeds_lower, mt_stream_t s) {
  if (!w || !b || !plan || M < 1 || D != 768) return MT_ERR_BAD_ARG;
  const MtGemmEpilogue none{};
  const mt_half* src16 = b->dh16;
  if (!dh16_valid) {
    MT_TRY(mt_cast_f32_to_f16(b->dh, b->dy16, (long)M * D, nullptr, 0, s));
    src16 = b->dy16;
  }
  MT_TRY(mt_gemm_nt_f16(src16, D, nullptr, w->wt_fc2, M, F, D, MT_EPI_BIAS, &none, b->dt16, F, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_gelu_f16_bwd(b->a1, b->dt16, b->da1, (long)M * F, s));
  MT_TRY(mt_gemm_nt_f16(b->da1, F, nullptr, w->wt_fc1, M, D, F, MT_EPI_BIAS, &none, b->dy16, D, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_layernorm_bwd(b->dy16, D, nullptr, MT_OUT_F16, b->hmid, D, nullptr, MT_OUT_F32, 0, w->n2_w, b->st2, b->dh, D, nullptr, MT_OUT_F32, 1,
                          nullptr, nullptr, b->dh16, nullptr, M, D, s));
  MT_TRY(mt_gemm_nt_f16(b->dh16, D, nullptr, w->wt_proj, M, D, D, MT_EPI_BIAS, &none, b->u16, D, nullptr, MT_OUT_F16, s));
  MT_TRY(mt_dense_attn_bwd(b->qkv, b->o16, b->u16, b->lse, plan, b->delta, b->dqkv16, MT_DENSE_BWD_ALL, s));
  MT_TRY(mt_gemm_nt_f16(b->dqkv16, 3 * D, nullptr, w->wt_qkv, M, D, 3 * D, MT_EPI_BIAS, &none, b->dy16, D, nullptr, MT_OUT_F16, s));
  return mt_layernorm_bwd(b->dy16, D, nullptr, MT_OUT_F16, b->hin, D, nullptr, MT_OUT_F32, 0, w->n1_w, b->st1, b->dh, D, nullptr, MT_OUT_F32, 1,
                          nullptr, nullptr, feeds_lower ? b->dh16 : nullptr, nullptr, M, D, s);
}
