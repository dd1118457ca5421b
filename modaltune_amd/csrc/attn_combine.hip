// Dilated attention (LongNet), backward: the combine pass over the per-branch compact gradients.
// (see attn.hip for the reference semantics and the forward; split out so that the translation unit can carry its own
// LLVM scheduling strategy -- attn_common.h)
#include "attn_common.h"

namespace {

// Sum the per-branch compact gradients into the dense fp16 dqkv [B*N, 2304] that feeds the dX GEMM.
// 192 threads per token row: thread -> 12 consecutive columns of one (q|k|v, head).  The workspace is token-major
// (attn_common.h: ws_slot): the heads a branch covers at a token are one contiguous run in the source AND in the dense row, so
// the threads of a wave read consecutive addresses.
__global__ __launch_bounds__(192) void dilated_attn_bwd_combine_kernel(const h16* __restrict__ ws, Plan p, h16* __restrict__ dqkv) {
  const long M = (long)p.B * p.N;
  const int t = threadIdx.x;
  const int col = t * 12, which = col / DM, h = (col % DM) / HD, d0 = col % HD;
  for (long m = blockIdx.x; m < M; m += gridDim.x) {
    const int pos = (int)(m % p.N);
    float acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 0.f;
#pragma unroll
    for (int br = 0; br < MT_MAX_BRANCHES; ++br) {
      if (br < p.nbranch) {
        const int dr = p.ratio[br], sg = p.seg[br], hb = H / dr;
        const int j = pos / sg, loc = pos - j * sg;
        if (loc % dr == h / hb) {
          const h16* src = ws + p.ws_off[br] + ((m * 3 + which) * hb + (h % hb)) * HD + d0;
          const h16x4 a0 = *reinterpret_cast<const h16x4*>(src), a1 = *reinterpret_cast<const h16x4*>(src + 4),
                      a2 = *reinterpret_cast<const h16x4*>(src + 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[e] += (float)a0[e]; acc[4 + e] += (float)a1[e]; acc[8 + e] += (float)a2[e]; }
        }
      }
    }
    h16* dst = dqkv + m * QKV_LD + col;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const h16x4 o = {(h16)acc[4 * k], (h16)acc[4 * k + 1], (h16)acc[4 * k + 2], (h16)acc[4 * k + 3]};
      *reinterpret_cast<h16x4*>(dst + 4 * k) = o;
    }
  }
}

}  // namespace

void mt_attn::launch_bwd_combine(const void* ws, const MtDilatedPlan* plan, mt_half* dqkv, hipStream_t s) {
  const Plan p = make_plan(plan, 128);
  const long M = (long)p.B * p.N;
  hipLaunchKernelGGL(dilated_attn_bwd_combine_kernel, dim3((int)min(M, 16384L)), dim3(192), 0, s, (const h16*)ws, p, (h16*)dqkv);
}
