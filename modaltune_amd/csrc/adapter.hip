// Modal-Adapter attention cores (12 heads x 16, E = 192):
//   inject  : every patch row attends over the T <= 128 modal tokens of its pass       (AM:225-229 in AM:359-369)
//   extract : the T modal tokens attend over the L patch rows of their pass (split-L)   (AM:225-229 in AM:321-335)
//   token   : T x T self-attention among the modal tokens                               (AM:87)
// These are tiny-FLOP, HBM/latency-bound ops; they run on the VALU in fp32 with the small operand (the token
// side) resident in LDS, one streaming pass over the patch-side operand.
#include "common.h"

namespace {

constexpr int AH = 12, AD = 16, AE = 192, TMAX = 128;
constexpr float ASCALE = 0.25f;   // 1/sqrt(16)

MT_DEVINL void load16(const h16* p, float* out) {
  const h16x8 a = ldg8(p), b = ldg8(p + 8);
#pragma unroll
  for (int i = 0; i < 8; ++i) { out[i] = (float)a[i]; out[8 + i] = (float)b[i]; }
}
MT_DEVINL void store16(h16* p, const float* v) {
  h16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (h16)v[i]; b[i] = (h16)v[8 + i]; }
  stg8(p, a); stg8(p + 8, b);
}

// ---------------------------------------------------------------- injector -----------------------
// forward on MFMA: grid (ceil(rows/128), 12 heads, B passes); wave = 32 patch rows (row = lane & 31).  With d = 16 one
// v_mfma_f32_32x32x16_f16 is a whole 32-token x 32-row score block: S^T = K . Q^T (K rows from an fp16 LDS image, Q^T
// straight from global memory), softmax lane-local (row = lane, tokens in registers + one cross-half shuffle),
// O^T += V^T . P^T with P^T taken from the score accumulators (V^T from a transposed fp16 LDS image).
constexpr int KP = 24;            // halves per row of the row-read K image (48 B: conflict-free 16-lane ds_read_b128 groups)
constexpr int TPV = TMAX + 8;     // halves per row of the transposed [16][tokens] images (272 B)
MT_DEVINL h16x8 cat8h(h16x4 lo, h16x4 hi) { return (h16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

__global__ __launch_bounds__(256) void inject_attn_fwd_kernel(const h16* __restrict__ q, int rows_per_pass, const float* __restrict__ k,
                                                              const float* __restrict__ v, int T, h16* __restrict__ a,
                                                              float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) h16 ksh[TMAX * KP];     // K[t][d]
  __shared__ __attribute__((aligned(16))) h16 vT[AD * TPV];       // V^T[d][t]
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < TMAX * AD; i += 256) {
    const int t = i / AD, d = i % AD;
    const bool ok = t < T;
    ksh[t * KP + d] = (h16)(ok ? k[((long)b * T + t) * AE + h * AD + d] : 0.f);
    vT[d * TPV + t] = (h16)(ok ? v[((long)b * T + t) * AE + h * AD + d] : 0.f);
  }
  __syncthreads();
  const int ntb = (T + 31) / 32;
  const int r = blockIdx.x * 128 + wave * 32 + l31;
  const bool valid = r < rows_per_pass;
  const long m = (long)b * rows_per_pass + (valid ? r : 0);
  const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const h16x8 qf = valid ? ldg8(q + m * AE + h * AD + 8 * hh) : zero8;     // B operand: Q^T[d = 8 hh + j][row]
  f32x16 sc[TMAX / 32];
  float mx = -1.0e30f;
#pragma unroll
  for (int tb = 0; tb < TMAX / 32; ++tb) {
    if (tb < ntb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[tb][i] = 0.f;
      const h16x8 kf = *reinterpret_cast<const h16x8*>(&ksh[(tb * 32 + l31) * KP + 8 * hh]);
      sc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf, sc[tb], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {      // accumulator rows are tokens: (i&3) + 8 (i>>2) + 4 hh
        if (tb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh >= T) sc[tb][i] = -1.0e30f;
        mx = fmaxf(mx, sc[tb][i]);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float c = ASCALE * 1.4426950408889634f;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float l = 0.f;
#pragma unroll
  for (int tb = 0; tb < TMAX / 32; ++tb) {
    if (tb < ntb) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        h16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pv = __builtin_amdgcn_exp2f((sc[tb][8 * s2 + e] - mx) * c);
          l += pv;
          pf[e] = (h16)pv;
        }
        // A operand: V^T[d = lane & 15][token 16 s2 + 8 (e>>2) + 4 hh + (e&3)] (the accumulator-order k permutation)
        const h16* vr = &vT[(l31 & 15) * TPV + tb * 32 + 16 * s2 + 4 * hh];
        const h16x8 vf = cat8h(*reinterpret_cast<const h16x4*>(vr), *reinterpret_cast<const h16x4*>(vr + 8));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, acc, 0, 0, 0);
      }
    }
  }
  l += __shfl_xor(l, 32, 64);
  if (valid) {
    const float inv = 1.0f / l;
    // O^T rows d = (i&3) + 8 (i>>2) + 4 hh, valid for i < 8: this lane holds d = 4 hh + {0..3} and 8 + 4 hh + {0..3}
    h16* dst = a + m * AE + h * AD;
    *reinterpret_cast<h16x4*>(dst + 4 * hh) = (h16x4){(h16)(acc[0] * inv), (h16)(acc[1] * inv), (h16)(acc[2] * inv), (h16)(acc[3] * inv)};
    *reinterpret_cast<h16x4*>(dst + 8 + 4 * hh) = (h16x4){(h16)(acc[4] * inv), (h16)(acc[5] * inv), (h16)(acc[6] * inv), (h16)(acc[7] * inv)};
    if (lse && hh == 0) lse[m * AH + h] = mx * ASCALE + __logf(l);
  }
}

// backward: workgroup = 4 waves = ITILES x 128 rows of one (pass, head); everything on MFMA.
// Phase 1 (wave = 32 rows, row = lane): S^T = K . Q^T and dP^T = V . dA^T (one 32x32x16 MFMA per 32-token block each),
// p = exp(s/4 - lse) (lse saved by the forward, same fp16 K), delta = a . da (flash identity), ds = p (dp - delta) / 4,
// dQ^T += K^T . dS^T with dS^T taken from the accumulators; p / ds / q / da go to LDS TRANSPOSED ([token][row], [dim][row]).
// Phase 2: dk[t,d] += sum_rows ds[row,t] q[row,d] and dv[t,d] += sum_rows p[row,t] da[row,d] are 32x32x16 products with
// the row index as the reduction dimension -- both operands are plain 16-byte row reads of the transposed images
// (wave w: product w & 1, token blocks w >> 1 and (w >> 1) + 2).  The accumulators live across the row tiles: one
// atomic per (token, dim) per workgroup at the very end.
constexpr int IBR = 128, ITILES = 4, RSTR = IBR + 8;   // RSTR: halves per transposed row (272 B: conflict-free b128)
__global__ __launch_bounds__(256) void inject_attn_bwd_kernel(const h16* __restrict__ q, const h16* __restrict__ a,
                                                              const float* __restrict__ lse, const h16* __restrict__ da,
                                                              int rows_per_pass, const float* __restrict__ k,
                                                              const float* __restrict__ v, int T, h16* __restrict__ dq,
                                                              float* __restrict__ dk, float* __restrict__ dv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ntb = (T + 31) / 32, TB = ntb * 32;
  h16* ksh = reinterpret_cast<h16*>(smem);            // [TB][KP]   K rows
  h16* vsh = ksh + TB * KP;                           // [TB][KP]   V rows
  h16* kT = vsh + TB * KP;                            // [16][TPV]  K^T
  h16* psT = kT + AD * TPV;                           // [T][RSTR]
  h16* dssT = psT + T * RSTR;                         // [T][RSTR]
  h16* qT = dssT + T * RSTR;                          // [16][RSTR]
  h16* daT = qT + AD * RSTR;                          // [16][RSTR]
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < TB * AD; i += 256) {
    const int t = i / AD, d = i % AD;
    const bool ok = t < T;
    const h16 kv_ = (h16)(ok ? k[((long)b * T + t) * AE + h * AD + d] : 0.f);
    ksh[t * KP + d] = kv_;
    kT[d * TPV + t] = kv_;
    vsh[t * KP + d] = (h16)(ok ? v[((long)b * T + t) * AE + h * AD + d] : 0.f);
  }
  f32x16 acc[TMAX / 64];            // phase 2: token blocks (wave >> 1) and (wave >> 1) + 2 of product (wave & 1)
#pragma unroll
  for (int j = 0; j < TMAX / 64; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const float c = ASCALE * 1.4426950408889634f;
  __syncthreads();
  for (int tile = 0; tile < ITILES; ++tile) {
    const int r0 = (blockIdx.x * ITILES + tile) * IBR;
    if (r0 >= rows_per_pass) break;                   // uniform
    const int rl = wave * 32 + l31;                   // row inside the tile
    const int r = r0 + rl;
    const bool valid = r < rows_per_pass;
    const long m = (long)b * rows_per_pass + (valid ? r : 0);
    const h16x8 qf = valid ? ldg8(q + m * AE + h * AD + 8 * hh) : zero8;
    const h16x8 daf = valid ? ldg8(da + m * AE + h * AD + 8 * hh) : zero8;
    const h16x8 af = ldg8(a + m * AE + h * AD + 8 * hh);
    const float nl2 = valid ? -lse[m * AH + h] * 1.4426950408889634f : -1.0e30f;
    float delta = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      delta = fmaf((float)af[e], (float)daf[e], delta);
      qT[(8 * hh + e) * RSTR + rl] = qf[e];
      daT[(8 * hh + e) * RSTR + rl] = daf[e];
    }
    delta += __shfl_xor(delta, 32, 64);
    f32x16 dqa;
#pragma unroll
    for (int i = 0; i < 16; ++i) dqa[i] = 0.f;
#pragma unroll
    for (int tb = 0; tb < TMAX / 32; ++tb) {
      if (tb < ntb) {
        f32x16 sc, dpv;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sc[i] = 0.f; dpv[i] = 0.f; }
        const h16x8 kf = *reinterpret_cast<const h16x8*>(&ksh[(tb * 32 + l31) * KP + 8 * hh]);
        const h16x8 vf = *reinterpret_cast<const h16x8*>(&vsh[(tb * 32 + l31) * KP + 8 * hh]);
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf, sc, 0, 0, 0);
        dpv = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, daf, dpv, 0, 0, 0);
        h16x8 dsf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = tb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;      // accumulator rows are tokens
          const float pv = t < T ? __builtin_amdgcn_exp2f(fmaf(sc[i], c, nl2)) : 0.f;
          const float ds = pv * (dpv[i] - delta) * ASCALE;
          dsf[i >> 3][i & 7] = (h16)ds;
          if (t < T) {
            psT[t * RSTR + rl] = (h16)pv;
            dssT[t * RSTR + rl] = (h16)ds;
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const h16* kr = &kT[(l31 & 15) * TPV + tb * 32 + 16 * s2 + 4 * hh];
          const h16x8 ktf = cat8h(*reinterpret_cast<const h16x4*>(kr), *reinterpret_cast<const h16x4*>(kr + 8));
          dqa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ktf, dsf[s2], dqa, 0, 0, 0);
        }
      }
    }
    if (valid) {
      h16* dst = dq + m * AE + h * AD;
      *reinterpret_cast<h16x4*>(dst + 4 * hh) = (h16x4){(h16)dqa[0], (h16)dqa[1], (h16)dqa[2], (h16)dqa[3]};
      *reinterpret_cast<h16x4*>(dst + 8 + 4 * hh) = (h16x4){(h16)dqa[4], (h16)dqa[5], (h16)dqa[6], (h16)dqa[7]};
    }
    __syncthreads();
    const h16* Asrc = (wave & 1) ? psT : dssT;
    const h16* Bsrc = (wave & 1) ? daT : qT;
#pragma unroll
    for (int j = 0; j < TMAX / 64; ++j) {
      const int tb = (wave >> 1) + 2 * j;
      if (tb < ntb) {
        const int trow = min(tb * 32 + l31, T - 1);   // rows >= T: valid memory, results never flushed
#pragma unroll
        for (int kk = 0; kk < IBR / 16; ++kk) {
          const h16x8 afr = *reinterpret_cast<const h16x8*>(&Asrc[trow * RSTR + kk * 16 + hh * 8]);
          const h16x8 bfr = *reinterpret_cast<const h16x8*>(&Bsrc[(l31 & 15) * RSTR + kk * 16 + hh * 8]);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr, bfr, acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // accumulator rows are tokens: row(i) = (i&3) + 8 (i>>2) + 4 hh; column = lane & 31 = dim (16 valid)
  float* dst = (wave & 1) ? dv : dk;
  if (l31 < AD) {
#pragma unroll
    for (int j = 0; j < TMAX / 64; ++j) {
      const int tb = (wave >> 1) + 2 * j;
      if (tb < ntb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = tb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (t < T) atomicAdd(&dst[((long)b * T + t) * AE + h * AD + l31], acc[j][i]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- extractor ----------------------
// forward on MFMA: grid (nsplit, 12, B); 4 waves, wave w sweeps the 32-key blocks w, w + 4, ... of the split.
// S^T[key, token] = K . Q^T: K rows straight from global memory (A operand, key = lane & 31), Q^T (pre-scaled, fp16) in
// registers, one 32x32x16 MFMA per 32-token block; online softmax lane-local (token = lane, keys in registers + one
// cross-half shuffle); O^T[d, token] += V^T . P^T with P^T from the accumulators and V^T read transposed
// (ds_read_b64_tr_b16) from a wave-private LDS copy of the 32 x 16 V block.  The four waves' partials are merged
// through LDS into one partial per (split, token) for the reduce kernel below.
constexpr int EKT = 32, EFT = 256, VP = 24;       // VP: halves per LDS row of the V block (48 B, 8-byte aligned tr reads)
MT_DEVINL h16x4 ad_tr4(const h16* p) {
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)p);
  return __builtin_bit_cast(h16x4, r);
}
__global__ __launch_bounds__(EFT) void extract_attn_fwd_kernel(const float* __restrict__ q, const h16* __restrict__ kv, int T, int L,
                                                               int keys_per_split, float* __restrict__ part_acc, float* __restrict__ part_ml) {
  __shared__ __attribute__((aligned(16))) h16 vsh[4][32 * VP];
  __shared__ float mrg[4][TMAX][AD + 2];
  const int sp = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int nsplit = gridDim.x;
  const int lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  const int li = lane & 15, tq = li >> 2, tp = li & 3;
  const int ntb = (T + 31) / 32;
  h16x8 qf[TMAX / 32];              // B operands: Q^T[d = 8 hh + j][token]
  f32x16 acc[TMAX / 32];
  float mrun[TMAX / 32], lrun[TMAX / 32];
#pragma unroll
  for (int tb = 0; tb < TMAX / 32; ++tb) {
    const int t = tb * 32 + l31;
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[tb][e] = (h16)(t < T ? q[((long)b * T + t) * AE + h * AD + 8 * hh + e] * ASCALE : 0.f);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[tb][i] = 0.f;
    mrun[tb] = -1.0e30f; lrun[tb] = 0.f;
  }
  const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const int kbeg = sp * keys_per_split, kend = min(L, kbeg + keys_per_split);
  h16* vw = vsh[wave];
  for (int k0 = kbeg + wave * EKT; k0 < kend; k0 += 4 * EKT) {
    const int key = k0 + l31;
    const bool valid = key < kend;
    const h16* row = kv + ((long)b * L + (valid ? key : kbeg)) * (2 * AE) + h * AD + 8 * hh;
    const h16x8 kf = valid ? ldg8(row) : zero8;
    const h16x8 vf = valid ? ldg8(row + AE) : zero8;
    *reinterpret_cast<h16x8*>(&vw[l31 * VP + 8 * hh]) = vf;      // wave-private: no workgroup barrier needed
    // V^T fragments (A operand of O^T += V^T . P^T), shared by the token blocks; lanes >= 16 of a half produce the unused
    // d rows 16..31 from the same columns
    h16x8 vt[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const h16* vr = &vw[(s2 * 16 + 4 * hh + tq) * VP + 4 * tp];
      const h16x4 lo = ad_tr4(vr), hi = ad_tr4(vr + 8 * VP);
      vt[s2] = (h16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int tb = 0; tb < TMAX / 32; ++tb) {
      if (tb < ntb) {
        f32x16 sc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = 0.f;
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[tb], sc, 0, 0, 0);
        float mx = -1.0e30f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {      // accumulator rows are keys (i&3) + 8 (i>>2) + 4 hh
          if (k0 + (i & 3) + 8 * (i >> 2) + 4 * hh >= kend) sc[i] = -1.0e30f;
          mx = fmaxf(mx, sc[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(mrun[tb], mx);
        const float al = __expf(mrun[tb] - mn);
        float ls = 0.f;
        h16x8 pf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float pv = __expf(sc[i] - mn);
          ls += pv;
          pf[i >> 3][i & 7] = (h16)pv;
        }
        ls += __shfl_xor(ls, 32, 64);
        lrun[tb] = lrun[tb] * al + ls;
        mrun[tb] = mn;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[tb][i] *= al;
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vt[0], pf[0], acc[tb], 0, 0, 0);
        acc[tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vt[1], pf[1], acc[tb], 0, 0, 0);
      }
    }
  }
  // per-wave partials -> LDS: O^T rows d = (i&3) + 8 (i>>2) + 4 hh (i < 8), column = token
#pragma unroll
  for (int tb = 0; tb < TMAX / 32; ++tb) {
    const int t = tb * 32 + l31;
    if (tb < ntb && t < T) {
#pragma unroll
      for (int i = 0; i < 8; ++i) mrg[wave][t][(i & 3) + 8 * (i >> 2) + 4 * hh] = acc[tb][i];
      if (hh == 0) { mrg[wave][t][AD] = mrun[tb]; mrg[wave][t][AD + 1] = lrun[tb]; }
    }
  }
  __syncthreads();
  if (tid < T) {
    float m2 = -1.0e30f;
    for (int wv = 0; wv < 4; ++wv) m2 = fmaxf(m2, mrg[wv][tid][AD]);
    float l2 = 0.f, a2[AD];
#pragma unroll
    for (int d = 0; d < AD; ++d) a2[d] = 0.f;
    for (int wv = 0; wv < 4; ++wv) {
      const float wgt = __expf(mrg[wv][tid][AD] - m2);
      l2 = fmaf(wgt, mrg[wv][tid][AD + 1], l2);
#pragma unroll
      for (int d = 0; d < AD; ++d) a2[d] = fmaf(wgt, mrg[wv][tid][d], a2[d]);
    }
    const long o = (((long)b * AH + h) * nsplit + sp) * T + tid;
#pragma unroll
    for (int d = 0; d < AD; ++d) part_acc[o * AD + d] = a2[d];
    part_ml[o * 2] = m2; part_ml[o * 2 + 1] = l2;
  }
}

__global__ void extract_attn_reduce_kernel(const float* __restrict__ part_acc, const float* __restrict__ part_ml, int T, int nsplit,
                                           float* __restrict__ out, float* __restrict__ lse) {
  // grid (B*12), block T threads (<=128): thread = token
  const int bh = blockIdx.x, b = bh / AH, h = bh % AH, t = threadIdx.x;
  if (t >= T) return;
  float mx = -1.0e30f;
  for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, part_ml[(((long)bh * nsplit + s) * T + t) * 2]);
  float l = 0.f, acc[AD];
#pragma unroll
  for (int d = 0; d < AD; ++d) acc[d] = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const long o = ((long)bh * nsplit + s) * T + t;
    const float w = __expf(part_ml[o * 2] - mx);
    l += w * part_ml[o * 2 + 1];
#pragma unroll
    for (int d = 0; d < AD; ++d) acc[d] = fmaf(w, part_acc[o * AD + d], acc[d]);
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int d = 0; d < AD; ++d) out[((long)b * T + t) * AE + h * AD + d] = acc[d] * inv;
  lse[((long)b * T + t) * AH + h] = mx + __logf(l);
}

// backward on MFMA: workgroup = 4 waves = ETILES x 128 keys of one (pass, head); wave = 32 keys (key = lane & 31).
// Phase 1: S[t,key] = Q . K^T and dP[t,key] = dO . V^T, one 32x32x16 MFMA per 32-token block each, with -lse[t] and
// -delta[t] as the initial accumulators (rows of the accumulators are tokens); p = exp(S'), ds = p dP';
// dK^T[d,key] += Q^T . dS and dV^T[d,key] += dO^T . P with dS / P taken from the accumulators (Q^T, dO^T from
// transposed fp16 LDS images); ds and k go to LDS transposed.  Phase 2: dq[t,d] += sum_keys ds[key,t] k[key,d] (wave w:
// token block w); one atomic per (token, dim) per workgroup at the end.  q (pre-scaled by 1/4) and dout are rounded to
// fp16 for the MFMA, as in the forward.
constexpr int EBK = 128, ETILES = 4;
__global__ __launch_bounds__(256) void extract_attn_bwd_kernel(const float* __restrict__ q, const h16* __restrict__ kv,
                                                               const float* __restrict__ out, const float* __restrict__ lse,
                                                               const float* __restrict__ dout, int T, int L, float* __restrict__ dq,
                                                               h16* __restrict__ dkv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ntb = (T + 31) / 32, TB = ntb * 32;
  float* nls = smem;                                  // [TB]  -lse (natural log) ; big negative past T
  float* ndl = nls + TMAX;                            // [TB]  -delta
  h16* qsh = reinterpret_cast<h16*>(ndl + TMAX);      // [TB][KP]   Q rows (scaled)
  h16* dosh = qsh + TB * KP;                          // [TB][KP]   dO rows
  h16* qT = dosh + TB * KP;                           // [16][TPV]  Q^T
  h16* doT = qT + AD * TPV;                           // [16][TPV]  dO^T
  h16* dssT = doT + AD * TPV;                         // [T][RSTR]
  h16* kT = dssT + T * RSTR;                          // [16][RSTR]
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, hh = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < TB * AD; i += 256) {
    const int t = i / AD, d = i % AD;
    const bool ok = t < T;
    const h16 qv_ = (h16)(ok ? q[((long)b * T + t) * AE + h * AD + d] * ASCALE : 0.f);
    const h16 dv_ = (h16)(ok ? dout[((long)b * T + t) * AE + h * AD + d] : 0.f);
    qsh[t * KP + d] = qv_; qT[d * TPV + t] = qv_;
    dosh[t * KP + d] = dv_; doT[d * TPV + t] = dv_;
  }
  for (int t = tid; t < TB; t += 256) {
    float d = 0.f, l = 1.0e30f;
    if (t < T) {
      l = lse[((long)b * T + t) * AH + h];
      for (int e = 0; e < AD; ++e) d = fmaf(dout[((long)b * T + t) * AE + h * AD + e], out[((long)b * T + t) * AE + h * AD + e], d);
    }
    nls[t] = -l; ndl[t] = -d;
  }
  f32x16 accq;                      // phase 2: token block `wave`
#pragma unroll
  for (int i = 0; i < 16; ++i) accq[i] = 0.f;
  const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
  for (int tile = 0; tile < ETILES; ++tile) {
    const int k0 = (blockIdx.x * ETILES + tile) * EBK;
    if (k0 >= L) break;             // uniform
    const int kl = wave * 32 + l31;
    const int key = k0 + kl;
    const bool valid = key < L;
    const h16* kvrow = kv + ((long)b * L + (valid ? key : 0)) * (2 * AE) + h * AD + 8 * hh;
    const h16x8 kf = valid ? ldg8(kvrow) : zero8;             // B operands: K^T[d = 8 hh + j][key], V^T likewise
    const h16x8 vf = valid ? ldg8(kvrow + AE) : zero8;
#pragma unroll
    for (int e = 0; e < 8; ++e) kT[(8 * hh + e) * RSTR + kl] = kf[e];
    f32x16 dka, dva;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dka[i] = 0.f; dva[i] = 0.f; }
#pragma unroll
    for (int tb = 0; tb < TMAX / 32; ++tb) {
      if (tb < ntb) {
        f32x16 sc, dpv;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {      // accumulator rows are tokens (i&3) + 8 (i>>2) + 4 hh
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(&nls[tb * 32 + 8 * g4 + 4 * hh]);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(&ndl[tb * 32 + 8 * g4 + 4 * hh]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { sc[4 * g4 + e] = l4[e]; dpv[4 * g4 + e] = d4[e]; }
        }
        const h16x8 qa = *reinterpret_cast<const h16x8*>(&qsh[(tb * 32 + l31) * KP + 8 * hh]);
        const h16x8 da = *reinterpret_cast<const h16x8*>(&dosh[(tb * 32 + l31) * KP + 8 * hh]);
        sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa, kf, sc, 0, 0, 0);
        dpv = __builtin_amdgcn_mfma_f32_32x32x16_f16(da, vf, dpv, 0, 0, 0);
        h16x8 pf[2], dsf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int t = tb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          const float pv = valid ? __expf(sc[i]) : 0.f;       // rows past T carry -1e30 -> 0
          const float ds = pv * dpv[i];
          pf[i >> 3][i & 7] = (h16)pv;
          dsf[i >> 3][i & 7] = (h16)ds;
          if (t < T) dssT[t * RSTR + kl] = (h16)(ds * ASCALE);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int toff = (l31 & 15) * TPV + tb * 32 + 16 * s2 + 4 * hh;
          const h16x8 qtf = cat8h(*reinterpret_cast<const h16x4*>(&qT[toff]), *reinterpret_cast<const h16x4*>(&qT[toff + 8]));
          const h16x8 dtf = cat8h(*reinterpret_cast<const h16x4*>(&doT[toff]), *reinterpret_cast<const h16x4*>(&doT[toff + 8]));
          dka = __builtin_amdgcn_mfma_f32_32x32x16_f16(qtf, dsf[s2], dka, 0, 0, 0);
          dva = __builtin_amdgcn_mfma_f32_32x32x16_f16(dtf, pf[s2], dva, 0, 0, 0);
        }
      }
    }
    if (valid) {      // rows d = (i&3) + 8 (i>>2) + 4 hh of the d x key accumulators, i < 8 (qsh already carries the 1/4)
      h16* dst = dkv + ((long)b * L + key) * (2 * AE) + h * AD;
      *reinterpret_cast<h16x4*>(dst + 4 * hh) = (h16x4){(h16)dka[0], (h16)dka[1], (h16)dka[2], (h16)dka[3]};
      *reinterpret_cast<h16x4*>(dst + 8 + 4 * hh) = (h16x4){(h16)dka[4], (h16)dka[5], (h16)dka[6], (h16)dka[7]};
      *reinterpret_cast<h16x4*>(dst + AE + 4 * hh) = (h16x4){(h16)dva[0], (h16)dva[1], (h16)dva[2], (h16)dva[3]};
      *reinterpret_cast<h16x4*>(dst + AE + 8 + 4 * hh) = (h16x4){(h16)dva[4], (h16)dva[5], (h16)dva[6], (h16)dva[7]};
    }
    __syncthreads();
    if (wave < ntb) {               // uniform per wave
      const int trow = min(wave * 32 + l31, T - 1);
#pragma unroll
      for (int kk = 0; kk < EBK / 16; ++kk) {
        const h16x8 af = *reinterpret_cast<const h16x8*>(&dssT[trow * RSTR + kk * 16 + hh * 8]);
        const h16x8 bf = *reinterpret_cast<const h16x8*>(&kT[(l31 & 15) * RSTR + kk * 16 + hh * 8]);
        accq = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, accq, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (l31 < AD && wave < ntb) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (t < T) atomicAdd(&dq[((long)b * T + t) * AE + h * AD + l31], accq[i]);
    }
  }
}

// ---------------------------------------------------------------- token self-attention -----------
// grid (heads, B), 512 threads: four threads per query token (sub = tid & 3), all T <= 128 tokens in ONE sweep.  K, V and the
// T x T score matrix live in LDS: the scores are written once, normalised in place, and leave as probs [B, heads, T, T] (saved
// for the backward) in one pass.  (The first form -- a thread per query looping over the keys with the score row in GLOBAL
// memory, written and re-read three times -- ran 24 us for 36 workgroups of 65 tokens; the step launches it between dependent
// token-side products, so its latency is exposed.  256 threads = 64 tokens per sweep: 13 us at T = 65, the 65th token costs a
// whole second sweep.)
MT_DEVINL f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// Score rows of the forward: T entries padded to a multiple of four, row stride S1 = 4 x odd >= that (sixteen tokens of a wave then start
// on sixteen different 16-byte bank groups), so that the value sweep reads FOUR probabilities with one 16-byte read -- every LDS read of
// that loop is in the 8 / 16-byte banking class (common.h: no counted lgkmcnt wait may span both classes).
MT_DEVINL int mha_t4(int T) { return (T + 3) & ~3; }
MT_DEVINL int mha_s1(int T) { const int t4 = mha_t4(T); return ((t4 >> 2) & 1) ? t4 : t4 + 4; }
__global__ __launch_bounds__(512) void token_mha_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            int T, int E, int heads, float* __restrict__ out, float* __restrict__ probs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T4 = mha_t4(T), S1 = mha_s1(T);
  float* ks = smem;                 // [T][16]
  float* vs = ks + T * AD;          // [T4][16], rows T .. T4 - 1 zero
  float* ss = vs + T4 * AD;         // [T][S1], entries T .. T4 - 1 of a row zero
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, sub = tid & 3;
  for (int i = tid; i < T4 * 4; i += 512) {
    const int tt = i >> 2, c = (i & 3) * 4;
    if (tt < T) {
      const long o = ((long)b * T + tt) * E + h * AD + c;
      *reinterpret_cast<f32x4*>(ks + tt * AD + c) = ld4(k + o);
      *reinterpret_cast<f32x4*>(vs + tt * AD + c) = ld4(v + o);
    } else {
      *reinterpret_cast<f32x4*>(vs + tt * AD + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  const int t = tid >> 2;
  const bool act = t < T;
  float qv[AD];
  if (act) {
    const float* qr = q + ((long)b * T + t) * E + h * AD;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 x = ld4(qr + 4 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) qv[4 * c + e] = x[e] * ASCALE;
    }
  }
  __syncthreads();
  if (act) {
    float mx = -1.0e30f;
    for (int j = sub; j < T; j += 4) {
      const float* kr = ks + j * AD;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < AD; ++d) s = fmaf(qv[d], kr[d], s);
      ss[t * S1 + j] = s;
      mx = fmaxf(mx, s);
    }
    if (T + sub < T4) ss[t * S1 + T + sub] = 0.f;      // (the row's padding: at most one entry per thread)
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float l = 0.f;
    for (int j = sub; j < T; j += 4) { const float p = __expf(ss[t * S1 + j] - mx); ss[t * S1 + j] = p; l += p; }
    l += __shfl_xor(l, 1, 64);
    l += __shfl_xor(l, 2, 64);
    const float inv = 1.0f / l;
    float* pr = probs + (((long)b * heads + h) * T + t) * T;
    for (int j = sub; j < T; j += 4) { const float p = ss[t * S1 + j] * inv; ss[t * S1 + j] = p; pr[j] = p; }
  }
  __syncthreads();          // a row's four writers -> its four readers
  if (act) {
    // value sweep, four keys per step: one 16-byte read of the row's probabilities, four 16-byte value rows (all one banking class)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < T4; j += 4) {
      const f32x4 p4 = ld4(ss + t * S1 + j);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += p4[u] * ld4(vs + (j + u) * AD + 4 * sub);
    }
    *reinterpret_cast<f32x4*>(out + ((long)b * T + t) * E + h * AD + 4 * sub) = acc;
  }
}

// dq / dk / dv of the above from the saved probs: dP = dO V^T, dS = P (dP - rowsum(P dP)) scale, dQ = dS K, dK = dS^T Q, dV = P^T dO.
// Q, K, V, dO and dS in LDS -- and P too while both T x T images fit (PL: T <= 127; beyond that P is read from global memory);
// four threads per token, each owning four of the sixteen head dimensions in the output sweeps.
template <bool PL>
__global__ __launch_bounds__(512) void token_mha_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ probs, const float* __restrict__ dout, int T, int E,
                                                            int heads, float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qs = smem;                 // [T][16]
  float* ks = qs + T * AD;
  float* vs = ks + T * AD;
  float* dos = vs + T * AD;
  float* dss = dos + T * AD;        // [T][T + 1]  dS (already scaled)
  float* pls = dss + T * (T + 1);   // [T][T + 1]  P (PL only)
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, sub = tid & 3, S1 = T + 1;
  const float* pbase = probs + ((long)b * heads + h) * T * T;
  for (int i = tid; i < T * 4; i += 512) {
    const int tt = i >> 2, c = (i & 3) * 4;
    const long o = ((long)b * T + tt) * E + h * AD + c;
    *reinterpret_cast<f32x4*>(qs + tt * AD + c) = ld4(q + o);
    *reinterpret_cast<f32x4*>(ks + tt * AD + c) = ld4(k + o);
    *reinterpret_cast<f32x4*>(vs + tt * AD + c) = ld4(v + o);
    *reinterpret_cast<f32x4*>(dos + tt * AD + c) = ld4(dout + o);
  }
  if (PL)
    for (int i = tid; i < T * T; i += 512) pls[(i / T) * S1 + i % T] = pbase[i];
  __syncthreads();
  const int t = tid >> 2;
  const bool act = t < T;
  const float* pr = PL ? pls + t * S1 : pbase + (long)t * T;      // row t of P
  const long pst = PL ? S1 : T;
  const float* pc = PL ? pls + t : pbase + t;                      // column t of P
  if (act) {          // a thread writes and re-reads only its own entries (j = sub mod 4) here
    float dov[AD];
#pragma unroll
    for (int d = 0; d < AD; ++d) dov[d] = dos[t * AD + d];
    // One banking class of LDS reads per loop (lds_f32, common.h): dP from the 16-byte value rows first, then delta from the 4-byte
    // probabilities and the thread's own dP entries.
    for (int j = sub; j < T; j += 4) {
      const float* vr = vs + j * AD;
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < AD; ++d) dp = fmaf(dov[d], vr[d], dp);
      dss[t * S1 + j] = dp;
    }
    float delta = 0.f;
    for (int j = sub; j < T; j += 4) delta = fmaf(lds_f32(&pr[j]), lds_f32(&dss[t * S1 + j]), delta);
    delta += __shfl_xor(delta, 1, 64);
    delta += __shfl_xor(delta, 2, 64);
    for (int j = sub; j < T; j += 4) dss[t * S1 + j] = lds_f32(&pr[j]) * (lds_f32(&dss[t * S1 + j]) - delta) * ASCALE;
  }
  __syncthreads();
  if (act) {          // dQ: thread (query t, dims 4 sub ..);  dK, dV: thread (key t, dims 4 sub ..)
    f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < T; ++j) {      // (4-byte LDS reads only, as in the forward's value sweep)
      aq += lds_f32(&dss[t * S1 + j]) * lds_f32x4_by_dword(ks + j * AD + 4 * sub);
      ak += lds_f32(&dss[j * S1 + t]) * lds_f32x4_by_dword(qs + j * AD + 4 * sub);
      av += lds_f32(&pc[j * pst]) * lds_f32x4_by_dword(dos + j * AD + 4 * sub);
    }
    const long o = ((long)b * T + t) * E + h * AD + 4 * sub;
    *reinterpret_cast<f32x4*>(dq + o) = aq;
    *reinterpret_cast<f32x4*>(dk + o) = ak;
    *reinterpret_cast<f32x4*>(dv + o) = av;
  }
}

}  // namespace

extern "C" int mt_inject_attn_fwd(const mt_half* q, int M, int rows_per_pass, const float* k, const float* v, int T,
                                  mt_half* a, float* lse, mt_stream_t stream) {
  if (!q || !k || !v || !a || M <= 0 || rows_per_pass <= 0 || M % rows_per_pass || T < 1 || T > TMAX) return MT_ERR_BAD_ARG;
  const int B = M / rows_per_pass;
  hipLaunchKernelGGL(inject_attn_fwd_kernel, dim3(cdiv(rows_per_pass, 128), AH, B), dim3(256), 0, (hipStream_t)stream,
                     (const h16*)q, rows_per_pass, k, v, T, (h16*)a, lse);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_inject_attn_bwd(const mt_half* q, const mt_half* a, const float* lse, const mt_half* da, int M,
                                  int rows_per_pass, const float* k, const float* v, int T, mt_half* dq, float* dk,
                                  float* dv, mt_stream_t stream) {
  if (!q || !a || !lse || !da || !k || !v || !dq || !dk || !dv || M <= 0 || rows_per_pass <= 0 || M % rows_per_pass || T < 1 ||
      T > TMAX)
    return MT_ERR_BAD_ARG;
  const int B = M / rows_per_pass;
  const int TBk = cdiv(T, 32) * 32;
  const size_t shm = sizeof(h16) * ((size_t)2 * TBk * KP + AD * TPV + (size_t)(2 * T + 2 * AD) * RSTR);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)inject_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(inject_attn_bwd_kernel, dim3(cdiv(rows_per_pass, IBR * ITILES), AH, B), dim3(256), shm, (hipStream_t)stream,
                     (const h16*)q, (const h16*)a, lse, (const h16*)da, rows_per_pass, k, v, T, (h16*)dq, dk, dv);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_extract_attn_fwd(const float* q, const mt_half* kv, int B, int T, int L, float* out, float* lse,
                                   float* part_acc, float* part_ml, int nsplit, mt_stream_t stream) {
  if (!q || !kv || !out || !lse || !part_acc || !part_ml || B < 1 || T < 1 || T > TMAX || L < 1 || nsplit < 1) return MT_ERR_BAD_ARG;
  const int kps = cdiv(cdiv(L, nsplit), EKT) * EKT;
  if ((long)kps * (nsplit - 1) >= L && nsplit > 1) return MT_ERR_BAD_ARG;   // every split must own >= 1 key
  hipLaunchKernelGGL(extract_attn_fwd_kernel, dim3(nsplit, AH, B), dim3(EFT), 0, (hipStream_t)stream, q, (const h16*)kv, T, L,
                     kps, part_acc, part_ml);
  hipLaunchKernelGGL(extract_attn_reduce_kernel, dim3(B * AH), dim3(TMAX), 0, (hipStream_t)stream, part_acc, part_ml, T,
                     nsplit, out, lse);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_extract_attn_bwd(const float* q, const mt_half* kv, const float* out, const float* lse,
                                   const float* dout, int B, int T, int L, float* dq, mt_half* dkv, mt_stream_t stream) {
  if (!q || !kv || !out || !lse || !dout || !dq || !dkv || B < 1 || T < 1 || T > TMAX || L < 1) return MT_ERR_BAD_ARG;
  const int TBk = cdiv(T, 32) * 32;
  const size_t shm = sizeof(float) * (2 * TMAX) + sizeof(h16) * ((size_t)2 * TBk * KP + 2 * AD * TPV + (size_t)(T + AD) * RSTR);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)extract_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(extract_attn_bwd_kernel, dim3(cdiv(L, EBK * ETILES), AH, B), dim3(256), shm, (hipStream_t)stream, q,
                     (const h16*)kv, out, lse, dout, T, L, dq, (h16*)dkv);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

static bool mha_ptrs_ok(const float* const* ps, int n, int E) {
  for (int i = 0; i < n; ++i)
    if (!ps[i] || ((uintptr_t)ps[i] & 15)) return false;
  return (E & 3) == 0;
}
extern "C" int mt_token_mha_fwd(const float* q, const float* k, const float* v, int B, int T, int E, int heads,
                                float* out, float* probs, mt_stream_t stream) {
  const float* ps[] = {q, k, v, out};
  if (!mha_ptrs_ok(ps, 4, E) || !probs || B < 1 || T < 1 || T > TMAX || E != heads * AD) return MT_ERR_BAD_ARG;
  const int T4 = (T + 3) & ~3, S1 = ((T4 >> 2) & 1) ? T4 : T4 + 4;      // (mha_t4 / mha_s1 of the kernel)
  const size_t shm = sizeof(float) * ((size_t)T * AD + (size_t)T4 * AD + (size_t)T * S1);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)token_mha_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(token_mha_fwd_kernel, dim3(heads, B), dim3(512), shm, (hipStream_t)stream, q, k, v, T, E, heads, out, probs);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_token_mha_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                                int B, int T, int E, int heads, float* dq, float* dk, float* dv, mt_stream_t stream) {
  const float* ps[] = {q, k, v, dout, dq, dk, dv};
  if (!mha_ptrs_ok(ps, 7, E) || !probs || B < 1 || T < 1 || T > TMAX || E != heads * AD) return MT_ERR_BAD_ARG;
  const bool pl = sizeof(float) * (4 * T * AD + 2 * T * (T + 1)) <= 160 * 1024;      // both T x T images (dS, P) fit up to T = 127
  const size_t shm = sizeof(float) * (4 * T * AD + (pl ? 2 : 1) * T * (T + 1));
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)token_mha_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)token_mha_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (pl)
    hipLaunchKernelGGL(token_mha_bwd_kernel<true>, dim3(heads, B), dim3(512), shm, (hipStream_t)stream, q, k, v, probs, dout, T, E,
                       heads, dq, dk, dv);
  else
    hipLaunchKernelGGL(token_mha_bwd_kernel<false>, dim3(heads, B), dim3(512), shm, (hipStream_t)stream, q, k, v, probs, dout, T, E,
                       heads, dq, dk, dv);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
