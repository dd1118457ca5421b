// Token-side STAGES: a row-blocked chain  [LayerNorm (+ pos rows)] -> Linear (+ act, + Dropout) [-> Linear (+ Dropout, DropPath,
// + scaled residual)]  of the modal-token path in ONE launch per direction (fp32 throughout, T <= ~200 rows).
//
// The token side of a train step is a chain of ~250 dependent launches of a few microseconds each (SelfAttentionLayer AM:81-94,
// the token half of CrossAttentionLayer AM:210-234, FFNLayer AM:284-293, the mixer feed-forwards GE:184-192, the fusion head
// LVA:341-347): the chain is bound by the NUMBER of launches.  Every op of such a stage is ROW-INDEPENDENT, so a workgroup that owns
// 16 rows can run the whole stage with workgroup barriers only -- there is no seam between workgroups anywhere (what costs as much
// as a launch, MI355X_MICROARCH.md "splitk-seam").  A workgroup = 16 waves; a wave computes whole 16 x 16 output tiles over the
// full K with v_mfma_f32_16x16x4_f32 on operands loaded straight from global memory / L2 into the MFMA lane layout (as
// sgemm_multi_kernel does); the intermediate rows go through global memory (they are the tensors the backward needs anyway) and
// are re-read by the same workgroup after a barrier.  What bounds a stage is the one CU streaming the stage's weights from L2.
//
// Backward = one launch for the activation-gradient chain (dy -> dh -> da -> LayerNorm backward, parameter gradients of the norm by
// atomics) + ONE mt_sgemm_multi launch for the weight / bias gradients of the stage's linears (dW = d^T x over the rows: those
// products are row REDUCTIONS, full-grid work, and nobody downstream waits for them).
#include "common.h"

namespace {

struct StageFwdArgs {
  int R, K0;
  const float* x;                       // [R, K0] dense
  // LayerNorm prologue (ln_w == nullptr: none, linear 1 reads x)
  const float* ln_w; const float* ln_b; float eps;
  const float* pe; int pe_period;       // rows added AFTER the affine (with_pos_embed, AM:64-65), or nullptr
  float* tn;                            // LN(x) w + b           (may be nullptr when pe is given and nobody needs the plain rows)
  float* a;                             // LN(x) w + b + pe      (== tn when pe is nullptr)
  float* stats;                         // [R, 2] mean, rstd
  // linear 1 and an optional sibling on the same input (k | v of one normed memory)
  const float* W1; const float* b1; int N1, act1; float* h; float* pre1; DropArgs drop1;
  const float* W1b; const float* b1b; int N1b; float* hb;
  // linear 2 (N2 == 0: none): y = resid_scale * resid + path(drop2(h W2^T + b2))
  const float* W2; const float* b2; int N2; const float* resid; float resid_scale; DropArgs drop2; float* y;
};

struct StageBwdArgs {
  int R, K0, N1, N1b, N2, act1;
  const float* dy; DropArgs drop2; float* dyp;         // dyp = path(drop2(dy)) (written when drop2 is active: the dW2 product reads it)
  float* dres; float resid_scale;                      // dres (+)= resid_scale * dy   (nullptr: no residual / handled by the caller)
  const float* W2; const float* pre1; DropArgs drop1;
  float* dh;                                           // with linear 2: OUT drop1(dyp W2) * act1'(pre1); without: IN (the gradient of h)
  const float* dhb;                                    // IN: gradient of the sibling's output
  const float* W1; const float* W1b;
  float* da; int da_accumulate;                        // da (+)= dh W1 (+ dhb W1b)   [R, K0]
  // LayerNorm backward (ln_w == nullptr: none)
  const float* x; const float* ln_w; const float* stats; const float* dtn_extra;
  float* dx; float* dlnw; float* dlnb; float* dpe; int pe_period;
};

MT_DEVINL float st_act(float v, int act) {
  switch (act) {
    case MT_ACT_RELU: return fmaxf(v, 0.f);
    case MT_ACT_GELU: return gelu_erf(v);
    case MT_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}
MT_DEVINL float st_act_grad(float v, int act) {
  switch (act) {
    case MT_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case MT_ACT_GELU: return gelu_erf_grad(v);
    case MT_ACT_ELU: return v > 0.f ? 1.f : __expf(v);
    default: return 1.f;
  }
}
MT_DEVINL float st_drop1(const DropArgs& d, long off) {        // element dropout factor of one element (linear index off)
  if (!(d.rng && d.p > 0.f)) return 1.f;
  return drop_keep1(d, d.site, (uint64_t)off) ? 1.f / (1.f - d.p) : 0.f;
}
MT_DEVINL float st_path(const DropArgs& d, int m) {
  if (!(d.rng && d.path_p > 0.f)) return 1.f;
  return drop_path_factor(d, m / d.rows_per_pass);
}

// acc += A[16 rows m0.., K] * B[16 rows n0.., K]^T for the calling wave: lane (r = lane & 15, kq = lane >> 4) feeds row r of either
// operand and four consecutive k per 16-wide block.  A(m, k) = A[m * lda + k] (k-contiguous, 16-byte aligned rows);
// B(n, k) = B[n * bs0 + k * bs1] (bs1 == 1: one 16-byte load, else four scalar loads coalesced across r).  Rows / columns beyond
// (M, N) are clamped (their products are never stored).  K % 4 == 0.
template <bool BK>
MT_DEVINL f32x4 st_tile(const float* __restrict__ A, long lda, int M, int m0, const float* __restrict__ B, long bs0, long bs1, int N, int n0,
                        int K, f32x4 acc) {
  const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
  const float* ap = A + (long)min(m0 + r, M - 1) * lda;
  const float* bp = B + (long)min(n0 + r, N - 1) * bs0;
  auto ldb = [&](int k) -> f32x4 {
    if (BK) return *reinterpret_cast<const f32x4*>(bp + k);
    return (f32x4){bp[(long)k * bs1], bp[(long)(k + 1) * bs1], bp[(long)(k + 2) * bs1], bp[(long)(k + 3) * bs1]};
  };
  int k0 = 0;
  constexpr int JB = BK ? 4 : 2;        // 16-wide k blocks in flight per batch (the strided operand costs an address pair per load)
  for (; k0 + 16 * JB <= K; k0 += 16 * JB) {
    f32x4 a[JB], b[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int k = k0 + 16 * j + 4 * kq;
      a[j] = *reinterpret_cast<const f32x4*>(ap + k);
      b[j] = ldb(k);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][e], b[j][e], acc, 0, 0, 0);
  }
  for (; k0 + 16 <= K; k0 += 16) {
    const int k = k0 + 4 * kq;
    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + k), b = ldb(k);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  }
  if (k0 < K) {        // ragged rest (K % 16 in {4, 8, 12}): lanes whose four k lie past K feed zeros
    const int k = k0 + 4 * kq;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (k < K) { a = *reinterpret_cast<const f32x4*>(ap + k); b = ldb(k); }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  }
  return acc;
}

constexpr int ST_WAVES = 16;

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(64 * ST_WAVES) void token_stage_fwd_kernel(StageFwdArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int r = lane & 15, kq = lane >> 4;
  // ---- phase 0: LayerNorm of this workgroup's 16 rows, one row per wave
  if (g.ln_w) {
    const int m = m0 + wave;
    if (m < g.R) {
      const float* xr = g.x + (long)m * g.K0;
      float s = 0.f;
      for (int c = lane * 4; c < g.K0; c += 256) { const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c); s += (v[0] + v[1]) + (v[2] + v[3]); }
      const float mean = wave_sum(s) / g.K0;
      float q = 0.f;
      for (int c = lane * 4; c < g.K0; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
      }
      const float rstd = rsqrtf(wave_sum(q) / g.K0 + g.eps);
      const float* per = g.pe ? g.pe + (long)(m % g.pe_period) * g.K0 : nullptr;
      for (int c = lane * 4; c < g.K0; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c), w = *reinterpret_cast<const f32x4*>(g.ln_w + c),
                    b = *reinterpret_cast<const f32x4*>(g.ln_b + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * w[e] + b[e];
        if (g.tn) *reinterpret_cast<f32x4*>(g.tn + (long)m * g.K0 + c) = o;
        if (per) { o += *reinterpret_cast<const f32x4*>(per + c); *reinterpret_cast<f32x4*>(g.a + (long)m * g.K0 + c) = o; }
      }
      if (lane == 0) { g.stats[2 * (long)m] = mean; g.stats[2 * (long)m + 1] = rstd; }
    }
    __syncthreads();
  }
  const float* ain = g.ln_w ? (g.pe ? g.a : g.tn) : g.x;
  // ---- phase 1: h = drop1(act1(a W1^T + b1)) (and the sibling hb = a W1b^T + b1b), tiles over the output columns
  const int t1 = (g.N1 + 15) >> 4, t1b = (g.N1b + 15) >> 4;
  for (int t = wave; t < t1 + t1b; t += ST_WAVES) {
    const bool sib = t >= t1;
    const int n0 = (sib ? t - t1 : t) * 16, N = sib ? g.N1b : g.N1;
    const float* W = sib ? g.W1b : g.W1;
    const float* bias = sib ? g.b1b : g.b1;
    float* out = sib ? g.hb : g.h;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = st_tile<true>(ain, g.K0, g.R, m0, W, g.K0, 1, N, n0, g.K0, acc);
    const int n = n0 + r;
    if (n < N) {
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + 4 * kq + e;
        if (m < g.R) {
          float v = acc[e] + bv;
          const long off = (long)m * N + n;
          if (!sib) {
            if (g.pre1) g.pre1[off] = v;
            v = st_act(v, g.act1) * st_drop1(g.drop1, off);
          }
          out[off] = v;
        }
      }
    }
  }
  if (g.N2 <= 0) return;
  __syncthreads();
  // ---- phase 2: y = resid_scale * resid + path(drop2(h W2^T + b2))
  const int t2 = (g.N2 + 15) >> 4;
  for (int t = wave; t < t2; t += ST_WAVES) {
    const int n0 = t * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = st_tile<true>(g.h, g.N1, g.R, m0, g.W2, g.N1, 1, g.N2, n0, g.N1, acc);
    const int n = n0 + r;
    if (n < g.N2) {
      const float bv = g.b2 ? g.b2[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + 4 * kq + e;
        if (m < g.R) {
          const long off = (long)m * g.N2 + n;
          float v = (acc[e] + bv) * st_drop1(g.drop2, off) * st_path(g.drop2, m);
          if (g.resid) v += g.resid_scale * g.resid[off];
          g.y[off] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward (activation gradients)
__global__ __launch_bounds__(64 * ST_WAVES) void token_stage_bwd_kernel(StageBwdArgs g) {
  __shared__ float red[2][ST_WAVES][64];       // LayerNorm parameter-gradient partials of one 256-column chunk
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int r = lane & 15, kq = lane >> 4;
  const float* dh_in = g.dh;
  if (g.N2 > 0) {
    // ---- phase A: dyp = path(drop2(dy)); dres += resid_scale * dy (one row per wave)
    const bool masked = g.drop2.rng && (g.drop2.p > 0.f || g.drop2.path_p > 0.f);
    const int m = m0 + wave;
    if (m < g.R && (masked || g.dres)) {
      const float pf = st_path(g.drop2, m);
      for (int c = lane; c < g.N2; c += 64) {
        const long off = (long)m * g.N2 + c;
        const float d = g.dy[off];
        if (masked) g.dyp[off] = d * st_drop1(g.drop2, off) * pf;
        if (g.dres) g.dres[off] += g.resid_scale * d;
      }
    }
    if (masked) __syncthreads();
    const float* dys = masked ? g.dyp : g.dy;
    // ---- phase B: dh = drop1(dys W2) * act1'(pre1): tiles over the hidden columns, K = N2, W2 read transposed
    const int t1 = (g.N1 + 15) >> 4;
    for (int t = wave; t < t1; t += ST_WAVES) {
      const int n0 = t * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = st_tile<false>(dys, g.N2, g.R, m0, g.W2, 1, g.N1, g.N1, n0, g.N2, acc);      // B(n = hidden j, k = out col) = W2[k * N1 + j]
      const int n = n0 + r;
      if (n < g.N1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + 4 * kq + e;
          if (m < g.R) {
            const long off = (long)m * g.N1 + n;
            float v = acc[e] * st_drop1(g.drop1, off);
            if (g.pre1) v *= st_act_grad(g.pre1[off], g.act1);
            g.dh[off] = v;
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- phase C: da (+)= dh W1 (+ dhb W1b): tiles over the K0 input columns
  const int t0 = (g.K0 + 15) >> 4;
  for (int t = wave; t < t0; t += ST_WAVES) {
    const int n0 = t * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = st_tile<false>(dh_in, g.N1, g.R, m0, g.W1, 1, g.K0, g.K0, n0, g.N1, acc);       // B(n = input col c, k = hidden j) = W1[k * K0 + c]
    if (g.dhb) acc = st_tile<false>(g.dhb, g.N1b, g.R, m0, g.W1b, 1, g.K0, g.K0, n0, g.N1b, acc);
    const int n = n0 + r;
    if (n < g.K0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + 4 * kq + e;
        if (m < g.R) {
          float* d = g.da + (long)m * g.K0 + n;
          *d = g.da_accumulate ? *d + acc[e] : acc[e];
        }
      }
    }
  }
  if (!g.ln_w) return;
  __syncthreads();
  // ---- phase D: LayerNorm backward of the 16 rows (one row per wave): dtn = da (+ dtn_extra); dx += ...; dpe[row % period] += da
  const int m = m0 + wave;
  const bool live = m < g.R;
  const float mean = live ? g.stats[2 * (long)m] : 0.f, rstd = live ? g.stats[2 * (long)m + 1] : 0.f;
  const float* xr = g.x + (long)(live ? m : 0) * g.K0;
  const float* dr = g.da + (long)(live ? m : 0) * g.K0;
  const float* er = g.dtn_extra ? g.dtn_extra + (long)(live ? m : 0) * g.K0 : nullptr;
  float s1 = 0.f, s2 = 0.f;
  if (live) {
    for (int c = lane * 4; c < g.K0; c += 256) {
      f32x4 d = *reinterpret_cast<const f32x4*>(dr + c);
      if (er) d += *reinterpret_cast<const f32x4*>(er + c);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c), w = *reinterpret_cast<const f32x4*>(g.ln_w + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float gg = d[e] * w[e]; s1 += gg; s2 += gg * (xv[e] - mean) * rstd; }
    }
  }
  const float c1 = wave_sum(s1) / g.K0, c2 = wave_sum(s2) / g.K0;
  for (int cb = 0; cb < g.K0; cb += 256) {          // 256-column chunks (uniform trip count: the barriers below are taken by every wave)
    const int c = cb + lane * 4;
    f32x4 gw = {0.f, 0.f, 0.f, 0.f}, gb = gw;
    if (live && c < g.K0) {
      f32x4 d = *reinterpret_cast<const f32x4*>(dr + c);
      const f32x4 dplain = d;
      if (er) d += *reinterpret_cast<const f32x4*>(er + c);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c), w = *reinterpret_cast<const f32x4*>(g.ln_w + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mean) * rstd;
        o[e] = rstd * (d[e] * w[e] - c1 - xh * c2);
        gw[e] = d[e] * xh; gb[e] = d[e];
      }
      float* dx = g.dx + (long)m * g.K0 + c;
      *reinterpret_cast<f32x4*>(dx) = *reinterpret_cast<const f32x4*>(dx) + o;
      if (g.dpe) {
        float* dp = g.dpe + (long)(m % g.pe_period) * g.K0 + c;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dp + e, dplain[e]);
      }
    }
    if (g.dlnw) {       // column sums over the workgroup's rows: waves meet in LDS, 64 threads x 4 columns add to the global gradient
#pragma unroll
      for (int e = 0; e < 4; ++e) { red[0][wave][lane] = gw[e]; red[1][wave][lane] = gb[e];
        __syncthreads();
        if (wave == 0) {
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < ST_WAVES; ++w2) { a += red[0][w2][lane]; b += red[1][w2][lane]; }
          if (c < g.K0) { atomicAdd(g.dlnw + c + e, a); atomicAdd(g.dlnb + c + e, b); }
        }
        __syncthreads();
      }
    }
  }
}

}  // namespace

extern "C" int mt_token_stage_fwd(const MtTokenStage* s, mt_stream_t stream) {
  if (!s || !s->x || s->R < 1 || s->K0 < 4 || (s->K0 & 3) || !s->W1 || s->N1 < 1 || !s->h) return MT_ERR_BAD_ARG;
  if (((uintptr_t)s->x & 15) || ((uintptr_t)s->W1 & 15)) return MT_ERR_BAD_ARG;
  const bool ln = s->ln_w != nullptr;
  if (ln && (!s->ln_b || !s->stats || !(s->eps > 0.f) || (s->pe ? (!s->a || s->pe_period < 1) : !s->tn))) return MT_ERR_BAD_ARG;
  if (s->N1b > 0 && (!s->W1b || !s->hb || s->N2 > 0 || ((uintptr_t)s->W1b & 15))) return MT_ERR_BAD_ARG;
  if (s->N2 > 0 && (!s->W2 || !s->y || (s->N1 & 3) || ((uintptr_t)s->W2 & 15) || ((uintptr_t)s->h & 15))) return MT_ERR_BAD_ARG;
  StageFwdArgs g;
  g.R = s->R; g.K0 = s->K0; g.x = s->x;
  g.ln_w = s->ln_w; g.ln_b = s->ln_b; g.eps = s->eps; g.pe = ln ? s->pe : nullptr; g.pe_period = s->pe_period;
  g.tn = s->tn; g.a = s->a; g.stats = s->stats;
  g.W1 = s->W1; g.b1 = s->b1; g.N1 = s->N1; g.act1 = s->act1; g.h = s->h; g.pre1 = s->pre1; g.drop1 = make_drop(&s->drop1);
  g.W1b = s->N1b > 0 ? s->W1b : nullptr; g.b1b = s->b1b; g.N1b = s->N1b > 0 ? s->N1b : 0; g.hb = s->hb;
  g.W2 = s->W2; g.b2 = s->b2; g.N2 = s->N2 > 0 ? s->N2 : 0; g.resid = s->resid; g.resid_scale = s->resid_scale; g.drop2 = make_drop(&s->drop2);
  g.y = s->y;
  if (g.drop1.path_p > 0.f) return MT_ERR_UNSUPPORTED;       // (DropPath is a property of the stage's OUTPUT branch)
  hipLaunchKernelGGL(token_stage_fwd_kernel, dim3(cdiv(s->R, 16)), dim3(64 * ST_WAVES), 0, (hipStream_t)stream, g);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_token_stage_bwd(const MtTokenStage* s, const MtTokenStageGrads* d, mt_stream_t stream) {
  if (!s || !d || s->R < 1 || s->K0 < 4 || (s->K0 & 3) || !s->W1 || s->N1 < 1 || !d->da) return MT_ERR_BAD_ARG;
  const bool ln = s->ln_w != nullptr;
  if (s->N2 > 0 && (!d->dy || !d->dh || !s->W2 || (s->N2 & 3) || (s->N1b > 0))) return MT_ERR_BAD_ARG;
  if (s->N2 <= 0 && !d->dh) return MT_ERR_BAD_ARG;
  if (s->N1b > 0 && (!d->dhb || !s->W1b || (s->N1b & 3))) return MT_ERR_BAD_ARG;
  if ((s->N1 & 3) || (s->act1 != MT_ACT_NONE && !s->pre1)) return MT_ERR_BAD_ARG;
  if (ln && (!s->stats || !d->dx || (!d->dlnw) != (!d->dlnb) || (d->dpe && (!s->pe || s->pe_period < 1)))) return MT_ERR_BAD_ARG;
  StageBwdArgs g;
  g.R = s->R; g.K0 = s->K0; g.N1 = s->N1; g.N1b = s->N1b > 0 ? s->N1b : 0; g.N2 = s->N2 > 0 ? s->N2 : 0; g.act1 = s->act1;
  g.dy = d->dy; g.drop2 = make_drop(&s->drop2); g.dyp = d->dyp;
  const bool masked = g.drop2.rng && (g.drop2.p > 0.f || g.drop2.path_p > 0.f);
  if (g.N2 > 0 && masked && !d->dyp) return MT_ERR_BAD_ARG;
  g.dres = d->dres; g.resid_scale = s->resid_scale;
  g.W2 = s->W2; g.pre1 = s->act1 != MT_ACT_NONE ? s->pre1 : nullptr; g.drop1 = make_drop(&s->drop1);
  g.dh = d->dh; g.dhb = g.N1b > 0 ? d->dhb : nullptr; g.W1 = s->W1; g.W1b = s->W1b;
  g.da = d->da; g.da_accumulate = d->da_accumulate;
  g.x = s->x; g.ln_w = s->ln_w; g.stats = s->stats; g.dtn_extra = ln ? d->dtn_extra : nullptr;
  g.dx = d->dx; g.dlnw = d->dlnw; g.dlnb = d->dlnb; g.dpe = ln ? d->dpe : nullptr; g.pe_period = s->pe_period;
  hipLaunchKernelGGL(token_stage_bwd_kernel, dim3(cdiv(s->R, 16)), dim3(64 * ST_WAVES), 0, (hipStream_t)stream, g);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
