// LayerNorm forward/backward (one wave per row, fp32 statistics) and the elementwise helpers.
// HBM-bound: every row is read once with 8/16-byte vector loads and written once.
#include "common.h"

namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static MT_DEVINL f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static MT_DEVINL void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4<h16> {
  static MT_DEVINL f32x4 load(const h16* p) {
    h16x4 v = *reinterpret_cast<const h16x4*>(p);
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
  static MT_DEVINL void store(h16* p, f32x4 v) {
    *reinterpret_cast<h16x4*>(p) = (h16x4){(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
  }
};

struct LnFwdArgs {
  const void* x; long ldx; RowMap xmap;
  const float* w; const float* b; const float* add_rows; int add_period;
  void* y; long ldy; RowMap ymap;
  float* stats; int M;
  float eps;
};

template <int D, typename InT, typename OutT, bool GELU>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdArgs a) {
  constexpr int NC = D / 256;   // 4-element chunks per lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 wr[NC], br[NC];         // a lane always owns the same columns: affine parameters stay in registers across rows
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    wr[c] = Vec4<float>::load(a.w + c * 256 + lane * 4);
    br[c] = Vec4<float>::load(a.b + c * 256 + lane * 4);
  }
  for (int m = blockIdx.x * 4 + wave; m < a.M; m += gridDim.x * 4) {
    const InT* x = reinterpret_cast<const InT*>(a.x) + a.xmap.map(m) * a.ldx;
    f32x4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      v[c] = Vec4<InT>::load(x + c * 256 + lane * 4);
      if (GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[c][e] = gelu_erf(v[c][e]);
      }
      s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + a.eps);
    OutT* y = reinterpret_cast<OutT*>(a.y) + a.ymap.map(m) * a.ldy;
    const float* add = a.add_rows ? a.add_rows + (long)(m % a.add_period) * D : nullptr;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * wr[c][e] + br[c][e];
      if (add) { const f32x4 p = Vec4<float>::load(add + col); o += p; }
      Vec4<OutT>::store(y + col, o);
    }
    if (a.stats && lane == 0) { a.stats[2 * (long)m] = mean; a.stats[2 * (long)m + 1] = rstd; }
  }
}

// h = x + drop(branch) ; y = LN(h) * w + b.  The residual add of a backbone sub-layer (branch = fp16 output of the
// out_proj / fc2 GEMM, Dropout + DropPath as in the GEMM's BIAS_RESID epilogue: same counters, same masks) rides on the
// LayerNorm that reads the sum anyway: the GEMM keeps a plain fp16 epilogue (MFMA-bound) and the fp32 residual stream
// is read and written by this HBM-bound pass only.
struct AddLnArgs {
  const float* x; const h16* branch; DropArgs drop;
  const float* w; const float* b;
  float* h; h16* y; float* stats; int M;
  float eps;
};

template <int D>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(AddLnArgs a) {
  constexpr int NC = D / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 wr[NC], br[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    wr[c] = Vec4<float>::load(a.w + c * 256 + lane * 4);
    br[c] = Vec4<float>::load(a.b + c * 256 + lane * 4);
  }
  const bool dropping = a.drop.active();
  for (int m = blockIdx.x * 4 + wave; m < a.M; m += gridDim.x * 4) {
    const float* x = a.x + (long)m * D;
    const h16* bp = a.branch + (long)m * D;
    f32x4 v[NC], bv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      v[c] = Vec4<float>::load(x + c * 256 + lane * 4);
      bv[c] = Vec4<h16>::load(bp + c * 256 + lane * 4);
    }
    const float pf = dropping ? drop_path_factor(a.drop, m / a.drop.rows_per_pass) : 1.f;
    float s = 0.f;
    float* hrow = a.h + (long)m * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      if (dropping) bv[c] *= drop_elem4(a.drop, ((uint64_t)m * D + col) >> 2, pf);
      v[c] += bv[c];
      Vec4<float>::store(hrow + col, v[c]);
      s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + a.eps);
    h16* y = a.y + (long)m * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * wr[c][e] + br[c][e];
      Vec4<h16>::store(y + c * 256 + lane * 4, o);
    }
    if (lane == 0) { a.stats[2 * (long)m] = mean; a.stats[2 * (long)m + 1] = rstd; }
  }
}

struct LnBwdArgs {
  const void* dy; long lddy; RowMap dymap;
  const void* x; long ldx; RowMap xmap;
  const float* w; const float* stats;
  void* dx; long lddx; RowMap dxmap; int accumulate;
  float* dw; float* db; h16* dx16; DropArgs drop16; int M;
};

template <int D, typename DyT, typename InT, typename DxT, bool GELU, bool PARAM>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs a) {
  constexpr int NC = D / 256;
  constexpr int NP = PARAM ? NC : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gw[NP], gb[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) { gw[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; gb[c] = gw[c]; }
  f32x4 wr[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) wr[c] = Vec4<float>::load(a.w + c * 256 + lane * 4);
  for (int m = blockIdx.x * 4 + wave; m < a.M; m += gridDim.x * 4) {
    const DyT* dy = reinterpret_cast<const DyT*>(a.dy) + a.dymap.map(m) * a.lddy;
    const InT* x = reinterpret_cast<const InT*>(a.x) + a.xmap.map(m) * a.ldx;
    const float mean = a.stats[2 * (long)m], rstd = a.stats[2 * (long)m + 1];
    f32x4 xh[NC], g[NC], raw[GELU ? NC : 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      f32x4 xv = Vec4<InT>::load(x + col);
      if (GELU) {
        raw[c] = xv;
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = gelu_erf(xv[e]);
      }
      const f32x4 d = Vec4<DyT>::load(dy + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[c][e] = (xv[e] - mean) * rstd;
        g[c][e] = d[e] * wr[c][e];
        s1 += g[c][e];
        s2 += g[c][e] * xh[c][e];
      }
      if (PARAM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { gw[c][e] += d[e] * xh[c][e]; gb[c][e] += d[e]; }
      }
    }
    const float c1 = wave_sum(s1) * (1.0f / D), c2 = wave_sum(s2) * (1.0f / D);
    DxT* dx = reinterpret_cast<DxT*>(a.dx) + a.dxmap.map(m) * a.lddx;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = rstd * (g[c][e] - c1 - xh[c][e] * c2);
        if (GELU) o[e] *= gelu_erf_grad(raw[c][e]);
      }
      if (a.accumulate) { const f32x4 old = Vec4<DxT>::load(dx + col); o += old; }
      Vec4<DxT>::store(dx + col, o);
      if (a.dx16) {
        if (a.drop16.active()) o *= drop_scale4(a.drop16, ((uint64_t)m * D + col) >> 2, m);
        Vec4<h16>::store(a.dx16 + (long)m * D + col, o);
      }
    }
  }
  if (PARAM) {
    __shared__ float red[2][4][D];
#pragma unroll
    for (int c = 0; c < NP; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[0][wave][c * 256 + lane * 4 + e] = gw[c][e];
        red[1][wave][c * 256 + lane * 4 + e] = gb[c][e];
      }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) {
      atomicAdd(&a.dw[i], red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i]);
      atomicAdd(&a.db[i], red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i]);
    }
  }
}


// ---------------------------------------------------------------- FFN sub-LayerNorm ---------------
// y = LN(gelu(a1)) over D = 3072 (feedforward_network.py:136-139) and its backward, fp16 in / fp16 out.  These two
// launches stream 0.37 / 0.55 GB per layer and were co-bound by the VALU (erf twice per element, scalar fp32 ops, the
// affine vectors re-read per row): here a lane owns 8 consecutive columns per chunk (16-byte loads/stores), keeps
// gamma/beta in registers across rows, evaluates erf ONCE for both gelu and gelu', and does the arithmetic with packed
// fp32 instructions (v_pk_fma/mul/add_f32).
MT_DEVINL f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
MT_DEVINL f32x2 splat2(float v) { return (f32x2){v, v}; }

// gelu(x) and d gelu / dx for two values; A&S 7.1.26 erf as in common.h (|err| <= 1.5e-7)
template <bool WANT_GRAD>
MT_DEVINL void gelu_pair(f32x2 x, f32x2& g, f32x2& dg) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 z = ax * splat2(0.70710678118654752440f);
  const f32x2 den = pk_fma2(z, splat2(0.3275911f), splat2(1.0f));
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 poly = pk_fma2(t, splat2(1.061405429f), splat2(-1.453152027f));
  poly = pk_fma2(poly, t, splat2(1.421413741f));
  poly = pk_fma2(poly, t, splat2(-0.284496736f));
  poly = pk_fma2(poly, t, splat2(0.254829592f));
  poly = poly * t;
  const f32x2 ea = z * z * splat2(-1.4426950408889634f);
  const f32x2 e = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
  const f32x2 r = pk_fma2(-poly, e, splat2(1.0f));                 // erf(|x| / sqrt 2)
  const f32x2 er = {copysignf(r[0], x[0]), copysignf(r[1], x[1])};
  const f32x2 h = pk_fma2(er, splat2(0.5f), splat2(0.5f));          // Phi(x)
  g = x * h;
  if (WANT_GRAD) dg = pk_fma2(x * splat2(0.39894228040143267794f), e, h);   // Phi(x) + x phi(x)
}

MT_DEVINL void cvt8(h16x8 v, f32x2 (&o)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (f32x2){(float)v[2 * i], (float)v[2 * i + 1]};
}
MT_DEVINL h16x8 pack8(const f32x2 (&o)[4]) {
  h16x8 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = (h16)o[i][0]; v[2 * i + 1] = (h16)o[i][1]; }
  return v;
}
MT_DEVINL void ldf8(const float* p, f32x2 (&o)[4]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  o[0] = (f32x2){a[0], a[1]}; o[1] = (f32x2){a[2], a[3]}; o[2] = (f32x2){b[0], b[1]}; o[3] = (f32x2){b[2], b[3]};
}

// Two waves per row (a lane owns D / 128 columns in chunks of 8): ~100 VGPRs, 4-5 waves per SIMD of loads in flight.
// Row statistics cross the wave pair through LDS.
template <int D>
__global__ __launch_bounds__(256) void ln_gelu_fwd_kernel(LnFwdArgs a) {
  constexpr int NC = D / 1024;   // 8-element chunks per lane
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = wave & 1, rsel = wave >> 1;
  const int col0 = half * (D / 2) + lane * 8;
  f32x2 wr[NC][4], br[NC][4];
#pragma unroll
  for (int c = 0; c < NC; ++c) { ldf8(a.w + col0 + c * 512, wr[c]); ldf8(a.b + col0 + c * 512, br[c]); }
  for (int base = blockIdx.x * 2; base < a.M; base += gridDim.x * 2) {
    const int m = base + rsel;
    const bool valid = m < a.M;
    const int mc = valid ? m : a.M - 1;
    const h16* x = reinterpret_cast<const h16*>(a.x) + a.xmap.map(mc) * a.ldx;
    h16x8 raw[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) raw[c] = ldg8(x + col0 + c * 512);
    f32x2 v[NC][4];
    f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x2 xv[4];
      cvt8(raw[c], xv);
#pragma unroll
      for (int i = 0; i < 4; ++i) { f32x2 dg; gelu_pair<false>(xv[i], v[c][i], dg); s2 += v[c][i]; }
    }
    const float ps = wave_sum(s2[0] + s2[1]);
    if (lane == 0) red[0][wave] = ps;
    __syncthreads();
    const float mean = (red[0][2 * rsel] + red[0][2 * rsel + 1]) * (1.0f / D);
    f32x2 q2 = {0.f, 0.f};
    const f32x2 nm = splat2(-mean);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[c][i] += nm; q2 = pk_fma2(v[c][i], v[c][i], q2); }
    const float pq = wave_sum(q2[0] + q2[1]);
    if (lane == 0) red[1][wave] = pq;
    __syncthreads();
    const float rstd = rsqrtf((red[1][2 * rsel] + red[1][2 * rsel + 1]) * (1.0f / D) + a.eps);
    if (valid) {
      h16* y = reinterpret_cast<h16*>(a.y) + a.ymap.map(m) * a.ldy;
      const f32x2 rs = splat2(rstd);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        f32x2 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pk_fma2(v[c][i] * rs, wr[c][i], br[c][i]);
        stg8(y + col0 + c * 512, pack8(o));
      }
      if (a.stats && lane == 0 && half == 0) { a.stats[2 * (long)m] = mean; a.stats[2 * (long)m + 1] = rstd; }
    }
  }
}

// backward: WPR waves per row (a lane owns D / (64 WPR) columns in chunks of 8), one row per workgroup iteration
template <int D, int WPR>
__global__ __launch_bounds__(64 * WPR) void ln_gelu_bwd_kernel(LnBwdArgs a) {
  constexpr int NC = D / (512 * WPR);
  static_assert(NC * 512 * WPR == D, "row must split into 8-element chunks over the waves");
  __shared__ float red[2][2][WPR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col0 = wave * (D / WPR) + lane * 8;
  f32x2 wr[NC][4];
#pragma unroll
  for (int c = 0; c < NC; ++c) ldf8(a.w + col0 + c * 512, wr[c]);
  int it = 0;
  for (int m = blockIdx.x; m < a.M; m += gridDim.x, it ^= 1) {
    const h16* dy = reinterpret_cast<const h16*>(a.dy) + a.dymap.map(m) * a.lddy;
    const h16* x = reinterpret_cast<const h16*>(a.x) + a.xmap.map(m) * a.ldx;
    h16x8 rx[NC], rd[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { rx[c] = ldg8(x + col0 + c * 512); rd[c] = ldg8(dy + col0 + c * 512); }
    const float mean = a.stats[2 * (long)m], rstd = a.stats[2 * (long)m + 1];
    const f32x2 rs = splat2(rstd), nmr = splat2(-mean * rstd);
    f32x2 xh[NC][4], g[NC][4], dgl[NC][4];
    f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x2 xv[4], dv[4];
      cvt8(rx[c], xv); cvt8(rd[c], dv);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 gl;
        gelu_pair<true>(xv[i], gl, dgl[c][i]);
        xh[c][i] = pk_fma2(gl, rs, nmr);
        g[c][i] = dv[i] * wr[c][i];
        s1 += g[c][i];
        s2 = pk_fma2(g[c][i], xh[c][i], s2);
      }
    }
    const float p1 = wave_sum(s1[0] + s1[1]), p2 = wave_sum(s2[0] + s2[1]);
    if (lane == 0) { red[it][0][wave] = p1; red[it][1][wave] = p2; }      // double-buffered: one barrier per row
    __syncthreads();
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < WPR; ++w2) { c1 += red[it][0][w2]; c2 += red[it][1][w2]; }
    c1 *= (1.0f / D); c2 *= (1.0f / D);
    h16* dx = reinterpret_cast<h16*>(a.dx) + a.dxmap.map(m) * a.lddx;
    const f32x2 nc1 = splat2(-c1), nc2 = splat2(-c2);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      f32x2 o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (pk_fma2(xh[c][i], nc2, g[c][i]) + nc1) * (dgl[c][i] * rs);
      if (a.accumulate) {
        f32x2 old[4];
        cvt8(ldg8(dx + col0 + c * 512), old);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += old[i];
      }
      stg8(dx + col0 + c * 512, pack8(o));
    }
  }
}

int ln_gelu_grid(int M) { return max(1, min(cdiv(M, 2), 8192)); }
int ln_grid(int M) { return max(1, min(cdiv(M, 4), 4096)); }

template <int D, typename InT, typename OutT>
int ln_fwd_dispatch(const LnFwdArgs& a, int gelu, hipStream_t s) {
  if (gelu) hipLaunchKernelGGL((ln_fwd_kernel<D, InT, OutT, true>), dim3(ln_grid(a.M)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((ln_fwd_kernel<D, InT, OutT, false>), dim3(ln_grid(a.M)), dim3(256), 0, s, a);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
template <int D>
int ln_fwd_types(const LnFwdArgs& a, int in_dt, int out_dt, int gelu, hipStream_t s) {
  if constexpr (D % 1024 == 0) {
    if (gelu && in_dt == MT_OUT_F16 && out_dt == MT_OUT_F16 && !a.add_rows && !(a.ldx & 7) && !(a.ldy & 7)) {
      hipLaunchKernelGGL((ln_gelu_fwd_kernel<D>), dim3(ln_gelu_grid(a.M)), dim3(256), 0, s, a);
      MT_CHECK_LAUNCH();
      return MT_OK;
    }
  }
  if (in_dt == MT_OUT_F32 && out_dt == MT_OUT_F16) return ln_fwd_dispatch<D, float, h16>(a, gelu, s);
  if (in_dt == MT_OUT_F32 && out_dt == MT_OUT_F32) return ln_fwd_dispatch<D, float, float>(a, gelu, s);
  if (in_dt == MT_OUT_F16 && out_dt == MT_OUT_F16) return ln_fwd_dispatch<D, h16, h16>(a, gelu, s);
  return MT_ERR_UNSUPPORTED;
}

template <int D, typename DyT, typename InT, typename DxT, bool GELU>
int ln_bwd_launch(const LnBwdArgs& a, hipStream_t s) {
  if (a.dw) {
    if constexpr (!GELU && sizeof(DyT) == 4) {            // trainable norms: token side (fp32) ...
      hipLaunchKernelGGL((ln_bwd_kernel<D, DyT, InT, DxT, GELU, true>), dim3(min(ln_grid(a.M), 512)), dim3(256), 0, s, a);
    } else if constexpr (D <= 768 && !GELU) {           // ... and the adapters' 768-wide norms over the patch rows
      hipLaunchKernelGGL((ln_bwd_kernel<D, DyT, InT, DxT, GELU, true>), dim3(min(ln_grid(a.M), 512)), dim3(256), 0, s, a);
    } else {
      return MT_ERR_UNSUPPORTED;
    }
  } else {
    hipLaunchKernelGGL((ln_bwd_kernel<D, DyT, InT, DxT, GELU, false>), dim3(ln_grid(a.M)), dim3(256), 0, s, a);
  }
  MT_CHECK_LAUNCH();
  return MT_OK;
}

// ---------------------------------------------------------------- elementwise ---------------------
__global__ void cast_drop_f32_f16_kernel(const float* x, h16* y, long n, DropArgs d, int D) {    // n % 4 == 0, D % 4 == 0
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    f32x4 v = Vec4<float>::load(x + i);
    v *= drop_scale4(d, (uint64_t)i >> 2, (int)(i / D));
    Vec4<h16>::store(y + i, v);
  }
}
// y(m,:) = drop(x(xmap(m),:)), dense fp32 y
__global__ void dropout_f32_kernel(const float* x, long ldx, RowMap xmap, float* y, int M, int D, DropArgs d) {
  const int per_row = D / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * per_row; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / per_row), c = (int)(i % per_row) * 4;
    f32x4 v = Vec4<float>::load(x + xmap.map(m) * ldx + c);
    v *= drop_scale4(d, ((uint64_t)m * D + c) >> 2, m);
    Vec4<float>::store(y + (long)m * D + c, v);
  }
}
__global__ void droppath_rows_kernel(float* x, int M, int D, DropArgs d) {
  const int per_row = D / 4;
  DropArgs dd = d;
  dd.p = 0.f;                           // path factor only
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * per_row; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / per_row), c = (int)(i % per_row) * 4;
    f32x4 v = Vec4<float>::load(x + (long)m * D + c);
    v *= drop_scale4(dd, 0, m);
    Vec4<float>::store(x + (long)m * D + c, v);
  }
}
__global__ void rng_advance_kernel(uint32_t* rng) { if (threadIdx.x == 0 && blockIdx.x == 0) rng[2] += 1u; }

__global__ void cast_f32_f16_kernel(const float* x, h16* y, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) Vec4<h16>::store(y + i, Vec4<float>::load(x + i));
  if (i < n && i + 3 >= n) for (long j = i; j < n; ++j) y[j] = (h16)x[j];
}
__global__ void cast_f16_f32_kernel(const h16* x, float* y, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) Vec4<float>::store(y + i, Vec4<h16>::load(x + i));
  if (i < n && i + 3 >= n) for (long j = i; j < n; ++j) y[j] = (float)x[j];
}
__global__ void act_fwd_kernel(const float* x, float* y, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = act == MT_ACT_RELU ? fmaxf(v, 0.f) : act == MT_ACT_GELU ? gelu_erf(v) : act == MT_ACT_ELU ? (v > 0.f ? v : expm1f(v)) : v;
  }
}
__global__ void act_bwd_kernel(const float* x, const float* dy, float* dx, long n, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    float d = 1.f;
    if (act == MT_ACT_RELU) d = v > 0.f ? 1.f : 0.f;
    else if (act == MT_ACT_GELU) d = gelu_erf_grad(v);
    else if (act == MT_ACT_ELU) d = v > 0.f ? 1.f : __expf(v);
    dx[i] = dy[i] * d;
  }
}
__global__ void axpy_kernel(const float* a, const float* b, float alpha, float* y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = a[i] + alpha * b[i];
}
__global__ void axpy_bcast_kernel(const float* a, const float* b, float alpha, float* y, long n, long period) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = a[i] + alpha * b[i % period];
}
// out[i] += sum_r x[r * period + i], r ascending: the adjoint of axpy_bcast (period % 4 == 0, 16-byte aligned)
__global__ void fold_rows_kernel(const float* __restrict__ x, int reps, long period, float* __restrict__ out) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < period; i += (long)gridDim.x * blockDim.x * 4) {
    f32x4 s = *reinterpret_cast<const f32x4*>(x + i);
    for (int r = 1; r < reps; ++r) s += *reinterpret_cast<const f32x4*>(x + r * period + i);
    *reinterpret_cast<f32x4*>(out + i) += s;
  }
}
__global__ void copy_rows_kernel(const float* src, long lds, RowMap smap, float* dst, long ldd, RowMap dmap, int M,
                                 int D, int accumulate) {
  const int per_row = D / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)M * per_row; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / per_row), c = (int)(i % per_row) * 4;
    f32x4 v = Vec4<float>::load(src + smap.map(m) * lds + c);
    float* d = dst + dmap.map(m) * ldd + c;
    if (accumulate) v += Vec4<float>::load(d);
    Vec4<float>::store(d, v);
  }
}

// Injector residual path backward.  y = (1+g) x + g * proj  (A.1), with proj = out(a) recomputed in fp16:
//   dx (+)= (1+g) dy ; dproj = g * dy (fp16, feeds the dX/dW GEMMs) ; dgamma += sum_m dy * (x + proj)
struct InjBwdArgs {
  const float* dy; long lddy; RowMap dymap;
  const float* x; long ldx; RowMap xmap;
  const h16* proj; const float* gamma;
  float* dx; long lddx; RowMap dxmap; int dx_accumulate;
  h16* dproj; float* dgamma; int M;
};
template <int D>
__global__ __launch_bounds__(256) void inject_resid_bwd_kernel(InjBwdArgs a) {
  constexpr int NC = D / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gg[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) gg[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int m = blockIdx.x * 4 + wave; m < a.M; m += gridDim.x * 4) {
    const float* dy = a.dy + a.dymap.map(m) * a.lddy;
    const float* x = a.x + a.xmap.map(m) * a.ldx;
    const h16* pr = a.proj + (long)m * D;
    float* dx = a.dx + a.dxmap.map(m) * a.lddx;
    h16* dp = a.dproj + (long)m * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int col = c * 256 + lane * 4;
      const f32x4 d = Vec4<float>::load(dy + col), xv = Vec4<float>::load(x + col), p = Vec4<h16>::load(pr + col);
      const f32x4 gm = Vec4<float>::load(a.gamma + col);
      f32x4 o, q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (1.0f + gm[e]) * d[e];
        q[e] = gm[e] * d[e];
        gg[c][e] += d[e] * (xv[e] + p[e]);
      }
      if (a.dx_accumulate) o += Vec4<float>::load(dx + col);
      Vec4<float>::store(dx + col, o);
      Vec4<h16>::store(dp + col, q);
    }
  }
  __shared__ float red[4][D];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][c * 256 + lane * 4 + e] = gg[c][e];
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) atomicAdd(&a.dgamma[i], red[0][i] + red[1][i] + red[2][i] + red[3][i]);
}

// fp32 [R, C] -> fp16 [R, C] (transpose = 0) or fp16 [C, R] (transpose = 1), 32x32 tiles through LDS
__global__ __launch_bounds__(256) void pack_f16_kernel(const float* __restrict__ src, int R, int C, h16* __restrict__ dst, int transpose) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (transpose) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;
      if (r < R && c < C) dst[(long)c * R + r] = (h16)tile[tx][ty + 8 * i];
    } else {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      if (r < R && c < C) dst[(long)r * C + c] = (h16)tile[ty + 8 * i][tx];
    }
  }
}

// grouped form: blockIdx.x = item (8 int64 per record, see mt_pack_weights_f16), blockIdx.y strides over its 32x32 tiles;
// one pass writes both the as-stored and the transposed fp16 copy
__global__ __launch_bounds__(256) void pack_group_kernel(const long long* __restrict__ items) {
  __shared__ float tile[32][33];
  const long long* it = items + (long)blockIdx.x * 8;
  const float* src = reinterpret_cast<const float*>(it[0]);
  h16* dst = reinterpret_cast<h16*>(it[1]);
  h16* dst_t = reinterpret_cast<h16*>(it[2]);
  const int R = (int)it[3], C = (int)it[4], roff = (int)it[5];
  const long ld = it[6], ld_t = it[7];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int tiles_c = (C + 31) / 32, tiles = ((R + 31) / 32) * tiles_c;
  for (int t = blockIdx.y; t < tiles; t += gridDim.y) {
    const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      const float v = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
      tile[ty + 8 * i][tx] = v;
      if (dst && r < R && c < C) dst[(long)(roff + r) * ld + c] = (h16)v;
    }
    __syncthreads();
    if (dst_t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < C) dst_t[(long)c * ld_t + roff + r] = (h16)tile[tx][ty + 8 * i];
      }
    }
    __syncthreads();
  }
}

int ew_grid(long n) { return (int)max(1L, min((n + 1023) / 1024, 4096L)); }

}  // namespace

extern "C" int mt_layernorm_fwd_eps(const void* x, long ldx, const MtRowMap* xmap, int in_dtype, int gelu_in,
                                    const float* w, const float* b, const float* add_rows, int add_period, void* y,
                                    long ldy, const MtRowMap* ymap, int out_dtype, float* stats, int M, int D, float eps,
                                    mt_stream_t stream) {
  if (!x || !y || !w || !b || M <= 0 || (ldx & 3) || (ldy & 3) || !(eps > 0.f)) return MT_ERR_BAD_ARG;
  LnFwdArgs a{x, ldx, make_rowmap(xmap), w, b, add_rows, add_period > 0 ? add_period : 1, y, ldy, make_rowmap(ymap), stats, M, eps};
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 256: return ln_fwd_types<256>(a, in_dtype, out_dtype, gelu_in, s);
    case 768: return ln_fwd_types<768>(a, in_dtype, out_dtype, gelu_in, s);
    case 2304: return ln_fwd_types<2304>(a, in_dtype, out_dtype, gelu_in, s);
    case 3072: return ln_fwd_types<3072>(a, in_dtype, out_dtype, gelu_in, s);
    default: return MT_ERR_UNSUPPORTED;
  }
}

extern "C" int mt_layernorm_fwd(const void* x, long ldx, const MtRowMap* xmap, int in_dtype, int gelu_in,
                                const float* w, const float* b, const float* add_rows, int add_period, void* y,
                                long ldy, const MtRowMap* ymap, int out_dtype, float* stats, int M, int D,
                                mt_stream_t stream) {
  return mt_layernorm_fwd_eps(x, ldx, xmap, in_dtype, gelu_in, w, b, add_rows, add_period, y, ldy, ymap, out_dtype, stats, M, D, 1e-5f,
                              stream);
}

extern "C" int mt_add_layernorm_fwd_eps(const float* x, const mt_half* branch, const MtDropout* drop, const float* w,
                                        const float* b, float* h, mt_half* y, float* stats, int M, int D, float eps,
                                        mt_stream_t stream) {
  if (!x || !branch || !w || !b || !h || !y || !stats || M <= 0 || h == x || !(eps > 0.f)) return MT_ERR_BAD_ARG;
  if (D != 768) return MT_ERR_UNSUPPORTED;
  AddLnArgs a{x, (const h16*)branch, make_drop(drop), w, b, h, (h16*)y, stats, M, eps};
  hipLaunchKernelGGL((add_ln_fwd_kernel<768>), dim3(ln_grid(M)), dim3(256), 0, (hipStream_t)stream, a);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_add_layernorm_fwd(const float* x, const mt_half* branch, const MtDropout* drop, const float* w,
                                    const float* b, float* h, mt_half* y, float* stats, int M, int D, mt_stream_t stream) {
  return mt_add_layernorm_fwd_eps(x, branch, drop, w, b, h, y, stats, M, D, 1e-5f, stream);
}

template <int D>
static int ln_bwd_types(const LnBwdArgs& a, int dy_dt, int in_dt, int dx_dt, int gelu, hipStream_t s) {
  const bool dyh = dy_dt == MT_OUT_F16, inh = in_dt == MT_OUT_F16, dxh = dx_dt == MT_OUT_F16;
  if (gelu) {
    if constexpr (D % 1024 == 0) {
      if (dyh && inh && dxh && !a.dw && !(a.lddy & 7) && !(a.ldx & 7) && !(a.lddx & 7)) {
        constexpr int WPR = (D % 1536 == 0) ? 3 : 2;        // 3072 -> three waves per row (16 columns per lane)
        hipLaunchKernelGGL((ln_gelu_bwd_kernel<D, WPR>), dim3(min(a.M, 16384)), dim3(64 * WPR), 0, s, a);
        MT_CHECK_LAUNCH();
        return MT_OK;
      }
    }
    if (dyh && inh && dxh) return ln_bwd_launch<D, h16, h16, h16, true>(a, s);
    return MT_ERR_UNSUPPORTED;
  }
  if (dyh && !inh && !dxh) return ln_bwd_launch<D, h16, float, float, false>(a, s);
  if (dyh && !inh && dxh) return ln_bwd_launch<D, h16, float, h16, false>(a, s);
  if (!dyh && !inh && !dxh) return ln_bwd_launch<D, float, float, float, false>(a, s);
  return MT_ERR_UNSUPPORTED;
}

extern "C" int mt_layernorm_bwd(const void* dy, long lddy, const MtRowMap* dymap, int dy_dtype, const void* x,
                                long ldx, const MtRowMap* xmap, int in_dtype, int gelu_in, const float* w,
                                const float* stats, void* dx, long lddx, const MtRowMap* dxmap, int dx_dtype,
                                int accumulate, float* dw, float* db, mt_half* dx_f16, const MtDropout* dx_f16_drop, int M,
                                int D, mt_stream_t stream) {
  if (!dy || !x || !w || !stats || !dx || M <= 0 || (!dw) != (!db)) return MT_ERR_BAD_ARG;
  if (dx_f16 && gelu_in) return MT_ERR_UNSUPPORTED;
  LnBwdArgs a{dy, lddy, make_rowmap(dymap), x, ldx, make_rowmap(xmap), w, stats, dx, lddx, make_rowmap(dxmap), accumulate, dw, db,
              (h16*)dx_f16, make_drop(dx_f16 ? dx_f16_drop : nullptr), M};
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 256: return ln_bwd_types<256>(a, dy_dtype, in_dtype, dx_dtype, gelu_in, s);
    case 768: return ln_bwd_types<768>(a, dy_dtype, in_dtype, dx_dtype, gelu_in, s);
    case 2304: return ln_bwd_types<2304>(a, dy_dtype, in_dtype, dx_dtype, gelu_in, s);
    case 3072: return ln_bwd_types<3072>(a, dy_dtype, in_dtype, dx_dtype, gelu_in, s);
    default: return MT_ERR_UNSUPPORTED;
  }
}

extern "C" int mt_cast_f32_to_f16(const float* x, mt_half* y, long n, const MtDropout* drop, int D, mt_stream_t stream) {
  if (!x || !y || n <= 0) return MT_ERR_BAD_ARG;
  const DropArgs d = make_drop(drop);
  if (d.active()) {
    if ((n & 3) || D <= 0 || (D & 3)) return MT_ERR_BAD_ARG;
    hipLaunchKernelGGL(cast_drop_f32_f16_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, (h16*)y, n, d, D);
  } else {
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, (h16*)y, n);
  }
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_rng_advance(unsigned* rng, mt_stream_t stream) {
  if (!rng) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rng);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_dropout_f32(const float* x, long ldx, const MtRowMap* xmap, float* y, int M, int D, const MtDropout* drop,
                              mt_stream_t stream) {
  if (!x || !y || M <= 0 || D <= 0 || (D & 3) || (ldx & 3)) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(ew_grid((long)M * D / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     make_rowmap(xmap), y, M, D, make_drop(drop));
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_droppath_rows_f32(float* x, int M, int D, const MtDropout* drop, mt_stream_t stream) {
  if (!x || M <= 0 || D <= 0 || (D & 3)) return MT_ERR_BAD_ARG;
  const DropArgs d = make_drop(drop);
  if (!d.active() || d.path_p <= 0.f) return MT_OK;
  hipLaunchKernelGGL(droppath_rows_kernel, dim3(ew_grid((long)M * D / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, M, D, d);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_cast_f16_to_f32(const mt_half* x, float* y, long n, mt_stream_t stream) {
  if (!x || !y || n <= 0) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(ew_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, (const h16*)x, y, n);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_pack_weight_f16(const float* src, int R, int C, mt_half* dst, int transpose, mt_stream_t stream) {
  if (!src || !dst || R <= 0 || C <= 0) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_f16_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, (hipStream_t)stream, src, R, C, (h16*)dst, transpose);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_pack_weights_f16(const long long* items, int n_items, mt_stream_t stream) {
  if (!items || n_items <= 0) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_group_kernel, dim3(n_items, 48), dim3(256), 0, (hipStream_t)stream, items);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_act_fwd(const float* x, float* y, long n, int act, mt_stream_t stream) {
  if (!x || !y || n <= 0) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, act);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_act_bwd(const float* x, const float* dy, float* dx, long n, int act, mt_stream_t stream) {
  if (!x || !dy || !dx || n <= 0) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, act);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_axpy(const float* a, const float* b, float alpha, float* y, long n, mt_stream_t stream) {
  if (!a || !b || !y || n <= 0) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, y, n);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_axpy_bcast(const float* a, const float* b, float alpha, float* y, long n, long period, mt_stream_t stream) {
  if (!a || !b || !y || n <= 0 || period <= 0) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(axpy_bcast_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, alpha, y, n, period);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_fold_rows(const float* x, int reps, long period, float* out, mt_stream_t stream) {
  if (!x || !out || reps < 1 || period <= 0 || (period & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(fold_rows_kernel, dim3(ew_grid(period / 4)), dim3(256), 0, (hipStream_t)stream, x, reps, period, out);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_copy_rows_f32(const float* src, long lds, const MtRowMap* smap, float* dst, long ldd,
                                const MtRowMap* dmap, int M, int D, int accumulate, mt_stream_t stream) {
  if (!src || !dst || M <= 0 || D <= 0 || (D & 3) || (lds & 3) || (ldd & 3)) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(ew_grid((long)M * D / 4)), dim3(256), 0, (hipStream_t)stream, src, lds,
                     make_rowmap(smap), dst, ldd, make_rowmap(dmap), M, D, accumulate);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
extern "C" int mt_inject_resid_bwd(const float* dy, long lddy, const MtRowMap* dymap, const float* x, long ldx,
                                   const MtRowMap* xmap, const mt_half* proj, const float* gamma, float* dx, long lddx,
                                   const MtRowMap* dxmap, int dx_accumulate, mt_half* dproj, float* dgamma, int M,
                                   int D, mt_stream_t stream) {
  if (!dy || !x || !proj || !gamma || !dx || !dproj || !dgamma || M <= 0 || D != 768) return MT_ERR_BAD_ARG;
  InjBwdArgs a{dy, lddy, make_rowmap(dymap), x, ldx, make_rowmap(xmap), (const h16*)proj, gamma, dx, lddx,
               make_rowmap(dxmap), dx_accumulate, (h16*)dproj, dgamma, M};
  hipLaunchKernelGGL(inject_resid_bwd_kernel<768>, dim3(min(cdiv(M, 4), 512)), dim3(256), 0, (hipStream_t)stream, a);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
