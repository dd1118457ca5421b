// Shared device helpers for the modaltune_hip kernels (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/modaltune_hip.h"

typedef _Float16 h16;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MT_WAVE 64
#define MT_DEVINL __device__ __forceinline__

// Logical row -> physical row of a "segmented" activation view: logical row m lives at
// (m / seg_rows) * seg_stride + row0 + (m % seg_rows).  With seg_rows = L, seg_stride = N, row0 = 1 this is
// "the patch rows of a [B, N, C] token buffer" (no cat/split copies around the cls row); seg_stride = 0
// broadcasts one slide to all task passes.  seg_rows <= 0 means identity.
struct RowMap {
  int seg_rows, seg_stride, row0;
  MT_DEVINL long map(int m) const {
    if (seg_rows <= 0) return m;
    int s = m / seg_rows;
    return (long)s * seg_stride + row0 + (m - s * seg_rows);
  }
};
static inline RowMap make_rowmap(const MtRowMap* r) {
  RowMap m;
  if (r) { m.seg_rows = r->seg_rows; m.seg_stride = r->seg_stride; m.row0 = r->row0; }
  else { m.seg_rows = 0; m.seg_stride = 0; m.row0 = 0; }
  return m;
}

MT_DEVINL float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
MT_DEVINL float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-level for the fp16-stored activations):
// erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z >= 0.  About 14 VALU ops against ~40 for erff();
// the FFN's GELU (feedforward_network.py:136) and its derivative were the VALU bound of the LN-3072 kernels.
// Returns erf(x / sqrt 2) and e = exp(-x^2 / 2) (shared with the derivative).
MT_DEVINL float erf_rsqrt2(float x, float& e) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float r = fmaf(-poly, e, 1.0f);
  return copysignf(r, x);
}
MT_DEVINL float gelu_erf(float x) { float e; return 0.5f * x * (1.0f + erf_rsqrt2(x, e)); }
MT_DEVINL float gelu_erf_grad(float x) {
  float e;
  const float er = erf_rsqrt2(x, e);
  return fmaf(x * 0.39894228040143267794f, e, 0.5f * (1.0f + er));   // Phi(x) + x phi(x)
}

// 16-byte global load/store helpers
MT_DEVINL h16x8 ldg8(const h16* p) { return *reinterpret_cast<const h16x8*>(p); }
MT_DEVINL void stg8(h16* p, h16x8 v) { *reinterpret_cast<h16x8*>(p) = v; }

#define MT_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return MT_ERR_LAUNCH;            \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
