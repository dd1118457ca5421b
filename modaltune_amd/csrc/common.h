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

MT_DEVINL float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
MT_DEVINL float gelu_erf_grad(float x) {
  const float k = 0.39894228040143267794f;  // 1/sqrt(2 pi)
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * k * __expf(-0.5f * x * x);
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
