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
// LDS reads of BOTH banking classes in flight (4-byte class: ds_read_b32 / ds_read2_b32; 8 / 16-byte class: ds_read_b64 / _b128) must
// not be consumed behind a COUNTED `s_waitcnt lgkmcnt(N > 0)`: beside another kernel's LDS traffic on the same CU (two HIP streams:
// the pass groups of the train step) the count was met while an older 16-byte read had not delivered lanes 48-63 -- stale registers,
// wrong sums, once in ~10 launches of mt_token_mha_fwd beside mt_gemm_tn_f16 (round 6; tools/diag/victim_stress2.py,
// profiles/r06_lds_counted_wait.txt; the aggressor triggers with its transposed reads and, rebuilt without them, with 4-byte writes next
// to 8-byte reads).  The same instructions behind ONE full wait, or with one class of reads per loop, never failed.
// hipcc places its waits itself (an asm wait does not hold back register-only arithmetic), so the token-side kernels whose loops mixed
// the two classes keep ONE class per loop instead -- either everything 16 bytes wide (mt_token_mha_fwd: padded score rows), or through
// these helpers: `volatile` 4-byte reads are never merged into 8 / 16-byte ones -- and their counted waits mean what they say.  tests/test_isa_lds_waits.py scans the ISA of every kernel
// for the pattern (tools/diag/lds_wait_scan.py).
MT_DEVINL float lds_f32(const float* p) { return *reinterpret_cast<const volatile float*>(p); }
MT_DEVINL f32x4 lds_f32x4_by_dword(const float* p) {
  const volatile float* q = reinterpret_cast<const volatile float*>(p);
  return (f32x4){q[0], q[1], q[2], q[3]};
}
#define MT_LDS_DRAIN_V1(a) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory")
#define MT_LDS_DRAIN_V2(a, b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory")
#define MT_LDS_DRAIN_V3(a, b, c) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) : : "memory")
#define MT_LDS_DRAIN_V4(a) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]) : : "memory")
#define MT_LDS_DRAIN_V8(a)                                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                            \
               : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]), "+v"((a)[4]), "+v"((a)[5]), "+v"((a)[6]), "+v"((a)[7]) \
               :                                                                                                                  \
               : "memory")

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

// ---- counter-based dropout masks (include/modaltune_hip.h: MtDropout)
struct DropArgs {
  const uint32_t* rng; uint32_t site; float p; uint32_t path_site; float path_p; int rows_per_pass;
  __host__ __device__ __forceinline__ bool active() const { return rng != nullptr && (p > 0.f || path_p > 0.f); }
};
static inline DropArgs make_drop(const MtDropout* d) {
  DropArgs a;
  if (d && d->rng && (d->p > 0.f || d->path_p > 0.f)) {
    a.rng = d->rng; a.site = d->site; a.p = d->p; a.path_site = d->path_site; a.path_p = d->path_p;
    a.rows_per_pass = d->rows_per_pass > 0 ? d->rows_per_pass : 1;
  } else {
    a.rng = nullptr; a.site = 0; a.p = 0.f; a.path_site = 0; a.path_p = 0.f; a.rows_per_pass = 1;
  }
  return a;
}
// Philox4x32 with 7 rounds (Salmon et al. 2011: passes BigCrush from 7 rounds up)
MT_DEVINL void philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
MT_DEVINL uint32_t drop_threshold(float p) { return (uint32_t)fminf(p * 4294967296.0f, 4294967040.0f); }
// DropPath factor of one task pass: 0 or 1 / (1 - path_p)
MT_DEVINL float drop_path_factor(const DropArgs& d, int pass) {
  if (!(d.path_p > 0.f)) return 1.f;
  uint32_t r[4];
  philox4x32_7((uint32_t)pass, 0x9E3779B9u, d.path_site, d.rng[2], d.rng[0], d.rng[1], r);
  return r[0] >= drop_threshold(d.path_p) ? 1.f / (1.f - d.path_p) : 0.f;
}
// scale factors of the 4 consecutive elements whose linear index starts at 4 * idx4, given their pass's DropPath factor
MT_DEVINL f32x4 drop_elem4(const DropArgs& d, uint64_t idx4, float pf) {
  f32x4 s = {pf, pf, pf, pf};
  if (d.p > 0.f) {
    uint32_t r[4];
    philox4x32_7((uint32_t)idx4, (uint32_t)(idx4 >> 32), d.site, d.rng[2], d.rng[0], d.rng[1], r);
    const uint32_t thr = drop_threshold(d.p);
    const float keep = pf / (1.f - d.p);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = r[e] >= thr ? keep : 0.f;
  }
  return s;
}
MT_DEVINL f32x4 drop_scale4(const DropArgs& d, uint64_t idx4, int m) {
  return drop_elem4(d, idx4, drop_path_factor(d, m / d.rows_per_pass));
}

// keep flag of ONE element (linear index idx) of an element-dropout site (token-side tensors, AlphaDropout)
MT_DEVINL bool drop_keep1(const DropArgs& d, uint32_t site, uint64_t idx) {
  uint32_t r[4];
  philox4x32_7((uint32_t)(idx >> 2), (uint32_t)(idx >> 34), site, d.rng[2], d.rng[0], d.rng[1], r);
  return r[idx & 3] >= drop_threshold(d.p);
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
