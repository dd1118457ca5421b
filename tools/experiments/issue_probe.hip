// Does a SIMD overlap one wave's MFMA block with another wave's VALU block?  (round 3: the attention kernels' MFMA-busy and
// VALU-busy fractions add up to ~90 % of the kernel time, as if the two pipes took turns.)
// Every wave loops over [NM x v_mfma_f32_32x32x16_f16 on 4 independent accumulators] [NE x v_exp_f32 + NV x v_fma_f32 on 8
// independent registers], the two blocks separated by scheduling barriers (PHASED) or left to the scheduler.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o build_variants/issue_probe.so tools/experiments/issue_probe.hip
#include <hip/hip_runtime.h>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NE, int NV, bool PHASED>
__global__ __launch_bounds__(256) void probe(float* __restrict__ sink, int iters) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  h16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    unsigned h = (t * 9781u + e * 6271u) * 2654435761u;
    a[e] = (h16)(((int)(h >> 20) & 1023) * (1.0f / 1024.0f) - 0.5f);
    b[e] = (h16)(((int)(h >> 8) & 1023) * (1.0f / 1024.0f) - 0.5f);
  }
  f32x16 c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
  float e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = -1.0f - 0.01f * (float)((t + i) & 63);
  const float k1 = 0.999f, k2 = -0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16((i & 1) ? a : b, (i & 2) ? a : b, c[i & 3], 0, 0, 0);
    if (PHASED) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NE; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i & 7]));
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[i & 7]) : "v"(k1), "v"(k2));
    if (PHASED) __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[j][i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += e[i];
  if (s == 123.456f) sink[0] = s;
}

#define CASE(id, NM, NE, NV, PH) case id: hipLaunchKernelGGL((probe<NM, NE, NV, PH>), dim3(wgs), dim3(256), 0, (hipStream_t)stream, sink, iters); break;
extern "C" int issue_probe(int which, float* sink, int wgs, int iters, void* stream) {
  switch (which) {
    CASE(0, 28, 0, 0, true)        // MFMA only: 28 per iteration (the dK/dV tile)
    CASE(1, 0, 32, 58, true)       // VALU only: 32 exp + 58 plain
    CASE(2, 28, 32, 58, true)      // both, phased
    CASE(3, 28, 32, 58, false)     // both, scheduler free
    CASE(4, 0, 32, 0, true)        // exp only
    CASE(5, 0, 0, 58, true)        // plain VALU only
    CASE(6, 28, 0, 58, true)       // MFMA + plain VALU
    CASE(7, 28, 32, 0, true)       // MFMA + exp
    CASE(8, 14, 0, 0, true)        // the forward tile: 14 MFMA
    CASE(9, 0, 32, 52, true)       //                   32 exp + 52 plain
    CASE(10, 14, 32, 52, true)     //                   both, phased
    CASE(11, 14, 32, 52, false)    //                   both, scheduler free
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
