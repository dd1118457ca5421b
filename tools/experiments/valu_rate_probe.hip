// Issue cost of the VALU instructions the attention kernels' softmax blocks are made of, alone and mixed (does a transcendental
// overlap plain VALU work or an MFMA of the same / another wave?).  One or two waves per SIMD, 64 instructions of each kind per
// iteration on 16 independent registers, s_memtime ticks (core clock) per iteration.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/experiments/valu_rate_probe.hip -o build_variants/valu_rate_probe.so
#include <hip/hip_runtime.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i) & 15]))
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[(i) & 15]))
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[(i) & 15]) : "v"(c))
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[(i) & 7]) : "v"(pc))
#define CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[(i) & 15]) : "v"(y[(i) & 15]), "v"(y[((i) + 1) & 15]))
#define FMAMIX(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(y[(i) & 15]) : "v"(u[(i) & 15]), "v"(c))
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(y[(i) & 15]) : "v"(y[((i) + 1) & 15]), "v"(y[((i) + 2) & 15]))
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[(i) & 15]) : "v"(y[(i) & 15]))
#define CVTRTZ(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[(i) & 15]) : "v"(y[(i) & 15]), "v"(y[((i) + 1) & 15]))
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(i) & 7]) : "v"(pc))
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i) & 7]) : "v"(pc))
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y[(i) & 15]) : "v"(c))
#define PERM(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[(i) & 15]) : "v"(u[((i) + 3) & 15]), "v"(u[((i) + 7) & 15]), "v"(u[((i) + 11) & 15]))
#define MFMA(i) acc[(i) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[(i) & 3], 0, 0, 0)
template <int MODE, int WAVES>
__global__ __launch_bounds__(256 * WAVES) void probe(unsigned long long* out, float* sink, int iters) {
  float x[16], y[16], c = 1.0001f;
  unsigned u[16];
  f32x2 p[8], pc = {1.0001f, 0.9999f};
  f32x16 acc[4];
  h16x8 a, b;
  for (int i = 0; i < 16; ++i) { x[i] = 0.001f * (threadIdx.x + i); y[i] = 0.5f + 0.001f * i; u[i] = threadIdx.x + i; }
  for (int i = 0; i < 8; ++i) p[i] = (f32x2){0.5f + i, 0.25f};
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  unsigned long long tot = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long t0 = now();
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      if (MODE == 0 || MODE == 2 || MODE == 7 || MODE == 12) EXP(i);
      if (MODE == 1 || MODE == 2 || MODE == 9) FMA(i);
      if (MODE == 12) { FMA(i); FMA(i + 5); FMA(i + 9); }
      if (MODE == 3) PKFMA(i);
      if (MODE == 4) SQRT(i);
      if (MODE == 5) CVTPK(i);
      if (MODE == 6) FMAMIX(i);
      if (MODE == 10) MAX3(i);
      if (MODE == 11) MOV(i);
      if (MODE == 13) CVTRTZ(i);
      if (MODE == 14) PKMUL(i);
      if (MODE == 15) PKADD(i);
      if (MODE == 16) MUL(i);
      if (MODE == 17) PERM(i);
      if ((MODE == 7 || MODE == 8 || MODE == 9) && (i & 3) == 0) MFMA(i >> 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = now();
    tot += t1 - t0;
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i] + y[i] + (float)u[i] + acc[i & 3][i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 * WAVES + (threadIdx.x >> 6)] = tot;
  if (s == 123.456f) sink[0] = s;
}
template <int WAVES>
static void launch(int mode, unsigned long long* o, float* k, int iters, int grid) {
#define CASE(m) if (mode == m) hipLaunchKernelGGL((probe<m, WAVES>), dim3(grid), dim3(256 * WAVES), 0, 0, o, k, iters)
  CASE(0); CASE(1); CASE(2); CASE(3); CASE(4); CASE(5); CASE(6); CASE(7); CASE(8); CASE(9); CASE(10); CASE(11); CASE(12); CASE(13); CASE(14); CASE(15); CASE(16); CASE(17);
}
extern "C" int run_probe(void* out, void* sink, int iters, int grid, int mode, int waves) {
  if (waves == 1) launch<1>(mode, (unsigned long long*)out, (float*)sink, iters, grid);
  else if (waves == 2) launch<2>(mode, (unsigned long long*)out, (float*)sink, iters, grid);
  else launch<3>(mode, (unsigned long long*)out, (float*)sink, iters, grid);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
