// Diagnostic (not part of the product library): a small-footprint VICTIM kernel that checks its own LDS reads.
// Every workgroup fills `lds_bytes` of LDS with a pattern that is a function of the byte address, then reads it back `iters` times
// with ds_read_b128, ds_read_b64 and ds_read2_b32 at lane-dependent addresses and counts words that differ from the pattern.
// Run beside a product kernel on another stream (tools/diag/lds_probe.py): a non-zero count means that kernel's waves disturb the
// LDS reads of a co-resident workgroup of another kernel (round 6: ds_read_b64_tr_b16 in mt_gemm_tn_f16 did).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pat(unsigned dword_index) { return dword_index * 2654435761u + 0x9e3779b9u; }

extern "C" __global__ void lds_probe_kernel(int iters, int lds_dwords, unsigned long long* __restrict__ result) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < lds_dwords; i += nt) smem[i] = pat(i);
  __syncthreads();
  unsigned long long bad128 = 0, bad64 = 0, bad32 = 0, bad128b = 0, bad64b = 0, bad32b = 0;
  unsigned first = 0xffffffffu, got = 0, want = 0;
  const int n16 = lds_dwords / 4;
  for (int it = 0; it < iters; ++it) {
    // b128: lane-contiguous 16-byte reads, start rotating with the iteration
    const int c = (tid + it * 67) % n16;
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned w = pat(4 * c + e);
      if (v[e] != w) { ++bad128; if (first == 0xffffffffu) { first = (unsigned)(4 * c + e) | (e << 28) | (0u << 30); got = v[e]; want = w; } }
    }
    // b64
    const int c2 = (tid * 2 + it * 131) % (lds_dwords / 2);
    const u32x2 v2 = *reinterpret_cast<const u32x2*>(smem + 2 * c2);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned w = pat(2 * c2 + e);
      if (v2[e] != w) { ++bad64; if (first == 0xffffffffu) { first = (unsigned)(2 * c2 + e) | (1u << 30); got = v2[e]; want = w; } }
    }
    // BROADCAST reads: the address depends on lane & 3 only (sixteen lanes of a wave share each address), as a kernel that sweeps a
    // small table with all its threads does (mt_token_mha_fwd's value rows)
    {
      const int cb = ((it * 4 + (tid & 3)) * 1) % n16;
      const u32x4 vb = *reinterpret_cast<const u32x4*>(smem + 4 * cb);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned w = pat(4 * cb + e);
        if (vb[e] != w) { ++bad128b; if (first == 0xffffffffu) { first = (unsigned)(4 * cb + e) | (e << 28) | (3u << 30); got = vb[e]; want = w; } }
      }
      const int cb2 = ((it * 4 + (tid & 3)) * 3 + 1) % (lds_dwords / 2);
      const u32x2 vb2 = *reinterpret_cast<const u32x2*>(smem + 2 * cb2);
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (vb2[e] != pat(2 * cb2 + e)) ++bad64b;
      const int cb3 = (it * 7 + (tid & 3)) % lds_dwords;
      if (smem[cb3] != pat(cb3)) ++bad32b;
    }
    // two separate dwords (ds_read2_b32 when the compiler pairs them)
    const int c3 = (tid + it * 29) % (lds_dwords - 8);
    const unsigned a = smem[c3], b = smem[c3 + 4];
    if (a != pat(c3)) { ++bad32; if (first == 0xffffffffu) { first = (unsigned)c3 | (2u << 30); got = a; want = pat(c3); } }
    if (b != pat(c3 + 4)) { ++bad32; if (first == 0xffffffffu) { first = (unsigned)(c3 + 4) | (2u << 30); got = b; want = pat(c3 + 4); } }
  }
  if (bad128) atomicAdd(&result[0], bad128);
  if (bad64) atomicAdd(&result[1], bad64);
  if (bad32) atomicAdd(&result[2], bad32);
  if (bad128b) atomicAdd(&result[6], bad128b);
  if (bad64b) atomicAdd(&result[7], bad64b);
  if (bad32b) atomicAdd(&result[8], bad32b);
  if (first != 0xffffffffu) {
    // one record per faulting thread is enough: last writer wins
    result[4] = ((unsigned long long)first << 32) | (unsigned)(tid | (blockIdx.x << 12));
    result[5] = ((unsigned long long)got << 32) | want;
    atomicAdd(&result[3], 1ull);
  }
}

extern "C" int lds_probe_launch(int grid, int threads, int lds_bytes, int iters, unsigned long long* result, hipStream_t s) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)lds_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(lds_probe_kernel, dim3(grid), dim3(threads), (size_t)lds_bytes, s, iters, lds_bytes / 4, result);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Second victim: the read pattern of mt_token_mha_fwd's third phase, densely -- a [rows][16] dword table swept by every thread with
// back-to-back 16-byte reads whose address depends on the sweep position and on (tid & 3) only.
extern "C" __global__ void lds_probe_sweep_kernel(int iters, int rows, unsigned long long* __restrict__ result) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int tid = threadIdx.x, nt = blockDim.x, sub = tid & 3;
  for (int i = tid; i < rows * 16; i += nt) smem[i] = pat(i);
  u32x4 want = {0, 0, 0, 0};
  for (int j = 0; j < rows; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) want[e] ^= pat(j * 16 + sub * 4 + e) * (unsigned)(j + 1);
  __syncthreads();
  unsigned long long bad[4] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 acc = {0, 0, 0, 0};
    for (int j = 0; j < rows; ++j) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + j * 16 + sub * 4);
      acc ^= v * (unsigned)(j + 1);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) bad[e] += acc[e] != want[e];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (bad[e]) { atomicAdd(&result[e], bad[e]); atomicAdd(&result[4 + ((tid & 63) >> 4)], bad[e]); }
}

extern "C" int lds_probe_sweep_launch(int grid, int threads, int rows, int iters, unsigned long long* result, hipStream_t s) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)lds_probe_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(lds_probe_sweep_kernel, dim3(grid), dim3(threads), (size_t)rows * 64 + 17160, s, iters, rows, result);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Third victim: does a COUNTED s_waitcnt lgkmcnt(N) still mean "all but the N youngest LDS operations have returned"?
// One asm block per iteration: clear the destination registers, issue [6 x ds_read_b128, 4 x {ds_read2_b32 | ds_read_b64}, 2 x ds_read_b128]
// (mt_token_mha_fwd's third phase), s_waitcnt lgkmcnt(5), COPY the seven destinations that the count says are back, s_waitcnt lgkmcnt(0),
// compare the copies with the registers' final contents.  MIXED = 1: the four middle reads are ds_read2_b32 (4-byte banking class),
// 0: ds_read_b64 (8 / 16-byte class, like the ds_read_b128 around them).
template <int MIXED>
__global__ void lds_probe_count_kernel(int iters, int rows, unsigned long long* __restrict__ result) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int tid = threadIdx.x, nt = blockDim.x, sub = tid & 3, t = tid >> 2;
  for (int i = tid; i < rows * 16 + 8192; i += nt) smem[i] = pat(i);
  __syncthreads();
  const unsigned a128 = (unsigned)(sub * 16);                       // byte address of the 16-byte sweep (the value rows)
  unsigned a32 = (unsigned)((rows * 16 + (t * 67) % 4096) * 4);     // byte address of the 4-byte sweep (this thread's score row)
  a32 &= ~7u;
  unsigned long long early = 0, early_any = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned b128 = a128 + (unsigned)((it % (rows > 8 ? rows - 8 : 1)) * 64);
    // fixed registers: v64..v95 the eight 16-byte destinations (the sixth is v[84:87]), v96..v103 the four 8-byte ones
    unsigned c5, f5, cq, fq, keep;
#define PROBE_HEAD                                                                                                                  \
    "v_mov_b32 v84, 0\n v_mov_b32 v96, 0\n v_mov_b32 v64, 0\n v_mov_b32 v68, 0\n s_nop 4\n"                                       \
    "ds_read_b128 v[64:67], %5\n ds_read_b128 v[68:71], %5 offset:64\n ds_read_b128 v[72:75], %5 offset:128\n"                    \
    "ds_read_b128 v[76:79], %5 offset:192\n ds_read_b128 v[80:83], %5 offset:256\n ds_read_b128 v[84:87], %5 offset:320\n"
#define PROBE_TAIL                                                                                                                  \
    "ds_read_b128 v[88:91], %5 offset:384\n ds_read_b128 v[92:95], %5 offset:448\n"                                               \
    "s_waitcnt lgkmcnt(5)\n"                                                                                                       \
    "v_mov_b32 %0, v84\n v_mov_b32 %2, v96\n"      /* the count says the sixth 16-byte read and the first 8-byte one are back */  \
    "s_waitcnt lgkmcnt(0)\n"                                                                                                       \
    "v_mov_b32 %1, v84\n v_mov_b32 %3, v96\n"                                                                                      \
    "v_xor_b32 %4, v65, v69\n v_xor_b32 %4, %4, v73\n v_xor_b32 %4, %4, v77\n v_xor_b32 %4, %4, v81\n v_xor_b32 %4, %4, v89\n"    \
    "v_xor_b32 %4, %4, v93\n v_xor_b32 %4, %4, v98\n v_xor_b32 %4, %4, v100\n v_xor_b32 %4, %4, v102\n"
#define PROBE_CLOB "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", \
    "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", \
    "v101", "v102", "v103"
    if (MIXED) {
      asm volatile(PROBE_HEAD
                   "ds_read2_b32 v[96:97], %6 offset1:1\n ds_read2_b32 v[98:99], %6 offset0:2 offset1:3\n"
                   "ds_read2_b32 v[100:101], %6 offset0:4 offset1:5\n ds_read2_b32 v[102:103], %6 offset0:6 offset1:7\n" PROBE_TAIL
                   : "=&v"(c5), "=&v"(f5), "=&v"(cq), "=&v"(fq), "=&v"(keep) : "v"(b128), "v"(a32) : PROBE_CLOB);
    } else {
      asm volatile(PROBE_HEAD
                   "ds_read_b64 v[96:97], %6\n ds_read_b64 v[98:99], %6 offset:8\n ds_read_b64 v[100:101], %6 offset:16\n"
                   "ds_read_b64 v[102:103], %6 offset:24\n" PROBE_TAIL
                   : "=&v"(c5), "=&v"(f5), "=&v"(cq), "=&v"(fq), "=&v"(keep) : "v"(b128), "v"(a32) : PROBE_CLOB);
    }
    const bool e1 = c5 != f5, e2 = cq != fq;
    early += e1; early_any += (e1 || e2);
    if (keep == 0x12345678u && iters < 0) early += 1000000;
  }
  if (early_any) { atomicAdd(&result[0], early); atomicAdd(&result[1], early_any); atomicAdd(&result[4 + ((tid & 63) >> 4)], early_any); }
}

extern "C" int lds_probe_count_launch(int mixed, int grid, int threads, int rows, int iters, unsigned long long* result, hipStream_t s) {
  const size_t shm = ((size_t)rows * 16 + 8192 + 64) * 4;
  if (mixed) hipLaunchKernelGGL(lds_probe_count_kernel<1>, dim3(grid), dim3(threads), shm, s, iters, rows, result);
  else hipLaunchKernelGGL(lds_probe_count_kernel<0>, dim3(grid), dim3(threads), shm, s, iters, rows, result);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
