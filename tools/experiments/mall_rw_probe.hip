// Does the Infinity Cache keep lines a kernel has just WRITTEN, and what is a hit worth to a streaming reader?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mall_rw_probe.hip -o /tmp/mall_rw_probe && /tmp/mall_rw_probe
// For S in 32 ... 768 MiB: (a) write S, read S in the same order; (b) write S, read S in the OPPOSITE order (the reader starts on the
// lines the writer wrote last: reuse distance of line x = 2 * (S - x)); (c) read S twice (second read timed: a read-allocated table);
// (d) read S after a 1 GiB fill (cold).  Streaming shape of the LayerNorm family: 256-thread workgroups, 16 B per lane, grid-stride
// over 16 KiB pieces, 2048 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void write_k(f32x4* p, long pieces, int rev, float v) {
  for (long i = blockIdx.x; i < pieces; i += gridDim.x) {
    const long pc = rev ? pieces - 1 - i : i;
    f32x4* q = p + pc * 1024 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j * 256] = (f32x4){v, v, v, v};
  }
}
__global__ __launch_bounds__(256) void read_k(const f32x4* p, long pieces, int rev, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = blockIdx.x; i < pieces; i += gridDim.x) {
    const long pc = rev ? pieces - 1 - i : i;
    const f32x4* q = p + pc * 1024 + threadIdx.x;
    f32x4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = q[j * 256];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += r[j];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[blockIdx.x] = acc[0];
}

// consumer shaped like the LayerNorm family: reads `hot` (just written by the producer), optionally `cold` (written long ago), writes out
__global__ __launch_bounds__(256) void copy_k(const f32x4* hot, const f32x4* cold, f32x4* out, long pieces, int rev) {
  for (long i = blockIdx.x; i < pieces; i += gridDim.x) {
    const long pc = rev ? pieces - 1 - i : i;
    const long o = pc * 1024 + threadIdx.x;
    f32x4 r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = hot[o + j * 256];
    if (cold) {
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] += cold[o + j * 256];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[o + j * 256] = r[j];
  }
}

int main() {
  const long MAXB = 768L << 20;
  f32x4 *buf, *junk; float* sink;
  CK(hipMalloc(&buf, MAXB)); CK(hipMalloc(&junk, 1L << 30)); CK(hipMalloc(&sink, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 2048;
  auto timed_read = [&](long pieces, int rev) { float ms; hipEventRecord(e0, 0); hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, buf, pieces, rev, sink); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; };
  auto timed_write = [&](long pieces, int rev) { float ms; hipEventRecord(e0, 0); hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, buf, pieces, rev, 1.0f); hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); return ms; };
  auto fill = [&]() { hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, junk, (1L << 30) / 16384, 0, 2.0f); };
  printf("   S MiB | read after write, same order | opposite order | second read | cold read | write cold | write after read (TB/s)\n");
  for (long mb : {32L, 64L, 96L, 128L, 192L, 256L, 320L, 384L, 512L, 768L}) {
    const long bytes = mb << 20, pieces = bytes / 16384;
    double r[6] = {0, 0, 0, 0, 0, 0};
    const int reps = 5;
    for (int it = 0; it < reps + 1; ++it) {
      float t[6];
      fill(); hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, buf, pieces, 0, 1.0f); t[0] = timed_read(pieces, 0);
      fill(); hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, buf, pieces, 0, 1.0f); t[1] = timed_read(pieces, 1);
      fill(); hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, buf, pieces, 0, sink); t[2] = timed_read(pieces, 0);
      fill(); t[3] = timed_read(pieces, 0);
      fill(); t[4] = timed_write(pieces, 0);
      fill(); hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, buf, pieces, 0, sink); t[5] = timed_write(pieces, 1);
      if (it) for (int k = 0; k < 6; ++k) r[k] += t[k];
    }
    printf("%8ld |", mb);
    for (int k = 0; k < 6; ++k) printf(" %6.2f (%6.1f us) |", bytes / (r[k] / reps * 1e-3) / 1e12, r[k] / reps * 1e3);
    printf("\n"); fflush(stdout);
  }
  // ---- producer (writes `hot` in ascending order behind a cold start) -> consumer copy in the same / the opposite order
  f32x4 *cold, *out; CK(hipMalloc(&cold, MAXB)); CK(hipMalloc(&out, MAXB));
  printf("consumer = copy (hot [+ cold] -> out); S MiB each | same order us | opposite us | with cold operand: same | opposite\n");
  for (long mb : {44L, 88L, 132L, 176L, 264L}) {
    const long bytes = mb << 20, pieces = bytes / 16384;
    double r[4] = {0, 0, 0, 0};
    const int reps = 5;
    for (int it = 0; it < reps + 1; ++it) {
      for (int k = 0; k < 4; ++k) {
        fill();
        hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, buf, pieces, 0, 1.0f);
        float ms; hipEventRecord(e0, 0);
        hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, buf, (k & 2) ? cold : (const f32x4*)nullptr, out, pieces, k & 1);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (it) r[k] += ms;
      }
    }
    printf("%8ld |", mb);
    for (int k = 0; k < 4; ++k) printf(" %7.1f us (%5.2f TB/s) |", r[k] / reps * 1e3, (k & 2 ? 3 : 2) * bytes / (r[k] / reps * 1e-3) / 1e12);
    printf("\n"); fflush(stdout);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
