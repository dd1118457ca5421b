// Distillation loss head, fused AdamW with GradScaler semantics, and library metadata.
#include "common.h"

namespace {

// One workgroup: R <= 8 rows of O <= 1024 logits.  train_modaltune.py:225-233:
//   z = logit / ||logit||; loss = 10 * sum_r sum_j p_rj (log p_rj - log_softmax(z_r)_j), p = softmax(target_r)
// d loss / d z_rj = 10 * (softmax(z_r)_j - p_rj)   (sum_j p_rj = 1);  d/d logit = (dz - z (z . dz)) / ||logit||
__global__ __launch_bounds__(256) void distill_loss_kernel(const float* __restrict__ logits, const float* __restrict__ target, int R, int O,
                                                           float loss_scale_host, const float* __restrict__ scale_dev,
                                                           float* __restrict__ loss, float* __restrict__ dlogits) {
  const float loss_scale = loss_scale_host * (scale_dev ? *scale_dev : 1.0f);
  __shared__ float red[256];
  __shared__ float bc[4];
  const int tid = threadIdx.x;
  auto block_sum = [&](float v) {
    red[tid] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float r = red[0]; __syncthreads(); return r;
  };
  auto block_max = [&](float v) {
    red[tid] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    const float r = red[0]; __syncthreads(); return r;
  };
  float total = 0.f;
  for (int r = 0; r < R; ++r) {
    const float* lg = logits + (long)r * O;
    const float* tg = target + (long)r * O;
    float ss = 0.f;
    for (int j = tid; j < O; j += 256) ss += lg[j] * lg[j];
    const float nrm = sqrtf(block_sum(ss));
    float zm = -1.0e30f, tm = -1.0e30f;
    for (int j = tid; j < O; j += 256) { zm = fmaxf(zm, lg[j] / nrm); tm = fmaxf(tm, tg[j]); }
    zm = block_max(zm); tm = block_max(tm);
    float ze = 0.f, te = 0.f;
    for (int j = tid; j < O; j += 256) { ze += expf(lg[j] / nrm - zm); te += expf(tg[j] - tm); }
    ze = block_sum(ze); te = block_sum(te);
    const float zl = zm + logf(ze), tl = tm + logf(te);
    float part = 0.f, zdz = 0.f;
    for (int j = tid; j < O; j += 256) {
      const float z = lg[j] / nrm;
      const float logq = z - zl, logp = tg[j] - tl, p = expf(logp);
      part += p * (logp - logq);
      const float dz = 10.0f * (expf(logq) - p);
      zdz += z * dz;
    }
    part = block_sum(part); zdz = block_sum(zdz);
    total += 10.0f * part;
    for (int j = tid; j < O; j += 256) {
      const float z = lg[j] / nrm;
      const float dz = 10.0f * (expf(z - zl) - expf(tg[j] - tl));
      dlogits[(long)r * O + j] = loss_scale * (dz - z * zdz) / nrm;
    }
  }
  if (tid == 0) *loss = total;
  (void)bc;
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             float lr, float b1, float b2, float omb1, float omb2, float eps, double wd, float decay, float bc1, float bc2s,
                             double b1d, double b2d, float grad_mult, const float* __restrict__ scale, const int* __restrict__ found_inf,
                             const int* __restrict__ step_dev, const float* __restrict__ lr_dev) {
  if (found_inf && *found_inf) return;       // GradScaler.step: skip the whole update when a grad is inf/nan
  // the schedule's current learning rate lives on the device (graph replays read it); 1 - lr * wd in double, as torch forms it
  if (lr_dev) { lr = *lr_dev; decay = (float)(1.0 - (double)lr * wd); }
  if (step_dev) {                            // optimiser step count lives on the device (skipped steps do not count)
    // bias corrections in DOUBLE, as torch's Python arithmetic forms them (1 - beta ** step): powf in fp32 is ~1e-5 off in the first
    // steps, where 1 - beta2 ** step is itself ~1e-3 (two pows per thread against >= 32 elements of 28 bytes each: not measurable)
    const double st = (double)(*step_dev + 1);
    bc1 = (float)(1.0 - pow(b1d, st));
    bc2s = (float)sqrt(1.0 - pow(b2d, st));
  }
  const float inv_scale = grad_mult * (scale ? 1.0f / *scale : 1.0f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * inv_scale;
    float pi = p[i] * decay;
    const float mi = b1 * m[i] + omb1 * gi;                 // (omb = 1 - beta formed in double on the host, as torch does)
    const float vi = b2 * v[i] + omb2 * gi * gi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

__global__ void check_finite_kernel(const float* __restrict__ g, long n, int* __restrict__ found_inf) {
  bool bad = false;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    bad |= !isfinite(g[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(found_inf, 1);
}

// y[r] = x[r] / ||x[r]||_2 : one workgroup per row (text / ||text||, TM:213)
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int O) {
  __shared__ float red[256];
  const float* xr = x + (long)blockIdx.x * O;
  float s = 0.f;
  for (int j = threadIdx.x; j < O; j += 256) s += xr[j] * xr[j];
  red[threadIdx.x] = s; __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  const float inv = 1.0f / sqrtf(red[0]);
  for (int j = threadIdx.x; j < O; j += 256) y[(long)blockIdx.x * O + j] = xr[j] * inv;
}

// torch.cuda.amp.GradScaler.update (TM:107,237): backoff 0.5 on overflow, x2 after `interval` clean steps.
__global__ void scaler_update_kernel(float* scale, int* tracker, int* found_inf, int* step_dev, float growth, float backoff, int interval) {
  if (threadIdx.x || blockIdx.x) return;
  if (*found_inf) { *scale *= backoff; *tracker = 0; }
  else {
    if (step_dev) ++(*step_dev);
    if (++(*tracker) >= interval) { *scale *= growth; *tracker = 0; }
  }
  *found_inf = 0;
}


// s[0] = target / max|x| (1 when the maximum is 0 or not finite), s[1] = 1 / s[0]: the device-side loss-scale of the
// nn.Module bridge (no host read-back of the incoming gradient's range).  One workgroup; n is a few hundred.
__global__ __launch_bounds__(256) void absmax_scale_kernel(const float* __restrict__ x, long n, float target, float* __restrict__ s) {
  __shared__ float red[256];
  float m = 0.f;
  bool bad = false;
  for (long i = threadIdx.x; i < n; i += 256) { const float v = fabsf(x[i]); bad |= !(v <= 3.0e38f); m = fmaxf(m, v); }
  red[threadIdx.x] = bad ? __builtin_inff() : m; __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + k]); __syncthreads(); }
  if (threadIdx.x == 0) {
    const float a = red[0];
    const float sc = (a > 0.f && a <= 3.0e38f) ? target / a : 1.0f;
    s[0] = sc; s[1] = 1.0f / sc;
  }
}

// y = a + (*alpha) * b (a may be null: y = (*alpha) * b)
__global__ void axpy_dev_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ alpha,
                                float* __restrict__ y, long n) {
  const float al = *alpha;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (a ? a[i] : 0.f) + al * b[i];
}

// grid cell of every patch: row = floor(coords[:, 0] / tile), col = floor(coords[:, 1] / tile) (slide_encoder.py:209-211);
// cells outside [0, ngrids) are clamped and reported through *err (checked by the host when it next synchronises)
__global__ void coords_to_grid_kernel(const float* __restrict__ coords, int L, float tile, int ngrids, int* __restrict__ prow,
                                      int* __restrict__ pcol, int* __restrict__ err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const float fr = floorf(coords[2 * i] / tile), fc = floorf(coords[2 * i + 1] / tile);
  // (int) of a NaN / out-of-range float is unspecified: test the floats (NaN fails every comparison -> bad)
  const bool bad = !(fr >= 0.f && fc >= 0.f && fr < (float)ngrids && fc < (float)ngrids);
  const int r = bad ? 0 : (int)fr, c = bad ? 0 : (int)fc;
  prow[i] = min(max(r, 0), ngrids - 1);
  pcol[i] = min(max(c, 0), ngrids - 1);
  if (bad && err) atomicOr(err, 1);
}

// dst(idx[m], :) (+)= src(m, :): rows scattered by index (TITAN feature gridding: index_add of the patch features into
// their grid cells, titan_adapter.py:318-320); atomics because several patches may fall into one cell
__global__ void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ src_idx, const int* __restrict__ idx,
                                    float* __restrict__ dst, int M, int D, int accumulate) {
  const long n = (long)M * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / D), d = (int)(i - (long)m * D);
    const float v = src[(long)(src_idx ? src_idx[m] : m) * D + d];
    float* p = dst + (long)idx[m] * D + d;
    if (accumulate) atomicAdd(p, v);
    else *p = v;
  }
}

// out[m] = max_d |x(m, d)|: one wave per row
__global__ __launch_bounds__(256) void row_absmax_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long m = (long)blockIdx.x * 4 + wave; m < M; m += (long)gridDim.x * 4) {
    float v = 0.f;
    for (int d = lane; d < D; d += 64) v = fmaxf(v, fabsf(x[m * D + d]));
    v = wave_max(v);
    if (lane == 0) out[m] = v;
  }
}

// Bare MFMA loop (bench.py: the chip's own fp16 MFMA rate next to the vendor peak): every wave issues
// `iters` x 4 independent v_mfma_f32_32x32x16_f16 on pseudo-random register operands (the clock the chip holds depends on the
// data: all-zero operands read ~19 % high, MI355X_MICROARCH.md "DVFS give-back"), 4 waves per workgroup = one per SIMD.
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* __restrict__ sink, int iters) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  h16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    unsigned h = (t * 9781u + e * 6271u) * 2654435761u;
    a[e] = (h16)(((int)(h >> 20) & 1023) * (1.0f / 1024.0f) - 0.5f);
    b[e] = (h16)(((int)(h >> 8) & 1023) * (1.0f / 1024.0f) - 0.5f);
  }
  f32x16 c0, c1, c2, c3;
#pragma unroll
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, c3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if (s == 123.456f) sink[0] = s;      // (keeps the chains alive; practically never taken)
}

}  // namespace

extern "C" int mt_l2norm_rows(const float* x, float* y, int R, int O, mt_stream_t stream) {
  if (!x || !y || R < 1 || O < 1) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, x, y, O);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_scaler_update(float* scale, int* growth_tracker, int* found_inf, int* step_dev, float growth,
                                float backoff, int interval, mt_stream_t stream) {
  if (!scale || !growth_tracker || !found_inf || interval < 1) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scale, growth_tracker, found_inf, step_dev,
                     growth, backoff, interval);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_distill_loss(const float* logits, const float* target, int R, int O, float loss_scale,
                               const float* scale_dev, float* loss, float* dlogits, mt_stream_t stream) {
  if (!logits || !target || !loss || !dlogits || R < 1 || O < 1) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(distill_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, R, O, loss_scale, scale_dev,
                     loss, dlogits);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_adamw_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int step_count, const int* step_dev, float grad_mult,
                             const float* scale, int* found_inf, const float* lr_dev, mt_stream_t stream) {
  if (!p || !g || !m || !v || n <= 0 || (step_count < 1 && !step_dev)) return MT_ERR_BAD_ARG;
  if (step_count < 1) step_count = 1;
  const float bc1 = (float)(1.0 - pow(beta1, (double)step_count));
  const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)step_count));
  const int grid = (int)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)lr, (float)beta1, (float)beta2,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, weight_decay, (float)(1.0 - lr * weight_decay), bc1, bc2s,
                     beta1, beta2, grad_mult, scale, (const int*)found_inf, step_dev, lr_dev);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_check_finite(const float* g, long n, int* found_inf, mt_stream_t stream) {
  if (!g || !found_inf || n <= 0) return MT_ERR_BAD_ARG;
  const int grid = (int)((n + 1023) / 1024 > 2048 ? 2048 : (n + 1023) / 1024);
  hipLaunchKernelGGL(check_finite_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, found_inf);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_absmax_scale(const float* x, long n, float target, float* s, mt_stream_t stream) {
  if (!x || !s || n <= 0 || !(target > 0.f)) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(absmax_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, target, s);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_axpy_dev(const float* a, const float* b, const float* alpha, float* y, long n, mt_stream_t stream) {
  if (!b || !alpha || !y || n <= 0) return MT_ERR_BAD_ARG;
  const int grid = (int)((n + 1023) / 1024 > 4096 ? 4096 : (n + 1023) / 1024);
  hipLaunchKernelGGL(axpy_dev_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, alpha, y, n);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_mfma_probe(float* sink, int workgroups, int iters, mt_stream_t stream) {
  if (!sink || workgroups < 1 || iters < 1) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, sink, iters);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_coords_to_grid(const float* coords, int L, float tile, int ngrids, int* prow, int* pcol, int* err,
                                 mt_stream_t stream) {
  if (!coords || !prow || !pcol || L < 1 || !(tile > 0.f) || ngrids < 1) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(coords_to_grid_kernel, dim3((L + 255) / 256), dim3(256), 0, (hipStream_t)stream, coords, L, tile, ngrids, prow,
                     pcol, err);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_scatter_rows_f32(const float* src, const int* src_idx, const int* idx, float* dst, int M, int D, int accumulate,
                                   mt_stream_t stream) {
  if (!src || !idx || !dst || M < 1 || D < 1) return MT_ERR_BAD_ARG;
  const long n = (long)M * D;
  const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, src_idx, idx, dst, M, D, accumulate);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_row_absmax_f32(const float* x, float* out, int M, int D, mt_stream_t stream) {
  if (!x || !out || M < 1 || D < 1) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(row_absmax_kernel, dim3((M + 3) / 4 > 4096 ? 4096 : (M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, out, M, D);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_version(void) { return 210; }      // (2.1: mt_build_id; the binary itself is identified by mt_build_id(), not by this number)

extern "C" const char* mt_status_string(int status) {
  switch (status) {
    case MT_OK: return "ok";
    case MT_ERR_BAD_ARG: return "bad argument (shape / alignment / null pointer)";
    case MT_ERR_LAUNCH: return "kernel launch failed";
    case MT_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}
