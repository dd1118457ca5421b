// TITAN configuration (BASELINE config 4), the pieces around the dense attention of dense_attn.hip:
//   * feature gridding of `TITANGeneAdapter.preprocess_features` + the background drop of `prepare_forward_features`
//     (models/aggregators/titan_adapter.py:295-327, 282-291) entirely on the device, without materialising the H x W grid:
//     the tokens a slide contributes are its OCCUPIED cells in row-major order, each the sum of its patches in patch order;
//   * erf-GELU of the ViT block's MLP on fp16 activations (forward / backward);
//   * the attentional pooling core (TA:401-402 `forward_attn_pool`): a few learned queries attending over all N tokens.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ feature gridding
// cells[i] = floor((coords[i] - min coords) / patch) (TA:304-312; the second offset of the reference is zero by construction),
// dims = {H, W} = max cell + 1.  ONE workgroup: L is a slide's patch count (<= a few 10^4).
__global__ __launch_bounds__(1024) void titan_grid_kernel(const float* __restrict__ coords, int L, float patch, int* __restrict__ cells,
                                                          int* __restrict__ dims, int* __restrict__ err) {
  __shared__ float red[2][16];
  __shared__ int redi[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m0 = 3.0e38f, m1 = 3.0e38f;
  bool bad = false;
  for (int i = tid; i < L; i += 1024) {
    const float a = coords[2 * i], b = coords[2 * i + 1];
    bad |= !(fabsf(a) < 1.0e9f) || !(fabsf(b) < 1.0e9f);      // NaN / inf / absurd
    m0 = fminf(m0, a); m1 = fminf(m1, b);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { m0 = fminf(m0, __shfl_xor(m0, o, 64)); m1 = fminf(m1, __shfl_xor(m1, o, 64)); }
  if (lane == 0) { red[0][wave] = m0; red[1][wave] = m1; }
  __syncthreads();
  m0 = red[0][0]; m1 = red[1][0];
  for (int k = 1; k < 16; ++k) { m0 = fminf(m0, red[0][k]); m1 = fminf(m1, red[1][k]); }
  int g0m = 0, g1m = 0;
  for (int i = tid; i < L; i += 1024) {
    const float f0 = floorf((coords[2 * i] - m0) / patch), f1 = floorf((coords[2 * i + 1] - m1) / patch);
    const bool ok = f0 >= 0.f && f1 >= 0.f && f0 < 32768.f && f1 < 32768.f;      // cell_key = (row << 16) | col stays a non-negative int, disjoint from the -1 / 0x7fffffff sentinels
    bad |= !ok;
    const int g0 = ok ? (int)f0 : 0, g1 = ok ? (int)f1 : 0;
    cells[2 * i] = g0; cells[2 * i + 1] = g1;
    g0m = max(g0m, g0); g1m = max(g1m, g1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { g0m = max(g0m, __shfl_xor(g0m, o, 64)); g1m = max(g1m, __shfl_xor(g1m, o, 64)); }
  if (lane == 0) { redi[0][wave] = g0m; redi[1][wave] = g1m; }
  if (bad && err) atomicOr(err, 1);
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < 16; ++k) { g0m = max(g0m, redi[0][k]); g1m = max(g1m, redi[1][k]); }
    dims[0] = max(g0m, redi[0][0]) + 1; dims[1] = max(g1m, redi[1][0]) + 1;
  }
}

MT_DEVINL int cell_key(const int* cells, int i) { return (cells[2 * i] << 16) | cells[2 * i + 1]; }

// Patches that share a cell form a chain in patch order: first[i] = no earlier patch has i's cell; next[i] = the next later
// patch of the same cell (or -1).  O(L^2) integer compares (L = 10^4: 10^8), no sort, no global atomics -- the sums below come out
// in patch order, bitwise reproducible (index_add_ on the CPU adds in that order).  One workgroup = 64 patches (one per lane) x 16
// waves: the workgroup stages 1024 keys at a time in LDS (one coalesced load per thread) and wave w scans the w-th 64 of them, four
// keys per broadcast ds_read_b128; the sixteen partial answers meet in LDS.  (Keys through the scalar cache, one dependent s_load
// pair per key: 47 us at L = 4 369 -- the kernel sits between two graph launches of the replayed step; one thread per patch
// scanning all L keys left 26 workgroups on the chip at L = 6 500: 220-350 us.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void titan_chain_kernel(const int* __restrict__ cells, int L, int* __restrict__ first, int* __restrict__ next) {
  __shared__ int s_first[64], s_next[64];
  __shared__ __attribute__((aligned(16))) int keys[1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int mine = i < L ? cell_key(cells, i) : -1;
  if (wave == 0) { s_first[lane] = 1; s_next[lane] = 0x7fffffff; }
  bool fst = true;
  int nxt = 0x7fffffff;
  for (int base = 0; base < L; base += 1024) {
    __syncthreads();
    const int jl = base + (int)threadIdx.x;
    keys[threadIdx.x] = jl < L ? cell_key(cells, jl) : -2;      // (-2: matches no patch, not even the -1 of the lanes past L)
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      const i32x4 kk = *reinterpret_cast<const i32x4*>(&keys[wave * 64 + k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = base + wave * 64 + k + e;
        if (kk[e] == mine) {
          if (j < i) fst = false;
          else if (j > i) nxt = min(nxt, j);
        }
      }
    }
  }
  if (!fst) s_first[lane] = 0;                  // (racing stores of the same value)
  if (nxt != 0x7fffffff) atomicMin(&s_next[lane], nxt);
  __syncthreads();
  if (wave == 0 && i < L) { first[i] = s_first[lane]; next[i] = s_next[lane] == 0x7fffffff ? -1 : s_next[lane]; }
}

// sums[i, :] = sum of the feature rows along i's chain (i the first patch of its cell), nz[i] = any(sum != 0): the reference's
// background mask is "any feature of the summed cell != 0" (TA:326).  One workgroup per patch; non-first patches exit.
__global__ __launch_bounds__(256) void titan_cell_sum_kernel(const float* __restrict__ feat, long ldf, int L, int C, const int* __restrict__ first,
                                                             const int* __restrict__ next, float* __restrict__ sums, int* __restrict__ nz) {
  const int i = blockIdx.x;
  if (!first[i]) { if (threadIdx.x == 0) nz[i] = 0; return; }
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  bool mine = false;
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    f32x4 acc = *reinterpret_cast<const f32x4*>(feat + (long)i * ldf + c);
    for (int j = next[i]; j >= 0; j = next[j]) acc += *reinterpret_cast<const f32x4*>(feat + (long)j * ldf + c);
    *reinterpret_cast<f32x4*>(sums + (long)i * C + c) = acc;
    mine |= acc[0] != 0.f || acc[1] != 0.f || acc[2] != 0.f || acc[3] != 0.f;
  }
  if (mine) any = 1;
  __syncthreads();
  if (threadIdx.x == 0) nz[i] = any;
}

// Token position of every occupied cell = its rank among the occupied cells in row-major order (the order `x[bg_mask]` keeps,
// TA:282-291); pos[i] = -1 for patches that own no token.  count[0] = number of tokens (without cls); cells_tok[pos] = (row, col).
// One workgroup = 64 patches (one per lane) x 4 waves: 1024 candidate keys at a time in LDS, wave w counts the smaller ones in its
// quarter (four keys per broadcast ds_read_b128); the four partial ranks meet in LDS.  (256 patches per workgroup, each thread walking
// every key one LDS read at a time: 18 workgroups and 53 us at L = 4 369.)
__global__ __launch_bounds__(256) void titan_order_kernel(const int* __restrict__ cells, const int* __restrict__ first, const int* __restrict__ nz,
                                                          int L, int* __restrict__ pos, int* __restrict__ cells_tok, int* __restrict__ count) {
  __shared__ __attribute__((aligned(16))) int keys[1024];
  __shared__ int s_rank[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const bool own = i < L && first[i] && nz[i];
  const int mine = i < L ? cell_key(cells, i) : 0;
  if (wave == 0) s_rank[lane] = 0;
  int rank = 0;
  for (int base = 0; base < L; base += 1024) {
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += 256) {
      const int j = base + k;
      keys[k] = (j < L && first[j] && nz[j]) ? cell_key(cells, j) : 0x7fffffff;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 256; k += 4) {
      const i32x4 kk = *reinterpret_cast<const i32x4*>(&keys[wave * 256 + k]);
      rank += (kk[0] < mine ? 1 : 0) + (kk[1] < mine ? 1 : 0) + (kk[2] < mine ? 1 : 0) + (kk[3] < mine ? 1 : 0);
    }
  }
  atomicAdd(&s_rank[lane], rank);
  __syncthreads();
  rank = s_rank[lane];
  if (wave == 0 && i < L) {
    pos[i] = own ? rank : -1;
    if (own) {
      cells_tok[2 * rank] = cells[2 * i]; cells_tok[2 * rank + 1] = cells[2 * i + 1];
      atomicAdd(count, 1);
    }
  }
}

// x16[pos[i], :] = fp16(sums[i, :]) for the patches that own a token (the A operand of the patch-embedding GEMM)
__global__ __launch_bounds__(256) void titan_gather_kernel(const float* __restrict__ sums, const int* __restrict__ pos, int L, int C,
                                                           h16* __restrict__ x16) {
  const int i = blockIdx.x;
  const int p = pos[i];
  if (p < 0) return;
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(sums + (long)i * C + c);
    const h16x4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
    *reinterpret_cast<h16x4*>(x16 + (long)p * C + c) = o;
  }
}

// ------------------------------------------------------------------------------------------------ MLP activation
__global__ __launch_bounds__(256) void gelu_f16_fwd_kernel(const h16* __restrict__ x, h16* __restrict__ y, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const h16x8 v = ldg8(x + i * 8);
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)gelu_erf((float)v[e]);
    stg8(y + i * 8, o);
  }
}
__global__ __launch_bounds__(256) void gelu_f16_bwd_kernel(const h16* __restrict__ x, const h16* __restrict__ dy, h16* __restrict__ dx, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const h16x8 v = ldg8(x + i * 8), g = ldg8(dy + i * 8);
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)((float)g[e] * gelu_erf_grad((float)v[e]));
    stg8(dx + i * 8, o);
  }
}

// ------------------------------------------------------------------------------------------------ attentional pooling core
// A few learned queries attend over the N tokens of every pass (flash-decoding form: the keys are split over workgroups).
// q fp32 [nq, E] (already projected, frozen); kv fp16 [B*N, 2E] (k | v, projected); hd = E / heads <= 128, hd % 8 == 0.
// Phase 1: workgroup = (pass, head, query, split): one key per thread; raw scaled scores saved (the backward recomputes
// p = exp(s - lse)), partial (max, sum, sum p v) per split.  Phase 2: combine the splits -> out, lse.
constexpr int POOL_KEYS = 256;      // keys per split = threads per workgroup
__global__ __launch_bounds__(256) void pool_attn_part_kernel(const float* __restrict__ q, const h16* __restrict__ kv, int B, int N, int E, int heads,
                                                             int nq, int nsplit, float scale, float* __restrict__ scores,
                                                             float* __restrict__ part) {
  __shared__ float qs[128];
  __shared__ float ps[POOL_KEYS];
  __shared__ float red[4];
  __shared__ float accs[4][128];
  const int hd = E / heads;
  const int sp = blockIdx.x % nsplit, iq = (blockIdx.x / nsplit) % nq, h = (blockIdx.x / (nsplit * nq)) % heads, b = blockIdx.x / (nsplit * nq * heads);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < hd) qs[tid] = q[(long)iq * E + h * hd + tid] * scale;
  __syncthreads();
  const int n = sp * POOL_KEYS + tid;
  float s = -3.0e38f;
  if (n < N) {
    const h16* kr = kv + ((long)b * N + n) * 2 * E + h * hd;
    s = 0.f;
    for (int d = 0; d < hd; d += 8) {
      const h16x8 kk = ldg8(kr + d);
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(qs[d + e], (float)kk[e], s);
    }
    scores[(((long)b * heads + h) * nq + iq) * N + n] = s;
  }
  float mx = wave_max(s);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float p = n < N ? __expf(s - mx) : 0.f;
  ps[tid] = p;
  float sum = wave_sum(p);
  __syncthreads();
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  // sum_n p[n] v[n][d]: thread -> (d = tid % 64 (+ 64), key group tid / 64); consecutive threads read consecutive halves of a row
  const int g = tid >> 6;
  float a0 = 0.f, a1 = 0.f;
  const int nk = min(POOL_KEYS, N - sp * POOL_KEYS);
  for (int j = g; j < nk; j += 4) {
    const h16* vr = kv + ((long)b * N + sp * POOL_KEYS + j) * 2 * E + E + h * hd;
    const float pj = ps[j];
    if (lane < hd) a0 = fmaf(pj, (float)vr[lane], a0);
    if (lane + 64 < hd) a1 = fmaf(pj, (float)vr[lane + 64], a1);
  }
  accs[g][lane] = a0; accs[g][lane + 64] = a1;
  __syncthreads();
  float* pp = part + (long)blockIdx.x * (2 + 128);
  if (tid < hd) pp[2 + tid] = accs[0][tid] + accs[1][tid] + accs[2][tid] + accs[3][tid];
  if (tid == 0) { pp[0] = mx; pp[1] = sum; }
}
__global__ __launch_bounds__(128) void pool_attn_combine_kernel(const float* __restrict__ part, int E, int heads, int nq, int nsplit,
                                                                float* __restrict__ out, float* __restrict__ lse) {
  const int hd = E / heads;
  const int iq = blockIdx.x % nq, h = (blockIdx.x / nq) % heads, b = blockIdx.x / (nq * heads);
  const float* pp = part + (long)blockIdx.x * nsplit * (2 + 128);
  float mx = -3.0e38f;
  for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, pp[s * 130]);
  float l = 0.f, acc = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float w = __expf(pp[s * 130] - mx);
    l = fmaf(w, pp[s * 130 + 1], l);
    if (threadIdx.x < hd) acc = fmaf(w, pp[s * 130 + 2 + threadIdx.x], acc);
  }
  if (threadIdx.x < hd) out[((long)b * nq + iq) * E + h * hd + threadIdx.x] = acc / l;
  if (threadIdx.x == 0) lse[((long)b * heads + h) * nq + iq] = mx + __logf(l);
}

// dkv (fp16 [B*N, 2E], overwritten): p = exp(s - lse) ; dV[n] = sum_q p dout_q ; dP = dout . v[n] ; dS = p (dP - dout . out) ;
// dK[n] = scale sum_q dS q.  One key per thread (the flash identity sum_m p[m] dP[m] = dout . out needs no second pass).
__global__ __launch_bounds__(256) void pool_attn_bwd_kernel(const float* __restrict__ q, const h16* __restrict__ kv, const float* __restrict__ scores,
                                                            const float* __restrict__ lse, const float* __restrict__ out,
                                                            const float* __restrict__ dout, int B, int N, int E, int heads, int nq, int nsplit,
                                                            float scale, h16* __restrict__ dkv) {
  __shared__ float qs[128], dos[128];
  __shared__ float tots;
  const int hd = E / heads;
  const int sp = blockIdx.x % nsplit, h = (blockIdx.x / nsplit) % heads, b = blockIdx.x / (nsplit * heads);
  const int tid = threadIdx.x;
  const int n = sp * POOL_KEYS + tid;
  for (int iq = 0; iq < nq; ++iq) {
    __syncthreads();
    if (tid < hd) {
      qs[tid] = q[(long)iq * E + h * hd + tid] * scale;
      dos[tid] = dout[((long)b * nq + iq) * E + h * hd + tid];
    }
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int d = 0; d < hd; ++d) t = fmaf(dos[d], out[((long)b * nq + iq) * E + h * hd + d], t);
      tots = t;
    }
    __syncthreads();
    if (n < N) {
      const h16* vr = kv + ((long)b * N + n) * 2 * E + E + h * hd;
      float dp = 0.f;
      for (int d = 0; d < hd; d += 8) {
        const h16x8 vv = ldg8(vr + d);
#pragma unroll
        for (int e = 0; e < 8; ++e) dp = fmaf(dos[d + e], (float)vv[e], dp);
      }
      const long row = ((long)b * heads + h) * nq + iq;
      const float p = __expf(scores[row * N + n] - lse[row]);
      const float ds = p * (dp - tots);
      h16* okp = dkv + ((long)b * N + n) * 2 * E + h * hd;
      h16* ovp = okp + E;
      for (int d = 0; d < hd; d += 8) {      // (further queries accumulate onto what the earlier ones wrote: same thread, same key)
        h16x8 a, c;
        if (iq > 0) { a = ldg8(okp + d); c = ldg8(ovp + d); }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = ds * qs[d + e], y = p * dos[d + e];
          a[e] = (h16)(iq > 0 ? (float)a[e] + x : x);
          c[e] = (h16)(iq > 0 ? (float)c[e] + y : y);
        }
        stg8(okp + d, a); stg8(ovp + d, c);
      }
    }
  }
}

}  // namespace

extern "C" int mt_titan_grid(const float* coords, int L, float patch, int* cells, int* dims, int* err, mt_stream_t stream) {
  if (!coords || !cells || !dims || L < 1 || !(patch > 0.f)) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(titan_grid_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, coords, L, patch, cells, dims, err);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_titan_cell_sums(const float* feat, long ldf, const int* cells, int L, int C, int* first, int* next, float* sums, int* nz,
                                  mt_stream_t stream) {
  if (!feat || !cells || !first || !next || !sums || !nz || L < 1 || C < 4 || (C & 3) || (ldf & 3)) return MT_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(titan_chain_kernel, dim3((L + 63) / 64), dim3(1024), 0, s, cells, L, first, next);
  hipLaunchKernelGGL(titan_cell_sum_kernel, dim3(L), dim3(256), 0, s, feat, ldf, L, C, (const int*)first, (const int*)next, sums, nz);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_titan_token_order(const int* cells, const int* first, const int* nz, int L, int* pos, int* cells_tok, int* count,
                                    mt_stream_t stream) {
  if (!cells || !first || !nz || !pos || !cells_tok || !count || L < 1) return MT_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(count, 0, sizeof(int), s) != hipSuccess) return MT_ERR_LAUNCH;
  hipLaunchKernelGGL(titan_order_kernel, dim3((L + 63) / 64), dim3(256), 0, s, cells, first, nz, L, pos, cells_tok, count);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_titan_gather_tokens(const float* sums, const int* pos, int L, int C, mt_half* x16, mt_stream_t stream) {
  if (!sums || !pos || !x16 || L < 1 || C < 4 || (C & 3)) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(titan_gather_kernel, dim3(L), dim3(256), 0, (hipStream_t)stream, sums, pos, L, C, (h16*)x16);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_gelu_f16_fwd(const mt_half* x, mt_half* y, long n, mt_stream_t stream) {
  if (!x || !y || n < 8 || (n & 7)) return MT_ERR_BAD_ARG;
  const long n8 = n / 8;
  hipLaunchKernelGGL(gelu_f16_fwd_kernel, dim3((int)min((n8 + 255) / 256, 16384L)), dim3(256), 0, (hipStream_t)stream, (const h16*)x,
                     (h16*)y, n8);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_gelu_f16_bwd(const mt_half* x, const mt_half* dy, mt_half* dx, long n, mt_stream_t stream) {
  if (!x || !dy || !dx || n < 8 || (n & 7)) return MT_ERR_BAD_ARG;
  const long n8 = n / 8;
  hipLaunchKernelGGL(gelu_f16_bwd_kernel, dim3((int)min((n8 + 255) / 256, 16384L)), dim3(256), 0, (hipStream_t)stream, (const h16*)x,
                     (const h16*)dy, (h16*)dx, n8);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" long mt_pool_attn_workspace_floats(int B, int N, int heads, int nq) {
  return (long)B * heads * nq * cdiv(N, POOL_KEYS) * (2 + 128);
}

extern "C" int mt_pool_attn_fwd(const float* q, const mt_half* kv, int B, int N, int E, int heads, int nq, float* out, float* scores,
                                float* lse, float* workspace, mt_stream_t stream) {
  if (!q || !kv || !out || !scores || !lse || !workspace || B < 1 || N < 1 || heads < 1 || nq < 1 || E % heads) return MT_ERR_BAD_ARG;
  const int hd = E / heads;
  if (hd > 128 || (hd & 7)) return MT_ERR_UNSUPPORTED;
  const int nsplit = cdiv(N, POOL_KEYS);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pool_attn_part_kernel, dim3(B * heads * nq * nsplit), dim3(256), 0, s, q, (const h16*)kv, B, N, E, heads, nq, nsplit,
                     1.0f / sqrtf((float)hd), scores, workspace);
  hipLaunchKernelGGL(pool_attn_combine_kernel, dim3(B * heads * nq), dim3(128), 0, s, (const float*)workspace, E, heads, nq, nsplit, out, lse);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_pool_attn_bwd(const float* q, const mt_half* kv, const float* scores, const float* lse, const float* out,
                                const float* dout, int B, int N, int E, int heads, int nq, mt_half* dkv, mt_stream_t stream) {
  if (!q || !kv || !scores || !lse || !out || !dout || !dkv || B < 1 || N < 1 || heads < 1 || nq < 1 || E % heads) return MT_ERR_BAD_ARG;
  const int hd = E / heads;
  if (hd > 128 || (hd & 7)) return MT_ERR_UNSUPPORTED;
  const int nsplit = cdiv(N, POOL_KEYS);
  hipLaunchKernelGGL(pool_attn_bwd_kernel, dim3(B * heads * nsplit), dim3(256), 0, (hipStream_t)stream, q, (const h16*)kv, scores, lse, out,
                     dout, B, N, E, heads, nq, nsplit, 1.0f / sqrtf((float)hd), (h16*)dkv);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
