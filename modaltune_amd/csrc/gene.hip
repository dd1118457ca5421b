// Grouped pathway networks of the gene encoder (gene_encoder.py:97-131,194-207): for each of G pathways
//   z_i = ELU(W2_i ELU(W1_i g_i + b1_i) + b2_i),  W1_i [latent, n_i], W2_i [latent, latent], g_i [n_i]
// With the real grouping (G = 331, n_i = 1..199, 24 M parameters) the reference runs 662 tiny nn.Linear modules; here
// ONE launch per direction walks all pathways: workgroup = pathway, thread = output unit, weights streamed through LDS
// in 32-column chunks with coalesced loads.  HBM-bound on the weights (forward reads them once; backward reads W2 once
// and read-modify-writes the gradient slots).  Parameters are addressed by element offsets into the flat fp32
// parameter / gradient buffers (the per-pathway tensors keep their reference state_dict names and order).
#include "common.h"

namespace {

constexpr int GL = 256;      // latent width = workgroup size
constexpr int GC = 32;       // chunk width

MT_DEVINL float elu(float v) { return v > 0.f ? v : expm1f(v); }
MT_DEVINL float elu_grad(float pre) { return pre > 0.f ? 1.f : __expf(pre); }

struct GeneArgs {
  const float* params; float* grads;
  const long* offs;      // [G][4]: w1, b1, w2, b2 (element offsets)
  const int* sizes;      // [G]
  const long* goff;      // [G] offset of g_i in `genes`
  const float* genes;
  float* a1; float* a2;  // pre-activations (saved): a1 [G][latent] (the same in every pass), a2 [G][P][latent]
  float* z;              // [G][P][latent] forward output (pathway-major: the mixer's group axis stays outermost)
  const float* dz;       // [G][P][latent]
  DropArgs adrop;        // nn.AlphaDropout after each ELU (gene_encoder.py:178-181): sites adrop.site, adrop.site + 1
  int G, P;              // P task passes: one forward call of the reference each, so each draws its own masks
};
constexpr int GP_MAX = 4;

// AlphaDropout(p): kept values pass, dropped ones become alpha' = -selu_scale * selu_alpha; then the affine (a, b) that
// restores zero mean / unit variance (torch.nn.functional.alpha_dropout)
constexpr float ALPHA_P = -1.7580993408473766f;
struct AlphaAff { float a, b; };
MT_DEVINL AlphaAff alpha_affine(float p) {
  const float a = rsqrtf((1.f - p) * (1.f + p * ALPHA_P * ALPHA_P));
  return AlphaAff{a, -a * ALPHA_P * p};
}

// y_p[j] = bias[j] + sum_k W[j][k] x_p[k]  (thread j) for P input vectors at once: W row-major [GL][n] is streamed through
// LDS ONCE for all passes, x_p in LDS
template <int P>
MT_DEVINL void gemv_rows(const float* __restrict__ W, int n, const float (*xs)[GL], float (*Ws)[GC + 1], float (&acc)[P]) {
  const int j = threadIdx.x;
  for (int k0 = 0; k0 < n; k0 += GC) {
    const int kc = min(GC, n - k0);
    __syncthreads();
    for (int i = j; i < GL * GC; i += GL) {       // 8 rows of 32 columns per pass: 128-B coalesced row segments
      const int r = i / GC, c = i - r * GC;
      Ws[r][c] = c < kc ? W[(long)r * n + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < GC; ++c) {
      const float w = lds_f32(&Ws[j][c]);      // (every LDS read of the loop in the 4-byte class: lds_f32, common.h)
#pragma unroll
      for (int p = 0; p < P; ++p) acc[p] = fmaf(w, (k0 + c < n) ? lds_f32(&xs[p][k0 + c]) : 0.f, acc[p]);
    }
  }
}

template <int P>
__global__ __launch_bounds__(GL) void gene_snn_fwd_kernel(GeneArgs a) {
  __shared__ float Ws[GL][GC + 1];
  __shared__ float xs[P][GL];
  const int i = blockIdx.x, j = threadIdx.x;
  const int n = a.sizes[i];
  const long* o = a.offs + 4L * i;
  const float* g = a.genes + a.goff[i];
  float acc = a.params[o[1] + j];
  // first layer (the same in every pass: the input is data): n can exceed the LDS vector; walk it in pieces of GL
  for (int p0 = 0; p0 < n; p0 += GL) {
    const int pn = min(GL, n - p0);
    __syncthreads();
    if (j < pn) xs[0][j] = g[p0 + j];
    __syncthreads();
    // rows of W1 restricted to columns [p0, p0 + pn): row stride n
    for (int k0 = 0; k0 < pn; k0 += GC) {
      const int kc = min(GC, pn - k0);
      __syncthreads();
      for (int t = j; t < GL * GC; t += GL) {
        const int r = t / GC, c = t - r * GC;
        Ws[r][c] = c < kc ? a.params[o[0] + (long)r * n + p0 + k0 + c] : 0.f;
      }
      __syncthreads();
#pragma unroll 8
      for (int c = 0; c < GC; ++c) acc = fmaf(lds_f32(&Ws[j][c]), (k0 + c < pn) ? lds_f32(&xs[0][k0 + c]) : 0.f, acc);
    }
  }
  a.a1[(long)i * GL + j] = acc;
  __syncthreads();
  const bool ad = a.adrop.active() && a.adrop.p > 0.f;
  const AlphaAff af = alpha_affine(a.adrop.p);
  const float e1 = elu(acc);
  float acc2[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {         // pass p draws its own masks: element index (i P + p) 256 + j
    float h1 = e1;
    if (ad) h1 = fmaf(af.a, drop_keep1(a.adrop, a.adrop.site, ((uint64_t)i * P + p) * GL + j) ? h1 : ALPHA_P, af.b);
    xs[p][j] = h1;
    acc2[p] = a.params[o[3] + j];
  }
  gemv_rows<P>(a.params + o[2], GL, xs, Ws, acc2);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const long e = ((long)i * P + p) * GL + j;
    a.a2[e] = acc2[p];
    float zz = elu(acc2[p]);
    if (ad) zz = fmaf(af.a, drop_keep1(a.adrop, a.adrop.site + 1, (uint64_t)e) ? zz : ALPHA_P, af.b);
    a.z[e] = zz;
  }
}

template <int P>
__global__ __launch_bounds__(GL) void gene_snn_bwd_kernel(GeneArgs a) {
  __shared__ float Ws[GC][GL + 1];   // 32 rows x 256 columns of W2
  __shared__ float da2s[P][GL], h1s[P][GL], da1s[GL];
  const int i = blockIdx.x, j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const int n = a.sizes[i];
  const long* o = a.offs + 4L * i;
  const float pre1 = a.a1[(long)i * GL + j];
  const bool ad = a.adrop.active() && a.adrop.p > 0.f;
  const AlphaAff af = alpha_affine(a.adrop.p);
  float g1[P], db2 = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const long e = ((long)i * P + p) * GL + j;
    const float pre2 = a.a2[e];
    const bool keep1 = !ad || drop_keep1(a.adrop, a.adrop.site, (uint64_t)e);
    const bool keep2 = !ad || drop_keep1(a.adrop, a.adrop.site + 1, (uint64_t)e);
    const float g2 = ad ? (keep2 ? af.a : 0.f) : 1.f;
    g1[p] = ad ? (keep1 ? af.a : 0.f) : 1.f;
    const float da2 = a.dz[e] * g2 * elu_grad(pre2);
    da2s[p][j] = da2;
    h1s[p][j] = ad ? fmaf(af.a, keep1 ? elu(pre1) : ALPHA_P, af.b) : elu(pre1);
    db2 += da2;
  }
  a.grads[o[3] + j] += db2;
  __syncthreads();
  // dW2[r][c] += sum_p da2_p[r] h1_p[c]: one wave per row, 16-byte accesses
  for (int r = wave; r < GL; r += 4) {
    float* dst = a.grads + o[2] + (long)r * GL + lane * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float d = lds_f32(&da2s[p][r]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(d, lds_f32(&h1s[p][lane * 4 + e]), v[e]);
    }
    *reinterpret_cast<f32x4*>(dst) = v;
  }
  // dh1_p[k] = sum_r W2[r][k] da2_p[r] (thread k): W2 streamed ONCE, 32 rows at a time, coalesced
  float dh1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) dh1[p] = 0.f;
  for (int r0 = 0; r0 < GL; r0 += GC) {
    __syncthreads();
    for (int t = j; t < GC * GL; t += GL) {
      const int r = t / GL, c = t - r * GL;
      Ws[r][c] = a.params[o[2] + (long)(r0 + r) * GL + c];
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < GC; ++r) {
      const float w = lds_f32(&Ws[r][j]);
#pragma unroll
      for (int p = 0; p < P; ++p) dh1[p] = fmaf(w, lds_f32(&da2s[p][r0 + r]), dh1[p]);
    }
  }
  // the first layer's input is the same in every pass: da1 = sum_p dh1_p * mask_p * elu'(pre1)
  float da1 = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) da1 += dh1[p] * g1[p];
  da1 *= elu_grad(pre1);
  a.grads[o[1] + j] += da1;
  da1s[j] = da1;
  __syncthreads();
  // dW1[r][k] += da1[r] g[k]: rows of n floats (4-byte accesses: the slots are only 4-byte aligned inside a row)
  const float* g = a.genes + a.goff[i];
  for (long t = j; t < (long)GL * n; t += GL) {
    const int r = (int)(t / n), k = (int)(t - (long)r * n);
    a.grads[o[0] + t] += lds_f32(&da1s[r]) * g[k];
  }
}

// ---- few pathways (the 6-pathway configuration of the headline bench): a workgroup per pathway leaves the chip to 6 workgroups that
// each pull a 256 KB W2 through one CU (fwd 37 us, bwd 83 us inside the step).  Here GS workgroups share a pathway by INDEX RANGES that
// need no reduction across workgroups: forward = GR = 256 / GS rows of the second layer each (the small first layer is recomputed by
// all of them); backward = rows [GR s, GR s + GR) of dW2 / db2 and COLUMNS [GR s, ...) of W2^T da2, hence of da1, db1 and the rows of dW1.
constexpr int GS = 8, GR = GL / GS;      // 8 workgroups x 32 rows / columns

template <int P>
__global__ __launch_bounds__(GL) void gene_snn_fwd_split_kernel(GeneArgs a) {
  __shared__ float Ws[GL][GC + 1];
  __shared__ float xs[P][GL];
  const int i = blockIdx.x / GS, sp = blockIdx.x % GS, j = threadIdx.x;
  const int n = a.sizes[i];
  const long* o = a.offs + 4L * i;
  const float* g = a.genes + a.goff[i];
  float acc = a.params[o[1] + j];
  for (int p0 = 0; p0 < n; p0 += GL) {        // first layer, as gene_snn_fwd_kernel
    const int pn = min(GL, n - p0);
    __syncthreads();
    if (j < pn) xs[0][j] = g[p0 + j];
    __syncthreads();
    for (int k0 = 0; k0 < pn; k0 += GC) {
      const int kc = min(GC, pn - k0);
      __syncthreads();
      for (int t = j; t < GL * GC; t += GL) {
        const int r = t / GC, c = t - r * GC;
        Ws[r][c] = c < kc ? a.params[o[0] + (long)r * n + p0 + k0 + c] : 0.f;
      }
      __syncthreads();
#pragma unroll 8
      for (int c = 0; c < GC; ++c) acc = fmaf(lds_f32(&Ws[j][c]), (k0 + c < pn) ? lds_f32(&xs[0][k0 + c]) : 0.f, acc);
    }
  }
  if (sp == 0) a.a1[(long)i * GL + j] = acc;
  __syncthreads();
  const bool ad = a.adrop.active() && a.adrop.p > 0.f;
  const AlphaAff af = alpha_affine(a.adrop.p);
  const float e1 = elu(acc);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float h1 = e1;
    if (ad) h1 = fmaf(af.a, drop_keep1(a.adrop, a.adrop.site, ((uint64_t)i * P + p) * GL + j) ? h1 : ALPHA_P, af.b);
    xs[p][j] = h1;
  }
  __syncthreads();
  // second layer, rows [GR sp, GR sp + GR): the rows are staged in LDS by all threads (8 per row: a wave reads 8 rows x 1 KB,
  // contiguous), then ONE thread per row runs the k = 0 .. 255 chain in the order of gene_snn_fwd_kernel (bias first, k ascending):
  // the two kernels give the same bits, so a model does not change with the number of pathways it is grouped into
  float (*Wl)[GL + 1] = reinterpret_cast<float (*)[GL + 1]>(&Ws[0][0]);      // [GR][GL + 1] in the chunk buffer (8 224 of its 8 448 floats)
  {
    const int r = j >> 3, part = j & 7;
    const float* wr = a.params + o[2] + (long)(GR * sp + r) * GL + part * 32;
#pragma unroll
    for (int c = 0; c < 32; c += 4) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) Wl[r][part * 32 + c + e] = w4[e];
    }
  }
  __syncthreads();
  if (j < GR) {
    const int r = GR * sp + j;
    float a2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) a2[p] = a.params[o[3] + r];
#pragma unroll 8
    for (int k = 0; k < GL; ++k) {
      const float w = lds_f32(&Wl[j][k]);
#pragma unroll
      for (int p = 0; p < P; ++p) a2[p] = fmaf(w, lds_f32(&xs[p][k]), a2[p]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const long e = ((long)i * P + p) * GL + r;
      a.a2[e] = a2[p];
      float zz = elu(a2[p]);
      if (ad) zz = fmaf(af.a, drop_keep1(a.adrop, a.adrop.site + 1, (uint64_t)e) ? zz : ALPHA_P, af.b);
      a.z[e] = zz;
    }
  }
}

template <int P>
__global__ __launch_bounds__(GL) void gene_snn_bwd_split_kernel(GeneArgs a) {
  __shared__ float da2s[P][GL], h1s[P][GL], Wc[GL][GR + 1], da1s[GR];
  const int i = blockIdx.x / GS, sp = blockIdx.x % GS, j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const int n = a.sizes[i];
  const long* o = a.offs + 4L * i;
  const float pre1 = a.a1[(long)i * GL + j];
  const bool ad = a.adrop.active() && a.adrop.p > 0.f;
  const AlphaAff af = alpha_affine(a.adrop.p);
  float db2 = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const long e = ((long)i * P + p) * GL + j;
    const float pre2 = a.a2[e];
    const bool keep1 = !ad || drop_keep1(a.adrop, a.adrop.site, (uint64_t)e);
    const bool keep2 = !ad || drop_keep1(a.adrop, a.adrop.site + 1, (uint64_t)e);
    const float g2 = ad ? (keep2 ? af.a : 0.f) : 1.f;
    const float da2 = a.dz[e] * g2 * elu_grad(pre2);
    da2s[p][j] = da2;
    h1s[p][j] = ad ? fmaf(af.a, keep1 ? elu(pre1) : ALPHA_P, af.b) : elu(pre1);
    db2 += da2;
  }
  if (j / GR == sp) a.grads[o[3] + j] += db2;
  __syncthreads();
  // dW2 rows [GR sp, GR sp + GR): one wave per row, 16-byte accesses
  for (int r = GR * sp + wave; r < GR * sp + GR; r += 4) {
    float* dst = a.grads + o[2] + (long)r * GL + lane * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float d = lds_f32(&da2s[p][r]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(d, lds_f32(&h1s[p][lane * 4 + e]), v[e]);
    }
    *reinterpret_cast<f32x4*>(dst) = v;
  }
  // dh1_p[k] = sum_r W2[r][k] da2_p[r] for the columns k in [GR sp, GR sp + GR): the 256 x GR block of W2 is staged in LDS (each
  // row's 128 bytes by 8 threads), then one thread per column runs r = 0 .. 255 in the order of gene_snn_bwd_kernel (same bits)
  {
    const int part = j & 7;
    for (int r = j >> 3; r < GL; r += GL / 8) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.params + o[2] + (long)r * GL + GR * sp + part * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) Wc[r][part * 4 + e] = w4[e];
    }
  }
  __syncthreads();
  if (j < GR) {
    const int k = GR * sp + j;
    float dh1[P];
#pragma unroll
    for (int p = 0; p < P; ++p) dh1[p] = 0.f;
#pragma unroll 8
    for (int r = 0; r < GL; ++r) {
      const float w = lds_f32(&Wc[r][j]);
#pragma unroll
      for (int p = 0; p < P; ++p) dh1[p] = fmaf(w, lds_f32(&da2s[p][r]), dh1[p]);
    }
    const float prek = a.a1[(long)i * GL + k];
    float da1 = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const bool keep1 = !ad || drop_keep1(a.adrop, a.adrop.site, (uint64_t)(((long)i * P + p) * GL + k));
      da1 += dh1[p] * (ad ? (keep1 ? af.a : 0.f) : 1.f);
    }
    da1 *= elu_grad(prek);
    a.grads[o[1] + k] += da1;
    da1s[j] = da1;
  }
  __syncthreads();
  // dW1 rows [GR sp, GR sp + GR): dW1[r][c] += da1[r] g[c]
  const float* g = a.genes + a.goff[i];
  for (long t = j; t < (long)GR * n; t += GL) {
    const int r = (int)(t / n), c = (int)(t - (long)r * n);
    a.grads[o[0] + ((long)GR * sp + r) * n + c] += lds_f32(&da1s[r]) * g[c];
  }
}

// below this many pathways a pathway is shared by GS workgroups (a workgroup per pathway fills the chip from a few hundred on)
#ifndef MT_GENE_SPLIT_BELOW
#define MT_GENE_SPLIT_BELOW 64
#endif
constexpr int GENE_SPLIT_BELOW = MT_GENE_SPLIT_BELOW;

}  // namespace

extern "C" int mt_gene_snn_fwd(const float* params, const long* offs, const int* sizes, const long* goff, const float* genes,
                               int G, int latent, int passes, float* a1, float* a2, float* z, const MtDropout* alpha_drop,
                               mt_stream_t stream) {
  if (!params || !offs || !sizes || !goff || !genes || !a1 || !a2 || !z || G < 1 || passes < 1) return MT_ERR_BAD_ARG;
  if (latent != GL || passes > GP_MAX) return MT_ERR_UNSUPPORTED;
  GeneArgs a{params, nullptr, offs, sizes, goff, genes, a1, a2, z, nullptr, make_drop(alpha_drop), G, passes};
  hipStream_t s = (hipStream_t)stream;
  if (G < GENE_SPLIT_BELOW) {
    switch (passes) {
      case 1: hipLaunchKernelGGL(gene_snn_fwd_split_kernel<1>, dim3(G * GS), dim3(GL), 0, s, a); break;
      case 2: hipLaunchKernelGGL(gene_snn_fwd_split_kernel<2>, dim3(G * GS), dim3(GL), 0, s, a); break;
      case 3: hipLaunchKernelGGL(gene_snn_fwd_split_kernel<3>, dim3(G * GS), dim3(GL), 0, s, a); break;
      default: hipLaunchKernelGGL(gene_snn_fwd_split_kernel<4>, dim3(G * GS), dim3(GL), 0, s, a); break;
    }
    MT_CHECK_LAUNCH();
    return MT_OK;
  }
  switch (passes) {
    case 1: hipLaunchKernelGGL(gene_snn_fwd_kernel<1>, dim3(G), dim3(GL), 0, s, a); break;
    case 2: hipLaunchKernelGGL(gene_snn_fwd_kernel<2>, dim3(G), dim3(GL), 0, s, a); break;
    case 3: hipLaunchKernelGGL(gene_snn_fwd_kernel<3>, dim3(G), dim3(GL), 0, s, a); break;
    default: hipLaunchKernelGGL(gene_snn_fwd_kernel<4>, dim3(G), dim3(GL), 0, s, a); break;
  }
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_gene_snn_bwd(const float* params, float* grads, const long* offs, const int* sizes, const long* goff,
                               const float* genes, int G, int latent, int passes, const float* a1, const float* a2,
                               const float* dz, const MtDropout* alpha_drop, mt_stream_t stream) {
  if (!params || !grads || !offs || !sizes || !goff || !genes || !a1 || !a2 || !dz || G < 1 || passes < 1) return MT_ERR_BAD_ARG;
  if (latent != GL || passes > GP_MAX) return MT_ERR_UNSUPPORTED;
  GeneArgs a{params, grads, offs, sizes, goff, genes, const_cast<float*>(a1), const_cast<float*>(a2), nullptr, dz,
             make_drop(alpha_drop), G, passes};
  hipStream_t s = (hipStream_t)stream;
  if (G < GENE_SPLIT_BELOW) {
    switch (passes) {
      case 1: hipLaunchKernelGGL(gene_snn_bwd_split_kernel<1>, dim3(G * GS), dim3(GL), 0, s, a); break;
      case 2: hipLaunchKernelGGL(gene_snn_bwd_split_kernel<2>, dim3(G * GS), dim3(GL), 0, s, a); break;
      case 3: hipLaunchKernelGGL(gene_snn_bwd_split_kernel<3>, dim3(G * GS), dim3(GL), 0, s, a); break;
      default: hipLaunchKernelGGL(gene_snn_bwd_split_kernel<4>, dim3(G * GS), dim3(GL), 0, s, a); break;
    }
    MT_CHECK_LAUNCH();
    return MT_OK;
  }
  switch (passes) {
    case 1: hipLaunchKernelGGL(gene_snn_bwd_kernel<1>, dim3(G), dim3(GL), 0, s, a); break;
    case 2: hipLaunchKernelGGL(gene_snn_bwd_kernel<2>, dim3(G), dim3(GL), 0, s, a); break;
    case 3: hipLaunchKernelGGL(gene_snn_bwd_kernel<3>, dim3(G), dim3(GL), 0, s, a); break;
    default: hipLaunchKernelGGL(gene_snn_bwd_kernel<4>, dim3(G), dim3(GL), 0, s, a); break;
  }
  MT_CHECK_LAUNCH();
  return MT_OK;
}
