// Dense multi-head flash attention with a 2-D ALiBi bias: the attention of the TITAN slide encoder's pre-norm ViT blocks
// (reference call sites: models/aggregators/titan_adapter.py:253-293 `get_alibi`, :359-361,394
// `blocks.modules_list[i](x, attn_bias, bg_mask)`; models/vitadapter/adapter_modules.py:526-558).  H heads of 64, one
// sequence of N tokens per task pass (cls + the slide's foreground cells), no dilation, no padding keys.
//
// Layout: q | k | v stay TOKEN-MAJOR, exactly as the qkv GEMM writes them ([B*N, 3*H*64] fp16): with a head dimension of 64 a
// head's row is one aligned 128-byte line, so a 64-key tile is 64 full lines whatever the row stride -- no head-major epilogue,
// no combine pass; o, dO and dq | dk | dv are token-major too and feed / come from plain GEMMs.
//
// ALiBi: bias[h, i, j] = -slope_h * |cell_i - cell_j| (euclidean, in grid cells; 0 to and from cls).  The reference's
// [H, N, N] fp32 table would cost more HBM traffic than K and V together (N = 4097: 805 MB per layer and pass).  The distance
// itself is head-independent, so ONE fp16 [N, N] table per slide (mt_alibi_dist: 34 MB at N = 4097, shared by every head, pass,
// layer and kernel of the step) is kept in the order the accumulator registers want it: a wave reads the 64 distances each of
// its lanes needs for a key tile as four coalesced 16-byte loads straight into registers (no LDS), and
// `s = fma(dist, -slope_h log2 e, -m)` is written INTO the accumulators the Q.K^T chain then starts from -- the bias costs
// one v_fma_mix_f32 per score and nothing else.  (Rounds 3 / early 4 produced the squared distance on the matrix pipe from two
// [N, 8] side tables and took v_sqrt_f32 per score: a transcendental costs two plain VALU slots on gfx950 --
// tools/experiments/valu_rate_probe -- and these kernels are VALU-bound at d = 64.)
//
// Kernel structure = the dilated kernels of attn.hip: swapped products with the query (forward, dQ) or the key (dK / dV) in the
// lane, 64-row LDS-DMA tile images with XOR-swizzled 16-byte chunks (attn_common.h: img_off -- all eight chunks are data here),
// double-buffered, one barrier per tile; q pre-scaled by 64^-1/2 log2 e inside the frozen qkv weight cache; per-query constants
// as initial accumulators; deferred exact rescale; P from the accumulators as the next operand.  The row sum is accumulated on
// the VALU in packed pairs (there is no spare column for a ones trick at d = 64).
#include "attn_common.h"

namespace {

constexpr int DH = 64;

struct DenseArgs {
  const h16* qkv; long ld;            // [B*N, ld]: q at column 0, k at D, v at 2 D (D = H * 64); ld = 3 D
  const h16* dist;                    // blocked distance table (mt_alibi_dist), or nullptr: no bias
  const float* nslope;                // [H]: -slope_h * log2(e)
  int N, B, H, D, qtiles;
};

struct DItem { int b, h, qt; bool live; };
// Workgroup -> (pass, head, 128-row tile).  Blocks b and b + 8 share an XCD.  The B * H (pass, head) groups do not divide by the eight
// XCDs (36 = 4 x 8 + 4): with whole groups per XCD four XCDs carry five groups and four carry four -- at N = 4 097 (33 tiles) that is
// 165 against 132 workgroups on 96 slots, the launch ends when the loaded half is done, and the 33rd tile pushes its CUs from five
// workgroups to six (measured: 4 096 -> 4 097 tokens costs the forward 13 %, tools/experiments/README.md).  So: XCD x owns the whole groups
// x, x + 8, ... (every tile of a group re-reads the same K / V rows from ONE L2; 4-5 groups x 1 MB at N = 4 097 stay resident) and an
// EQUAL share of the tiles of the B * H mod 8 left-over groups (a contiguous run of the (group, tile) list: a left-over group's
// K / V are read through two L2s at most).  An XCD walks its lanes -- whole groups, plus the left-over run -- with their tiles
// interleaved (tile 0 of each, then tile 1 of each, ...): the workgroups of one tile index that run side by side share that tile's
// stripe of the distance table (same-box: -1..-3 % on all three kernels against group after group; two groups at a time: +10 %).
MT_DEVINL DItem ddecode(const DenseArgs& a, int bid) {
  const int x = bid & 7, j = bid >> 3;
  const int G = a.B * a.H, full = G >> 3, rem = G & 7;
  const int nl = full + (rem ? 1 : 0);      // lanes per XCD
  const int lane = __builtin_amdgcn_readfirstlane(j % nl), pos = __builtin_amdgcn_readfirstlane(j / nl);
  int gid, qt;
  bool live = true;
  if (lane < full) {
    gid = lane * 8 + x;
    qt = pos;
  } else {
    const int items = rem * a.qtiles, share = (items + 7) >> 3;
    const int e = x * share + pos;
    live = pos < share && e < items;
    gid = full * 8 + e / a.qtiles;
    qt = e % a.qtiles;
  }
  DItem w;
  w.live = live;
  w.qt = __builtin_amdgcn_readfirstlane(qt);
  w.h = __builtin_amdgcn_readfirstlane(live ? gid % a.H : 0);
  w.b = __builtin_amdgcn_readfirstlane(live ? gid / a.H : 0);
  return w;
}

struct DmaLane64 {      // this thread's two 16-byte pieces of a [64][128 B] tile image: piece p = 2 * wave + i covers rows 8p..8p+7
  uint32_t voff[2]; int lds_halves[2];
  MT_DEVINL DmaLane64(int tid, int row_stride_bytes) {
    const int wave = tid >> 6, j = tid & 63;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = 2 * wave + i, row = 8 * piece + (j >> 3);
      const int c = (j & 7) ^ img_f(row);
      voff[i] = (uint32_t)(row * row_stride_bytes + c * 16);
      lds_halves[i] = piece * 512;
    }
  }
};
MT_DEVINL void dma_tile64(h16* img, __amdgpu_buffer_rsrc_t rs, const DmaLane64& d) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(img + d.lds_halves[i]), 16, d.voff[i], 0, 0, 0);
}
MT_DEVINL f32x16 splat16(float v) {
  f32x16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = v;
  return r;
}
MT_DEVINL float sum_halves(float x) { return x + __shfl_xor(x, 32, 64); }

// Distance table: block (A, t) = the distances between the 32 lane-side tokens 32 A .. 32 A + 31 and the 64 tile-side tokens of
// tile t, stored as the four 16-byte pieces each lane of a wave consumes: piece j = 2 sub + half holds accumulator registers
// 8 half .. 8 half + 7 of sub-block sub, i.e. the tile-side tokens sub * 32 + (i & 3) + 8 (i >> 2) + 4 hh of lane (l31, hh).
// The matrix is symmetric, so the same table serves the kernels with the query in the lane (forward, dQ) and the one with the
// key in the lane (dK / dV).
constexpr int DBLK = 2048;            // halves per (A, t) block
struct DistRegs { h16x8 p[4]; };
MT_DEVINL void dist_load(DistRegs& r, const h16* stripe, int t) {
#pragma unroll
  for (int j = 0; j < 4; ++j) r.p[j] = ldg8(stripe + (long)t * DBLK + j * 512);
}
// nslope * (fp16 half of h2) + c in ONE instruction (hipcc turns the C expression into two conversions and half a v_pk_fma_f32)
MT_DEVINL float fma_mix_lo(uint32_t h2, float nslope, float c) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "s"(nslope), "v"(c));
  return r;
}
MT_DEVINL float fma_mix_hi(uint32_t h2, float nslope, float c) {
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "s"(nslope), "v"(c));
  return r;
}
MT_DEVINL float dist_bias(const DistRegs& r, int sub, int i, float nslope, float c) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 w = __builtin_bit_cast(u32x4, r.p[2 * sub + (i >> 3)]);
  return (i & 1) ? fma_mix_hi(w[(i & 7) >> 1], nslope, c) : fma_mix_lo(w[(i & 7) >> 1], nslope, c);
}
// accumulator initialiser of sub-block sub: nslope * dist + c
MT_DEVINL f32x16 dist_init(const DistRegs& r, int sub, float nslope, float c) {
  f32x16 s;
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = dist_bias(r, sub, i, nslope, c);
  return s;
}

// ------------------------------------------------------------------------------------------------ forward
template <bool BIAS>
__global__ __launch_bounds__(256, 3) void dense_attn_fwd_kernel(DenseArgs a, h16* __restrict__ o, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) h16 smem[4 * IMG_HALVES];      // K0 | K1 | V0 | V1
  h16* const Ks = smem;
  h16* const Vs = smem + 2 * IMG_HALVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const DItem w = ddecode(a, blockIdx.x);
  if (!w.live) return;
  const int N = a.N;
  const long ld = a.ld;

  const int iq = w.qt * 128 + wave * 32 + l31;
  const bool qvalid = iq < N;
  const long qrow = (long)w.b * N + min(iq, N - 1);
  h16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = sel8(qvalid, ldg8(a.qkv + qrow * ld + w.h * DH + ks * 16 + hh * 8));
  const float nslope = BIAS ? a.nslope[w.h] : 0.f;

  const int ntile = (N + 63) >> 6;
  const h16* const dstripe = BIAS ? a.dist + (long)(w.qt * 4 + wave) * ntile * DBLK + lane * 8 : nullptr;
  DistRegs dr;
  const int row_bytes = (int)ld * 2;
  const long valid_bytes = (long)(N - 1) * row_bytes + DH * 2;
  const long tile_bytes = 64L * row_bytes;
  const h16* const kseq = a.qkv + (long)w.b * N * ld + a.D + w.h * DH;
  const h16* const vseq = kseq + a.D;
  const DmaLane64 dl(tid, row_bytes);
  auto dma = [&](int t) {
    dma_tile64(Ks + (t & 1) * IMG_HALVES, tile_rsrc(kseq, t * tile_bytes, valid_bytes), dl);
    dma_tile64(Vs + (t & 1) * IMG_HALVES, tile_rsrc(vseq, t * tile_bytes, valid_bytes), dl);
  };
  const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  int krd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) krd[ks] = img_off(l31, 2 * ks + hh);
  const int vc = 2 * (grp & 1) + (tp >> 1), vo = 4 * (tp & 1);
  const int va0 = img_off(4 * hh + tq, vc) + vo, va1 = img_off(4 * hh + tq, vc + 4) + vo;
  const int vb0 = img_off(4 * hh + tq + 8, vc) + vo, vb1 = img_off(4 * hh + tq + 8, vc + 4) + vo;

  // logits of one 64-key tile in log2 units relative to the running reference: s[sub][reg] = q'.k + nslope * dist - m2; the
  // chains start from `init` (BIAS: the bias - m2 written by dist_init; otherwise the splat of -m2)
  auto qk = [&](int t, f32x16 (&s)[2], const f32x16& init0, const f32x16& init1) {
    const h16* Kb = Ks + (t & 1) * IMG_HALVES;
    h16x8 kf[2][4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[sub][ks] = *reinterpret_cast<const h16x8*>(&Kb[sub * 32 * IMG_ROW + krd[ks]]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][ks], qf[ks], ks == 0 ? init0 : s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1][ks], qf[ks], ks == 0 ? init1 : s[1], 0, 0, 0);
    }
    // all eight fragment reads in flight before the first product (hipcc otherwise serialises read -> wait -> MFMA through one
    // fragment register: eight exposed LDS latencies per tile)
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  };

  f32x16 o0 = splat16(0.f), o1 = splat16(0.f);
  f32x2 lsum2 = {0.f, 0.f};
  dma(0);
  if (BIAS) dist_load(dr, dstripe, 0);
  dma_wait_all();
  __syncthreads();
  // running reference m2 (log2 units); starts at the row maximum of tile 0 (bias included: the ALiBi term can push a whole tile
  // far below its raw scores)
  float m2;
  f32x16 minit = splat16(0.f);      // (!BIAS: the splat of -m2, the C operand of both chains)
  {
    f32x16 s0[2];
    if (BIAS) qk(0, s0, dist_init(dr, 0, nslope, 0.f), dist_init(dr, 1, nslope, 0.f));
    else qk(0, s0, minit, minit);
    float mx = NEG_BIG;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kidx = sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        mx = fmaxf(mx, kidx < N ? s0[sub][i] : NEG_BIG);
      }
    m2 = max_halves(mx);
    if (!BIAS) minit = splat16(-m2);
  }

  auto tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int kb = t * 64;
    const h16* Vb = Vs + (t & 1) * IMG_HALVES;
    if (t + 1 < ntile) dma(t + 1);
    f32x16 s_cur[2];
    if (BIAS) {
      const f32x16 i0 = dist_init(dr, 0, nslope, -m2), i1 = dist_init(dr, 1, nslope, -m2);
      if (t + 1 < ntile) dist_load(dr, dstripe, t + 1);      // (same registers: the two lines above were their last use)
      qk(t, s_cur, i0, i1);
    } else {
      qk(t, s_cur, minit, minit);
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (TAIL) {
          const int kidx = kb + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (kidx >= N) s_cur[sub][i] = NEG_BIG;
        }
        mx = fmaxf(mx, s_cur[sub][i]);
      }
    mx = max_halves(mx);
    if (__any(mx > RESCALE_LOG2)) {      // deferred exact rescale (attn.hip)
      const float up = fmaxf(mx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-up);
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; s_cur[0][i] -= up; s_cur[1][i] -= up; }
      lsum2 *= alpha;
      m2 += up;
      if (!BIAS) minit = splat16(-m2);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        h16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 p = pk_exp2((f32x2){s_cur[sub][8 * s2 + e], s_cur[sub][8 * s2 + e + 1]});
          pf[e] = (h16)p[0]; pf[e + 1] = (h16)p[1];
          lsum2 += p;
        }
        const h16* vblk = Vb + (sub * 32 + s2 * 16) * IMG_ROW;
        const h16x8 v0 = cat8(lds_tr4(vblk + va0), lds_tr4(vblk + vb0));
        const h16x8 v1 = cat8(lds_tr4(vblk + va1), lds_tr4(vblk + vb1));
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf, o1, 0, 0, 0);
      }
    dma_wait_all();
    __syncthreads();
  };
  const bool tail_last = (N & 63) != 0;
  const int nplain = tail_last ? ntile - 1 : ntile;
  for (int t = 0; t < nplain; ++t) tile(t, std::false_type{});
  if (tail_last) tile(ntile - 1, std::true_type{});

  const float l = sum_halves(lsum2[0] + lsum2[1]);
  if (qvalid) {
    const float inv = 1.0f / l;
    h16* orow = o + qrow * a.D + w.h * DH;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const h16x4 v0 = {(h16)(o0[4 * gq] * inv), (h16)(o0[4 * gq + 1] * inv), (h16)(o0[4 * gq + 2] * inv), (h16)(o0[4 * gq + 3] * inv)};
      const h16x4 v1 = {(h16)(o1[4 * gq] * inv), (h16)(o1[4 * gq + 1] * inv), (h16)(o1[4 * gq + 2] * inv), (h16)(o1[4 * gq + 3] * inv)};
      *reinterpret_cast<h16x4*>(orow + 8 * gq + 4 * hh) = v0;
      *reinterpret_cast<h16x4*>(orow + 32 + 8 * gq + 4 * hh) = v1;
    }
    if (hh == 0) lse[qrow * a.H + w.h] = (m2 + __log2f(l)) * LN2;
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (query = lane)
//   P'^T = exp2(S'^T + bias - L2[q] + log2 ln2) ; dP^T = V . dO^T - delta[q] ; dS^T = P'^T dP^T ; dQ'^T += K^T . dS^T
template <bool BIAS>
__global__ __launch_bounds__(256, 3) void dense_attn_bwd_q_kernel(DenseArgs a, const h16* __restrict__ d_o, const float* __restrict__ lse,
                                                               const float* __restrict__ delta, h16* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) h16 smem[4 * IMG_HALVES];      // K0 | K1 | V0 | V1
  h16* const Ks = smem;
  h16* const Vs = smem + 2 * IMG_HALVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const DItem w = ddecode(a, blockIdx.x);
  if (!w.live) return;
  const int N = a.N;
  const long ld = a.ld;

  const int iq = w.qt * 128 + wave * 32 + l31;
  const bool qvalid = iq < N;
  const long qrow = (long)w.b * N + min(iq, N - 1);
  h16x8 qf[4], dof[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = sel8(qvalid, ldg8(a.qkv + qrow * ld + w.h * DH + ks * 16 + hh * 8));
    dof[ks] = sel8(qvalid, ldg8(d_o + qrow * a.D + w.h * DH + ks * 16 + hh * 8));
  }
  const float nslope = BIAS ? a.nslope[w.h] : 0.f;
  const float L2raw = lse[qrow * a.H + w.h], dlraw = delta[qrow * a.H + w.h];
  const float nl2 = qvalid ? fmaf(-L2raw, LOG2E, LOG2_LN2) : -1.0e30f;      // invalid queries: P' = 0
  const f32x16 nl2i = splat16(nl2);
  const f32x16 ndli = splat16(qvalid ? -dlraw : 0.f);

  const int ntile = (N + 63) >> 6;
  const h16* const dstripe = BIAS ? a.dist + (long)(w.qt * 4 + wave) * ntile * DBLK + lane * 8 : nullptr;
  DistRegs dr;
  const int row_bytes = (int)ld * 2;
  const long valid_bytes = (long)(N - 1) * row_bytes + DH * 2;
  const long tile_bytes = 64L * row_bytes;
  const h16* const kseq = a.qkv + (long)w.b * N * ld + a.D + w.h * DH;
  const h16* const vseq = kseq + a.D;
  const DmaLane64 dl(tid, row_bytes);
  auto dma = [&](int t) {
    dma_tile64(Ks + (t & 1) * IMG_HALVES, tile_rsrc(kseq, t * tile_bytes, valid_bytes), dl);
    dma_tile64(Vs + (t & 1) * IMG_HALVES, tile_rsrc(vseq, t * tile_bytes, valid_bytes), dl);
  };
  const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  int rrd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rrd[ks] = img_off(l31, 2 * ks + hh);
  const int kc = 2 * (grp & 1) + (tp >> 1), ko = 4 * (tp & 1);
  const int ka0 = img_off(4 * hh + tq, kc) + ko, ka1 = img_off(4 * hh + tq, kc + 4) + ko;
  const int kb0 = img_off(4 * hh + tq + 8, kc) + ko, kb1 = img_off(4 * hh + tq + 8, kc + 4) + ko;

  f32x16 dq0 = splat16(0.f), dq1 = splat16(0.f);
  dma(0);
  if (BIAS) dist_load(dr, dstripe, 0);
  dma_wait_all();
  __syncthreads();
  auto tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int kb = t * 64;
    const h16* Kb = Ks + (t & 1) * IMG_HALVES;
    const h16* Vb = Vs + (t & 1) * IMG_HALVES;
    if (t + 1 < ntile) dma(t + 1);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s, dp;
      if (BIAS) {
        s = dist_init(dr, sub, nslope, nl2);
        if (sub == 1 && t + 1 < ntile) dist_load(dr, dstripe, t + 1);      // (same registers: last use just above)
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const h16x8 kf = *reinterpret_cast<const h16x8*>(&Kb[sub * 32 * IMG_ROW + rrd[ks]]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 && !BIAS ? nl2i : s, 0, 0, 0);
        const h16x8 vf = *reinterpret_cast<const h16x8*>(&Vb[sub * 32 * IMG_ROW + rrd[ks]]);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, dof[ks], ks == 0 ? ndli : dp, 0, 0, 0);
      }
      h16x8 dsf[2];
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f32x2 pt = pk_exp2((f32x2){s[i], s[i + 1]});
        if (TAIL) {
          const int kidx = kb + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (kidx >= N) pt[0] = 0.f;
          if (kidx + 1 >= N) pt[1] = 0.f;
        }
        const f32x2 d = pt * (f32x2){dp[i], dp[i + 1]};
        dsf[i >> 3][i & 7] = (h16)d[0];
        dsf[i >> 3][(i & 7) + 1] = (h16)d[1];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const h16* kblk = Kb + (sub * 32 + s2 * 16) * IMG_ROW;
        const h16x8 k0 = cat8(lds_tr4(kblk + ka0), lds_tr4(kblk + kb0));
        const h16x8 k1 = cat8(lds_tr4(kblk + ka1), lds_tr4(kblk + kb1));
        dq0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, dsf[s2], dq0, 0, 0, 0);
        dq1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, dsf[s2], dq1, 0, 0, 0);
      }
    }
    dma_wait_all();
    __syncthreads();
  };
  const bool tail_last = (N & 63) != 0;
  const int nplain = tail_last ? ntile - 1 : ntile;
  for (int t = 0; t < nplain; ++t) tile(t, std::false_type{});
  if (tail_last) tile(ntile - 1, std::true_type{});
  if (qvalid) {
    h16* out = dqkv + qrow * ld + w.h * DH;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const h16x4 v0 = {(h16)dq0[4 * gq], (h16)dq0[4 * gq + 1], (h16)dq0[4 * gq + 2], (h16)dq0[4 * gq + 3]};
      const h16x4 v1 = {(h16)dq1[4 * gq], (h16)dq1[4 * gq + 1], (h16)dq1[4 * gq + 2], (h16)dq1[4 * gq + 3]};
      *reinterpret_cast<h16x4*>(out + 8 * gq + 4 * hh) = v0;
      *reinterpret_cast<h16x4*>(out + 32 + 8 * gq + 4 * hh) = v1;
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV (key = lane)
//   S[q,key] = Q' . K^T + bias ; dP = dO . V^T ; P' = exp2(S - L2[q] + log2 ln2) ; dS = P' (dP - delta[q])
//   dV^T[d,key] += dO^T . P' (x 1 / ln2 at the end) ; dK^T[d,key] += Q'^T . dS
// (3 waves per SIMD without the bias: 168 registers, 8 spilled dwords outside the tile loop, -3.6 %; with the 16 distance registers
// the same bound spills inside the loop: 0.50 -> 0.63 ms)
template <bool BIAS>
__global__ __launch_bounds__(256, BIAS ? 2 : 3) void dense_attn_bwd_kv_kernel(DenseArgs a, const h16* __restrict__ d_o, const float* __restrict__ lse,
                                                                const float* __restrict__ delta, h16* __restrict__ dqkv) {
  __shared__ __attribute__((aligned(16))) h16 smem[4 * IMG_HALVES];      // Q0 | Q1 | D0 | D1
  __shared__ __attribute__((aligned(16))) float L2s[2][64];
  __shared__ __attribute__((aligned(16))) float Dls[2][64];
  h16* const Qx = smem;
  h16* const Dx = smem + 2 * IMG_HALVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const DItem w = ddecode(a, blockIdx.x);
  if (!w.live) return;
  const int N = a.N;
  const long ld = a.ld;

  const int ik = w.qt * 128 + wave * 32 + l31;
  const bool kvalid = ik < N;
  const long krow = (long)w.b * N + min(ik, N - 1);
  h16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kf[ks] = sel8(kvalid, ldg8(a.qkv + krow * ld + a.D + w.h * DH + ks * 16 + hh * 8));
    vf[ks] = sel8(kvalid, ldg8(a.qkv + krow * ld + 2 * a.D + w.h * DH + ks * 16 + hh * 8));
  }
  const float nslope = BIAS ? a.nslope[w.h] : 0.f;

  const int ntile = (N + 63) >> 6;
  const h16* const dstripe = BIAS ? a.dist + (long)(w.qt * 4 + wave) * ntile * DBLK + lane * 8 : nullptr;
  DistRegs dr;
  const int qrow_bytes = (int)ld * 2, drow_bytes = a.D * 2;
  const long qvalid_bytes = (long)(N - 1) * qrow_bytes + DH * 2, dvalid_bytes = (long)(N - 1) * drow_bytes + DH * 2;
  const h16* const qseq = a.qkv + (long)w.b * N * ld + w.h * DH;
  const h16* const dseq = d_o + (long)w.b * N * a.D + w.h * DH;
  const DmaLane64 dlq(tid, qrow_bytes), dld(tid, drow_bytes);
  const float* const lbase = lse + (long)w.b * N * a.H + w.h;
  const float* const dbase = delta + (long)w.b * N * a.H + w.h;
  float rl2 = 0.f, rdl = 0.f;
  auto issue = [&](int t) {
    dma_tile64(Qx + (t & 1) * IMG_HALVES, tile_rsrc(qseq, t * 64L * qrow_bytes, qvalid_bytes), dlq);
    dma_tile64(Dx + (t & 1) * IMG_HALVES, tile_rsrc(dseq, t * 64L * drow_bytes, dvalid_bytes), dld);
    if (tid < 64) {
      // RAW loads only; the arithmetic waits in publish() at the end of the tile (a use here parks wave 0 on s_waitcnt vmcnt(0),
      // the DMA just issued included: attn.hip, dK/dV kernel)
      const long off = (long)min(t * 64 + lane, N - 1) * a.H;
      rl2 = lbase[off];
      rdl = dbase[off];
    }
  };
  auto publish = [&](int t) {
    if (tid < 64) {
      const bool ok = t * 64 + lane < N;      // (Q = dO = 0 past the end: P' meets zeros)
      L2s[t & 1][tid] = ok ? fmaf(-rl2, LOG2E, LOG2_LN2) : 0.f;
      Dls[t & 1][tid] = ok ? -rdl : 0.f;
    }
  };

  f32x16 dk0 = splat16(0.f), dk1 = splat16(0.f), dv0 = splat16(0.f), dv1 = splat16(0.f);
  const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  int rrd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rrd[ks] = img_off(l31, 2 * ks + hh);
  const int trc = 2 * (grp & 1) + (tp >> 1), tro = 4 * (tp & 1);
  const int tr_a0 = img_off(4 * hh + tq, trc) + tro, tr_a1 = img_off(4 * hh + tq, trc + 4) + tro;
  const int tr_b0 = img_off(4 * hh + tq + 8, trc) + tro, tr_b1 = img_off(4 * hh + tq + 8, trc + 4) + tro;

  issue(0);
  if (BIAS) dist_load(dr, dstripe, 0);
  publish(0);
  dma_wait_all();
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const h16* Qb = Qx + (t & 1) * IMG_HALVES;
    const h16* Db = Dx + (t & 1) * IMG_HALVES;
    const float* L2b = L2s[t & 1];
    const float* Dlb = Dls[t & 1];
    if (t + 1 < ntile) issue(t + 1);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(&L2b[sub * 32 + 8 * g4 + 4 * hh]);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(&Dlb[sub * 32 + 8 * g4 + 4 * hh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * g4 + e;
          s[i] = BIAS ? dist_bias(dr, sub, i, nslope, x0[e]) : x0[e];
          dp[i] = x1[e];
        }
      }
      if (BIAS && sub == 1 && t + 1 < ntile) dist_load(dr, dstripe, t + 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const h16x8 qa = *reinterpret_cast<const h16x8*>(&Qb[sub * 32 * IMG_ROW + rrd[ks]]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa, kf[ks], s, 0, 0, 0);
        const h16x8 da = *reinterpret_cast<const h16x8*>(&Db[sub * 32 * IMG_ROW + rrd[ks]]);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(da, vf[ks], dp, 0, 0, 0);
      }
      h16x8 pf[2], dsf[2];
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const f32x2 pt = pk_exp2((f32x2){s[i], s[i + 1]});
        const f32x2 d = pt * (f32x2){dp[i], dp[i + 1]};
        pf[i >> 3][i & 7] = (h16)pt[0]; pf[i >> 3][(i & 7) + 1] = (h16)pt[1];
        dsf[i >> 3][i & 7] = (h16)d[0]; dsf[i >> 3][(i & 7) + 1] = (h16)d[1];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int rb = (sub * 32 + s2 * 16) * IMG_ROW;
        const h16x8 d0 = cat8(lds_tr4(&Db[rb + tr_a0]), lds_tr4(&Db[rb + tr_b0]));
        const h16x8 d1 = cat8(lds_tr4(&Db[rb + tr_a1]), lds_tr4(&Db[rb + tr_b1]));
        dv0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d0, pf[s2], dv0, 0, 0, 0);
        dv1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, pf[s2], dv1, 0, 0, 0);
        const h16x8 q0 = cat8(lds_tr4(&Qb[rb + tr_a0]), lds_tr4(&Qb[rb + tr_b0]));
        const h16x8 q1 = cat8(lds_tr4(&Qb[rb + tr_a1]), lds_tr4(&Qb[rb + tr_b1]));
        dk0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0, dsf[s2], dk0, 0, 0, 0);
        dk1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1, dsf[s2], dk1, 0, 0, 0);
      }
    }
    if (t + 1 < ntile) publish(t + 1);
    dma_wait_all();
    __syncthreads();
  }
  if (kvalid) {
    h16* outk = dqkv + krow * ld + a.D + w.h * DH;
    h16* outv = outk + a.D;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const h16x4 k0 = {(h16)dk0[4 * gq], (h16)dk0[4 * gq + 1], (h16)dk0[4 * gq + 2], (h16)dk0[4 * gq + 3]};
      const h16x4 k1 = {(h16)dk1[4 * gq], (h16)dk1[4 * gq + 1], (h16)dk1[4 * gq + 2], (h16)dk1[4 * gq + 3]};
      const h16x4 v0 = {(h16)(dv0[4 * gq] * INV_LN2), (h16)(dv0[4 * gq + 1] * INV_LN2), (h16)(dv0[4 * gq + 2] * INV_LN2),
                        (h16)(dv0[4 * gq + 3] * INV_LN2)};
      const h16x4 v1 = {(h16)(dv1[4 * gq] * INV_LN2), (h16)(dv1[4 * gq + 1] * INV_LN2), (h16)(dv1[4 * gq + 2] * INV_LN2),
                        (h16)(dv1[4 * gq + 3] * INV_LN2)};
      *reinterpret_cast<h16x4*>(outk + 8 * gq + 4 * hh) = k0;
      *reinterpret_cast<h16x4*>(outk + 32 + 8 * gq + 4 * hh) = k1;
      *reinterpret_cast<h16x4*>(outv + 8 * gq + 4 * hh) = v0;
      *reinterpret_cast<h16x4*>(outv + 32 + 8 * gq + 4 * hh) = v1;
    }
  }
}

// delta[m, h] = sum_d dO[m, h, d] * O[m, h, d] (token-major fp16 rows of H * 64): eight lanes per (row, head), 16 bytes each
__global__ __launch_bounds__(256) void dense_attn_delta_kernel(const h16* __restrict__ o, const h16* __restrict__ d_o,
                                                               float* __restrict__ delta, long M, int H) {
  const long nchunk = M * H * 8;
  for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunk; c += (long)gridDim.x * blockDim.x) {
    const h16x8 x = ldg8(o + c * 8), y = ldg8(d_o + c * 8);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf((float)x[e], (float)y[e], s);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if ((c & 7) == 0) delta[c >> 3] = s;
  }
}

// Blocked distance table (see DistRegs): grid (tile t, block A), thread = (piece j, lane).
__global__ __launch_bounds__(256) void alibi_dist_kernel(const int* __restrict__ cells, int N, int ntile, h16* __restrict__ tab) {
  const int t = blockIdx.x, A = blockIdx.y, j = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l31 = lane & 31, hh = lane >> 5, sub = j >> 1, half = j & 1;
  const int ia = A * 32 + l31;
  const bool aok = ia >= 1 && ia < N;
  const float xa = aok ? (float)cells[2 * (ia - 1)] : 0.f, ya = aok ? (float)cells[2 * (ia - 1) + 1] : 0.f;
  h16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = 8 * half + e, ib = t * 64 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
    float d = 0.f;
    if (aok && ib >= 1 && ib < N) {
      const float dx = xa - (float)cells[2 * (ib - 1)], dy = ya - (float)cells[2 * (ib - 1) + 1];
      d = __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy));
    }
    v[e] = (h16)d;
  }
  *reinterpret_cast<h16x8*>(tab + ((long)A * ntile + t) * DBLK + j * 512 + lane * 8) = v;
}

bool dense_plan_ok(const MtDensePlan* p) {
  if (!p || p->N < 1 || p->B < 1 || p->H < 1 || p->H > 64) return false;
  if (p->dist && !p->nslope) return false;
  return true;
}
DenseArgs dense_args(const mt_half* qkv, const MtDensePlan* p) {
  DenseArgs a;
  a.qkv = (const h16*)qkv; a.N = p->N; a.B = p->B; a.H = p->H; a.D = p->H * DH; a.ld = 3L * a.D;
  a.nslope = p->nslope; a.dist = (const h16*)p->dist;
  a.qtiles = cdiv(p->N, 128);
  return a;
}
int dense_grid(const DenseArgs& a) { return cdiv(a.B * a.H, 8) * 8 * a.qtiles; }

}  // namespace

extern "C" long mt_alibi_dist_halves(int N) { return N < 1 ? 0 : 4L * cdiv(N, 128) * cdiv(N, 64) * DBLK; }
extern "C" int mt_alibi_dist(const int* cells, int N, mt_half* table, mt_stream_t stream) {
  if (!table || N < 1 || (N > 1 && !cells)) return MT_ERR_BAD_ARG;
  const int ntile = cdiv(N, 64), nA = 4 * cdiv(N, 128);
  if (nA > 65535) return MT_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(alibi_dist_kernel, dim3(ntile, nA), dim3(256), 0, (hipStream_t)stream, cells, N, ntile, (h16*)table);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_dense_attn_fwd(const mt_half* qkv, const MtDensePlan* plan, mt_half* o, float* lse, mt_stream_t stream) {
  if (!qkv || !o || !lse || !dense_plan_ok(plan)) return MT_ERR_BAD_ARG;
  const DenseArgs a = dense_args(qkv, plan);
  if (a.dist)
    hipLaunchKernelGGL(dense_attn_fwd_kernel<true>, dim3(dense_grid(a)), dim3(256), 0, (hipStream_t)stream, a, (h16*)o, lse);
  else
    hipLaunchKernelGGL(dense_attn_fwd_kernel<false>, dim3(dense_grid(a)), dim3(256), 0, (hipStream_t)stream, a, (h16*)o, lse);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_dense_attn_bwd(const mt_half* qkv, const mt_half* o, const mt_half* d_o, const float* lse, const MtDensePlan* plan,
                                 float* delta, mt_half* dqkv, int phases, mt_stream_t stream) {
  if (!qkv || !o || !d_o || !lse || !delta || !dqkv || !dense_plan_ok(plan) || !(phases & 7)) return MT_ERR_BAD_ARG;
  const DenseArgs a = dense_args(qkv, plan);
  hipStream_t s = (hipStream_t)stream;
  const long M = (long)a.B * a.N;
  if (phases & MT_DENSE_BWD_DELTA)
    hipLaunchKernelGGL(dense_attn_delta_kernel, dim3((int)min((M * a.H * 8 + 255) / 256, 16384L)), dim3(256), 0, s, (const h16*)o,
                       (const h16*)d_o, delta, M, a.H);
  const dim3 grid(dense_grid(a));
  if (phases & MT_DENSE_BWD_KV) {
    if (a.dist) hipLaunchKernelGGL(dense_attn_bwd_kv_kernel<true>, grid, dim3(256), 0, s, a, (const h16*)d_o, lse, delta, (h16*)dqkv);
    else hipLaunchKernelGGL(dense_attn_bwd_kv_kernel<false>, grid, dim3(256), 0, s, a, (const h16*)d_o, lse, delta, (h16*)dqkv);
  }
  if (phases & MT_DENSE_BWD_Q) {
    if (a.dist) hipLaunchKernelGGL(dense_attn_bwd_q_kernel<true>, grid, dim3(256), 0, s, a, (const h16*)d_o, lse, delta, (h16*)dqkv);
    else hipLaunchKernelGGL(dense_attn_bwd_q_kernel<false>, grid, dim3(256), 0, s, a, (const h16*)d_o, lse, delta, (h16*)dqkv);
  }
  MT_CHECK_LAUNCH();
  return MT_OK;
}
