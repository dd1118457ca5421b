// Dilated attention (LongNet) for the frozen Prov-GigaPath backbone: 16 heads x 48, five branches.
//
// Reference semantics (torchscale/component/dilated_attention.py:22-59,82-144,212-255):
//   branch (s, r): the sequence is cut in segments of s tokens; head h (group g = h / (16/r)) attends, inside
//   each segment, over the positions g, g+r, g+2r, ... (n = ceil(s/r) entries).  Entries past the segment end
//   or past N are ALL-ZERO rows that still act as keys (logit 0, value 0).  Branch outputs are mixed with
//   softmax-over-branches of the per-(position, head) LSE, treated as constants in backward.
//
// Forward kernel: one workgroup = 128 queries (4 waves x 32) of one (pass, branch, segment, head); K/V tiles of
// 64 keys are gathered in place from the fused qkv activation (no diag_embed / pad copies) into LDS.
// Per wave, "swapped" products keep the softmax lane-local (query = lane, keys = registers):
//   S^T[key, q] = K . Q^T           v_mfma_f32_32x32x16_f16, K rows from LDS (ds_read_b128), Q^T in registers
//   O^T[d, q]  += V^T[d, key] . P^T  P^T taken straight from the S accumulators (no LDS round trip),
//                                    V^T via ds_read_b64_tr_b16; V carries a ones column so O^T row 48 = sum(P)
#include "attn_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// forward.  K and V tiles live in two LDS-DMA images each (attn_common.h: img_off), one barrier per tile: the DMA of tile
// t + 1 is issued at the top of tile t and awaited at its end.  No staging registers, stores or per-tile address
// arithmetic; the only tile variant is the last one of a sequence whose length is not a multiple of 64.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dilated_attn_fwd_kernel(const h16* __restrict__ qkv, Plan p, h16* __restrict__ o_br,
                                                               float* __restrict__ lse_br) {
  __shared__ __attribute__((aligned(16))) h16 smem[4 * IMG_HALVES];      // K0 | K1 | V0 | V1
  h16* const Ks = smem;
  h16* const Vs = smem + 2 * IMG_HALVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const WorkItem w = decode(p, blockIdx.x);
  const Seq sq = make_seq(p, w);
  const long M = (long)p.B * p.N;
  // Entries [nv, n) of the sparse sequence are zero padding (segment / sequence end): as QUERIES they produce nothing
  // that is ever read, as KEYS they all have logit 0 and value 0.  A workgroup of padded queries exits; key tiles
  // made only of padding are not computed -- their sum(P) share is added in closed form after the loop.
  const int nv = __builtin_amdgcn_readfirstlane(sq.nvalid());
  const int nvq = min(nv, p.qlimit[w.br]);                 // entries that act as queries (sequence-parallel plans: a prefix)
  if (w.qt * 128 >= nvq) return;

  // constant chunks of the V images: logical chunk 6 = ones at d = 48 and 52 (O^T row 48 of both lane halves accumulates
  // sum(P)), chunk 7 = zeros; written once (the DMA never touches them)
  {
    const int buf = tid >> 7, row = (tid >> 1) & 63, which = 6 + (tid & 1);
    const h16x8 one = {(h16)1.f, 0, 0, 0, (h16)1.f, 0, 0, 0}, zero = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<h16x8*>(&Vs[buf * IMG_HALVES + img_off(row, which)]) = which == 6 ? one : zero;
  }

  // Q^T fragments (B operand): lane = query, element j of k-step ks = Q[q][16 ks + 8 hh + j]
  const int iq = w.qt * 128 + wave * 32 + l31;
  const bool qvalid = sq.valid(iq) && iq < nvq;
  const long qrow = sq.row_clamped(iq);
  // q arrives pre-scaled (attn_common.h: QK_SCALE_LOG2): S' = K . Q'^T is the exp2 argument as it leaves the MFMA chain, up
  // to the running reference m2 -- which rides in as the INITIAL accumulator (below).
  h16x8 qf[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
    qf[ks] = sel8(qvalid, ldg8(hm_ptr(qkv, M, w.h, qrow) + ks * 16 + hh * 8));

  const int ntile = (sq.n + 63) / 64;
  const int nproc = (nv + 63) >> 6;      // tiles holding at least one real key
  const int row_bytes = sq.dr * HD * 2;                             // distance of two sparse entries in memory
  const long valid_bytes = (long)(nv - 1) * row_bytes + HD * 2;     // entries [0, nv) are real rows
  const long tile_bytes = 64L * row_bytes;
  const h16* const kseq = hm_ptr(qkv, M, H + w.h, sq.row(0));
  const h16* const vseq = hm_ptr(qkv, M, 2 * H + w.h, sq.row(0));
  const DmaLane dl(tid, row_bytes);
  auto dma = [&](int t) {
    dma_tile(Ks + (t & 1) * IMG_HALVES, tile_rsrc(kseq, t * tile_bytes, valid_bytes), dl);
    dma_tile(Vs + (t & 1) * IMG_HALVES, tile_rsrc(vseq, t * tile_bytes, valid_bytes), dl);
  };
  // per-lane read offsets (halves): K rows sub * 32 + l31 at chunk 2 ks + hh; V transposed reads of rows 4 hh + tq (+ 8),
  // d blocks 0..31 / 32..63 (the sub / s2 row-block offsets are multiples of 16 rows: they do not change img_f)
  const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  int krd[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) krd[ks] = img_off(l31, 2 * ks + hh);
  const int vc = 2 * (grp & 1) + (tp >> 1), vo = 4 * (tp & 1);
  const int va0 = img_off(4 * hh + tq, vc) + vo, va1 = img_off(4 * hh + tq, vc + 4) + vo;
  const int vb0 = img_off(4 * hh + tq + 8, vc) + vo, vb1 = img_off(4 * hh + tq + 8, vc + 4) + vo;

  // scores of one 64-key tile relative to the reference: s[sub][reg] = c q.k - m2 (key = row, query = lane); `init` is the
  // accumulator the chains start from (splat(-m2): the query is the lane, so one value per lane)
  auto qk = [&](const h16* Kb, f32x16 (&s)[2], const f32x16& init) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const h16x8 kf = *reinterpret_cast<const h16x8*>(&Kb[sub * 32 * IMG_ROW + krd[ks]]);
        s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? init : s[sub], 0, 0, 0);
      }
    }
  };

  f32x16 o0, o1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }

  // prologue: tile 0 -> LDS
  dma(0);
  dma_wait_all();
  __syncthreads();
  // Running reference m2 of the scaled logits (log2 units), carried as the accumulator initialiser minit = splat(-m2).  It
  // starts at the row maximum over tile 0 (one extra S product per workgroup; tile 0 is then processed by the loop like
  // every other tile) and moves up only through the deferred rescale below.
  float m2;
  f32x16 minit;
  {
    f32x16 s0[2], zero;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
    qk(Ks, s0, zero);
    float mx = NEG_BIG;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kidx = sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        mx = fmaxf(mx, kidx < sq.n ? s0[sub][i] : NEG_BIG);
      }
    m2 = max_halves(mx);
#pragma unroll
    for (int i = 0; i < 16; ++i) minit[i] = -m2;
  }

  auto tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int kb = t * 64;
    const h16* Kb = Ks + (t & 1) * IMG_HALVES;
    const h16* Vb = Vs + (t & 1) * IMG_HALVES;
    if (t + 1 < nproc) dma(t + 1);
    f32x16 s_cur[2];
    qk(Kb, s_cur, minit);
    // keys >= n are tile padding (excluded, last tile only); zero-padded keys keep logit 0 (DA:98-101)
    float mx = NEG_BIG;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (TAIL) {
          const int kidx = kb + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (kidx >= sq.n) s_cur[sub][i] = NEG_BIG;
        }
        mx = fmaxf(mx, s_cur[sub][i]);
      }
    mx = max_halves(mx);
    // Deferred rescale (exact): the reference m2 only moves when some row's maximum grew by more than 2^RESCALE_LOG2 past
    // it; until then P = exp2(c s - m2) <= 2^RESCALE_LOG2, which fp16 P / fp32 O hold without loss.
    if (__any(mx > RESCALE_LOG2)) {
      const float up = fmaxf(mx, 0.f);               // this row's reference moves up by `up`
      const float alpha = __builtin_amdgcn_exp2f(-up);
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; s_cur[0][i] -= up; s_cur[1][i] -= up; }
      m2 += up;
#pragma unroll
      for (int i = 0; i < 16; ++i) minit[i] = -m2;
    }
    // O^T += V^T . P^T ; A fragment element e of lane half hh = V[key 16 s2 + 8 (e>>2) + 4 hh + (e&3)][d = lane & 31]
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        h16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 a = pk_exp2((f32x2){s_cur[sub][8 * s2 + e], s_cur[sub][8 * s2 + e + 1]});
          pf[e] = (h16)a[0]; pf[e + 1] = (h16)a[1];
        }
        const h16* vblk = Vb + (sub * 32 + s2 * 16) * IMG_ROW;
        const h16x8 v0 = cat8(lds_tr4(vblk + va0), lds_tr4(vblk + vb0));
        const h16x8 v1 = cat8(lds_tr4(vblk + va1), lds_tr4(vblk + vb1));
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, pf, o1, 0, 0, 0);
      }
    dma_wait_all();            // tile t + 1 has landed (this wave's pieces) ...
    __syncthreads();           // ... everybody's, and everybody is done reading tile t's images
  };
  const bool tail_last = nproc == ntile && (sq.n & 63);
  const int nplain = tail_last ? nproc - 1 : nproc;
  for (int t = 0; t < nplain; ++t) tile(t, std::false_type{});
  if (tail_last) tile(nproc - 1, std::true_type{});
  const int rest = sq.n - nproc * 64;      // padded keys in the tiles not computed: logit 0, value 0
  if (rest > 0) {
    if (__any(-m2 > RESCALE_LOG2)) {             // logit 0 lies more than the threshold above the reference
      const float up = fmaxf(-m2, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-up);
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
      m2 += up;
    }
    o1[8] += (float)rest * __builtin_amdgcn_exp2f(-m2);
  }

  if (qvalid) {
    const float l = o1[8];           // O^T row 48 (lane half 0) / row 52 (lane half 1): both carry sum(P)
    const float inv = 1.0f / l;
    h16* orow = o_br + ((long)w.br * M + qrow) * DM + w.h * HD;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      h16x4 v = {(h16)(o0[4 * gq] * inv), (h16)(o0[4 * gq + 1] * inv), (h16)(o0[4 * gq + 2] * inv), (h16)(o0[4 * gq + 3] * inv)};
      *reinterpret_cast<h16x4*>(orow + 8 * gq + 4 * hh) = v;
    }
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      h16x4 v = {(h16)(o1[4 * gq] * inv), (h16)(o1[4 * gq + 1] * inv), (h16)(o1[4 * gq + 2] * inv), (h16)(o1[4 * gq + 3] * inv)};
      *reinterpret_cast<h16x4*>(orow + 32 + 8 * gq + 4 * hh) = v;
    }
    if (hh == 0) lse_br[((long)w.br * M + qrow) * H + w.h] = (m2 + __log2f(l)) * LN2;
  }
}

// ------------------------------------------------------------------------------------------------
// branch mix + inner_attn_ln (forward / backward): one wave per token row (16 heads x 48); a lane owns 12 consecutive
// columns = a quarter of one head.  HBM-bound passes over the covered branch outputs, so the row loop is kept
// branch-free: a lane's group id per branch is computed once, the row's residue per branch is wave-uniform, and a
// branch that does not cover (position, head) reads the (always covered) ratio-1 branch's line instead -- an L1 hit --
// and is discarded with a select.  (A branch around each load made hipcc wait for every load separately.)
// ------------------------------------------------------------------------------------------------
struct MixGeom {
  int grp[MT_MAX_BRANCHES];     // this lane's head group per branch: h / (16 / ratio)
  MT_DEVINL MixGeom(const Plan& p, int h) {
#pragma unroll
    for (int b = 0; b < MT_MAX_BRANCHES; ++b) grp[b] = b < p.nbranch ? h / (H / p.ratio[b]) : -1;
  }
};
// position -> (segment index, offset inside the segment) without an integer division: q = floor(pos / seg) from the
// float reciprocal, corrected by one step (exact for pos < 2^23)
MT_DEVINL void seg_split(const Plan& p, int b, int pos, int& j, int& loc) {
  const int sg = p.seg[b];
  if (sg >= p.N) { j = 0; loc = pos; return; }
  j = (int)((float)pos * p.inv_seg[b]);
  loc = pos - j * sg;
  if (loc < 0) { loc += sg; --j; }
  if (loc >= sg) { loc -= sg; ++j; }
}
// residue of position pos inside its segment, modulo the dilation
MT_DEVINL int residue(const Plan& p, int b, int pos) {
  int j, loc;
  seg_split(p, b, pos, j, loc);
  const int dr = p.ratio[b];
  return (dr & (dr - 1)) == 0 ? (loc & (dr - 1)) : loc % dr;
}
MT_DEVINL int dense_branch(const Plan& p) {      // a branch with ratio 1 covers every (position, head)
  int d = 0;
#pragma unroll
  for (int b = 0; b < MT_MAX_BRANCHES; ++b) if (b < p.nbranch && p.ratio[b] == 1) d = b;
  return d;
}
// Two token rows per wave: 32 lanes per row, a lane owns 24 consecutive columns = half of one head (three 16-byte loads
// per branch; the 12-column / 8-byte form ran at 3.6 TB/s, tools: profiles/r02 roofline table).
constexpr int MIX_CPL = 24;
struct H24 { h16x8 a, b, c; };
MT_DEVINL H24 ld24(const h16* p) {
  return H24{*reinterpret_cast<const h16x8*>(p), *reinterpret_cast<const h16x8*>(p + 8), *reinterpret_cast<const h16x8*>(p + 16)};
}
MT_DEVINL float h24(const H24& v, int e) { return (float)(e < 8 ? v.a[e & 7] : e < 16 ? v.b[e & 7] : v.c[e & 7]); }
MT_DEVINL float half_sum(float v) {       // over the 32 lanes of a row
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NB>
__global__ __launch_bounds__(256) void mix_ln_fwd_kernel(const h16* __restrict__ o_br, const float* __restrict__ lse_br, Plan p,
                                                         const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                         h16* __restrict__ y, float* __restrict__ stats, float* __restrict__ lse_tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 5, l5 = lane & 31;
  const long M = (long)p.B * p.N;
  const int h = l5 >> 1, c0 = l5 * MIX_CPL;
  const MixGeom geo(p, h);
  const int db = dense_branch(p);
  float w[MIX_CPL], bb[MIX_CPL];
#pragma unroll
  for (int e = 0; e < MIX_CPL; ++e) { w[e] = ln_w[c0 + e]; bb[e] = ln_b[c0 + e]; }
  for (long m2 = ((long)blockIdx.x * 4 + wave) * 2; m2 < M; m2 += (long)gridDim.x * 8) {
    const bool live = m2 + sub < M;
    const long m = live ? m2 + sub : M - 1;
    const int pos = (int)(m % p.N);
    bool cov[NB];
    float lse[NB];
    H24 ob[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      cov[b] = b < p.nbranch && geo.grp[b] == residue(p, b, pos);
      const int sb = cov[b] ? b : db;
      lse[b] = lse_br[((long)sb * M + m) * H + h];
      ob[b] = ld24(o_br + ((long)sb * M + m) * DM + c0);
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int b = 0; b < NB; ++b) { lse[b] = cov[b] ? lse[b] : NEG_BIG; mx = fmaxf(mx, lse[b]); }
    float den = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) den += cov[b] ? __expf(lse[b] - mx) : 0.f;
    const float tot = mx + __logf(den);
    float v[MIX_CPL];
#pragma unroll
    for (int e = 0; e < MIX_CPL; ++e) v[e] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float wgt = cov[b] ? __expf(lse[b] - tot) : 0.f;
#pragma unroll
      for (int e = 0; e < MIX_CPL; ++e) v[e] = fmaf(wgt, cov[b] ? h24(ob[b], e) : 0.f, v[e]);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < MIX_CPL; ++e) s += v[e];
    const float mean = half_sum(s) * (1.0f / DM);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < MIX_CPL; ++e) { const float d = v[e] - mean; q += d * d; }
    const float rstd = rsqrtf(half_sum(q) * (1.0f / DM) + 1e-5f);
    if (live) {
      h16* dst = y + m * DM + c0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16)((v[8 * k + e] - mean) * rstd * w[8 * k + e] + bb[8 * k + e]);
        *reinterpret_cast<h16x8*>(dst + 8 * k) = o;
      }
      if (l5 == 0) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
      if ((l5 & 1) == 0) lse_tot[m * H + h] = tot;
    }
  }
}

// backward of mix + LN: recompute mixed from the branch outputs, LayerNorm backward (frozen affine: no dw/db),
// dmixed (fp16, head-major) and delta_b = sum_d dmixed * O_b per (row, head, branch).
template <int NB>
__global__ __launch_bounds__(256) void mix_ln_bwd_kernel(const h16* __restrict__ dy, const h16* __restrict__ o_br,
                                                         const float* __restrict__ lse_br, const float* __restrict__ lse_tot, Plan p,
                                                         const float* __restrict__ ln_w, const float* __restrict__ stats,
                                                         h16* __restrict__ dmixed, float* __restrict__ delta_br) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 5, l5 = lane & 31;
  const long M = (long)p.B * p.N;
  const int h = l5 >> 1, c0 = l5 * MIX_CPL;
  const MixGeom geo(p, h);
  const int db = dense_branch(p);
  float w[MIX_CPL];
#pragma unroll
  for (int e = 0; e < MIX_CPL; ++e) w[e] = ln_w[c0 + e];
  for (long m2 = ((long)blockIdx.x * 4 + wave) * 2; m2 < M; m2 += (long)gridDim.x * 8) {
    const bool live = m2 + sub < M;
    const long m = live ? m2 + sub : M - 1;
    const int pos = (int)(m % p.N);
    const float tot = lse_tot[m * H + h];
    const float mean = stats[2 * m], rstd = stats[2 * m + 1];
    const H24 dyv = ld24(dy + m * DM + c0);
    bool cov[NB];
    float lse[NB];
    H24 ob[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      cov[b] = b < p.nbranch && geo.grp[b] == residue(p, b, pos);
      const int sb = cov[b] ? b : db;
      lse[b] = lse_br[((long)sb * M + m) * H + h];
      ob[b] = ld24(o_br + ((long)sb * M + m) * DM + c0);
    }
    float v[MIX_CPL];
#pragma unroll
    for (int e = 0; e < MIX_CPL; ++e) v[e] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float wgt = cov[b] ? __expf(lse[b] - tot) : 0.f;
#pragma unroll
      for (int e = 0; e < MIX_CPL; ++e) v[e] = fmaf(wgt, cov[b] ? h24(ob[b], e) : 0.f, v[e]);
    }
    float g[MIX_CPL], xh[MIX_CPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MIX_CPL; ++i) {
      xh[i] = (v[i] - mean) * rstd;
      g[i] = h24(dyv, i) * w[i];
      s1 += g[i]; s2 = fmaf(g[i], xh[i], s2);
    }
    const float c1 = half_sum(s1) * (1.0f / DM), c2 = half_sum(s2) * (1.0f / DM);
    float dm[MIX_CPL];
    h16* dst = dmixed + ((long)h * M + m) * HD + (c0 - h * HD);      // head-major [head][B*N][48]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      h16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = 8 * k + e;
        o[e] = (h16)(rstd * (g[i] - c1 - xh[i] * c2));
        dm[i] = (float)o[e];          // delta must match the fp16 dmixed the attention backward consumes
      }
      if (live) *reinterpret_cast<h16x8*>(dst + 8 * k) = o;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < MIX_CPL; ++e) d = fmaf(dm[e], h24(ob[b], e), d);
      d += __shfl_xor(d, 1, 64);
      if (live && (l5 & 1) == 0 && cov[b]) delta_br[((long)b * M + m) * H + h] = d;
    }
  }
}

// (backward kernels: attn_bwd_q.hip, attn_bwd_kv.hip, attn_combine.hip -- one translation unit per LLVM scheduling strategy)

}  // namespace

extern "C" int mt_dilated_attn_fwd(const mt_half* qkv, const MtDilatedPlan* plan, mt_half* o_br, float* lse_br,
                                   mt_stream_t stream) {
  if (!qkv || !o_br || !lse_br || !plan_ok(plan)) return MT_ERR_BAD_ARG;
  const Plan p = make_plan(plan, 128);
  const int nblk = p.blk_off[p.nbranch];
  hipLaunchKernelGGL(dilated_attn_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const h16*)qkv, p,
                     (h16*)o_br, lse_br);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_dilated_mix_ln_fwd(const mt_half* o_br, const float* lse_br, const MtDilatedPlan* plan,
                                     const float* ln_w, const float* ln_b, mt_half* y, float* stats, float* lse_tot,
                                     mt_stream_t stream) {
  if (!o_br || !lse_br || !ln_w || !ln_b || !y || !stats || !lse_tot || !plan_ok(plan)) return MT_ERR_BAD_ARG;
  const Plan p = make_plan(plan, 128);
  const long M = (long)p.B * p.N;
  const dim3 grid((int)min((M + 7) / 8, 8192L));
  if (p.nbranch <= 5)
    hipLaunchKernelGGL(mix_ln_fwd_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, (const h16*)o_br, lse_br, p, ln_w, ln_b,
                       (h16*)y, stats, lse_tot);
  else
    hipLaunchKernelGGL(mix_ln_fwd_kernel<MT_MAX_BRANCHES>, grid, dim3(256), 0, (hipStream_t)stream, (const h16*)o_br, lse_br, p,
                       ln_w, ln_b, (h16*)y, stats, lse_tot);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_dilated_mix_ln_bwd(const mt_half* dy, const mt_half* o_br, const float* lse_br, const float* lse_tot,
                                     const MtDilatedPlan* plan, const float* ln_w, const float* stats, mt_half* dmixed,
                                     float* delta_br, mt_stream_t stream) {
  if (!dy || !o_br || !lse_br || !lse_tot || !ln_w || !stats || !dmixed || !delta_br || !plan_ok(plan)) return MT_ERR_BAD_ARG;
  const Plan p = make_plan(plan, 128);
  const long M = (long)p.B * p.N;
  const dim3 grid((int)min((M + 7) / 8, 8192L));
  if (p.nbranch <= 5)
    hipLaunchKernelGGL(mix_ln_bwd_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, (const h16*)dy, (const h16*)o_br, lse_br,
                       lse_tot, p, ln_w, stats, (h16*)dmixed, delta_br);
  else
    hipLaunchKernelGGL(mix_ln_bwd_kernel<MT_MAX_BRANCHES>, grid, dim3(256), 0, (hipStream_t)stream, (const h16*)dy,
                       (const h16*)o_br, lse_br, lse_tot, p, ln_w, stats, (h16*)dmixed, delta_br);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" long mt_dilated_attn_bwd_workspace_bytes(const MtDilatedPlan* plan) {
  if (!plan_ok(plan)) return MT_ERR_BAD_ARG;
  const Plan p = make_plan(plan, 128);
  return p.ws_off[p.nbranch] * (long)sizeof(h16);
}

extern "C" int mt_dilated_attn_bwd(const mt_half* qkv, const mt_half* dmixed, const float* lse_tot,
                                   const float* delta_br, const MtDilatedPlan* plan, void* workspace, mt_half* dqkv,
                                   int phases, mt_stream_t stream) {
  if (!qkv || !dmixed || !lse_tot || !delta_br || !workspace || !dqkv || !plan_ok(plan) || !(phases & 7)) return MT_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  // every (branch, position, head) slot of the workspace is written exactly once by each of the two kernels
  if (phases & MT_ATTN_BWD_KV) mt_attn::launch_bwd_kv(qkv, dmixed, lse_tot, delta_br, plan, workspace, s);
  if (phases & MT_ATTN_BWD_Q) mt_attn::launch_bwd_q(qkv, dmixed, lse_tot, delta_br, plan, workspace, s);
  if (phases & MT_ATTN_BWD_COMBINE) mt_attn::launch_bwd_combine(workspace, plan, dqkv, s);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
