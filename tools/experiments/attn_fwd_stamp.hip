// DIAGNOSTIC build of the forward attention kernel (guide §7, in-kernel stamps): where does a tile's time go?
// Generated from modaltune_amd/csrc/attn.hip by tools/experiments/make_attn_stamp.py; never shipped, never timed as a whole.
// Per wave and tile: [t0 top .. t1 scores + row maximum + rescale decision done] [t1 .. t2 exp / convert / P.V issued and
// retired] [t2 .. t3 DMA wait + barrier].  Sums over all waves land in dbg[0..2], the tile count in dbg[3].
#include "../../modaltune_amd/csrc/attn_common.h"

namespace {
MT_DEVINL unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
__global__ __launch_bounds__(256) void dilated_attn_fwd_stamp_kernel(const h16* __restrict__ qkv, Plan p, h16* __restrict__ o_br,
                                                               float* __restrict__ lse_br, unsigned long long* __restrict__ dbg) {
  unsigned long long acc_s = 0, acc_p = 0, acc_b = 0, acc_n = 0;
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
  if (w.qt * 128 >= nv) return;

  // constant chunks of the V images: logical chunk 6 = ones at d = 48 and 52 (O^T row 48 of both lane halves accumulates
  // sum(P)), chunk 7 = zeros; written once (the DMA never touches them)
  {
    const int buf = tid >> 7, row = (tid >> 1) & 63, which = 6 + (tid & 1);
    const h16x8 one = {(h16)1.f, 0, 0, 0, (h16)1.f, 0, 0, 0}, zero = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<h16x8*>(&Vs[buf * IMG_HALVES + img_off(row, which)]) = which == 6 ? one : zero;
  }

  // Q^T fragments (B operand): lane = query, element j of k-step ks = Q[q][16 ks + 8 hh + j]
  const int iq = w.qt * 128 + wave * 32 + l31;
  const bool qvalid = sq.valid(iq);
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
    m2 = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
    for (int i = 0; i < 16; ++i) minit[i] = -m2;
  }

  auto tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int kb = t * 64;
    const h16* Kb = Ks + (t & 1) * IMG_HALVES;
    const h16* Vb = Vs + (t & 1) * IMG_HALVES;
    const unsigned long long t0 = stamp();
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
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
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
    const unsigned long long t1 = stamp();
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
    asm volatile("" :: "v"(o0), "v"(o1));
    const unsigned long long t2 = stamp();
    dma_wait_all();            // tile t + 1 has landed (this wave's pieces) ...
    __syncthreads();           // ... everybody's, and everybody is done reading tile t's images
    const unsigned long long t3 = stamp();
    acc_s += t1 - t0; acc_p += t2 - t1; acc_b += t3 - t2; acc_n += 1;
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

  if (lane == 0) {
    atomicAdd(&dbg[0], acc_s); atomicAdd(&dbg[1], acc_p); atomicAdd(&dbg[2], acc_b); atomicAdd(&dbg[3], acc_n);
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


}  // namespace

extern "C" int mt_dbg_attn_fwd_stamps(const mt_half* qkv, const MtDilatedPlan* plan, mt_half* o_br, float* lse_br,
                                      unsigned long long* dbg, mt_stream_t stream) {
  if (!plan_ok(plan)) return MT_ERR_BAD_ARG;
  const Plan p = make_plan(plan, 128);
  hipLaunchKernelGGL(dilated_attn_fwd_stamp_kernel, dim3(p.blk_off[p.nbranch]), dim3(256), 0, (hipStream_t)stream, (const h16*)qkv, p,
                     (h16*)o_br, lse_br, dbg);
  MT_CHECK_LAUNCH();
  return MT_OK;
}
