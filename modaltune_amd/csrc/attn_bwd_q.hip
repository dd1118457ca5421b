// Dilated attention (LongNet), backward: the dQ kernel.
// (see attn.hip for the reference semantics and the forward; split out so that the translation unit can carry its own
// LLVM scheduling strategy -- attn_common.h)
#include "attn_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// backward, kernel Q: dQ.  Same decomposition as the forward (query = lane).
//   P'^T = exp2(S'^T - L2[q] + log2(ln 2))     (S' = K . Q'^T with the pre-scaled q'; L2 = lse_tot * log2e; P' = ln2 w_b P_b)
//   dP^T[key,q] = V . dO^T                      V rows from LDS, dO^T in registers
//   dS^T = P'^T (dP^T - delta_b[q])             (= dL/dS': the ln 2 rides inside P')
//   dQ^T[d,q] += K^T[d,key] . dS^T              K^T via transposed LDS reads
// The elementwise block is written with packed fp32 ops (v_pk_fma/add/mul_f32): these kernels issue about as many
// VALU cycles as MFMA cycles, and the two did not overlap (PMC: VALU 49 %, MFMA 37 % busy before this form).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dilated_attn_bwd_q_kernel(const h16* __restrict__ qkv, const h16* __restrict__ dmixed,
                                                                 const float* __restrict__ lse_tot, const float* __restrict__ delta_br,
                                                                 Plan p, h16* __restrict__ ws) {
  // K and V tiles in LDS-DMA images (attn_common.h: img_off), double-buffered, one barrier per tile; the K image serves
  // both the row reads (S) and the transposed reads (dQ): one image instead of two, no staging stores
  __shared__ __attribute__((aligned(16))) h16 smem[4 * IMG_HALVES];      // K0 | K1 | V0 | V1
  h16* const Ks = smem;
  h16* const Vs = smem + 2 * IMG_HALVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const WorkItem w = decode(p, blockIdx.x);
  const Seq sq = make_seq(p, w);
  const long M = (long)p.B * p.N;
  const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // padded queries get no gradient; padded keys have K = 0 and add nothing to dQ: neither is computed
  const int nv = __builtin_amdgcn_readfirstlane(sq.nvalid());
  const int nvq = min(nv, p.qlimit[w.br]);
  if (w.qt * 128 >= nvq) return;

  {   // constant chunks 6, 7 (zeros, read as d rows 48..63 of K^T) of both K images; written once
    const int buf = tid >> 7, row = (tid >> 1) & 63, which = 6 + (tid & 1);
    *reinterpret_cast<h16x8*>(&Ks[buf * IMG_HALVES + img_off(row, which)]) = zero8;
  }

  const int iq = w.qt * 128 + wave * 32 + l31;
  const bool qvalid = sq.valid(iq) && iq < nvq;
  const long qrow = sq.row_clamped(iq);
  h16x8 qf[3], dof[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    qf[ks] = sel8(qvalid, ldg8(hm_ptr(qkv, M, w.h, qrow) + ks * 16 + hh * 8));
    dof[ks] = sel8(qvalid, ldg8(hm_ptr(dmixed, M, w.h, qrow) + ks * 16 + hh * 8));
  }
  // invalid queries: -L2 = -big -> P' = 0
  const float L2raw = lse_tot[qrow * H + w.h], dlraw = delta_br[((long)w.br * M + qrow) * H + w.h];
  const float nl2 = qvalid ? fmaf(-L2raw, LOG2E, LOG2_LN2) : -1.0e30f;
  const float ndl = qvalid ? -dlraw : 0.f;
  f32x16 nl2i, ndli;
#pragma unroll
  for (int i = 0; i < 16; ++i) { nl2i[i] = nl2; ndli[i] = ndl; }

  const int ntile = (nv + 63) >> 6;      // tiles holding at least one real key
  const int row_bytes = sq.dr * HD * 2;
  const long valid_bytes = (long)(nv - 1) * row_bytes + HD * 2;     // entries [0, nv) are real rows; the rest read as zeros
  const long tile_bytes = 64L * row_bytes;
  const h16* const kseq = hm_ptr(qkv, M, H + w.h, sq.row(0));
  const h16* const vseq = hm_ptr(qkv, M, 2 * H + w.h, sq.row(0));
  const DmaLane dl(tid, row_bytes);
  auto dma = [&](int t) {
    // (both images of a tile under one exec mask per piece, wave-uniform LDS destinations: dQ -1.6 %; the same form is
    // neutral in the forward and costs the dK/dV kernel 2 %, so those keep the two-call form)
    dma_tile_pair(Ks + (t & 1) * IMG_HALVES, tile_rsrc(kseq, t * tile_bytes, valid_bytes), Vs + (t & 1) * IMG_HALVES,
                  tile_rsrc(vseq, t * tile_bytes, valid_bytes), dl, __builtin_amdgcn_readfirstlane(tid >> 6));
  };
  const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  int rrd[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) rrd[ks] = img_off(l31, 2 * ks + hh);
  const int kc = 2 * (grp & 1) + (tp >> 1), ko = 4 * (tp & 1);
  const int ka0 = img_off(4 * hh + tq, kc) + ko, ka1 = img_off(4 * hh + tq, kc + 4) + ko;
  const int kb0 = img_off(4 * hh + tq + 8, kc) + ko, kb1 = img_off(4 * hh + tq + 8, kc + 4) + ko;

  f32x16 dq0, dq1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dq0[i] = 0.f; dq1[i] = 0.f; }
  dma(0);
  dma_wait_all();
  __syncthreads();
  // tail_tag: the tile holds keys >= n (tile padding, excluded)
  auto tile = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int kb = t * 64;
    const h16* Kb = Ks + (t & 1) * IMG_HALVES;
    const h16* Vb = Vs + (t & 1) * IMG_HALVES;
    if (t + 1 < ntile) dma(t + 1);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      // the per-query constants (query = lane: one value per lane) ride in as the INITIAL accumulators:
      // S' - L2 + log2(ln 2) and dP - delta leave the chains ready
      f32x16 s, dp;
      // The six row fragments of the two chains are requested AHEAD of the products -- three reads up front, then one read per
      // MFMA (sched_group_barrier: DS_READ x3, (MFMA, DS_READ) x3, MFMA x3) -- so every fragment is in flight for >= 2 MFMAs
      // (64+ cycles) before its use.  Left alone the compiler reuses ONE fragment register: read -> s_waitcnt lgkmcnt(0) -> MFMA,
      // six exposed LDS round trips per 32 keys (same-box A/B, tools/attn3_microbench.py: dQ 0.537 -> 0.517 ms, dK/dV 0.802 ->
      // 0.777; requesting all six at once costs registers: dK/dV drops to 2 waves per SIMD and loses 9 %).
      h16x8 ka[3], va[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        ka[ks] = *reinterpret_cast<const h16x8*>(&Kb[sub * 32 * IMG_ROW + rrd[ks]]);
        va[ks] = *reinterpret_cast<const h16x8*>(&Vb[sub * 32 * IMG_ROW + rrd[ks]]);
      }
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka[ks], qf[ks], ks == 0 ? nl2i : s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(va[ks], dof[ks], ks == 0 ? ndli : dp, 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
      h16x8 dsf[2];
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f32x2 pt = pk_exp2((f32x2){s[i], s[i + 1]});
        if (TAIL) {
          const int kidx = kb + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (kidx >= sq.n) pt[0] = 0.f;
          if (kidx + 1 >= sq.n) pt[1] = 0.f;
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
  const bool tail_last = ntile * 64 > sq.n;      // the last processed tile contains keys >= n
  const int nplain = tail_last ? ntile - 1 : ntile;
  for (int t = 0; t < nplain; ++t) tile(t, std::false_type{});
  if (tail_last) tile(ntile - 1, std::true_type{});
  if (qvalid) {
    h16* out = ws + ws_slot(p, w, qrow);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const h16x4 v = {(h16)dq0[4 * gq], (h16)dq0[4 * gq + 1], (h16)dq0[4 * gq + 2], (h16)dq0[4 * gq + 3]};
      *reinterpret_cast<h16x4*>(out + 8 * gq + 4 * hh) = v;
    }
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const h16x4 v = {(h16)dq1[4 * gq], (h16)dq1[4 * gq + 1], (h16)dq1[4 * gq + 2], (h16)dq1[4 * gq + 3]};
      *reinterpret_cast<h16x4*>(out + 32 + 8 * gq + 4 * hh) = v;
    }
  }
}

}  // namespace

void mt_attn::launch_bwd_q(const mt_half* qkv, const mt_half* dmixed, const float* lse_tot, const float* delta_br, const MtDilatedPlan* plan, void* ws, hipStream_t s) {
  const Plan p = make_plan(plan, 128);
  hipLaunchKernelGGL(dilated_attn_bwd_q_kernel, dim3(p.blk_off[p.nbranch]), dim3(256), 0, s, (const h16*)qkv, (const h16*)dmixed, lse_tot, delta_br, p, (h16*)ws);
}
