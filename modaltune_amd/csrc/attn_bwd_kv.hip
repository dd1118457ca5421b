// Dilated attention (LongNet), backward: the dK / dV kernel.
// (see attn.hip for the reference semantics and the forward; split out so that the translation unit can carry its own
// LLVM scheduling strategy -- attn_common.h)
#include "attn_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// backward, kernel KV: dK, dV.  One workgroup = 128 keys (key = lane) of one (pass, branch, segment, head),
// sweeping the queries of the same sparse sequence in tiles of 64.
//   S[q,key]  = Q . K^T            Q rows from LDS, K^T in registers
//   dP[q,key] = dO . V^T           dO rows from LDS, V^T in registers
//   P' = exp2(S' - L2[q] + log2(ln 2)) ; dS = P' (dP - delta[q])          (S' = Q' . K^T, q pre-scaled; P' = ln2 P~)
//   dV^T[d,key] += dO^T[d,q] . P'  (rescaled by 1 / ln 2 once at the end) ; dK^T[d,key] += Q'^T[d,q] . dS
// LDS holds -L2 + log2(ln 2) and -delta per query; they are read straight into the S / dP accumulators before the
// MFMA chains, and q carries the softmax scale, so the elementwise block is 1 packed mul, 2 exp and 2 packed
// converts per element pair.
// ------------------------------------------------------------------------------------------------
// (launch bound: 3 waves per SIMD = 168 VGPRs; the prefetched fragments would otherwise push the kernel to 175 and a wave per SIMD less)
__global__ __launch_bounds__(256, 3) void dilated_attn_bwd_kv_kernel(const h16* __restrict__ qkv, const h16* __restrict__ dmixed,
                                                                  const float* __restrict__ lse_tot, const float* __restrict__ delta_br,
                                                                  Plan p, h16* __restrict__ ws) {
  // Q and dO tiles in LDS-DMA images (attn_common.h: img_off: each read both by rows and transposed), double-buffered
  // together with the per-query constants; one barrier per tile
  __shared__ __attribute__((aligned(16))) h16 smem[4 * IMG_HALVES];      // Q0 | Q1 | D0 | D1
  __shared__ __attribute__((aligned(16))) float L2s[2][64];
  __shared__ __attribute__((aligned(16))) float Dls[2][64];
  h16* const Qx = smem;
  h16* const Dx = smem + 2 * IMG_HALVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const WorkItem w = decode(p, blockIdx.x);
  const Seq sq = make_seq(p, w);
  const long M = (long)p.B * p.N;
  const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // padded keys get no gradient; padded queries read as Q = dO = 0 (range check of the DMA descriptor) and add nothing
  const int nv = __builtin_amdgcn_readfirstlane(sq.nvalid());
  if (w.qt * 128 >= nv) return;

  {   // the zero columns d = 48..63 (logical chunks 6, 7) of all four images, written once: 4 x 64 x 2 chunks, two per thread
    const int img = tid >> 6, row = tid & 63;
    *reinterpret_cast<h16x8*>(&smem[img * IMG_HALVES + img_off(row, 6)]) = zero8;
    *reinterpret_cast<h16x8*>(&smem[img * IMG_HALVES + img_off(row, 7)]) = zero8;
  }

  // this lane's key: K^T / V^T fragments (B operands), element j of k-step ks = K[key][16 ks + 8 hh + j]
  const int ik = w.qt * 128 + wave * 32 + l31;
  const bool kvalid = sq.valid(ik);
  const long krow = sq.row_clamped(ik);
  h16x8 kf[3], vf[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    kf[ks] = sel8(kvalid, ldg8(hm_ptr(qkv, M, H + w.h, krow) + ks * 16 + hh * 8));
    vf[ks] = sel8(kvalid, ldg8(hm_ptr(qkv, M, 2 * H + w.h, krow) + ks * 16 + hh * 8));
  }

  const int nvq = min(nv, p.qlimit[w.br]);   // entries that act as queries (sequence-parallel plans: a prefix)
  const int ntile = (nvq + 63) >> 6;     // tiles holding at least one real query
  const int row_bytes = sq.dr * HD * 2;
  const long valid_bytes = (long)(nvq - 1) * row_bytes + HD * 2;
  const long tile_bytes = 64L * row_bytes;
  const h16* const qseq = hm_ptr(qkv, M, w.h, sq.row(0));
  const h16* const dseq = hm_ptr(dmixed, M, w.h, sq.row(0));
  const DmaLane dl(tid, row_bytes);
  // per-query constants of a tile: wave 0 loads them (one query per lane), neutral values past the end of the sequence
  const float* const lbase = lse_tot + sq.row(0) * H + w.h;
  const float* const dbase = delta_br + ((long)w.br * M + sq.row(0)) * H + w.h;
  float rl2 = 0.f, rdl = 0.f;
  auto issue = [&](int t) {
    dma_tile(Qx + (t & 1) * IMG_HALVES, tile_rsrc(qseq, t * tile_bytes, valid_bytes), dl);
    dma_tile(Dx + (t & 1) * IMG_HALVES, tile_rsrc(dseq, t * tile_bytes, valid_bytes), dl);
    if (tid < 64) {      // RAW loads only: the arithmetic on them waits in publish(), at the END of the tile (a use here parks wave 0
      const long off = (long)min(t * 64 + lane, nvq - 1) * sq.dr * H;      // on s_waitcnt vmcnt(0) -- the DMA just issued included)
      rl2 = lbase[off];
      rdl = dbase[off];
    }
  };
  auto publish = [&](int t) {      // constants of tile t into their buffer (written by wave 0, read after the barrier)
    if (tid < 64) {
      const bool ok = t * 64 + lane < nvq;      // (Q = dO = 0 past the end: P' is multiplied by zeros)
      L2s[t & 1][tid] = ok ? fmaf(-rl2, LOG2E, LOG2_LN2) : 0.f;
      Dls[t & 1][tid] = ok ? -rdl : 0.f;
    }
  };

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk0[i] = 0.f; dk1[i] = 0.f; dv0[i] = 0.f; dv1[i] = 0.f; }
  const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  // per-lane offsets into the images: row reads of chunk 2 ks + hh, transposed reads of rows 4 hh + tq (a) and + 8 (b),
  // column blocks d 0..31 (0) and 32..63 (1); the sub / s2 row-block offsets are multiples of 16 rows (img_f unchanged)
  int rrd[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) rrd[ks] = img_off(l31, 2 * ks + hh);
  const int trc = 2 * (grp & 1) + (tp >> 1), tro = 4 * (tp & 1);
  const int tr_a0 = img_off(4 * hh + tq, trc) + tro, tr_a1 = img_off(4 * hh + tq, trc + 4) + tro;
  const int tr_b0 = img_off(4 * hh + tq + 8, trc) + tro, tr_b1 = img_off(4 * hh + tq + 8, trc + 4) + tro;

  issue(0);
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
      // row constants ride in as the INITIAL accumulators (rows of the accumulators are queries:
      // row(i) = (i&3) + 8 (i>>2) + 4 hh): S' - L2 + log2(ln 2), dP' = dO.V^T - delta
      f32x16 s, dp;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&L2b[sub * 32 + 8 * g4 + 4 * hh]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&Dlb[sub * 32 + 8 * g4 + 4 * hh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[4 * g4 + e] = a[e]; dp[4 * g4 + e] = b[e]; }
      }
      // row fragments requested ahead of the products, three deep (see the dQ kernel); the full barrier keeps the constants'
      // reads above out of the read / MFMA groups
      __builtin_amdgcn_sched_barrier(0);
      h16x8 qa[3], da[3];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        qa[ks] = *reinterpret_cast<const h16x8*>(&Qb[sub * 32 * IMG_ROW + rrd[ks]]);
        da[ks] = *reinterpret_cast<const h16x8*>(&Db[sub * 32 * IMG_ROW + rrd[ks]]);
      }
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[ks], kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(da[ks], vf[ks], dp, 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);
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
        const int rb = (sub * 32 + s2 * 16) * IMG_ROW;      // rows rb + 4 hh + tq and + 8; cols d 0..31 / 32..63
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
    dma_wait_all();            // tile t + 1 has landed ...
    __syncthreads();           // ... for everybody, and everybody has left tile t
  }
  if (kvalid) {
    h16* outk = ws + ws_slot(p, w, krow) + ws_which_stride(p, w.br);
    h16* outv = outk + ws_which_stride(p, w.br);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const h16x4 a = {(h16)dk0[4 * gq], (h16)dk0[4 * gq + 1], (h16)dk0[4 * gq + 2], (h16)dk0[4 * gq + 3]};
      const h16x4 b = {(h16)(dv0[4 * gq] * INV_LN2), (h16)(dv0[4 * gq + 1] * INV_LN2), (h16)(dv0[4 * gq + 2] * INV_LN2),
                       (h16)(dv0[4 * gq + 3] * INV_LN2)};
      *reinterpret_cast<h16x4*>(outk + 8 * gq + 4 * hh) = a;
      *reinterpret_cast<h16x4*>(outv + 8 * gq + 4 * hh) = b;
    }
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const h16x4 a = {(h16)dk1[4 * gq], (h16)dk1[4 * gq + 1], (h16)dk1[4 * gq + 2], (h16)dk1[4 * gq + 3]};
      const h16x4 b = {(h16)(dv1[4 * gq] * INV_LN2), (h16)(dv1[4 * gq + 1] * INV_LN2), (h16)(dv1[4 * gq + 2] * INV_LN2),
                       (h16)(dv1[4 * gq + 3] * INV_LN2)};
      *reinterpret_cast<h16x4*>(outk + 32 + 8 * gq + 4 * hh) = a;
      *reinterpret_cast<h16x4*>(outv + 32 + 8 * gq + 4 * hh) = b;
    }
  }
}

}  // namespace

void mt_attn::launch_bwd_kv(const mt_half* qkv, const mt_half* dmixed, const float* lse_tot, const float* delta_br, const MtDilatedPlan* plan, void* ws, hipStream_t s) {
  const Plan p = make_plan(plan, 128);
  hipLaunchKernelGGL(dilated_attn_bwd_kv_kernel, dim3(p.blk_off[p.nbranch]), dim3(256), 0, s, (const h16*)qkv, (const h16*)dmixed, lse_tot, delta_br, p, (h16*)ws);
}
