// EXPERIMENT (not built).  Measured on MI355X at L = 10000:
//   * first form (straight per-sub loop): 9 % SLOWER than the 32-key kernel in attn.hip (1.00 vs 0.92 ms): with one wave
//     per SIMD the compiler-made schedule left the exp/convert block exposed;
//   * this form (software pipeline A(u+1), V(u), C(u-1) over the (sub, key-block) steps, row constants as initial
//     accumulators): on par with the 32-key kernel (10.79 vs 10.75 ms/step; LDS pipe 23 % busy instead of 50 %, MFMA 37 %,
//     VALU 36 %, wave 25 % in issue stalls + 25 % parked) -- the LDS pressure is gone but one wave per SIMD does not
//     overlap its own MFMA and VALU streams well enough; forcing an interleave with sched_group_barrier (-DMT_SGB) spills
//     into AGPR copies and is 1-2 % slower.
// Kept as the starting point for a hand-placed schedule.
// Dilated attention backward, dK/dV kernel with 64 keys per wave (see attn.hip for the algorithm and the 32-key form).
//
// Why a second form: PMC counters on the 32-keys-per-wave kernel (profiles/r01_pmc_attn_bwd.txt) show the LDS pipe as
// its busiest unit (50-70 %), ahead of MFMA (40 %) and VALU (39 %): every wave re-reads the whole 64-query Q / dO tile
// (row reads for S and dP, transposed reads for dV and dK, the per-query L2 / delta vectors) for only 32 keys.  Here a
// wave owns 64 keys = two 32-key column blocks that share every LDS fragment, so LDS bytes per MFMA halve; the four
// accumulator sets live in AGPRs (one wave per SIMD, 512 registers), and the two independent column blocks give the
// scheduler MFMA work to put under the exp / convert block of the other.
#include "attn_common.h"

namespace {

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void dilated_attn_bwd_kv64_kernel(const h16* __restrict__ qkv, const h16* __restrict__ dmixed,
                                  const float* __restrict__ lse_tot, const float* __restrict__ delta_br, Plan p,
                                  h16* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) h16 Qs[2][64 * KSTR];
  __shared__ __attribute__((aligned(16))) h16 Qt[2][64 * VSTR];
  __shared__ __attribute__((aligned(16))) h16 Ds[2][64 * KSTR];
  __shared__ __attribute__((aligned(16))) h16 Dt[2][64 * VSTR];
  __shared__ __attribute__((aligned(16))) float L2s[2][64];
  __shared__ __attribute__((aligned(16))) float Dls[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, l31 = lane & 31;
  const WorkItem w = decode(p, blockIdx.x);
  const Seq sq = make_seq(p, w);
  const long M = (long)p.B * p.N;
  const float c = 0.14433756729740643f * LOG2E;
  const f32x2 c2 = {c, c};
  const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  if (tid < 128) {
    const int buf = tid >> 6, row = tid & 63;
    *reinterpret_cast<h16x8*>(&Qt[buf][row * VSTR + 48]) = zero8; *reinterpret_cast<h16x8*>(&Qt[buf][row * VSTR + 56]) = zero8;
    *reinterpret_cast<h16x8*>(&Dt[buf][row * VSTR + 48]) = zero8; *reinterpret_cast<h16x8*>(&Dt[buf][row * VSTR + 56]) = zero8;
  }

  // this lane's two keys: K^T / V^T fragments (B operands), element j of k-step ks = K[key][16 ks + 8 hh + j]
  int ik[2]; bool kvalid[2];
  h16x8 kf[2][3], vf[2][3];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    ik[kb] = w.qt * 256 + wave * 64 + kb * 32 + l31;
    kvalid[kb] = sq.valid(ik[kb]);
    const long krow = sq.row_clamped(ik[kb]);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      kf[kb][ks] = sel8(kvalid[kb], ldg8(hm_ptr(qkv, M, H + w.h, krow) + ks * 16 + hh * 8));
      vf[kb][ks] = sel8(kvalid[kb], ldg8(hm_ptr(qkv, M, 2 * H + w.h, krow) + ks * 16 + hh * 8));
    }
  }

  const StageIdx st(tid);
  const int ntile = (sq.n + 63) / 64;
  const int nfull = __builtin_amdgcn_readfirstlane(sq.nvalid() >> 6);   // tiles [0, nfull) hold only real rows
  const h16* qbase = hm_ptr(qkv, M, w.h, sq.row(0));
  const h16* dbase = hm_ptr(dmixed, M, w.h, sq.row(0));
  const float* lbase = lse_tot + sq.row(0) * H + w.h;
  const float* dlbase = delta_br + ((long)w.br * M + sq.row(0)) * H + w.h;
  const uint32_t c0 = (uint32_t)(st.row0 * sq.dr * HD + st.part0 * 8) * 2u;
  const uint32_t c1 = st.has1 ? (uint32_t)(st.row1 * sq.dr * HD + st.part1 * 8) * 2u : c0;
  const uint32_t cl = (uint32_t)(lane * sq.dr * H) * 4u;
  h16x8 rq0, rq1, rd0, rd1;
  float rl2 = 0.f, rdl = 0.f;
  bool ok0 = false, ok1 = false, ok2 = false;
  auto gload = [&](int t, auto full_tag) {      // first touched in lstore()
    const int qb = t * 64;
    if (decltype(full_tag)::value) {
      const long adv = (long)qb * sq.dr * HD, advl = (long)qb * sq.dr * H;
      rq0 = ldg8_off(qbase + adv, c0); rd0 = ldg8_off(dbase + adv, c0);
      rq1 = ldg8_off(qbase + adv, c1); rd1 = ldg8_off(dbase + adv, c1);
      rl2 = ldf_off(lbase + advl, cl); rdl = ldf_off(dlbase + advl, cl);
    } else {      // ragged tile: clamped rows, neutralised in lstore()
      const int i0 = qb + st.row0, i1 = qb + st.row1, i2 = qb + lane;
      const long r0 = sq.row_clamped(i0), r1 = sq.row_clamped(i1), r2 = sq.row_clamped(i2);
      rq0 = ldg8(hm_ptr(qkv, M, w.h, r0) + st.part0 * 8); rd0 = ldg8(hm_ptr(dmixed, M, w.h, r0) + st.part0 * 8);
      rq1 = ldg8(hm_ptr(qkv, M, w.h, r1) + st.part1 * 8); rd1 = ldg8(hm_ptr(dmixed, M, w.h, r1) + st.part1 * 8);
      rl2 = lse_tot[r2 * H + w.h]; rdl = delta_br[((long)w.br * M + r2) * H + w.h];
      ok0 = sq.valid(i0); ok1 = sq.valid(i1); ok2 = sq.valid(i2);
    }
  };
  auto lstore = [&](int buf, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const h16x8 q0 = FULL ? rq0 : sel8(ok0, rq0), d0 = FULL ? rd0 : sel8(ok0, rd0);
    *reinterpret_cast<h16x8*>(&Qs[buf][st.row0 * KSTR + st.part0 * 8]) = q0;
    *reinterpret_cast<h16x8*>(&Qt[buf][st.row0 * VSTR + st.part0 * 8]) = q0;
    *reinterpret_cast<h16x8*>(&Ds[buf][st.row0 * KSTR + st.part0 * 8]) = d0;
    *reinterpret_cast<h16x8*>(&Dt[buf][st.row0 * VSTR + st.part0 * 8]) = d0;
    if (st.has1) {
      const h16x8 q1 = FULL ? rq1 : sel8(ok1, rq1), d1 = FULL ? rd1 : sel8(ok1, rd1);
      *reinterpret_cast<h16x8*>(&Qs[buf][st.row1 * KSTR + st.part1 * 8]) = q1;
      *reinterpret_cast<h16x8*>(&Qt[buf][st.row1 * VSTR + st.part1 * 8]) = q1;
      *reinterpret_cast<h16x8*>(&Ds[buf][st.row1 * KSTR + st.part1 * 8]) = d1;
      *reinterpret_cast<h16x8*>(&Dt[buf][st.row1 * VSTR + st.part1 * 8]) = d1;
    }
    if (tid < 64) {      // padded / out-of-range queries contribute nothing: -L2 = -big -> P' = 0
      const bool ok = FULL || ok2;
      L2s[buf][tid] = ok ? fmaf(-rl2, LOG2E, LOG2_SCALE) * (1.0f / c) : -1.0e30f;     // initial accumulator of the S chain
      Dls[buf][tid] = ok ? -rdl : 0.f;
    }
  };

  f32x16 dk0[2], dk1[2], dv0[2], dv1[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk0[kb][i] = 0.f; dk1[kb][i] = 0.f; dv0[kb][i] = 0.f; dv1[kb][i] = 0.f; }
  const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;

  // tile t sits in LDS buffer t & 1; tile t + 1 travels global -> registers during the MFMAs and registers -> the other
  // buffer after them: one barrier per tile.  next_tag: tile t + 1 is a full tile.
  // Software pipeline over the four (sub, kb) steps u of a tile: A(u) = S / dP chains of 32 queries x 32 keys,
  // V(u) = the exp / convert block, C(u) = the dV / dK products.  Program order is A(u+1), V(u), C(u-1): the VALU block
  // of step u has the MFMAs of its two neighbours to hide under (they do not depend on it).
  auto tile = [&](int t, auto next_tag) {
    const int buf = t & 1;
    if (t + 1 < ntile) gload(t + 1, next_tag);
    f32x16 s[2], dp[2];            // two pipeline slots: step u in slot u & 1
    h16x8 pf[2][2], dsf[2][2];     // slot u & 1
    h16x8 qa[3], da[3];
    h16x8 d0[2], d1[2], q0[2], q1[2];
    auto load_rows = [&](int sub) {
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        qa[ks] = *reinterpret_cast<const h16x8*>(&Qs[buf][(sub * 32 + l31) * KSTR + ks * 16 + hh * 8]);
        da[ks] = *reinterpret_cast<const h16x8*>(&Ds[buf][(sub * 32 + l31) * KSTR + ks * 16 + hh * 8]);
      }
    };
    auto load_tr = [&](int sub) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int roff = (sub * 32 + s2 * 16 + 4 * hh + tq) * VSTR + 16 * (grp & 1) + 4 * tp;
        d0[s2] = cat8(lds_tr4(&Dt[buf][roff]), lds_tr4(&Dt[buf][roff + 8 * VSTR]));
        d1[s2] = cat8(lds_tr4(&Dt[buf][roff + 32]), lds_tr4(&Dt[buf][roff + 8 * VSTR + 32]));
        q0[s2] = cat8(lds_tr4(&Qt[buf][roff]), lds_tr4(&Qt[buf][roff + 8 * VSTR]));
        q1[s2] = cat8(lds_tr4(&Qt[buf][roff + 32]), lds_tr4(&Qt[buf][roff + 8 * VSTR + 32]));
      }
    };
    auto stepA = [&](int u) {      // S, dP of step u; row constants ride in as the initial accumulators
      const int sub = u >> 1, kb = u & 1, sl = u & 1;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&L2s[buf][sub * 32 + 8 * g4 + 4 * hh]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&Dls[buf][sub * 32 + 8 * g4 + 4 * hh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[sl][4 * g4 + e] = a[e]; dp[sl][4 * g4 + e] = b[e]; }
      }
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        s[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[ks], kf[kb][ks], s[sl], 0, 0, 0);
        dp[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(da[ks], vf[kb][ks], dp[sl], 0, 0, 0);
      }
    };
    auto stepV = [&](int u) {
      const int sl = u & 1;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const f32x2 pt = pk_exp2((f32x2){s[sl][i], s[sl][i + 1]} * c2);
        const f32x2 d = pt * (f32x2){dp[sl][i], dp[sl][i + 1]};
        pf[sl][i >> 3][i & 7] = (h16)pt[0]; pf[sl][i >> 3][(i & 7) + 1] = (h16)pt[1];
        dsf[sl][i >> 3][i & 7] = (h16)d[0]; dsf[sl][i >> 3][(i & 7) + 1] = (h16)d[1];
      }
    };
    auto stepC = [&](int u) {
      const int kb = u & 1, sl = u & 1;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        dv0[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d0[s2], pf[sl][s2], dv0[kb], 0, 0, 0);
        dv1[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1[s2], pf[sl][s2], dv1[kb], 0, 0, 0);
        dk0[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0[s2], dsf[sl][s2], dk0[kb], 0, 0, 0);
        dk1[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1[s2], dsf[sl][s2], dk1[kb], 0, 0, 0);
      }
    };
    // sub 0
    load_rows(0);
    stepA(0);
    load_tr(0);
    stepA(1); stepV(0);
    load_rows(1);
    stepA(2); stepV(1); stepC(0);
    stepC(1);
    // sub 1 (tr fragments of sub 0 are dead after C(1))
    load_tr(1);
    stepA(3); stepV(2);
    stepV(3); stepC(2);
    stepC(3);
#ifdef MT_SGB
#pragma unroll
    for (int i = 0; i < 56; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
    }
#endif
    if (t + 1 < ntile) lstore(buf ^ 1, next_tag);
    __syncthreads();
  };
  gload(0, std::false_type{});
  lstore(0, std::false_type{});
  __syncthreads();
  int t = 0;
  for (; t + 1 < nfull; ++t) tile(t, std::true_type{});
  for (; t < ntile; ++t) tile(t, std::false_type{});

#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    if (kvalid[kb]) {
      h16* outk = ws + ws_slot(p, w, ik[kb]) + HD;
      h16* outv = outk + HD;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const h16x4 a = {(h16)dk0[kb][4 * gq], (h16)dk0[kb][4 * gq + 1], (h16)dk0[kb][4 * gq + 2], (h16)dk0[kb][4 * gq + 3]};
        const h16x4 b = {(h16)(dv0[kb][4 * gq] * INV_SCALE), (h16)(dv0[kb][4 * gq + 1] * INV_SCALE),
                         (h16)(dv0[kb][4 * gq + 2] * INV_SCALE), (h16)(dv0[kb][4 * gq + 3] * INV_SCALE)};
        *reinterpret_cast<h16x4*>(outk + 8 * gq + 4 * hh) = a;
        *reinterpret_cast<h16x4*>(outv + 8 * gq + 4 * hh) = b;
      }
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const h16x4 a = {(h16)dk1[kb][4 * gq], (h16)dk1[kb][4 * gq + 1], (h16)dk1[kb][4 * gq + 2], (h16)dk1[kb][4 * gq + 3]};
        const h16x4 b = {(h16)(dv1[kb][4 * gq] * INV_SCALE), (h16)(dv1[kb][4 * gq + 1] * INV_SCALE),
                         (h16)(dv1[kb][4 * gq + 2] * INV_SCALE), (h16)(dv1[kb][4 * gq + 3] * INV_SCALE)};
        *reinterpret_cast<h16x4*>(outk + 32 + 8 * gq + 4 * hh) = a;
        *reinterpret_cast<h16x4*>(outv + 32 + 8 * gq + 4 * hh) = b;
      }
    }
  }
}

}  // namespace

// internal launcher used by mt_dilated_attn_bwd (attn.hip)
int mt_launch_bwd_kv64(const void* qkv, const void* dmixed, const float* lse_tot, const float* delta_br,
                       const MtDilatedPlan* plan, void* workspace, hipStream_t s) {
  const Plan p = make_plan(plan, 256);
  const int nblk = p.blk_off[p.nbranch];
  hipLaunchKernelGGL(dilated_attn_bwd_kv64_kernel, dim3(nblk), dim3(256), 0, s, (const h16*)qkv, (const h16*)dmixed, lse_tot,
                     delta_br, p, (h16*)workspace);
  return hipGetLastError() == hipSuccess ? MT_OK : MT_ERR_LAUNCH;
}
