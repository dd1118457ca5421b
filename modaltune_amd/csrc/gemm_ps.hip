// gemm_nt, persistent form for the backbone shapes (M ~ 30 000 rows, N and K in {768, 2304, 3072}): C[M,N] = A[M,K] . W[N,K]^T (+ bias),
// fp16 operands, fp32 accumulation, fp16 output (row-major or the head-major q|k|v of MT_EPI_QKV_HM).
//
// Why a second kernel (gemm.hip holds the 8-wave ping-pong form): at K = 768 a 256 x 256 output tile is only 12 K-tiles deep, and
// the ping-pong kernel's tile boundary -- accumulators staged through LDS, stores drained, workgroup retired, next workgroup's cold
// prologue -- is 40 % of a tile (stamps: prologue 2.5 us, main loop 16.5 us, epilogue 8 us).  Eight waves at 256 registers cannot
// hold a finished tile while the next one accumulates.  This kernel runs ONE wave per SIMD (4 waves, 512 registers each):
//   * output tile 192 x 256, a wave owns 96 x 128 of it: 48 accumulator tiles of v_mfma_f32_16x16x32_f16 = 192 AccVGPRs (the
//     256 x 256 tile was built first: 256 accumulators + the 128 registers of the held tile leave the allocator ~100 registers for
//     fragments, staging and addresses, and hipcc spilled 30-130 of them into the loop);
//   * the workgroup is PERSISTENT: it walks a list of tiles and its operand stream (LDS-DMA into a ring of four 32-deep K-slices)
//     never stops at a tile boundary -- the first slices of tile t + 1 land while tile t finishes;
//   * at the end of a tile the accumulators are converted into 96 VGPRs of packed fp16 (two thirds of that under the last slice's own
//     MFMAs) and the next tile starts at once: its first K-slice's MFMAs take the bias (or zero) as their C operand, so there is no
//     clearing pass and no bias pass; the held tile is DRAINED under the next tile's main loop, one 16-row slab per K-slice:
//     registers -> a private 4 KB LDS staging area (wave-local, no workgroup barrier) -> 16-byte stores of 256-byte row segments;
//   * per K-slice (32 deep): 48 MFMAs (768 matrix-pipe cycles), 14 ds_read_b128, 7 LDS-DMA pieces of 1 KB, ONE workgroup barrier
//     after the first third of the slice, so that the fragments of slice q + 1 are fetched under the rest of slice q.
// LDS: 4 x 28 KB operand ring + 4 x 4352 B staging + 2 x 1 KB bias = 134 144 B (one workgroup per CU).
//
// Synchronisation of the ring (slice q lives in slot q & 3; a slot = A rows 0..191 | W rows 0..255, 64 B per row, 16-byte chunks
// XOR-swizzled by -(row >> 2) & 3 so that every lane group of a ds_read_b128 hits 16 distinct 16-byte slots: see `swz`):
//   after pair 0 of slice q:  s_waitcnt vmcnt(n) -- this wave's DMA pieces of slice q + 1 have landed (issued during slice q - 2)
//                             s_barrier          -- ... and everybody's: slot (q + 1) & 3 is readable; every wave has issued (and, its
//                                                   MFMAs having consumed them, completed) its reads of slice q - 1: slot (q - 1) & 3 is free
//   pairs 1, 2:               DMA pieces of slice q + 3 go to slot (q + 3) & 3 = (q - 1) & 3; fragments of slice q + 1 are read.
// vmcnt is one in-order counter for loads and stores: the wait names the number of YOUNGER operations that may stay in flight (the
// pieces of slice q + 2 -- with the bias piece if that slice opens a tile -- and the drain stores of slices q - 2 and q - 1).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int PS_BM = 192, PS_BN = 256;        // output tile
constexpr int PS_AREG = PS_BM * 64;            // A region of a slot (192 rows x 64 B); the W region (256 rows) follows
constexpr int PS_SLOT = PS_AREG + PS_BN * 64;  // bytes per ring slot (28 672)
constexpr int PS_STAGE = 4 * PS_SLOT;          // staging areas (one per wave)
constexpr int PS_STAGE_ROW = 272;              // bytes per staged row: 128 halves + 8 (16-byte aligned rows, conflict-free 8-byte writes)
constexpr int PS_STAGE_WAVE = 16 * PS_STAGE_ROW;
constexpr int PS_BIAS = PS_STAGE + 4 * PS_STAGE_WAVE;
constexpr int PS_LDS = PS_BIAS + 2 * 1024;

struct GemmPsArgs {
  const h16* A; long lda;
  const h16* W;
  const float* bias;       // may be null
  h16* C; long ldc;
  int M, N, K;
  int nbm, nbn;            // tiles along M / N
  int gc;                  // column tiles per group of the tile order (see tile_of)
};

#ifdef PS_STAMP      // diagnostic build (tools/experiments/gemm_ps_stamp.py): phase durations of the steady-state slice, wave 0 of every workgroup
__device__ unsigned long long* ps_stamp_buf;
MT_DEVINL unsigned long long ps_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PS_T(var) const unsigned long long var = ps_now()
#define PS_ACC(slot, a, b) stamp_acc[slot] += (b) - (a)
#else
#define PS_T(var)
#define PS_ACC(slot, a, b)
#endif

#ifdef PS_GELU_EPI    // TIMING ONLY (tools/experiments): erf-GELU applied where the finished tile is converted -- what folding the FFN's
#define PS_OUT(x) gelu_erf(x)      // activation into fc1 would cost inside this kernel
#else
#define PS_OUT(x) (x)
#endif

template <int N> MT_DEVINL void ps_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
MT_DEVINL void ps_barrier() { asm volatile("s_barrier" ::: "memory"); }
// one LDS-DMA piece: 64 lanes x 16 B from (descriptor base + per-lane voff + uniform soff) to LDS bytes [dst, dst + 1024); lanes whose
// address lies past the descriptor's extent deliver zeros (the rows of the last row tile beyond M)
MT_DEVINL void ps_dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_byte_off, char* smem) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + lds_byte_off), 16, voff, soff, 0, 0);
}
MT_DEVINL __amdgpu_buffer_rsrc_t ps_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// Tile order: column tiles in groups of `gc`; inside a group row-major (row tile, then the group's columns).  The tiles in flight on
// an XCD at one time are a run of this order: with gc * 256 rows of W (gc * K * 512 bytes) instead of all of W they keep hitting the
// XCD's 4 MB L2, and every A row block is still shared by gc workgroups while it is hot.
template <typename Args>
MT_DEVINL void tile_of(const Args& g, int idx, int& mt, int& nt) {
  const int per = g.nbm * g.gc;
  const int cg = idx / per, rem = idx - cg * per;
  const int w = min(g.gc, g.nbn - cg * g.gc);      // (the last group may be narrower)
  mt = rem / w;
  nt = cg * g.gc + (rem - mt * w);
}

// (all operand / output extents are below 4 GiB: byte offsets are 32-bit, the bases live in buffer descriptors / scalar registers)
template <int EPI, bool HAS_BIAS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_nt_ps_kernel(GemmPsArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[PS_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- this workgroup's tile list: XCD x = blockIdx & 7 owns a contiguous range of the row-major tile order; its workgroups take
  // that range strided by their count, so the tiles in flight on an XCD are neighbours (same A rows, neighbouring W rows in its L2)
  const int ntiles = g.nbm * g.nbn;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int tq = ntiles / 8, trm = ntiles % 8;
  const int t_begin = xcd * tq + min(xcd, trm), t_count = tq + (xcd < trm ? 1 : 0);
  const int my_tiles = t_count > slot_id ? (t_count - slot_id + nslot - 1) / nslot : 0;
  if (my_tiles == 0) return;
  const int S = g.K >> 5;                      // K-slices per tile

  // ---- DMA side: the tile being FETCHED runs up to three slices ahead of the tile being computed.
  // piece = 16 rows x 64 B; lane -> (row lane >> 2, physical chunk lane & 3, logical chunk (lane & 3) ^ ((lane >> 4) & 3));
  // wave w moves A pieces 3w .. 3w + 2 and W pieces 4w .. 4w + 3 of every slice.  Past the last slice of the list the last tile is
  // simply fetched again into slots nobody reads any more: no "is there a next slice" branch anywhere.
  const int prow = lane >> 2, lchunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const unsigned lda2 = (unsigned)g.lda * 2u, K2 = (unsigned)g.K * 2u;
  const __amdgpu_buffer_rsrc_t rsA = ps_rsrc(g.A, (unsigned)g.M * lda2), rsW = ps_rsrc(g.W, (unsigned)g.N * K2),
                               rsB = ps_rsrc(HAS_BIAS ? (const void*)g.bias : (const void*)g.W, (unsigned)g.N * 4u),
                               rsC = ps_rsrc(g.C, (unsigned)g.M * (unsigned)g.N * 2u);
  unsigned voffA[3];                           // per-lane A offsets of the tile being fetched (rows clamped to M - 1 in the last row tile)
  const unsigned voffW = (unsigned)prow * K2 + lchunk * 16, voffB = lane * 16;      // W / bias: per-lane constant + uniform tile part
  unsigned soffW = 0, soffB = 0;
  int d_tile = 0, d_ks = 0;                    // tile (index into my list) and slice of the next DMA group
  auto dma_tile_setup = [&](int t) {
    const int idx = t_begin + slot_id + min(t, my_tiles - 1) * nslot;
    int mt, nt; tile_of(g, idx, mt, nt);
#pragma unroll
    for (int p = 0; p < 3; ++p) voffA[p] = (unsigned)min(mt * PS_BM + wave * 48 + p * 16 + prow, g.M - 1) * lda2 + lchunk * 16;
    soffW = (unsigned)(nt * PS_BN + wave * 64) * K2;
    soffB = (unsigned)(nt * PS_BN) * 4u;
  };
  // piece p of the group of slice `qslice`: p 0-2 A, 3-6 W, 7 the tile's bias (256 floats; every wave writes the same bytes)
  auto dma_piece = [&](auto p_c, int qslice) {
    constexpr int p = decltype(p_c)::value;

    const unsigned sb = (unsigned)(qslice & 3) * PS_SLOT, k0 = (unsigned)d_ks * 64;
    if constexpr (p < 3) ps_dma16(rsA, voffA[p], k0, sb + (unsigned)wave * 3072 + p * 1024, smem);
    else if constexpr (p < 7) ps_dma16(rsW, voffW, soffW + (p - 3) * 16 * K2 + k0, sb + PS_AREG + (unsigned)wave * 4096 + (p - 3) * 1024, smem);
    else ps_dma16(rsB, voffB, soffB, PS_BIAS + (unsigned)(d_tile & 1) * 1024, smem);
  };
  auto dma_advance = [&]() { if (++d_ks == S) { d_ks = 0; ++d_tile; dma_tile_setup(d_tile); } };
  using std::integral_constant;
  auto dma_group_all = [&](int qslice) {       // (prologue only: the slices issue their pieces one by one between MFMAs)
    dma_piece(integral_constant<int, 0>{}, qslice); dma_piece(integral_constant<int, 1>{}, qslice);
    dma_piece(integral_constant<int, 2>{}, qslice); dma_piece(integral_constant<int, 3>{}, qslice);
    dma_piece(integral_constant<int, 4>{}, qslice); dma_piece(integral_constant<int, 5>{}, qslice);
    dma_piece(integral_constant<int, 6>{}, qslice);
    if (HAS_BIAS && d_ks == 0) dma_piece(integral_constant<int, 7>{}, qslice);
    dma_advance();
  };

  // ---- compute side
  // physical chunk = logical chunk ^ (-(row >> 2) & 3).  ds_read_b128 serves a wave in four groups of 16 lanes that are NOT runs of
  // consecutive lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: MI355X_MICROARCH.md, LDS): a group holds every fr once, half of
  // them from the next column quad.  With 64-byte rows the 16 reads of a group fall on 16 distinct 16-byte slots of the 256-byte bank
  // window only if the four rows that share (fr & 3) get four different chunks: fq ^ g(fr >> 2) with g = (0, 3, 2, 1) does that for all
  // four groups; the obvious g = identity leaves every group 2-way conflicted.
  const unsigned swz = (unsigned)((fq ^ (0 - (fr >> 2))) & 3) * 16;
  const unsigned a_lane = (unsigned)(wr * 96 + fr) * 64 + swz;
  const unsigned b_lane = PS_AREG + (unsigned)(wc * 128 + fr) * 64 + swz;
  auto ld_frag = [&](unsigned off) -> h16x8 { return *reinterpret_cast<const h16x8*>(smem + off); };

  f32x4 acc[6][8];
  h16x4 held[6][8];                            // the finished tile, packed fp16 (96 VGPRs), drained under the next tile
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) held[i][j] = (h16x4){(h16)0.f, (h16)0.f, (h16)0.f, (h16)0.f};
  // Fragments: ONE set of the wave's eight W fragments (bf) and the A rows in three pairs (afA: rows 0,1; afB: 2,3; afC: 4,5).  Every
  // pair runs column-major -- (r, j), (r + 1, j) for j = 0..7 -- so in the LAST pair bf[j] is dead two MFMAs after it started and is
  // refilled for the next slice at once: it is needed again 14 MFMAs (~220 cycles) later, in the same order.
  h16x8 bf[8], afA[2], afB[2], afC[2];
  f32x4 cinit[8];                              // C operand of a tile's first slice: the bias of the lane's 4 columns per column tile

  // drain of the held tile: slab c (16 rows) -> this wave's staging area (8-byte writes: row fr, column quad fq) -> back as 16-byte
  // row chunks (row t * 4 + fq, chunk fr) -> global; the uniform row term rides in the scalar offset of the buffer store, rows >= M
  // get an out-of-range vector offset (the descriptor drops them): no address arithmetic and no control flow per store.
  const unsigned st_w = PS_STAGE + wave * PS_STAGE_WAVE + fr * PS_STAGE_ROW + fq * 8;
  const unsigned st_r = PS_STAGE + wave * PS_STAGE_WAVE + fq * PS_STAGE_ROW + fr * 16;
  const unsigned row_bytes = EPI == MT_EPI_QKV_HM ? 96u : (unsigned)(g.ldc * 2);
  unsigned held_off = 0;                       // byte offset of (first row of the wave's quarter + fq, the lane's 8 columns)
  int held_rows_left = 0;                      // rows of the wave's quarter that exist (< M), minus fq; 0 before the first handoff
  h16x8 dr[4];
  auto drain_write1 = [&](auto c_c, auto j_c) {
    constexpr int c = decltype(c_c)::value, j = decltype(j_c)::value;
    *reinterpret_cast<h16x4*>(smem + st_w + j * 32) = held[c][j];
  };
  auto drain_read1 = [&](auto t_c) {
    constexpr int t = decltype(t_c)::value;
    dr[t] = *reinterpret_cast<const h16x8*>(smem + st_r + t * 4 * PS_STAGE_ROW);
  };
  auto drain_store1 = [&](auto c_c, auto t_c) {
    constexpr int c = decltype(c_c)::value, t = decltype(t_c)::value;
    const unsigned vo = (c * 16 + t * 4 < held_rows_left) ? held_off : 0xffffff00u;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, dr[t]), rsC, vo,
                                           (unsigned)(c * 16 + t * 4) * row_bytes, 0);
  };
  auto convert_pair = [&](auto i0_c) {         // accumulator rows i0, i0 + 1 -> held (the tile's last slice has finished with them)
    constexpr int i0 = decltype(i0_c)::value;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        held[i0 + r][j] = (h16x4){(h16)PS_OUT(acc[i0 + r][j][0]), (h16)PS_OUT(acc[i0 + r][j][1]), (h16)PS_OUT(acc[i0 + r][j][2]), (h16)PS_OUT(acc[i0 + r][j][3])};
  };

  // ---- prologue: slices 0, 1, 2 in flight; slice 0 awaited and published; its first fragments fetched
  dma_tile_setup(0);
  dma_group_all(0);
  dma_group_all(1);
  dma_group_all(2);
  ps_wait_vmcnt<0>();                          // (cold start: all three; the steady state waits are counted)
  ps_barrier();
#pragma unroll
  for (int j = 0; j < 7; ++j) bf[j] = ld_frag(b_lane + j * 1024);       // (bf[7] is fetched by the slice itself)
  afA[0] = ld_frag(a_lane); afA[1] = ld_frag(a_lane + 1024);

#ifdef PS_STAMP
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long k_begin = ps_now();
#endif
  int q = 0;
  for (int t = 0; t < my_tiles; ++t) {
    // One K-slice = 48 MFMAs in three row pairs (rows 2p, 2p + 1), each pair column-major; EVERY other instruction of the slice rides
    // in the gap behind one MFMA, at most one memory operation per gap, in an order pinned by scheduling fences: at one wave per SIMD
    // nothing else covers an instruction's issue time, and left to itself hipcc clusters the seven LDS-DMA pieces behind the barrier
    // (stamps: 460 cycles of a 1670-cycle slice, all four waves queueing at the CU's one texture-address unit at once).
    //   MODE 0: first slice of a tile (C operand = bias / zero), 1: middle, 2: last (rows are converted as their pair finishes).
    //   DRAIN >= 0: slab of the held tile drained under it.  NWAIT: vector-memory operations younger than the DMA group awaited at the
    //   top (the group of slice q + 2 and the drain stores of slice q - 1).  BIASGRP: the group issued here opens a tile.
    auto slice = [&](auto mode_c, auto drain_c, auto nwait_c, auto biasgrp_c) {
      constexpr int MODE = decltype(mode_c)::value, DRAIN = decltype(drain_c)::value, NWAIT = decltype(nwait_c)::value;
      constexpr bool BIASGRP = decltype(biasgrp_c)::value != 0;
      constexpr int DC = DRAIN >= 0 ? DRAIN : 0;
      const unsigned sa = (unsigned)(q & 3) * PS_SLOT, sn = (unsigned)((q + 1) & 3) * PS_SLOT;
      PS_T(s0);
      // ---- publish slice q + 1 (its fragments are read below), free slot q - 1 (the DMA group of slice q + 3 goes there)
      __builtin_amdgcn_sched_barrier(0);
      ps_wait_vmcnt<NWAIT>();
      PS_T(s1);
      ps_barrier();
      __builtin_amdgcn_sched_barrier(0);
      PS_T(s2);
#pragma unroll
      for (int n = 0; n < 48; ++n) {
        const int pr = n >> 4, k = (n & 15) >> 1, r = n & 1, i = 2 * pr + r;
        const h16x8& af = pr == 0 ? afA[r] : pr == 1 ? afB[r] : afC[r];
        acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[k], af, MODE == 0 ? (HAS_BIAS ? cinit[k] : (f32x4){0.f, 0.f, 0.f, 0.f}) : acc[i][k], 0, 0, 0);
        // -- fragments of this slice
        if (n == 0) bf[7] = ld_frag(sa + b_lane + 7 * 1024);
        if (n == 1) afB[0] = ld_frag(sa + a_lane + 2 * 1024);
        if (n == 2) afB[1] = ld_frag(sa + a_lane + 3 * 1024);
        if (n == 8) afC[0] = ld_frag(sa + a_lane + 4 * 1024);
        if (n == 9) afC[1] = ld_frag(sa + a_lane + 5 * 1024);
        // -- fragments of the next slice: rows 0, 1 once pair 0 is through (n = 15), W column k once pair 2 is through with it
        if (n == 16) afA[0] = ld_frag(sn + a_lane);
        if (n == 17) afA[1] = ld_frag(sn + a_lane + 1024);
        if (n >= 33 && n <= 45 && (n & 1)) bf[(n - 33) >> 1] = ld_frag(sn + b_lane + ((n - 33) >> 1) * 1024);
        // -- the DMA group of slice q + 3, one piece every six or seven MFMAs
#ifndef PS_NO_DMA     // (-DPS_NO_DMA: TIMING ONLY, wrong results -- what the in-loop LDS-DMA pieces cost)
        if (n == 3) dma_piece(integral_constant<int, 0>{}, q + 3);
        if (n == 10) dma_piece(integral_constant<int, 1>{}, q + 3);
        if (n == 18) dma_piece(integral_constant<int, 2>{}, q + 3);
        if (n == 24) dma_piece(integral_constant<int, 3>{}, q + 3);
        if (n == 30) dma_piece(integral_constant<int, 4>{}, q + 3);
        if (n == 36) dma_piece(integral_constant<int, 5>{}, q + 3);
        if (n == 42) dma_piece(integral_constant<int, 6>{}, q + 3);
        if constexpr (HAS_BIAS && BIASGRP) { if (n == 46) dma_piece(integral_constant<int, 7>{}, q + 3); }
#endif
        // -- the slab of the held tile: to staging, back as row chunks, out
        if constexpr (DRAIN >= 0) {
          if (n == 4) drain_write1(integral_constant<int, DC>{}, integral_constant<int, 0>{});
          if (n == 5) drain_write1(integral_constant<int, DC>{}, integral_constant<int, 1>{});
          if (n == 6) drain_write1(integral_constant<int, DC>{}, integral_constant<int, 2>{});
          if (n == 7) drain_write1(integral_constant<int, DC>{}, integral_constant<int, 3>{});
          if (n == 11) drain_write1(integral_constant<int, DC>{}, integral_constant<int, 4>{});
          if (n == 12) drain_write1(integral_constant<int, DC>{}, integral_constant<int, 5>{});
          if (n == 13) drain_write1(integral_constant<int, DC>{}, integral_constant<int, 6>{});
          if (n == 14) drain_write1(integral_constant<int, DC>{}, integral_constant<int, 7>{});
          if (n == 19) drain_read1(integral_constant<int, 0>{});
          if (n == 20) drain_read1(integral_constant<int, 1>{});
          if (n == 21) drain_read1(integral_constant<int, 2>{});
          if (n == 22) drain_read1(integral_constant<int, 3>{});
          if (n == 26) drain_store1(integral_constant<int, DC>{}, integral_constant<int, 0>{});
          if (n == 27) drain_store1(integral_constant<int, DC>{}, integral_constant<int, 1>{});
          if (n == 28) drain_store1(integral_constant<int, DC>{}, integral_constant<int, 2>{});
          if (n == 29) drain_store1(integral_constant<int, DC>{}, integral_constant<int, 3>{});
        }
        // -- last slice of the tile: one finished accumulator tile per gap becomes packed fp16 (rows 0-3 here, rows 4, 5 below)
        if constexpr (MODE == 2) {
          if (n >= 16) {
            const int ci = (n - 16) >> 3, cj = (n - 16) & 7;
            held[ci][cj] = (h16x4){(h16)PS_OUT(acc[ci][cj][0]), (h16)PS_OUT(acc[ci][cj][1]), (h16)PS_OUT(acc[ci][cj][2]), (h16)PS_OUT(acc[ci][cj][3])};
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      dma_advance();
      if constexpr (MODE == 2) convert_pair(integral_constant<int, 4>{});
      PS_T(s3);
      if (MODE == 1 && DRAIN < 0 && !BIASGRP) { PS_ACC(0, s0, s1); PS_ACC(1, s1, s2); PS_ACC(2, s2, s3); PS_ACC(5, s3 - 1, s3); }
      ++q;
    };
    using I0 = integral_constant<int, 0>; using I1 = integral_constant<int, 1>; using I2 = integral_constant<int, 2>;
    using NoDrain = integral_constant<int, -1>;
    constexpr int G = 7, GB = HAS_BIAS ? 8 : 7;
    // the first slice's C operand: the tile's bias (landed with the tile's first DMA group, published one slice ago)
    if constexpr (HAS_BIAS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) cinit[j] = *reinterpret_cast<const f32x4*>(smem + PS_BIAS + (t & 1) * 1024 + (wc * 128 + j * 16 + fq * 4) * 4);
    }
    slice(I0{}, NoDrain{}, integral_constant<int, G>{}, I0{});                                    // ks = 0
    slice(I1{}, integral_constant<int, 0>{}, integral_constant<int, G>{}, I0{});                  // 1
    slice(I1{}, integral_constant<int, 1>{}, integral_constant<int, G + 4>{}, I0{});              // 2 .. 7: + the stores of the slice before
    slice(I1{}, integral_constant<int, 2>{}, integral_constant<int, G + 4>{}, I0{});
    slice(I1{}, integral_constant<int, 3>{}, integral_constant<int, G + 4>{}, I0{});
    slice(I1{}, integral_constant<int, 4>{}, integral_constant<int, G + 4>{}, I0{});
    slice(I1{}, integral_constant<int, 5>{}, integral_constant<int, G + 4>{}, I0{});
    slice(I1{}, NoDrain{}, integral_constant<int, G + 4>{}, I0{});                                // 7
    for (int ks = 8; ks < S - 3; ++ks) slice(I1{}, NoDrain{}, integral_constant<int, G>{}, I0{});
    slice(I1{}, NoDrain{}, integral_constant<int, G>{}, I1{});                                    // S - 3: issues the next tile's first group (+ bias)
    slice(I1{}, NoDrain{}, integral_constant<int, GB>{}, I0{});                                   // S - 2: ... which is the younger group here
    __builtin_amdgcn_sched_barrier(0);
    slice(I2{}, NoDrain{}, integral_constant<int, G>{}, I0{});                                    // S - 1: converts its rows into `held`
    // (Scheduling fences around the handoff: the next tile's first MFMAs do not read the old accumulators, so nothing ties them to
    // the conversions -- left free, the scheduler starts them above the conversions and the allocator has to find 192 more registers.)
    {
      const int idx = t_begin + slot_id + t * nslot;
      int mt, nt; tile_of(g, idx, mt, nt);
      const int m = mt * PS_BM + wr * 96 + fq, n = nt * PS_BN + wc * 128 + fr * 8;
      held_rows_left = g.M - m;
      held_off = EPI == MT_EPI_QKV_HM ? (unsigned)(((n / 48) * g.M + m) * 48 + n % 48) * 2u
                                       : (unsigned)m * (unsigned)(g.ldc * 2) + (unsigned)n * 2u;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#ifdef PS_STAMP
  if (tid == 0) {
    const unsigned long long k_end = ps_now();
    unsigned long long* o = ps_stamp_buf + blockIdx.x * 8;
    for (int i = 0; i < 6; ++i) o[i] = stamp_acc[i];
    o[6] = k_end - k_begin; o[7] = my_tiles;
  }
#endif
  // ---- the last tile drains in the open
#define PS_DRAIN_OPEN(c)                                                                                                      \
  drain_write1(integral_constant<int, c>{}, integral_constant<int, 0>{}); drain_write1(integral_constant<int, c>{}, integral_constant<int, 1>{}); \
  drain_write1(integral_constant<int, c>{}, integral_constant<int, 2>{}); drain_write1(integral_constant<int, c>{}, integral_constant<int, 3>{}); \
  drain_write1(integral_constant<int, c>{}, integral_constant<int, 4>{}); drain_write1(integral_constant<int, c>{}, integral_constant<int, 5>{}); \
  drain_write1(integral_constant<int, c>{}, integral_constant<int, 6>{}); drain_write1(integral_constant<int, c>{}, integral_constant<int, 7>{}); \
  drain_read1(integral_constant<int, 0>{}); drain_read1(integral_constant<int, 1>{}); drain_read1(integral_constant<int, 2>{});       \
  drain_read1(integral_constant<int, 3>{});                                                                                            \
  drain_store1(integral_constant<int, c>{}, integral_constant<int, 0>{}); drain_store1(integral_constant<int, c>{}, integral_constant<int, 1>{}); \
  drain_store1(integral_constant<int, c>{}, integral_constant<int, 2>{}); drain_store1(integral_constant<int, c>{}, integral_constant<int, 3>{});
  PS_DRAIN_OPEN(0) PS_DRAIN_OPEN(1) PS_DRAIN_OPEN(2) PS_DRAIN_OPEN(3) PS_DRAIN_OPEN(4) PS_DRAIN_OPEN(5)
#undef PS_DRAIN_OPEN
  // The DMA groups issued by the last three slices (the last tile "fetched again into slots nobody reads") are still in flight here.
  // A wave must not end under them: when the workgroup is gone its LDS goes to the next workgroup the CU admits -- with two pass groups
  // on two streams that is a kernel of the OTHER stream, at once -- and the late 1 KB pieces land in ITS image (round 6: found as a
  // once-in-30-steps corruption of a 3 / 4 KB run of the prompt self-attention's score rows = one wave's A / W pieces; a single stream
  // hides it, the next kernel starts only behind the end-of-kernel flush).
  ps_wait_vmcnt<0>();
}

}  // namespace

int mt_gemm_ps_launch(const void* A, long lda, const void* W, int M, int N, int K, int epilogue, const float* bias, void* C, long ldc, hipStream_t s);
#ifdef PS_STAMP
extern "C" int mt_gemm_ps_set_stamps(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(ps_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
extern "C" int mt_gemm_ps_stamp_launch(const void* A, long lda, const void* W, int M, int N, int K, const float* bias, void* C, long ldc) {
  return mt_gemm_ps_launch(A, lda, W, M, N, K, MT_EPI_BIAS, bias, C, ldc, nullptr);
}
#endif

// C-ABI-internal entry (called by mt_gemm_nt_f16 in gemm.hip for the shapes this kernel serves).  Returns MT_OK or a negative status.
int mt_gemm_ps_launch(const void* A, long lda, const void* W, int M, int N, int K, int epilogue, const float* bias, void* C, long ldc,
                      hipStream_t s) {
  if (N % 256 != 0 || K % 64 != 0 || K < 768 || M < 256) return MT_ERR_UNSUPPORTED;      // (K >= 768: a tile is at least 24 slices, the drain window is 9)
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
      ncu = 256;
  }
  GemmPsArgs g;
  g.A = (const h16*)A; g.lda = lda; g.W = (const h16*)W; g.bias = bias; g.C = (h16*)C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.nbm = cdiv(M, PS_BM); g.nbn = N / PS_BN;
  {
    const char* e = getenv("MT_GEMM_GC");      // experiments: column tiles per group of the tile order (0 / unset: the default below)
    const int want = e ? atoi(e) : 0;
    // default: all columns when there are at most four column tiles; otherwise groups whose W rows (gc * 256 * K * 2 bytes) stay
    // around 1.5 MB -- tools/gemm_gc_sweep.py, operands from HBM: N = 3072, K = 768: 170 us row-major, 153 in groups of 4; N = 2304: 130 / 118
    g.gc = want > 0 ? min(want, g.nbn) : (g.nbn <= 4 ? g.nbn : max(2, min(g.nbn, 3072 / K)));
  }
  const int ntiles = g.nbm * g.nbn;
  // Worth it only when every CU walks at least a tile and a half on average and the last round is not mostly idle (the tiles are of
  // equal size and statically assigned): M = 30 003 gives 1884 / 1413 / 471 tiles for N = 3072 / 2304 / 768 = 92 % of 8 / 6 / 2 rounds.
  const int rounds = cdiv(ntiles, ncu);
  // (a SINGLE round is not worth it either: the cold prologue and the drain in the open are not amortised -- round 6, M = 12 291:
  // 768 x 768 on 195 tiles 37.6 us against 28.2 us for the 128 x 128 kernel, N = 768 / K = 2304 86.8 against 62.6 for the ping-pong kernel)
  static const char* force = getenv("MT_GEMM_FORCE");      // experiments (tools/gemm_candidates.py): "ps" lifts the fill rule, anything else declines
  if (force && force[0] && strcmp(force, "ps") != 0) return MT_ERR_UNSUPPORTED;
  // (the fill bound: 75 % -- tools/gemm_candidates.py, round 6: at 76-77 % fill this kernel still leads the others, M = 8 194 / N = 2304:
  // 37.7 us against 46.8 (ping-pong) and 43.5 (128 x 128); M = 6 147 / N = 3072: 38.5 against 47.6 / 42.7; at 67 % it trails, M = 8 194 / N = 3072)
  if (!(force && force[0]) && (2 * ntiles < 3 * ncu || 4L * ntiles < 3L * rounds * ncu)) return MT_ERR_UNSUPPORTED;
  if (ldc != N && epilogue != MT_EPI_QKV_HM) return MT_ERR_UNSUPPORTED;      // (the store descriptor spans M * N contiguous halves)
  // K = 3072 (fc2, dX of fc1): the A operand is 184 MB that the previous kernel has just written; with operands from HBM the issue of
  // the LDS-DMA pieces itself backs up (stamps: the slice body 1000 -> 1560 cycles, the vmcnt wait unchanged) and a wave that is alone on
  // its SIMD has nobody to run MFMAs meanwhile.  tools/gemm_cold_bench.py: 175 vs 161 us cold, 124 vs 133 warm; in the step: 3.82 vs 3.83 ms
  // for the 24 launches.  A five-slot ring (DMA four slices ahead, all 160 KB of LDS) changed neither.  Left to the ping-pong kernel.
  if (K > 2304 && !(force && force[0])) return MT_ERR_UNSUPPORTED;
  int grid = min(ncu, ntiles);
  grid = max(8, grid / 8 * 8);
  // 32-bit byte offsets: the A descriptor's extent and the per-lane A offsets are formed from lda (a strided A -- lda > K, e.g. a column
  // slice of a wider buffer -- must not wrap them either)
  if (lda < K || (long)M * lda * 2 >= (1L << 32) || (long)M * N * 2 >= (1L << 32) || (long)N * K * 2 >= (1L << 32)) return MT_ERR_UNSUPPORTED;
  if (epilogue == MT_EPI_QKV_HM) {
    if (!bias || N % 48 != 0) return MT_ERR_UNSUPPORTED;      // (head-major layout: N = 3 x heads x 48; the caller's fallback reports BAD_ARG)
    hipLaunchKernelGGL((gemm_nt_ps_kernel<MT_EPI_QKV_HM, true>), dim3(grid), dim3(256), 0, s, g);
  } else if (bias) {
    hipLaunchKernelGGL((gemm_nt_ps_kernel<MT_EPI_BIAS, true>), dim3(grid), dim3(256), 0, s, g);
  } else {
    hipLaunchKernelGGL((gemm_nt_ps_kernel<MT_EPI_BIAS, false>), dim3(grid), dim3(256), 0, s, g);
  }
  MT_CHECK_LAUNCH();
  return MT_OK;
}
