// EXPERIMENT, NOT BUILT (round 4): persistent + two waves per SIMD.  Correct (tools/gemm_ps_check.py), but no faster than gemm_ps.hip warm and
// slower on operands from HBM, in both forms tried (12-MFMA clusters / 4 barriers per slice; 24-MFMA clusters / 2 barriers): in the step
// 3072x768: 3.59 ms vs 3.49 (gemm_ps) vs 3.82 (ping-pong); 768x3072: 4.13 vs 3.89 (ping-pong).  tools/experiments/README.md.
// To build it again: copy next to gemm_ps.hip, add to __graft_entry__.SOURCES and call mt_gemm_pq_launch from mt_gemm_nt_f16.
//
// gemm_nt for the backbone shapes (M ~ 30 000 rows, N and K in {768, 2304, 3072}), persistent ping-pong form:
// C[M,N] = A[M,K] . W[N,K]^T (+ bias), fp16 operands, fp32 accumulation, fp16 output (row-major or the head-major q|k|v of MT_EPI_QKV_HM).
//
// Two earlier kernels each solve half of the problem.  gemm.hip's 8-wave ping-pong kernel keeps the matrix pipe fed inside a tile (the
// two waves of a SIMD alternate MFMA clusters and memory phases) but pays 40 % of a K = 768 tile at its boundary: accumulators staged
// through LDS, stores drained, workgroup retired, cold prologue of the next one.  gemm_ps.hip walks a list of tiles with ONE wave per
// SIMD, holds the finished tile in registers and drains it under the next one -- no boundary -- but a wave that is alone on its SIMD
// pays the issue time of every LDS-DMA piece and every back-pressure stall of the memory pipe in MFMA time (stamps: 61 % of the MFMA
// rate warm, worse on operands that come from HBM).  This kernel is both: EIGHT waves, two per SIMD, alternating as in the ping-pong
// kernel; the workgroup is persistent, its operand stream (LDS-DMA into a ring of four 32-deep K-slices) runs across tile boundaries,
// and the finished tile (96 x 64 per wave = 48 registers of packed fp16) leaves under the next tile's memory phases.
//
//   tile 192 x 256; wave (wr, wc) of 2 x 4 owns rows wr * 96 .. + 96, columns wc * 64 .. + 64: 24 accumulator tiles of
//   v_mfma_f32_16x16x32_f16 (96 VGPRs) + the held tile (48) + fragments (6 A + 4 W = 40) + the first slice's C operand (16): 2 waves / SIMD.
//
// One K-slice of one wave = ONE cluster of 24 MFMAs (rows 0-5 x the 4 column tiles, 384 matrix-pipe cycles) and one memory phase:
//     [C(q)]  bar  [MEM: A rows 0-5 and W of slice q + 1 | DMA pieces of slice q + 4 | drain: staged slab out, next slab staged | wait | convert]  bar
// Group 1 (waves 4-7) runs the same program one barrier behind group 0, so on every SIMD one wave is in its cluster while the other is in
// its memory phase: two barriers per slice (the first version alternated 12-MFMA clusters with two memory phases per slice, four
// barriers: 58 % of the MFMA rate, no better than gemm_ps.hip).  Ring protocol (slice q in slot q & 3):
//   * the memory phase that loads slice s ends with `wait`: this wave's pieces of slice s + 1 have landed (younger: its pieces of slices
//     s + 2, s + 3 and the drain stores of this and the previous phase).  Group 1 loads slice q in interval 0 of slice q, group 0 loads
//     slice q + 1 in interval 1: by the barrier between them every wave has waited for slice q + 1, which group 0 then reads first.
//   * the last reads of slot q are group 1's (interval 0 of slice q, retired by lgkmcnt(0) before the barrier): the pieces of slice q + 4
//     -> slot q & 3 are issued by the phases that load slice q + 1 (group 0: interval 1 of slice q) or later.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int PQ_BM = 192, PQ_BN = 256;
constexpr int PQ_AREG = PQ_BM * 64;              // A region of a slot (192 rows x 64 B); the W region (256 rows) follows: piece p at p * 1024
constexpr int PQ_SLOT = PQ_AREG + PQ_BN * 64;    // 28 672 B
constexpr int PQ_STAGE = 4 * PQ_SLOT;            // staging areas (one per wave)
constexpr int PQ_STAGE_ROW = 144;                // 64 halves + 8: 16-byte aligned rows
constexpr int PQ_STAGE_WAVE = 16 * PQ_STAGE_ROW;
constexpr int PQ_BIAS = PQ_STAGE + 8 * PQ_STAGE_WAVE;
constexpr int PQ_LDS = PQ_BIAS + 2 * 1024;       // 135 168 B

struct GemmPqArgs {
  const h16* A; long lda;
  const h16* W;
  const float* bias;       // may be null
  h16* C; long ldc;
  int M, N, K;
  int nbm, nbn;
  int gc;                  // column tiles per group of the tile order (see tile_of)
};

template <int N> MT_DEVINL void pq_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
MT_DEVINL void pq_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// interval boundary: nothing -- not even a register-only MFMA -- may be scheduled across it
MT_DEVINL void pq_bar() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
MT_DEVINL void pq_dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_byte_off, char* smem) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + lds_byte_off), 16, voff, soff, 0, 0);
}
MT_DEVINL __amdgpu_buffer_rsrc_t pq_rsrc(const void* base, unsigned bytes) {
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

template <int EPI, bool HAS_BIAS>
__global__ __launch_bounds__(512) void gemm_nt_pq_kernel(GemmPqArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[PQ_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  using std::integral_constant;

  // ---- tile list (as gemm_ps.hip): XCD x = blockIdx & 7 owns a contiguous range of the row-major tile order, strided over its workgroups
  const int ntiles = g.nbm * g.nbn;
  const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int tq = ntiles / 8, trm = ntiles % 8;
  const int t_begin = xcd * tq + min(xcd, trm), t_count = tq + (xcd < trm ? 1 : 0);
  const int my_tiles = t_count > slot_id ? (t_count - slot_id + nslot - 1) / nslot : 0;
  if (my_tiles == 0) return;
  const int S = g.K >> 5;

  // ---- DMA side.  A slice is 28 pieces of 16 rows x 64 B (A: 0-11, W: 12-27; piece p lands at slot + p * 1024); wave w moves pieces
  // w, w + 8, w + 16 and -- waves 0-3, i.e. group 0 -- w + 24.  lane -> (row lane >> 2, physical chunk lane & 3, logical chunk
  // (lane & 3) ^ (-(lane >> 4) & 3)): see gemm_ps.hip for the swizzle.  Past the end of the list the last tile is fetched again.
  const int prow = lane >> 2, lchunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const unsigned lda2 = (unsigned)g.lda * 2u, K2 = (unsigned)g.K * 2u;
  const __amdgpu_buffer_rsrc_t rsA = pq_rsrc(g.A, (unsigned)g.M * lda2), rsW = pq_rsrc(g.W, (unsigned)g.N * K2),
                               rsB = pq_rsrc(HAS_BIAS ? (const void*)g.bias : (const void*)g.W, (unsigned)g.N * 4u),
                               rsC = pq_rsrc(g.C, (unsigned)g.M * (unsigned)g.N * 2u);
  const bool g0 = wave < 4;                    // (uniform) group 0: four pieces per slice, group 1: three
  unsigned voffA0 = 0, voffA1 = 0;             // per-lane offsets of A pieces w and w + 8 (rows clamped to M - 1)
  const unsigned voffW = (unsigned)prow * K2 + lchunk * 16, voffB = lane * 16;
  unsigned soffW = 0, soffB = 0;               // uniform: the W tile's first row / the bias slice
  int d_tile = 0, d_ks = 0;
  auto dma_tile_setup = [&](int t) {
    const int idx = t_begin + slot_id + min(t, my_tiles - 1) * nslot;
    int mt, nt; tile_of(g, idx, mt, nt);
    voffA0 = (unsigned)min(mt * PQ_BM + wave * 16 + prow, g.M - 1) * lda2 + lchunk * 16;
    voffA1 = (unsigned)min(mt * PQ_BM + (wave + 8) * 16 + prow, g.M - 1) * lda2 + lchunk * 16;
    soffW = (unsigned)(nt * PQ_BN) * K2;
    soffB = (unsigned)(nt * PQ_BN) * 4u;
  };
  auto dma_entry = [&](auto e_c, unsigned slot) {      // entry e of this wave's pieces of the slice at the DMA cursor
    constexpr int e = decltype(e_c)::value;
    const unsigned sb = slot * PQ_SLOT, k0 = (unsigned)d_ks * 64;
    if constexpr (e == 0) pq_dma16(rsA, voffA0, k0, sb + (unsigned)wave * 1024, smem);
    if constexpr (e == 1) {
      if (g0) pq_dma16(rsA, voffA1, k0, sb + (unsigned)(wave + 8) * 1024, smem);
      else pq_dma16(rsW, voffW, soffW + (unsigned)(wave - 4) * 16 * K2 + k0, sb + (unsigned)(wave + 8) * 1024, smem);
    }
    if constexpr (e == 2) pq_dma16(rsW, voffW, soffW + (unsigned)(wave + 4) * 16 * K2 + k0, sb + (unsigned)(wave + 16) * 1024, smem);
    if constexpr (e == 3) { if (g0) pq_dma16(rsW, voffW, soffW + (unsigned)(wave + 12) * 16 * K2 + k0, sb + (unsigned)(wave + 24) * 1024, smem); }
    if constexpr (e == 4) pq_dma16(rsB, voffB, soffB, PQ_BIAS + (unsigned)(d_tile & 1) * 1024, smem);      // 256 floats; every wave writes the same bytes
  };
  auto dma_advance = [&]() { if (++d_ks == S) { d_ks = 0; ++d_tile; dma_tile_setup(d_tile); } };

  // ---- compute side
  const unsigned swz = (unsigned)((fq ^ (0 - (fr >> 2))) & 3) * 16;
  const unsigned a_lane = (unsigned)(wr * 96 + fr) * 64 + swz;
  const unsigned b_lane = PQ_AREG + (unsigned)(wc * 64 + fr) * 64 + swz;
  auto ld_frag = [&](unsigned off) -> h16x8 { return *reinterpret_cast<const h16x8*>(smem + off); };

  f32x4 acc[6][4];
  h16x4 held[6][4];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) held[i][j] = (h16x4){(h16)0.f, (h16)0.f, (h16)0.f, (h16)0.f};
  h16x8 af[6], bf[4];
  f32x4 cinit[4];

  // drain: slab c (16 rows x 64 columns) -> staging (8-byte writes: row fr, column quad fq) -> back as 16-byte chunks (row t * 8 +
  // (lane >> 3), chunk lane & 7) -> global: the uniform row term in the scalar offset, rows >= M dropped by an out-of-range vector offset
  const unsigned st_w = PQ_STAGE + wave * PQ_STAGE_WAVE + fr * PQ_STAGE_ROW + fq * 8;
  const unsigned st_r = PQ_STAGE + wave * PQ_STAGE_WAVE + (lane >> 3) * PQ_STAGE_ROW + (lane & 7) * 16;
  const unsigned row_bytes = EPI == MT_EPI_QKV_HM ? 96u : (unsigned)(g.ldc * 2);
  unsigned held_off = 0;
  int held_rows_left = 0;
  h16x8 dr[2];
  auto drain_write = [&](auto c_c) {
    constexpr int c = decltype(c_c)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<h16x4*>(smem + st_w + j * 32) = held[c][j];
  };
  auto drain_read = [&]() {
    dr[0] = *reinterpret_cast<const h16x8*>(smem + st_r);
    dr[1] = *reinterpret_cast<const h16x8*>(smem + st_r + 8 * PQ_STAGE_ROW);
  };
  auto drain_store = [&](auto c_c) {
    constexpr int c = decltype(c_c)::value;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const unsigned vo = (c * 16 + t * 8 < held_rows_left) ? held_off : 0xffffff00u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, dr[t]), rsC, vo,
                                             (unsigned)(c * 16 + t * 8) * row_bytes, 0);
    }
  };
  auto convert_rows = [&](auto i0_c) {         // accumulator rows i0 .. i0 + 2 -> held
    constexpr int i0 = decltype(i0_c)::value;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        held[i0 + r][j] = (h16x4){(h16)acc[i0 + r][j][0], (h16)acc[i0 + r][j][1], (h16)acc[i0 + r][j][2], (h16)acc[i0 + r][j][3]};
  };

  // ---- prologue: slices 0 .. 3 issued, landed, published; the fragments of slice 0 fetched
  dma_tile_setup(0);
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    dma_entry(integral_constant<int, 0>{}, sl); dma_entry(integral_constant<int, 1>{}, sl);
    dma_entry(integral_constant<int, 2>{}, sl); dma_entry(integral_constant<int, 3>{}, sl);
    if (HAS_BIAS && sl == 0) dma_entry(integral_constant<int, 4>{}, sl);
    dma_advance();
  }
  pq_wait_vmcnt<0>();
  pq_bar();
#pragma unroll
  for (int i = 0; i < 6; ++i) af[i] = ld_frag(a_lane + i * 1024);
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[j] = ld_frag(b_lane + j * 1024);
  if constexpr (HAS_BIAS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) cinit[j] = *reinterpret_cast<const f32x4*>(smem + PQ_BIAS + (wc * 64 + j * 16 + fq * 4) * 4);
  }
  pq_wait_lds();
  if (wr == 1) pq_bar();                       // group 1 runs one interval behind

  unsigned slot_cur = 0, slot_nxt = 1;         // slots of slices q and q + 1; the DMA of slice q + 4 goes to slot_cur (free once slice q is loaded)
  for (int t = 0; t < my_tiles; ++t) {
    // MODE 0 first / 1 middle / 2 last slice of a tile; RS: slab read back + stored (-1 none); WS: slab staged (-1 none); NX: stores + bias
    // pieces younger than the awaited group (on top of the DMA pieces); BIAS: this slice issues the next tile's bias
    auto body = [&](auto mode_c, auto rs_c, auto ws_c, auto nx_c, auto bias_c) {
      constexpr int MODE = decltype(mode_c)::value, RS = decltype(rs_c)::value, WS = decltype(ws_c)::value, NX = decltype(nx_c)::value;
      constexpr bool BIAS = decltype(bias_c)::value != 0;
      const unsigned sn = slot_nxt * PQ_SLOT;
      // ---- C(q): the wave's whole 96 x 64
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], MODE == 0 ? (HAS_BIAS ? cinit[j] : (f32x4){0.f, 0.f, 0.f, 0.f}) : acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      pq_bar();
      // ---- MEM: fragments of slice q + 1; the slab staged one phase ago leaves, the next one is staged; pieces of slice q + 4
#pragma unroll
      for (int i = 0; i < 6; ++i) af[i] = ld_frag(sn + a_lane + i * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = ld_frag(sn + b_lane + j * 1024);
      if constexpr (RS >= 0) {
        drain_read();
        drain_store(integral_constant<int, (RS >= 0 ? RS : 0)>{});
      }
      dma_entry(integral_constant<int, 0>{}, slot_cur);
      dma_entry(integral_constant<int, 1>{}, slot_cur);
      dma_entry(integral_constant<int, 2>{}, slot_cur);
      dma_entry(integral_constant<int, 3>{}, slot_cur);
      if constexpr (HAS_BIAS && BIAS) dma_entry(integral_constant<int, 4>{}, slot_cur);
      dma_advance();
      if constexpr (MODE == 2) {
        convert_rows(integral_constant<int, 0>{});
        convert_rows(integral_constant<int, 3>{});
        if constexpr (HAS_BIAS) {      // the next tile's bias: landed with its first slice (awaited one phase ago, published since)
#pragma unroll
          for (int j = 0; j < 4; ++j) cinit[j] = *reinterpret_cast<const f32x4*>(smem + PQ_BIAS + ((t + 1) & 1) * 1024 + (wc * 64 + j * 16 + fq * 4) * 4);
        }
      }
      if constexpr (WS >= 0) {                 // (after the read-back above: one staging area)
        if constexpr (RS >= 0) pq_wait_lds();
        drain_write(integral_constant<int, (WS >= 0 ? WS : 0)>{});
      }
      // younger than this wave's last piece of slice q + 2: its pieces of slices q + 3 and q + 4 (4 / 3 each), bias piece, drain stores
      if (g0) pq_wait_vmcnt<8 + NX>(); else pq_wait_vmcnt<6 + NX>();
      pq_wait_lds();
      pq_bar();
      slot_cur = slot_nxt; slot_nxt = (slot_nxt + 1) & 3;
    };
    using I0 = integral_constant<int, 0>; using I1 = integral_constant<int, 1>; using I2 = integral_constant<int, 2>; using I4 = integral_constant<int, 4>;
    using No = integral_constant<int, -1>;
    constexpr int NB = HAS_BIAS ? 1 : 0;
    body(I0{}, No{}, I0{}, I0{}, I0{});                                                            // ks = 0: stages slab 0
    body(I1{}, I0{}, I1{}, I2{}, I0{});                                                            // 1: slab 0 out (2 stores), slab 1 staged
    body(I1{}, I1{}, I2{}, I4{}, I0{});                                                            // 2 .. 6: + the previous phase's 2 stores
    body(I1{}, I2{}, integral_constant<int, 3>{}, I4{}, I0{});
    body(I1{}, integral_constant<int, 3>{}, I4{}, I4{}, I0{});
    body(I1{}, I4{}, integral_constant<int, 5>{}, I4{}, I0{});
    body(I1{}, integral_constant<int, 5>{}, No{}, I4{}, I0{});
    body(I1{}, No{}, No{}, I2{}, I0{});                                                            // 7: the stores of slice 6
    for (int ks = 8; ks < S - 4; ++ks) body(I1{}, No{}, No{}, I0{}, I0{});
    body(I1{}, No{}, No{}, integral_constant<int, NB>{}, I1{});                                    // S - 4: issues the next tile's first slice + its bias
    body(I1{}, No{}, No{}, integral_constant<int, NB>{}, I0{});                                    // S - 3: ... which is one of the two younger
    body(I1{}, No{}, No{}, I0{}, I0{});                                                            // S - 2       slices here; awaited at S - 2
    body(I2{}, No{}, No{}, I0{}, I0{});                                                            // S - 1: converts into `held`
    {
      const int idx = t_begin + slot_id + t * nslot;
      int mt, nt; tile_of(g, idx, mt, nt);
      const int m = mt * PQ_BM + wr * 96 + (lane >> 3), n = nt * PQ_BN + wc * 64 + (lane & 7) * 8;
      held_rows_left = g.M - m;
      held_off = EPI == MT_EPI_QKV_HM ? (unsigned)(((n / 48) * g.M + m) * 48 + n % 48) * 2u
                                       : (unsigned)m * (unsigned)(g.ldc * 2) + (unsigned)n * 2u;
    }
  }
  if (wr == 0) pq_bar();                       // group 0 matches group 1's extra barrier
  // ---- the last tile drains in the open
  pq_wait_lds();
#define PQ_DRAIN_OPEN(c) drain_write(integral_constant<int, c>{}); drain_read(); drain_store(integral_constant<int, c>{});
  PQ_DRAIN_OPEN(0) PQ_DRAIN_OPEN(1) PQ_DRAIN_OPEN(2) PQ_DRAIN_OPEN(3) PQ_DRAIN_OPEN(4) PQ_DRAIN_OPEN(5)
#undef PQ_DRAIN_OPEN
}

}  // namespace

// C-ABI-internal entry (called by mt_gemm_nt_f16 in gemm.hip).  Returns MT_OK, or MT_ERR_UNSUPPORTED for what this kernel does not serve.
int mt_gemm_pq_launch(const void* A, long lda, const void* W, int M, int N, int K, int epilogue, const float* bias, void* C, long ldc,
                      hipStream_t s) {
  if (N % 256 != 0 || K % 64 != 0 || K < 768 || M < 256) return MT_ERR_UNSUPPORTED;      // (K >= 768: a tile is >= 24 slices, the drain window is 8)
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
      ncu = 256;
  }
  GemmPqArgs g;
  g.A = (const h16*)A; g.lda = lda; g.W = (const h16*)W; g.bias = bias; g.C = (h16*)C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.nbm = cdiv(M, PQ_BM); g.nbn = N / PQ_BN;
  {
    const char* e = getenv("MT_GEMM_GC");      // experiments: column tiles per group of the tile order (0 / unset: the default below)
    const int want = e ? atoi(e) : 0;
    // default: all columns when there are at most four column tiles; otherwise groups whose W rows (gc * 256 * K * 2 bytes) stay
    // around 1.5 MB -- tools/gemm_gc_sweep.py, operands from HBM: N = 3072, K = 768: 170 us row-major, 153 in groups of 4; N = 2304: 130 / 118
    g.gc = want > 0 ? min(want, g.nbn) : (g.nbn <= 4 ? g.nbn : max(2, min(g.nbn, 3072 / K)));
  }
  const int ntiles = g.nbm * g.nbn;
  const int rounds = cdiv(ntiles, ncu);
  if (2 * ntiles < 3 * ncu || 5L * ntiles < 4L * rounds * ncu) return MT_ERR_UNSUPPORTED;      // (few or badly rounding tiles: gemm.hip's kernels)
  if (ldc != N && epilogue != MT_EPI_QKV_HM) return MT_ERR_UNSUPPORTED;
  if ((long)M * K * 2 >= (1L << 32) || (long)M * N * 2 >= (1L << 32) || (long)N * K * 2 >= (1L << 32)) return MT_ERR_UNSUPPORTED;   // 32-bit byte offsets
  int grid = min(ncu, ntiles);
  grid = max(8, grid / 8 * 8);
  if (epilogue == MT_EPI_QKV_HM) {
    if (!bias) return MT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((gemm_nt_pq_kernel<MT_EPI_QKV_HM, true>), dim3(grid), dim3(512), 0, s, g);
  } else if (bias) {
    hipLaunchKernelGGL((gemm_nt_pq_kernel<MT_EPI_BIAS, true>), dim3(grid), dim3(512), 0, s, g);
  } else {
    hipLaunchKernelGGL((gemm_nt_pq_kernel<MT_EPI_BIAS, false>), dim3(grid), dim3(512), 0, s, g);
  }
  MT_CHECK_LAUNCH();
  return MT_OK;
}
