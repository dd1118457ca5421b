// GEMM kernels of the Modal-Adapter hot path (gfx950, fp16 operands, fp32 MFMA accumulation).
//
//   gemm_nt : C[M,N] = epi(A[M,K] . W[N,K]^T)      every big-M nn.Linear forward and every dX GEMM
//   gemm_tn : C[N1,N2] += A[M,N1]^T . B[M,N2]      weight gradients of the trainable big-M linears
//   colsum  : out[N]  += sum_m A[m,N]              bias gradients
//   sgemm_small : strided fp32 GEMM for the token-side (T <= 66 rows) ops
//
// gemm_nt structure: 128 x BN x 64 tiles, 4 waves, v_mfma_f32_16x16x32_f16, double-buffered LDS filled by LDS-DMA
// (global_load_lds_dwordx4, issued for tile t+1 before the MFMAs of tile t; one barrier per K-tile), XOR-swizzled
// 128-B rows so the ds_read_b128 fragment reads are bank-conflict free.
#include <stdlib.h>
#include <string.h>

#include "common.h"

// gemm_ps.hip: persistent form for the backbone shapes (returns MT_ERR_UNSUPPORTED for shapes it does not serve)
int mt_gemm_ps_launch(const void* A, long lda, const void* W, int M, int N, int K, int epilogue, const float* bias, void* C, long ldc, hipStream_t s);

namespace {

constexpr int BK = 64;

#ifdef MT_GEMM_STAMP      // diagnostic build (tools/experiments): phase durations of the ping-pong kernel, summed over workgroups
MT_DEVINL unsigned long long stamp_now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define MT_STAMP(var) const unsigned long long var = stamp_now()
#define MT_STAMP_ADD(g, slot, a, b) do { if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(const_cast<float*>((g).pos_table)) + (slot), (b) - (a)); } while (0)
#else
#define MT_STAMP(var)
#define MT_STAMP_ADD(g, slot, a, b)
#endif

struct GemmNtArgs {
  const h16* A; long lda; RowMap amap;
  const h16* W;
  int M, N, K;
  int m_begin, m_end;     // row range of this launch (tile rows start at m_begin; rows >= m_end are not stored)
  const float* bias;
  const float* resid; long ldr; RowMap rmap;
  const float* colscale;
  const float* pos_table; const int* pos_row; const int* pos_col;
  void* C; long ldc; RowMap cmap;
  DropArgs drop;          // BIAS_RESID: C = resid + drop(acc + bias)
};

// LDS-DMA staging (global_load_lds_dwordx4): each wave-instruction drops 64 x 16 B = 1 KiB = 8 tile rows of 128 B
// straight into LDS, no staging VGPRs and no ds_write pass.  The DMA destination is lane-linear, so the
// bank-conflict fix is an XOR swizzle applied on the per-lane SOURCE address and again on the fragment reads
// (guide rule 21): the 16-B chunk c of tile row r lives at chunk position c ^ (r & 7); with it the 16-lane groups
// of ds_read_b128 hit 16 distinct 16-B slots.
constexpr int erows_for(int bm, int cs, int lds_bytes) {
  int e = bm;
  while (e * cs * 4 > lds_bytes) e /= 2;
  return e;
}

// ---- epilogue (shared by the tile kernels).  The MFMAs were issued with the operands swapped (W fragment as A,
// activation fragment as B), so each accumulator is a TRANSPOSED 16x16 tile: lane = (m = lane & 15, n = 4 * (lane >> 4)
// .. + 3), i.e. four consecutive output columns of one row per lane -> one 16-byte LDS write per tile instead of four
// scalar ones.  The tile is staged through LDS (the operand buffers are free now) so that every global access of the
// epilogue -- the residual read and the C write -- is a full-width row segment (16 B per lane, BN * 4 B per row).
// fp16 outputs with a plain epilogue (bias, or bias + head-major q|k|v): the tile is converted in registers and staged as
// fp16 -- half the LDS bytes of the fp32 staging, so a 256 x 256 tile takes two passes instead of four -- and leaves as
// 16-byte row segments (half the store instructions of the 8-byte form: the store tail is issue-bound, guide T21).
// Staging rows of BN + 4 halves: the 8-byte writes of a 16-lane group (16 rows, one column quad) land on 16 distinct
// bank pairs; read back as two ds_read_b64 per thread (rows are only 8-byte aligned).
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
constexpr int erows_h16(int bm, int csh, int lds_bytes) {
  int e = bm;
  while (e * csh * 2 > lds_bytes) e /= 2;
  return e;
}
template <int BM, int BN, int WM, int WN, int EPI, int LDS_BYTES>
MT_DEVINL void gemm_epilogue_h16(const GemmNtArgs& g, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], h16* smem, int m0, int n0) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  constexpr int CSH = BN + 4;
  constexpr int EROWS = erows_h16(BM, CSH, LDS_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  h16* C = reinterpret_cast<h16*>(g.C);
  f32x4 b4[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn * TN + j * 16 + fq * 4;
    b4[j] = (g.bias && n < g.N) ? *reinterpret_cast<const f32x4*>(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  constexpr int CPR = BN / 8;                                  // 16-byte chunks per tile row
  constexpr int RPP = NT / CPR;                                // rows per sweep
  const int cc = (tid % CPR) * 8, r0 = tid / CPR;
  const int n = n0 + cc;
#pragma unroll
  for (int pass = 0; pass < BM / EROWS; ++pass) {
    const int rbase = pass * EROWS;
    MT_STAMP(e0);
    if (pass > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * TM + i * 16 + fr - rbase;
      if (row >= 0 && row < EROWS) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const f32x4 v = acc[i][j] + b4[j];
          *reinterpret_cast<h16x4*>(&smem[row * CSH + wn * TN + j * 16 + fq * 4]) = (h16x4){(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
        }
      }
    }
    __syncthreads();
    MT_STAMP(e1);
#pragma unroll 4
    for (int rr = r0; rr < EROWS; rr += RPP) {
      const int m = m0 + rbase + rr;
      if (m >= g.m_end || n >= g.N) continue;
      const h16x4 lo = *reinterpret_cast<const h16x4*>(&smem[rr * CSH + cc]), hi = *reinterpret_cast<const h16x4*>(&smem[rr * CSH + cc + 4]);
      h16* dst = EPI == MT_EPI_QKV_HM ? C + ((long)(n / 48) * g.M + m) * 48 + (n % 48)      // 8 | 48: chunks never straddle a head
                                      : C + g.cmap.map(m) * g.ldc + n;
      *reinterpret_cast<h16x8*>(dst) = (h16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    MT_STAMP(e2);
    MT_STAMP_ADD(g, 2, e0, e1);
    MT_STAMP_ADD(g, 3, e1, e2);
  }
}

template <int BM, int BN, int WM, int WN, int EPI, typename OutT, int LDS_BYTES>
MT_DEVINL void gemm_epilogue(const GemmNtArgs& g, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], h16* smem, int m0, int n0) {
  if constexpr (sizeof(OutT) == 2 && (EPI == MT_EPI_BIAS || EPI == MT_EPI_QKV_HM)) {
    if ((g.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(g.C) & 15) == 0) {      // 16-byte row segments need aligned rows
      gemm_epilogue_h16<BM, BN, WM, WN, EPI, LDS_BYTES>(g, acc, smem, m0, n0);
      return;
    }
  }
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  constexpr int CS = BN + 4;
  constexpr int EROWS = erows_for(BM, CS, LDS_BYTES);
  static_assert(EROWS * CS * 4 <= LDS_BYTES, "epilogue staging must fit the operand LDS");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  float* Cs = reinterpret_cast<float*>(smem);
  OutT* C = reinterpret_cast<OutT*>(g.C);
  constexpr int CPR = BN / 4;                                  // float4 chunks per tile row
  constexpr int RPP = NT / CPR;                                // rows per pass
  const int cc = (tid % CPR) * 4, r0 = tid / CPR;
  const int n = n0 + cc;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, gm4 = {0.f, 0.f, 0.f, 0.f};
  if (n < g.N) {
    if (g.bias) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);
    if (EPI == MT_EPI_INJECT) gm4 = *reinterpret_cast<const f32x4*>(g.colscale + n);
  }
  // DropPath factors of the (at most two) task passes this tile touches, once per thread
  const bool dropping = EPI == MT_EPI_BIAS_RESID && g.drop.active();
  const int dpass0 = dropping ? m0 / g.drop.rows_per_pass : 0;
  const int dsplit = (dpass0 + 1) * g.drop.rows_per_pass;        // first row of the next pass
  const bool dfast = dropping && g.drop.rows_per_pass >= BM;
  const float dpf0 = dfast ? drop_path_factor(g.drop, dpass0) : 1.f, dpf1 = dfast ? drop_path_factor(g.drop, dpass0 + 1) : 1.f;
#pragma unroll
  for (int pass = 0; pass < BM / EROWS; ++pass) {
    const int rbase = pass * EROWS;
    if (pass > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * TM + i * 16 + fr - rbase;
      if (row >= 0 && row < EROWS) {
#pragma unroll
        for (int j = 0; j < NI; ++j) *reinterpret_cast<f32x4*>(&Cs[row * CS + wn * TN + j * 16 + fq * 4]) = acc[i][j];
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int rr = r0; rr < EROWS; rr += RPP) {
      const int m = m0 + rbase + rr;
      if (m >= g.m_end || n >= g.N) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[rr * CS + cc]);
      v += bias4;
      if (EPI == MT_EPI_BIAS_RESID) {
        if (dropping)       // dropout / DropPath of the branch
          v *= dfast ? drop_elem4(g.drop, ((uint64_t)m * g.N + n) >> 2, m < dsplit ? dpf0 : dpf1)
                     : drop_scale4(g.drop, ((uint64_t)m * g.N + n) >> 2, m);
        v += *reinterpret_cast<const f32x4*>(g.resid + g.rmap.map(m) * g.ldr + n);
      }
      if (EPI == MT_EPI_INJECT) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(g.resid + g.rmap.map(m) * g.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (1.0f + gm4[e]) * x[e] + gm4[e] * v[e];
      }
      if (EPI == MT_EPI_POSEMB) {
        const int half = g.N >> 1;   // first half encodes the grid column, second half the row (A.8)
        const float* tab = (n < half) ? g.pos_table + (long)g.pos_col[m] * half + n : g.pos_table + (long)g.pos_row[m] * half + (n - half);
        v += *reinterpret_cast<const f32x4*>(tab);
      }
      OutT* dst = C + g.cmap.map(m) * g.ldc + n;
      if (EPI == MT_EPI_QKV_HM)     // [q|k|v][head][M][48]: column n -> slab n / 48, offset n % 48 (4 | 48: chunks never straddle)
        dst = C + ((long)(n / 48) * g.M + m) * 48 + (n % 48);
      if constexpr (sizeof(OutT) == 4) {
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
        *reinterpret_cast<h16x4*>(dst) = (h16x4){(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
      }
    }
  }
}

// Tile choice: 128 x {128, 64} with 4 waves (two workgroups per CU) for the narrow / shallow shapes, and the 256 x 256
// ping-pong kernel below (8 waves, 128 KB of LDS, one workgroup per CU) for N >= 2304 or K >= 2304.  A K-tile of the 128^2
// form moves 32 KB for 1024 MFMA cycles per SIMD, i.e. it wants ~134 GB/s per CU from L2 at full MFMA rate against
// ~70 GB/s deliverable (MI355X_MICROARCH.md, gather into LDS from L2), and it ends every K-tile on a vmcnt(0) + barrier.
template <int BM, int BN, int WM, int WN, int EPI, typename OutT>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(GemmNtArgs g) {
  constexpr int NT = WM * WN * 64;              // threads
  constexpr int TM = BM / WM, TN = BN / WN;     // wave tile
  constexpr int MI = TM / 16, NI = TN / 16;     // 16x16 MFMA tiles per wave
  constexpr int ACH = BM * 8 / NT, BCH = BN * 8 / NT;  // 16-B chunks per thread per K-tile
  constexpr int CS = BN + 4;                    // fp32 epilogue staging stride (floats)
  constexpr int EROWS = erows_for(BM, CS, 2 * (BM + BN) * BK * 2);   // epilogue rows per pass
  static_assert(EROWS * CS * 4 <= 2 * (BM + BN) * BK * 2, "epilogue staging must fit the operand LDS");
  __shared__ __attribute__((aligned(16))) h16 smem[2 * (BM + BN) * BK];
  h16* const As0 = smem;
  h16* const Bs0 = smem + 2 * BM * BK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: consecutive logical tiles share an A row-panel; keep them on one XCD (T1).
  const int nbn = (g.N + BN - 1) / BN;
  const int nbm = (g.m_end - g.m_begin + BM - 1) / BM;
  const int nwg = nbn * nbm;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, x = bid % 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
  }
  const int m0 = g.m_begin + (bid / nbn) * BM, n0 = (bid % nbn) * BN;

  // per-thread DMA sources: chunk id c = i * 256 + tid -> tile row c >> 3, physical chunk c & 7, logical chunk
  // (c & 7) ^ (row & 7); rows past the edge are clamped (their products are never stored)
  const int srow = tid >> 3, lchunk = (tid & 7) ^ (srow & 7);
  const h16* aptr[ACH];
  const h16* bptr[BCH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int m = min(m0 + i * (NT / 8) + srow, g.M - 1);
    aptr[i] = g.A + g.amap.map(m) * g.lda + lchunk * 8;
  }
#pragma unroll
  for (int i = 0; i < BCH; ++i) {
    const int n = min(n0 + i * (NT / 8) + srow, g.N - 1);
    bptr[i] = g.W + (long)n * g.K + lchunk * 8;
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < ACH; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(aptr[i] + k0),
                                       (__attribute__((address_space(3))) void*)(As0 + buf * BM * BK + (i * NT + wave * 64) * 8), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < BCH; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bptr[i] + k0),
                                       (__attribute__((address_space(3))) void*)(Bs0 + buf * BN * BK + (i * NT + wave * 64) * 8), 16, 0, 0);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = fr & 7;                         // row & 7 of every fragment row this lane reads
  stage(0, 0);
  __syncthreads();                               // (hipcc drains the LDS-DMA with vmcnt(0) before the barrier)
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) stage(buf ^ 1, (t + 1) * BK);
    const h16* As = As0 + buf * BM * BK;
    const h16* Bs = Bs0 + buf * BN * BK;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      const int pc = ((kk * 4 + fq) ^ sw) * 8;
      h16x8 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const h16x8*>(&As[(wm * TM + i * 16 + fr) * BK + pc]);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const h16x8*>(&Bs[(wn * TN + j * 16 + fr) * BK + pc]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile: see epilogue
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
  }

  gemm_epilogue<BM, BN, WM, WN, EPI, OutT, 2 * (BM + BN) * BK * 2>(g, acc, smem, m0, n0);
}

// ------------------------------------------------------------------------------------------------
// gemm_nt, 256 x 256 ping-pong form (8 waves, wave tile 128 x 64, one workgroup per CU, 128 KB LDS).
//
// The two waves of a SIMD (wave i of row-group 0, wave i + 4 of row-group 1) alternate roles in lock step: while one
// runs a CLUSTER of 16 MFMAs (one 64 x 32 quadrant of its output tile x a 64-deep K-tile), the other pulls the next
// fragments out of LDS and issues LDS-DMA for a later K-tile; a raw s_barrier ends every interval and group 1 runs one
// interval behind group 0.  Four intervals of each kind per K-tile:
//   p0: read A(rows 0-63) + B(cols 0-31)   issue DMA A-half 0 of tile kt+1     MFMA quadrant (0,0)
//   p1: read B(cols 32-63)                 issue DMA A-half 1 of tile kt+1     MFMA quadrant (0,1)
//   p2: read A(rows 64-127)                issue DMA B-half 0 of tile kt+2     MFMA quadrant (1,1)
//   p3: -                                  issue DMA B-half 1 of tile kt+2     MFMA quadrant (1,0)
// LDS holds two K-tiles of four 16-KB half-tiles (A rows 0-127 / 128-255, B cols 0-127 / 128-255).  Fragments are in
// registers after p0-p2, so a half-tile slot is refilled long before the tile that reuses the buffer is read: the DMA
// runs 1-2 K-tiles ahead with only two buffers and is awaited ONCE per K-tile with a counted s_waitcnt vmcnt(4) in p3
// (the two B half-tiles just issued stay in flight).
//   RAW: a wave's vmcnt in p3 of tile kt retires its pieces of tile kt+1 (A) and earlier; group 1 does the same one
//        interval later, and group 0 first reads tile kt+1 two barriers after its own wait = one barrier after group 1's.
//   WAR: every ds_read is retired (lgkmcnt(0)) before the barrier that closes its interval; B slots of buffer d are
//        last read in p1 of tile kt (group 1: during group 0's MFMA p1) and refilled from p2; A slots last read in p2
//        and refilled in p0/p1 of the next tile.
// ------------------------------------------------------------------------------------------------
template <int N> MT_DEVINL void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
MT_DEVINL void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// RH = rows per A half-tile = rows per wave group: 128 (256 x 256 tile) or 64 (128 x 256 tile for the last partial round)
template <int RH, int EPI, typename OutT>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(GemmNtArgs g) {
  constexpr int BM = 2 * RH, BN = 256, WM = 2, WN = 4;
  constexpr int AH = RH * BK, HALF = 128 * BK;   // halves per A / B half-tile
  constexpr int TILE = 2 * AH + 2 * HALF;        // A0 | A1 | B0 | B1
  constexpr int AP = RH / 64;                    // DMA pieces per thread per A half-tile (B: 2)
  constexpr int RT = RH / 32;                    // 16-row fragment tiles per A quadrant
  __shared__ __attribute__((aligned(16))) h16 smem[2 * TILE];

  MT_STAMP(t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int nbn = (g.N + BN - 1) / BN, nbm = (g.m_end - g.m_begin + BM - 1) / BM, nwg = nbn * nbm;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, x = bid % 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / 8;
  }
  const int m0 = g.m_begin + (bid / nbn) * BM, n0 = (bid % nbn) * BN;

  // DMA pieces: 64 rows x 8 chunks of 16 B per piece (one per thread): row = tid >> 3, physical chunk tid & 7, logical
  // chunk (tid & 7) ^ (row & 7); an A half-tile is AP pieces, a B half-tile 2
  const int srow = tid >> 3, lchunk = (tid & 7) ^ (srow & 7);
  const h16* asrc[2][AP];                        // [half][piece]
  const h16* bsrc[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = min(m0 + h * RH + i * 64 + srow, g.M - 1);
      asrc[h][i] = g.A + g.amap.map(m) * g.lda + lchunk * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = min(n0 + h * 128 + i * 64 + srow, g.N - 1);
      bsrc[h][i] = g.W + (long)n * g.K + lchunk * 8;
    }
  }
  auto dma_a = [&](int h, int slot_halves, int k0) {
#pragma unroll
    for (int i = 0; i < AP; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[h][i] + k0),
                                       (__attribute__((address_space(3))) void*)(smem + slot_halves + (i * 512 + wave * 64) * 8), 16, 0, 0);
  };
  auto dma_b = [&](int h, int slot_halves, int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[h][i] + k0),
                                       (__attribute__((address_space(3))) void*)(smem + slot_halves + (i * 512 + wave * 64) * 8), 16, 0, 0);
  };

  f32x4 acc[2 * RT][4];
#pragma unroll
  for (int i = 0; i < 2 * RT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
  const int a_off = wr * AH;                                     // this wave's A half-tile inside a K-tile buffer
  const int b_off = 2 * AH + (wc >> 1) * HALF + (wc & 1) * 64 * BK;     // its 64 B columns inside the B half-tile

  h16x8 af[RT][2], bf[2][2][2];                  // af[row tile][k step]; bf[qn][col tile][k step]
  auto read_a = [&](const h16* buf, int qm) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        af[i][kk] = *reinterpret_cast<const h16x8*>(&buf[a_off + (qm * (RH / 2) + i * 16 + fr) * BK + (((kk * 4 + fq) ^ sw) * 8)]);
  };
  auto read_b = [&](const h16* buf, int qn) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        bf[qn][j][kk] = *reinterpret_cast<const h16x8*>(&buf[b_off + (qn * 32 + j * 16 + fr) * BK + (((kk * 4 + fq) ^ sw) * 8)]);
  };
  auto cluster = [&](int qm, int qn) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[qm * RT + i][qn * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[qn][j][kk], af[i][kk], acc[qm * RT + i][qn * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto bar = [&]() { __builtin_amdgcn_s_barrier(); };

  // prologue: tile 0 complete, B halves of tile 1 in flight
  constexpr int B0 = 2 * AH, B1 = 2 * AH + HALF;                 // slots inside a K-tile buffer
  dma_a(0, 0, 0); dma_a(1, AH, 0); dma_b(0, B0, 0); dma_b(1, B1, 0);
  if (nk > 1) { dma_b(0, TILE + B0, BK); dma_b(1, TILE + B1, BK); wait_vmcnt<4>(); }
  else wait_vmcnt<0>();
  bar();
  MT_STAMP(t1);
  if (wr == 1) bar();                            // group 1 runs one interval behind

  for (int kt = 0; kt < nk; ++kt) {
    const h16* buf = smem + (kt & 1) * TILE;
    const int nb = ((kt + 1) & 1) * TILE;        // buffer of tile kt+1; tile kt+2 reuses this tile's buffer
    const int cb = (kt & 1) * TILE;
    const bool has1 = kt + 1 < nk, has2 = kt + 2 < nk;
    // p0
    read_b(buf, 0); read_a(buf, 0);
    if (has1) dma_a(0, nb, (kt + 1) * BK);
    wait_lds(); bar();
    cluster(0, 0); bar();
    // p1
    read_b(buf, 1);
    if (has1) dma_a(1, nb + AH, (kt + 1) * BK);
    wait_lds(); bar();
    cluster(0, 1); bar();
    // p2
    read_a(buf, 1);
    if (has2) dma_b(0, cb + B0, (kt + 2) * BK);
    wait_lds(); bar();
    cluster(1, 1); bar();
    // p3: await tile kt+1 (its A halves; its B halves are older), leave tile kt+2's B halves in flight
    if (has2) { dma_b(1, cb + B1, (kt + 2) * BK); wait_vmcnt<4>(); }
    else wait_vmcnt<0>();
    bar();
    cluster(1, 0); bar();
  }
  if (wr == 0) bar();                            // group 0 matches group 1's extra barrier
  __syncthreads();
  MT_STAMP(t2);
  MT_STAMP_ADD(g, 0, t0, t1);
  MT_STAMP_ADD(g, 1, t1, t2);
  gemm_epilogue<BM, BN, WM, WN, EPI, OutT, 2 * TILE * 2>(g, acc, smem, m0, n0);
  MT_STAMP(t3);
  MT_STAMP_ADD(g, 4, t2, t3);
  MT_STAMP_ADD(g, 5, t3 - 1, t3);
}

template <int BN, int EPI, typename OutT>
int launch_nt(const GemmNtArgs& a, hipStream_t s) {
  if constexpr (BN == 256) {
    // Ping-pong 256 x 256 tiles, one workgroup per CU.  When the tile count leaves a short last round, the rows of that
    // round are covered by 128 x 256 tiles instead (half the time per tile): e.g. N = 768, M = 30003 is 354 tiles on
    // 256 CUs = 2 rounds; 255 big + 195 small tiles finish in 1.5.
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
        ncu = 256;
    }
    // Which rows get 256-row tiles: the cheapest of {all big, whole rounds of big tiles + one round of small ones, all small} under
    // cost(round of big tiles) = 10, cost(round of small tiles) = 6 (a 128 x 256 tile takes 57 us at K = 3072, a 256 x 256 one 97).
    // All-small matters below half a round of big tiles: N = 768, K = 3072 at M = 9 219 is 108 big tiles on 256 CUs -- 72.7 us whatever
    // M is -- or 219 small ones (round 6; hipBLASLt: 50 us).
    const int nbn = cdiv(a.N, 256), nrt = cdiv(a.M, 256);
    int big_rt = nrt;                                     // row tiles covered by 256-row tiles
    long best = 10L * cdiv(nrt * nbn, ncu);
    if (nrt * nbn > ncu) {
      const int full = (nrt * nbn) / ncu;                 // whole rounds of big tiles
      const int rt = (full * ncu) / nbn;
      const int rest_rows = a.M - rt * 256;
      if (rest_rows > 0 && cdiv(rest_rows, 128) * nbn <= ncu && 10L * full + 6 < best) { best = 10L * full + 6; big_rt = rt; }
    }
    if (6L * cdiv(cdiv(a.M, 128) * nbn, ncu) < best) big_rt = 0;
    static const char* force = getenv("MT_GEMM_FORCE");    // experiments (tools/gemm_candidates.py)
    if (force && !strcmp(force, "pp_big")) big_rt = nrt;
    if (force && !strcmp(force, "pp_small")) big_rt = 0;
    GemmNtArgs g1 = a;
    g1.m_begin = 0; g1.m_end = min(a.M, big_rt * 256);
    if (big_rt > 0) hipLaunchKernelGGL((gemm_nt_pp_kernel<128, EPI, OutT>), dim3(big_rt * nbn), dim3(512), 0, s, g1);
    if (g1.m_end < a.M) {
      GemmNtArgs g2 = a;
      g2.m_begin = g1.m_end; g2.m_end = a.M;
      hipLaunchKernelGGL((gemm_nt_pp_kernel<64, EPI, OutT>), dim3(cdiv(a.M - g1.m_end, 128) * nbn), dim3(512), 0, s, g2);
    }
    MT_CHECK_LAUNCH();
    return MT_OK;
  } else {
    constexpr int BM = 128;
    const int nwg = cdiv(a.M, BM) * cdiv(a.N, BN);
    if (BN == 128)
      hipLaunchKernelGGL((gemm_nt_kernel<BM, 128, 2, 2, EPI, OutT>), dim3(nwg), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<BM, 64, 4, 1, EPI, OutT>), dim3(nwg), dim3(256), 0, s, a);
    MT_CHECK_LAUNCH();
    return MT_OK;
  }
}

template <int EPI, typename OutT>
int launch_nt_bn(const GemmNtArgs& a, hipStream_t s) {
  // N = 192 / 384 / 576 (adapter projections) tile exactly with BN = 64; everything else uses 128
  if (a.N % 128 != 0) return launch_nt<64, EPI, OutT>(a, s);
  static const bool pp_ok = getenv("MT_GEMM_NOPP") == nullptr;
  static const char* force = getenv("MT_GEMM_FORCE");      // experiments (tools/gemm_candidates.py)
  if (force && !strncmp(force, "pp", 2) && a.N % 256 == 0) return launch_nt<256, EPI, OutT>(a, s);
  if (force && !strcmp(force, "k128")) return launch_nt<128, EPI, OutT>(a, s);
  const long tiles256 = (long)cdiv(a.M, 256) * (a.N / 256);
  if (pp_ok && a.N % 256 == 0 && a.M >= 8192 && (a.N >= 2304 || a.K >= 2304)) return launch_nt<256, EPI, OutT>(a, s);
  // (N = 768 / K >= 2304 at M = 8 194 ... 10 922: alone and warm the 128 x 128 kernel leads the ping-pong kernel's 128-row tiles by 6-9 %
  // -- tools/gemm_candidates.py -- but INSIDE the step, next to the other pass group's kernels, sending M = 10 001 there cost 0.15-0.25 ms
  // at 10 000 patches, same box, two pairs: not done)
  // A wide product below the M >= 8 192 gate whose 256-row tiles fill three quarters of a round: M = 4 097 / N = 3072: 28.0 us
  // against 32.2 on 128 x 128 tiles, M = 6 147 / N = 2304: 28.2 against 33.3 (in the step at 4 096 patches: 18.33 -> 18.27 ms)
  if (pp_ok && a.N % 256 == 0 && a.M >= 4096 && a.M < 8192 && a.N >= 2304 && tiles256 <= 256 && 4 * tiles256 >= 3 * 256)
    return launch_nt<256, EPI, OutT>(a, s);
  // the small square shape (N = K = 768: attention output projection and its dX) where 256-row tiles fill their rounds: M = 18 435: 32.9 us
  // against 38.6 for the 128 x 128 kernel, M = 20 002 (the B = 2 pass group at 10 000 patches): 33.3 against 36.3; at half-filled rounds
  // (M = 10 001: 26.7 against 22.4) the small tiles stay (tools/gemm_candidates.py, round 6)
  if (pp_ok && a.N % 256 == 0 && a.M >= 8192) {
    const long tiles = (long)cdiv(a.M, 256) * (a.N / 256), ncu = 256;
    if (5 * tiles >= 4 * cdiv(tiles, ncu) * ncu) return launch_nt<256, EPI, OutT>(a, s);
  }
  return launch_nt<128, EPI, OutT>(a, s);
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: C[N1,N2] += sum_m A[m,n1] B[m,n2]  (+ colsum[n1] += sum_m A[m,n1], the bias gradient, on the same pass).
// TM x TN output tile per workgroup (64 / 128 / 192 on each side: the adapter's 192 / 384 / 768 divide exactly), 32 rows of m
// per step, split over M with fp32 atomics.  Both operands are "k-strided" in memory, so the tiles are staged as they lie
// ([m][n]) and the MFMA fragments come from ds_read_b64_tr_b16 (hardware transposed read).
// What bounds it at M = 30 000 is traffic, not the 9-18 GFLOP: the M-split's atomics (split x N1 x N2 x 4 bytes at the
// ~1.3 TB/s memory-side atomic rate), the operand re-reads (A: N2 / TN times, B: N1 / TM times) and how many DISTINCT bytes
// are in flight (tools/experiments/README.md, round 2).  Workgroup id -> (XCD = id % 8, slot = id / 8): the tiles of one row
// range run on one XCD and share its L2, which is worth 20-30 % over the tile-major order at every shape of the step.
// ------------------------------------------------------------------------------------------------
MT_DEVINL h16x4 lds_tr4(const h16* p) {
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)p);
  return __builtin_bit_cast(h16x4, r);
}

struct GemmTnArgs {
  const h16* A; long lda; RowMap amap;
  const h16* B; long ldb; RowMap bmap;
  int M, N1, N2, rows_per_split;
  float* C; long ldc;
  float* colsum;      // optional [N1]
};

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs g) {
  constexpr int SA = TM + 8, SB = TN + 8;       // halves per LDS row (+16 B: 8-byte aligned tr reads, 16-B aligned writes)
  constexpr int CA = TM / 64, CB = TN / 64;     // 16-byte chunks per thread per step (32 rows x T/8 chunks over 256 threads)
  constexpr int MI = TM / 32, NI = TN / 32;     // 16 x 16 MFMA tiles per wave (wave tile TM/2 x TN/2)
  __shared__ __attribute__((aligned(16))) h16 As[2][32 * SA];
  __shared__ __attribute__((aligned(16))) h16 Bs[2][32 * SB];
  __shared__ float csum[TM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_x = g.N1 / TM, tiles = tiles_x * (g.N2 / TN);
  const int slot = blockIdx.x >> 3, tile = slot % tiles, split_id = (slot / tiles) * 8 + (blockIdx.x & 7);
  const int n1_0 = (tile % tiles_x) * TM, n2_0 = (tile / tiles_x) * TN;
  const int mbeg = split_id * g.rows_per_split;
  const int mend = min(g.M, mbeg + g.rows_per_split);
  if (mbeg >= mend) return;
  const bool want_cs = g.colsum != nullptr && n2_0 == 0;
  // staging: chunk c = i * 256 + tid of a 32-row tile: row c / (T / 8), chunk column c % (T / 8)
  h16x8 ra[CA], rb[CB];
  float cs[CA][8];
#pragma unroll
  for (int i = 0; i < CA; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[i][e] = 0.f;
  const h16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  auto gload = [&](int mt) {      // branch-free: clamp the row, select zero past the end of the split
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = i * 256 + tid, row = c / (TM / 8), kc = c % (TM / 8);
      const int m = mt + row, mc = min(m, mend - 1);
      const h16x8 a = ldg8(g.A + g.amap.map(mc) * g.lda + n1_0 + kc * 8);
      ra[i] = m < mend ? a : zero;
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int c = i * 256 + tid, row = c / (TN / 8), kc = c % (TN / 8);
      const int m = mt + row, mc = min(m, mend - 1);
      const h16x8 b = ldg8(g.B + g.bmap.map(mc) * g.ldb + n2_0 + kc * 8);
      rb[i] = m < mend ? b : zero;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = i * 256 + tid, row = c / (TM / 8), kc = c % (TM / 8);
      *reinterpret_cast<h16x8*>(&As[buf][row * SA + kc * 8]) = ra[i];
      if (want_cs) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[i][e] += (float)ra[i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int c = i * 256 + tid, row = c / (TN / 8), kc = c % (TN / 8);
      *reinterpret_cast<h16x8*>(&Bs[buf][row * SB + kc * 8]) = rb[i];
    }
  };
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = (mend - mbeg + 31) / 32;
  gload(mbeg);
  lstore(0);
  if (tid < TM) csum[tid] = 0.f;
  __syncthreads();
  // transposed-read addressing (T10): 16-lane group grp covers k rows 8*grp + {0..3} (+4 for the second read);
  // lane 4q+p of the group supplies the address of row q, columns 4p..4p+3 of the 16-column block.
  const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) gload(mbeg + (t + 1) * 32);
    h16x8 af[MI], bf[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const h16* pa = &As[buf][(8 * grp + tq) * SA + wm * (TM / 2) + i * 16 + 4 * tp];
      const h16x4 lo = lds_tr4(pa), hi = lds_tr4(pa + 4 * SA);
      af[i] = (h16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const h16* pb = &Bs[buf][(8 * grp + tq) * SB + wn * (TN / 2) + j * 16 + 4 * tp];
      const h16x4 lo = lds_tr4(pb), hi = lds_tr4(pb + 4 * SB);
      bf[j] = (h16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    if (t + 1 < nt) lstore(buf ^ 1);
    __syncthreads();
  }
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n1 = n1_0 + wm * (TM / 2) + i * 16 + fq * 4 + r;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n2 = n2_0 + wn * (TN / 2) + j * 16 + fr;
        atomicAdd(&g.C[(long)n1 * g.ldc + n2], acc[i][j][r]);
      }
    }
  if (want_cs) {      // the threads of one chunk column hold partial sums over their rows: meet in LDS, one global atomic per column
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int kc = (i * 256 + tid) % (TM / 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(&csum[kc * 8 + e], cs[i][e]);
    }
    __syncthreads();
    if (tid < TM) atomicAdd(&g.colsum[n1_0 + tid], csum[tid]);
  }
}

// block = 32 column groups of 8 (16-byte loads) x 8 row lanes over a slab of rows; one atomic per column per block
__global__ __launch_bounds__(256) void colsum_kernel(const h16* A, long lda, RowMap amap, int M, int N,
                                                     int rows_per_block, float* out) {
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + cg * 8;
  const int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (col < N)
    for (int m = mbeg + rl; m < mend; m += 8) {
      const h16x8 v = ldg8(A + amap.map(m) * lda + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
  __shared__ float red[8][256 + 8];
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = s[e];
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][c];
    atomicAdd(&out[blockIdx.x * 256 + c], t);
  }
}

// ------------------------------------------------------------------------------------------------
// Small strided fp32 GEMM (token side): fp32 MFMA, operands straight from global memory.
// The token side is a chain of ~5 us launches, so what counts is how few there are: one launch carries up to
// MT_SGEMM_MAX independent products (the dX and dW products of one nn.Linear backward; sibling projections of one input),
// the nn.Linear forward's activation / dropout / residual ride on the epilogue, and the backward's dropout mask and
// activation derivative are applied to the dy operand as it is loaded (no dpre tensor, no elementwise launches).
// ------------------------------------------------------------------------------------------------
struct SgemmArgs {
  const float* A; long as0, as1, a_bs;
  const float* B; long bs0, bs1, b_bs;
  const float* bias; int bias_on_m;
  float* C; long cs0, cs1, c_bs;
  int M, N, K, act, accumulate;
  float* rowsum;      // optional: rowsum[m] += sum_k A'(m,k) (the bias gradient riding on a dW = dy^T x product)
  float* pre_out;     // optional: the value before the activation, addressed like C (what the backward differentiates)
  const float* resid; // optional: added after activation and dropout, addressed like C
  DropArgs cdrop;     // element dropout of the activation output; mask index = element offset in C (a dense tensor)
  const float* a_aux; int a_act;    // A'(m,k) = drop(A(m,k)) * act'(a_aux(m,k)), a_aux addressed like A
  DropArgs adrop;     // mask index = element offset in A (the dense dy tensor whose forward drew the same mask)
  float resid_scale;  // C = resid_scale * resid + ...
  int a_ld;           // row length of the dense tensor behind A (DropPath of a_drop: pass = (offset / a_ld) / rows_per_pass)
  int ak, bk;         // operand rows k-contiguous and 16-byte aligned
  int nx, ny, nblk;   // 16 x 16 tiles along n / m; workgroups of this product (nx * ny * batch)
};
struct SgemmMulti { SgemmArgs p[MT_SGEMM_MAX]; int n; };

MT_DEVINL float apply_act(float v, int act) {
  switch (act) {
    case MT_ACT_RELU: return fmaxf(v, 0.f);
    case MT_ACT_GELU: return gelu_erf(v);
    case MT_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}
MT_DEVINL float act_grad(float v, int act) {
  switch (act) {
    case MT_ACT_RELU: return v > 0.f ? 1.f : 0.f;
    case MT_ACT_GELU: return gelu_erf_grad(v);
    case MT_ACT_ELU: return v > 0.f ? 1.f : __expf(v);
    default: return 1.f;
  }
}
// scale factors of 4 consecutive elements (offset off, off % 4 == 0) / of one element of an element-dropout site
MT_DEVINL f32x4 sg_drop4(const DropArgs& d, long off) { return drop_elem4(d, (uint64_t)off >> 2, 1.f); }
MT_DEVINL float sg_drop1(const DropArgs& d, long off) { return drop_keep1(d, d.site, (uint64_t)off) ? 1.f / (1.f - d.p) : 0.f; }

// 4 consecutive k of operand row `p` (k-contiguous: one 16-byte load; strided: 4 scalar loads)
template <bool KC>
MT_DEVINL f32x4 sg_load4(const float* p, long k, long s1) {
  if (KC) return *reinterpret_cast<const f32x4*>(p + k);
  return (f32x4){p[k * s1], p[(k + 1) * s1], p[(k + 2) * s1], p[(k + 3) * s1]};
}
// A' = drop(A) * act'(aux) on 4 consecutive k; off0 = element offset of (row, k) from the tensor base
template <bool KC>
MT_DEVINL f32x4 sg_xform4(const SgemmArgs& g, f32x4 a, f32x4 aux, long off0) {
  if (g.adrop.rng && g.adrop.p > 0.f) {
    if (KC) a *= sg_drop4(g.adrop, off0);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= sg_drop1(g.adrop, off0 + e * g.as1);
    }
  }
  if (g.adrop.rng && g.adrop.path_p > 0.f) {      // DropPath of the branch: a factor per task pass of the dense dy tensor's rows
    if (KC) a *= drop_path_factor(g.adrop, (int)(off0 / g.a_ld) / g.adrop.rows_per_pass);      // (4 consecutive k: one row)
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= drop_path_factor(g.adrop, (int)((off0 + e * g.as1) / g.a_ld) / g.adrop.rows_per_pass);
    }
  }
  if (g.a_aux) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] *= act_grad(aux[e], g.a_act);
  }
  return a;
}

// One workgroup = one 16x16 output tile; its four waves split K and each runs v_mfma_f32_16x16x4_f32 (fp32 operands,
// fp32 accumulate -- the token side stays fp32) on operands loaded STRAIGHT from global memory into the MFMA lane
// layout: lane (r = lane & 15, kq = lane >> 4) supplies row r of the tile and four consecutive k per 16-wide k block,
// one 16-byte load when the operand is k-contiguous (AK / BK), coalesced scalar loads across r otherwise (the
// transposed operands of the dW products).  No LDS in the k loop; the four partial tiles meet in LDS once at the end.
// (The previous form -- one output per thread, both operands re-read from LDS for every FMA -- was LDS-bound at
// 17 us for 195 x 192 x 768; these GEMMs are latency chains, so what matters is one deep batch of independent loads.)
template <bool AK, bool BK_>
MT_DEVINL void sgemm_tile(const SgemmArgs& g, int bx, int by, int bz, float (*part)[16][17], float (*rsum)[16]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int m0 = by * 16, n0 = bx * 16;
  const long arow = (long)bz * g.a_bs + (long)min(m0 + r, g.M - 1) * g.as0;     // element offset of this lane's A row
  const float* ap = g.A + arow;
  const float* xp = g.a_aux ? g.a_aux + arow : nullptr;
  const float* bp = g.B + (long)bz * g.b_bs + (long)min(n0 + r, g.N - 1) * g.bs0;
  const bool xf = g.a_aux != nullptr || g.adrop.rng != nullptr;
  // this wave's k range: a multiple of 16 per wave
  const int kper = ((g.K + 63) / 64) * 16;
  const int kb = wave * kper, ke = min(g.K, kb + kper);
  const bool want_rs = g.rowsum != nullptr && bx == 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float rs = 0.f;
  int k0 = kb;
  // full 64-wide blocks: 4 (x 2 or 3 operands) independent loads per lane in flight, then 16 MFMAs
  for (; k0 + 64 <= ke; k0 += 64) {
    f32x4 a[4], b[4], x[4] = {};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 16 * j + 4 * kq;
      a[j] = sg_load4<AK>(ap, k, g.as1);
      b[j] = sg_load4<BK_>(bp, k, g.bs1);
      if (xp) x[j] = sg_load4<AK>(xp, k, g.as1);
    }
    if (xf) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = sg_xform4<AK>(g, a[j], x[j], arow + (long)(k0 + 16 * j + 4 * kq) * g.as1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][e], b[j][e], acc, 0, 0, 0);
      if (want_rs) rs += (a[j][0] + a[j][1]) + (a[j][2] + a[j][3]);
    }
  }
  // full 16-wide blocks (a wave's share of a short K is below 64)
  for (; k0 + 16 <= ke; k0 += 16) {
    const int k = k0 + 4 * kq;
    f32x4 a = sg_load4<AK>(ap, k, g.as1), x = {0.f, 0.f, 0.f, 0.f};
    const f32x4 b = sg_load4<BK_>(bp, k, g.bs1);
    if (xp) x = sg_load4<AK>(xp, k, g.as1);
    if (xf) a = sg_xform4<AK>(g, a, x, arow + (long)k * g.as1);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    if (want_rs) rs += (a[0] + a[1]) + (a[2] + a[3]);
  }
  // ragged rest: guarded scalar loads (zero beyond the range)
  for (; k0 < ke; k0 += 16) {
    float a[4], b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + 4 * kq + e;
      const bool ok = k < ke;
      const int kc = ok ? k : kb;
      float av = ap[(long)kc * g.as1];
      const float bv = bp[(long)kc * g.bs1];
      if (g.adrop.rng && g.adrop.p > 0.f) av *= sg_drop1(g.adrop, arow + (long)kc * g.as1);
      if (g.adrop.rng && g.adrop.path_p > 0.f) av *= drop_path_factor(g.adrop, (int)((arow + (long)kc * g.as1) / g.a_ld) / g.adrop.rows_per_pass);
      if (xp) av *= act_grad(xp[(long)kc * g.as1], g.a_act);
      a[e] = ok ? av : 0.f; b[e] = ok ? bv : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    if (want_rs) rs += (a[0] + a[1]) + (a[2] + a[3]);
  }
  // accumulator element e of lane = C[m = 4 * kq + e][n = r]
#pragma unroll
  for (int e = 0; e < 4; ++e) part[wave][4 * kq + e][r] = acc[e];
  if (want_rs) {
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    if (kq == 0) rsum[wave][r] = rs;
  }
  __syncthreads();
  const int tm = tid >> 4, tn = tid & 15;
  const int m = m0 + tm, n = n0 + tn;
  if (want_rs && tid < 16 && m0 + tid < g.M)
    g.rowsum[m0 + tid] += (rsum[0][tid] + rsum[1][tid]) + (rsum[2][tid] + rsum[3][tid]);
  if (m < g.M && n < g.N) {
    float v = (part[0][tm][tn] + part[1][tm][tn]) + (part[2][tm][tn] + part[3][tm][tn]);
    if (g.bias) v += g.bias[g.bias_on_m ? m : n];
    const long off = (long)bz * g.c_bs + m * g.cs0 + n * g.cs1;
    if (g.pre_out) g.pre_out[off] = v;
    v = apply_act(v, g.act);
    if (g.cdrop.rng && g.cdrop.p > 0.f) v *= sg_drop1(g.cdrop, off);
    if (g.cdrop.rng && g.cdrop.path_p > 0.f) v *= drop_path_factor(g.cdrop, m / g.cdrop.rows_per_pass);
    if (g.resid) v += g.resid_scale * g.resid[off];
    float* c = &g.C[off];
    *c = g.accumulate ? *c + v : v;
  }
}

__global__ __launch_bounds__(256) void sgemm_multi_kernel(SgemmMulti mm) {
  __shared__ float part[4][16][17];
  __shared__ float rsum[4][16];
  int blk = blockIdx.x, pi = 0;
#pragma unroll
  for (int i = 0; i + 1 < MT_SGEMM_MAX; ++i)
    if (pi == i && i + 1 < mm.n && blk >= mm.p[i].nblk) { blk -= mm.p[i].nblk; pi = i + 1; }
  const SgemmArgs& g = mm.p[pi];
  const int bx = blk % g.nx, by = (blk / g.nx) % g.ny, bz = blk / (g.nx * g.ny);
  if (g.ak && g.bk) sgemm_tile<true, true>(g, bx, by, bz, part, rsum);
  else if (g.ak) sgemm_tile<true, false>(g, bx, by, bz, part, rsum);
  else if (g.bk) sgemm_tile<false, true>(g, bx, by, bz, part, rsum);
  else sgemm_tile<false, false>(g, bx, by, bz, part, rsum);
}

}  // namespace

extern "C" int mt_gemm_nt_f16(const mt_half* A, long lda, const MtRowMap* amap, const mt_half* W, int M, int N, int K,
                              int epilogue, const MtGemmEpilogue* epi, void* C, long ldc, const MtRowMap* cmap,
                              int out_dtype, mt_stream_t stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return MT_ERR_BAD_ARG;
  if (K % BK != 0 || lda % 8 != 0 || (N % 64) != 0 || (ldc % 4) != 0) return MT_ERR_BAD_ARG;
  if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 7)) return MT_ERR_BAD_ARG;
  if (epi && epi->resid && (((uintptr_t)epi->resid & 15) || (epi->ldr % 4))) return MT_ERR_BAD_ARG;
  GemmNtArgs a;
  a.A = (const h16*)A; a.lda = lda; a.amap = make_rowmap(amap);
  a.W = (const h16*)W; a.M = M; a.N = N; a.K = K;
  a.m_begin = 0; a.m_end = M;
  a.bias = epi ? epi->bias : nullptr;
  a.resid = epi ? epi->resid : nullptr; a.ldr = epi ? epi->ldr : 0;
  a.rmap = make_rowmap(epi ? &epi->rmap : nullptr);
  a.colscale = epi ? epi->colscale : nullptr;
  a.pos_table = epi ? epi->pos_table : nullptr;
  a.pos_row = epi ? epi->pos_row : nullptr; a.pos_col = epi ? epi->pos_col : nullptr;
  a.drop = make_drop(epi ? &epi->drop : nullptr);
  if (a.drop.active() && epilogue != MT_EPI_BIAS_RESID) return MT_ERR_UNSUPPORTED;
  a.C = C; a.ldc = ldc; a.cmap = make_rowmap(cmap);
  hipStream_t s = (hipStream_t)stream;
  const bool f32 = out_dtype == MT_OUT_F32;
  // the frozen backbone's big-M shapes (and their dX counterparts) run on the persistent one-wave-per-SIMD kernel (gemm_ps.hip), which
  // declines (MT_ERR_UNSUPPORTED) what it does not serve or would serve badly (few tiles per CU).  MT_GEMM_PS=0 switches it off (A/B runs).
  const char* ps_env = getenv("MT_GEMM_PS");
  if (!(ps_env && ps_env[0] == '0') && !f32 && (epilogue == MT_EPI_BIAS || epilogue == MT_EPI_QKV_HM) && a.amap.seg_rows <= 0 &&
      a.cmap.seg_rows <= 0 && (ldc % 8) == 0 && !((uintptr_t)C & 15) && (epilogue != MT_EPI_QKV_HM || a.bias) && !a.drop.active()) {
    const int rc = mt_gemm_ps_launch(A, lda, W, M, N, K, epilogue, a.bias, C, ldc, s);
    if (rc != MT_ERR_UNSUPPORTED) return rc;
  }
  switch (epilogue) {
    case MT_EPI_BIAS:
      return f32 ? launch_nt_bn<MT_EPI_BIAS, float>(a, s) : launch_nt_bn<MT_EPI_BIAS, h16>(a, s);
    case MT_EPI_BIAS_RESID:
      if (!a.resid) return MT_ERR_BAD_ARG;
      return f32 ? launch_nt_bn<MT_EPI_BIAS_RESID, float>(a, s) : launch_nt_bn<MT_EPI_BIAS_RESID, h16>(a, s);
    case MT_EPI_INJECT:
      if (!a.resid || !a.colscale || !f32) return MT_ERR_BAD_ARG;
      return launch_nt_bn<MT_EPI_INJECT, float>(a, s);
    case MT_EPI_POSEMB:
      if (!a.pos_table || !a.pos_row || !a.pos_col || !f32) return MT_ERR_BAD_ARG;
      return launch_nt_bn<MT_EPI_POSEMB, float>(a, s);
    case MT_EPI_QKV_HM:
      if (f32 || (N % 48) != 0 || cmap) return MT_ERR_BAD_ARG;
      return launch_nt_bn<MT_EPI_QKV_HM, h16>(a, s);
    default:
      return MT_ERR_BAD_ARG;
  }
}

extern "C" int mt_gemm_tn_f16(const mt_half* A, long lda, const MtRowMap* amap, const mt_half* B, long ldb,
                              const MtRowMap* bmap, int M, int N1, int N2, float* C, long ldc, float* colsum,
                              mt_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N1 % 64 || N2 % 64 || lda % 8 || ldb % 8) return MT_ERR_BAD_ARG;
  GemmTnArgs g;
  g.A = (const h16*)A; g.lda = lda; g.amap = make_rowmap(amap);
  g.B = (const h16*)B; g.ldb = ldb; g.bmap = make_rowmap(bmap);
  g.M = M; g.N1 = N1; g.N2 = N2; g.C = C; g.ldc = ldc; g.colsum = colsum;
  // 64 x 64 tiles and ~1024 workgroups (512 for the small products) measured best at M = 30 000 once the tiles of a row range
  // share an XCD (tools/experiments/tn_sweep.sh: 63 / 39 / 39 / 19 us for 384x768 / 192x768 / 768x192 / 192x192; the larger
  // tiles the kernel template allows re-read less but need more splits -- more atomics -- for the same number of workgroups)
  const int tiles = (N1 / 64) * (N2 / 64);
  int split = max(1, min(cdiv(M, 256), cdiv(tiles >= 16 ? 1024 : 512, tiles)));
  split = cdiv(split, 8) * 8;                                   // a multiple of the 8 XCDs
  g.rows_per_split = cdiv(cdiv(M, split), 32) * 32;
  hipLaunchKernelGGL((gemm_tn_kernel<64, 64>), dim3(tiles * split), dim3(256), 0, (hipStream_t)stream, g);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_colsum_f16(const mt_half* A, long lda, const MtRowMap* amap, int M, int N, float* out,
                             mt_stream_t stream) {
  if (!A || !out || M <= 0 || N <= 0 || (N & 7) || (lda & 7)) return MT_ERR_BAD_ARG;
  const int rows_per_block = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 256), cdiv(M, rows_per_block)), dim3(256), 0, (hipStream_t)stream,
                     (const h16*)A, lda, make_rowmap(amap), M, N, rows_per_block, out);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

static int sgemm_fill(const MtSgemm& q, SgemmArgs& g) {
  if (!q.A || !q.B || !q.C || q.M <= 0 || q.N <= 0 || q.K <= 0 || q.batch <= 0 || (q.rowsum && q.batch != 1)) return MT_ERR_BAD_ARG;
  g.A = q.A; g.as0 = q.as0; g.as1 = q.as1; g.a_bs = q.a_bs;
  g.B = q.B; g.bs0 = q.bs0; g.bs1 = q.bs1; g.b_bs = q.b_bs;
  g.bias = q.bias; g.bias_on_m = q.bias_on_m;
  g.C = q.C; g.cs0 = q.cs0; g.cs1 = q.cs1; g.c_bs = q.c_bs;
  g.M = q.M; g.N = q.N; g.K = q.K; g.act = q.act; g.accumulate = q.accumulate;
  g.rowsum = q.rowsum; g.pre_out = q.pre_out; g.resid = q.resid;
  g.cdrop = make_drop(&q.c_drop); g.a_aux = q.a_aux; g.a_act = q.a_act; g.adrop = make_drop(&q.a_drop);
  g.resid_scale = q.resid_scale != 0.f ? q.resid_scale : 1.f;
  g.a_ld = q.a_ld;
  // DropPath of a_drop needs the row length of the dense tensor behind A; of c_drop a dense one-batch C (row m = its row)
  if (g.adrop.rng && g.adrop.path_p > 0.f && q.a_ld < 1) return MT_ERR_BAD_ARG;
  if (g.cdrop.rng && g.cdrop.path_p > 0.f && q.batch != 1) return MT_ERR_UNSUPPORTED;
  if (!(g.cdrop.p > 0.f || g.cdrop.path_p > 0.f)) g.cdrop.rng = nullptr;
  if (!(g.adrop.p > 0.f || g.adrop.path_p > 0.f)) g.adrop.rng = nullptr;
  // 16-byte loads along k need k-contiguous, 16-byte aligned rows (of the auxiliary operand too)
  g.ak = q.as1 == 1 && (q.as0 % 4) == 0 && (q.a_bs % 4) == 0 && ((uintptr_t)q.A & 15) == 0 && ((uintptr_t)q.a_aux & 15) == 0;
  g.bk = q.bs1 == 1 && (q.bs0 % 4) == 0 && (q.b_bs % 4) == 0 && ((uintptr_t)q.B & 15) == 0;
  g.nx = cdiv(q.N, 16); g.ny = cdiv(q.M, 16); g.nblk = g.nx * g.ny * q.batch;
  return MT_OK;
}

extern "C" int mt_sgemm_multi(const MtSgemm* probs, int n, mt_stream_t stream) {
  if (!probs || n < 1 || n > MT_SGEMM_MAX) return MT_ERR_BAD_ARG;
  SgemmMulti mm;
  mm.n = n;
  long blocks = 0;
  for (int i = 0; i < MT_SGEMM_MAX; ++i) {
    const int rc = sgemm_fill(probs[i < n ? i : 0], mm.p[i]);
    if (rc != MT_OK) return rc;
    if (i < n) blocks += mm.p[i].nblk;
  }
  if (blocks > 0x7fffffffL) return MT_ERR_BAD_ARG;
  hipLaunchKernelGGL(sgemm_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mm);
  MT_CHECK_LAUNCH();
  return MT_OK;
}

extern "C" int mt_sgemm_small(const float* A, long as0, long as1, long a_bs, const float* B, long bs0, long bs1,
                              long b_bs, const float* bias, int bias_on_m, float* C, long cs0, long cs1, long c_bs,
                              int M, int N, int K, int batch, int act, int accumulate, float* rowsum,
                              mt_stream_t stream) {
  MtSgemm q = {};
  q.A = A; q.as0 = as0; q.as1 = as1; q.a_bs = a_bs; q.B = B; q.bs0 = bs0; q.bs1 = bs1; q.b_bs = b_bs;
  q.bias = bias; q.bias_on_m = bias_on_m; q.C = C; q.cs0 = cs0; q.cs1 = cs1; q.c_bs = c_bs;
  q.M = M; q.N = N; q.K = K; q.batch = batch; q.act = act; q.accumulate = accumulate; q.rowsum = rowsum;
  return mt_sgemm_multi(&q, 1, stream);
}
