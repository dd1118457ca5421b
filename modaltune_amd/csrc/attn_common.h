// Shared pieces of the dilated-attention kernels (attn.hip, attn_bwd64.hip): plan, work decode, sparse-sequence
// geometry, LDS layouts and small device helpers.  Internal linkage: each translation unit gets its own copy.
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

constexpr int H = 16, HD = 48, DM = 768, QKV_LD = 2304;
constexpr int KSTR = 56;    // halves per K-layout LDS row (112 B: conflict-free ds_read_b128 row reads)
constexpr int VSTR = 96;    // halves per V-layout LDS row (192 B: conflict-free ds_read_b64_tr_b16)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_LOG2 = 8.0f;
// The q slab of the head-major q|k|v buffer holds q' = (48^-1/2 log2 e) q: the softmax scale, in log2 units, is baked into
// the q rows of the frozen projection's fp16 weight cache (engine.py), so s' = q'.k is the exp2 argument as it leaves the
// MFMA chain and no kernel multiplies by it.  Backward: dL/ds' = ln2 P (dP - delta), so P'' = ln2 P = exp2(s' - L2 + log2 ln2)
// carries the factor; dq' = dS''.k and dk = dS''^T.q' are the gradients the dX GEMM (which reads the same scaled cache) needs,
// dv = P^T dO = P''^T dO / ln2.
constexpr float QK_SCALE_LOG2 = 0.14433756729740643f * 1.4426950408889634f;   // 48^-1/2 * log2(e)
constexpr float LOG2_LN2 = -0.52876637294489768f;    // log2(ln 2)
constexpr float INV_LN2 = 1.4426950408889634f;

struct Plan {
  int nbranch, N, B;
  int seg[MT_MAX_BRANCHES], ratio[MT_MAX_BRANCHES], nseg[MT_MAX_BRANCHES], n[MT_MAX_BRANCHES];
  int order[MT_MAX_BRANCHES], qtiles[MT_MAX_BRANCHES], blk_off[MT_MAX_BRANCHES + 1];
  long ws_off[MT_MAX_BRANCHES + 1];   // backward workspace: per-branch compact, TOKEN-major [pass][token][q|k|v][16 / ratio heads][48] fp16
  float inv_seg[MT_MAX_BRANCHES];     // 1 / seg (position -> segment index without an integer division per row)
  int qlimit[MT_MAX_BRANCHES];        // sparse entries [0, qlimit) act as queries (= n unless sequence-parallel)
};

Plan make_plan(const MtDilatedPlan* p, int qtile) {
  Plan d;
  d.nbranch = p->nbranch; d.N = p->N; d.B = p->B;
  for (int i = 0; i < p->nbranch; ++i) {
    d.seg[i] = p->seg[i]; d.ratio[i] = p->ratio[i]; d.nseg[i] = p->nseg[i]; d.n[i] = p->n[i];
    d.order[i] = i;
    d.qtiles[i] = cdiv(p->n[i], qtile);
    d.inv_seg[i] = 1.0f / (float)p->seg[i];
    d.qlimit[i] = p->qlimit[i] > 0 ? p->qlimit[i] : p->n[i];
  }
  // longest sparse sequences first (their workgroups run longest)
  for (int i = 0; i < d.nbranch; ++i)
    for (int j = i + 1; j < d.nbranch; ++j)
      if (d.n[d.order[j]] > d.n[d.order[i]]) { int t = d.order[i]; d.order[i] = d.order[j]; d.order[j] = t; }
  d.blk_off[0] = 0;
  for (int i = 0; i < d.nbranch; ++i) {
    const int b = d.order[i];
    d.blk_off[i + 1] = d.blk_off[i] + d.B * d.nseg[b] * H * d.qtiles[b];
  }
  d.ws_off[0] = 0;
  for (int b = 0; b < d.nbranch; ++b) d.ws_off[b + 1] = d.ws_off[b] + (long)d.B * d.N * 3 * (H / d.ratio[b]) * HD;
  return d;
}

bool plan_ok(const MtDilatedPlan* p) {
  if (!p || p->nbranch < 1 || p->nbranch > MT_MAX_BRANCHES || p->N < 1 || p->B < 1) return false;
  for (int i = 0; i < p->nbranch; ++i) {
    const int r = p->ratio[i], s = p->seg[i];
    if (r < 1 || r > H || (H % r) != 0 || s < 1 || s > p->N) return false;
    if (p->nseg[i] != cdiv(p->N, s) || p->n[i] != cdiv(s, r)) return false;
    if (p->qlimit[i] < 0 || p->qlimit[i] > p->n[i]) return false;
  }
  return true;
}

struct WorkItem { int br, b, j, h, qt; };

// Workgroup -> work item.  Blocks b and b + 8 share an XCD (round-robin dispatch), so XCD x = bid % 8 walks its own
// list j = bid / 8: per branch (longest sequences first) the (pass, segment, head) groups x, x + 8, x + 16, ... and,
// inside a group, the query tiles back to back.  All query tiles of a group -- which re-read the same K/V rows --
// therefore run on ONE XCD's L2, and every XCD gets the same mix of long and short sequences (each branch has a
// multiple of 8 groups because there are 16 heads).  Placement only affects speed, never results.
MT_DEVINL WorkItem decode(const Plan& p, int bid) {
  const int x = bid & 7, j = bid >> 3;
  int oi = 0;
#pragma unroll
  for (int i = 1; i < MT_MAX_BRANCHES; ++i)
    if (i < p.nbranch && j >= (p.blk_off[i] >> 3)) oi = i;
  WorkItem w;
  w.br = p.order[oi];
  int local = j - (p.blk_off[oi] >> 3);
  w.qt = local % p.qtiles[w.br];
  const int gid = (local / p.qtiles[w.br]) * 8 + x;     // group index inside the branch
  w.h = gid % H;
  const int r = gid / H;
  w.j = r % p.nseg[w.br];
  w.b = r / p.nseg[w.br];
  // the divisions above run on the VALU: pin the (wave-uniform) results in SGPRs so everything derived from them --
  // base pointers, tile counts, loop bounds -- stays scalar
  w.br = __builtin_amdgcn_readfirstlane(w.br); w.qt = __builtin_amdgcn_readfirstlane(w.qt);
  w.h = __builtin_amdgcn_readfirstlane(w.h); w.j = __builtin_amdgcn_readfirstlane(w.j);
  w.b = __builtin_amdgcn_readfirstlane(w.b);
  return w;
}

// single-branch launches (backward): same XCD-balanced walk over one branch
MT_DEVINL WorkItem decode_branch(const Plan& p, int br, int bid) {
  const int x = bid & 7, j = bid >> 3;
  WorkItem w;
  w.br = br;
  w.qt = j % p.qtiles[br];
  const int gid = (j / p.qtiles[br]) * 8 + x;
  w.h = gid % H;
  const int r = gid / H;
  w.j = r % p.nseg[br];
  w.b = r / p.nseg[br];
  return w;
}

// ONE LDS image of a 64-row x 48-col fp16 tile that serves both the row reads (ds_read_b128: A fragments of Q . K^T
// style products) and the transposed reads (ds_read_b64_tr_b16: A fragments of Q^T . dS style products): rows of VSTR
// halves (48 dwords: the four rows of a transposed read land 16 banks apart), and the 16-byte chunk c of row r stored
// at chunk c ^ ((r >> 2) & 3) so that the 16 rows of a ds_read_b128 lane group, which share c, spread over all 64 banks.
// Both patterns are conflict-free (checked exhaustively); chunks 6 and 7 of every row hold the zeros read as d = 48..63.
MT_DEVINL int swz(int row, int chunk) { return row * VSTR + ((chunk ^ ((row >> 2) & 3)) << 3); }

MT_DEVINL h16x4 lds_tr4(const h16* p) {
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)p);
  return __builtin_bit_cast(h16x4, r);
}
MT_DEVINL h16x8 cat8(h16x4 lo, h16x4 hi) { return (h16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

// geometry of one (branch, segment, head) sparse sequence
struct Seq {
  int n, s, dr, r, seg_base, N;
  long row_base;   // b * N
  MT_DEVINL bool valid(int i) const {
    const int loc = r + i * dr;
    return i < n && loc < s && seg_base + loc < N;
  }
  MT_DEVINL long row(int i) const { return row_base + seg_base + r + (long)i * dr; }
  // always-in-bounds row for unconditional loads (the value is discarded with a select when !valid(i)):
  // a branch around each load would make hipcc wait for every load separately (guide §5 trap (c))
  MT_DEVINL long row_clamped(int i) const { return min(row(i), row_base + (long)N - 1); }
  // valid(i) holds exactly for i < nvalid(): r + i dr < min(s, N - seg_base)
  MT_DEVINL int nvalid() const {
    const int lim = min(s, N - seg_base) - r;
    return lim <= 0 ? 0 : min(n, (lim + dr - 1) / dr);
  }
};

// q / k / v are HEAD-MAJOR: [which = q|k|v][head][B*N rows][48] (written that way by the QKV GEMM epilogue), so a
// head's dilated key walk touches rows 96 * r bytes apart (contiguous for r = 1) instead of 4608 * r in the
// token-major [B*N, 2304] layout; dmixed is head-major too ([head][B*N][48]).
MT_DEVINL const h16* hm_ptr(const h16* base, long M, int slab, long row) { return base + ((long)slab * M + row) * HD; }

// uniform base + 32-bit byte offset: one VGPR of address state per thread (global_load ... v_off, s[base]), no 64-bit
// per-lane arithmetic in the tile loops
MT_DEVINL h16x8 ldg8_off(const h16* base, uint32_t byte_off) {
  return *reinterpret_cast<const h16x8*>(reinterpret_cast<const char*>(base) + byte_off);
}
MT_DEVINL float ldf_off(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
// Full tiles go through BUFFER loads: resource = the sparse sequence's (wave-uniform) base, voffset = the lane's constant
// 32-bit byte offset, soffset = the tile's advance in an SGPR -- no per-lane address arithmetic at all (with global
// loads the loop-strength reducer kept one 64-bit pointer per load and lane: 11 v_lshl_add_u64 per tile in the dK/dV
// kernel).  Offsets stay far below 2^31 (a sparse sequence spans at most N rows of 96 B x dilation).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
MT_DEVINL __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
MT_DEVINL h16x8 buf_ldg8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
MT_DEVINL float buf_ldf(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
MT_DEVINL f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
MT_DEVINL f32x2 pk_exp2(f32x2 a) { return (f32x2){__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])}; }

// max over the two lane halves (lane l and l ^ 32) without the LDS round trip of ds_bpermute: v_permlane32_swap
// exchanges the upper half of one register with the lower half of another (guide T12)
MT_DEVINL float max_halves(float x) {
  float lo = x, hi = x;      // two registers: the instruction rewrites BOTH operands in place (lo <- {x_lo, x_lo}, hi <- {x_hi, x_hi})
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));      // (2 wait states after a VALU write)
  return fmaxf(lo, hi);
}

MT_DEVINL h16x8 sel8(bool ok, h16x8 v) {
  const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return ok ? v : z;
}

// Backward workspace of a branch: TOKEN-major [pass][token][which = dq | dk | dv][slot][48] with slot = head % (16 / ratio): a
// token is visited by the 16 / ratio heads of ONE group (the group whose index is the token's residue inside its segment), so
// the (16 / ratio) x 48 halves of one `which` are exactly the contiguous run of columns that group owns in the dense row
// [q | k | v][16 heads x 48] -- the combine pass then reads and writes contiguous runs per (token, branch, which) instead of
// gathering 288-byte entries that lie n x 288 bytes apart (same-box A/B: 0.113 -> 0.105 ms per launch; 16-byte loads with two
// rows per workgroup were slower: 0.118).
// Returns the offset (halves) of `which` = 0 of (token, head); `which` advances by ws_which_stride().
MT_DEVINL long ws_slot(const Plan& p, const WorkItem& w, long token_row) {
  const int hb = H / p.ratio[w.br];
  return p.ws_off[w.br] + (token_row * 3) * (hb * HD) + (w.h % hb) * HD;
}
MT_DEVINL int ws_which_stride(const Plan& p, int br) { return (H / p.ratio[br]) * HD; }

MT_DEVINL Seq make_seq(const Plan& p, const WorkItem& w) {
  Seq q;
  q.n = p.n[w.br]; q.s = p.seg[w.br]; q.dr = p.ratio[w.br];
  q.r = __builtin_amdgcn_readfirstlane(w.h / (H / q.dr));
  q.seg_base = w.j * q.s; q.N = p.N; q.row_base = (long)w.b * p.N;
  return q;
}

// ---- LDS-DMA tile image (forward kernel; tools/lds_bank_check.py) ---------------------------------------------------
// [64 rows][128 B]: logical 16-byte chunks 0..5 = the 48 halves of a row, 6..7 = constants (ones / zeros columns read as
// d = 48..63); chunk c of row r sits at chunk position c ^ img_f(r).  With it BOTH the row reads (ds_read_b128, 16-lane
// groups of rows at one chunk) and the transposed reads (ds_read_b64_tr_b16, four consecutive rows x 64 B) are
// bank-conflict free, and -- rows being exactly 8 chunks -- a tile is filled by LDS-DMA (buffer_load_dwordx4 ... lds writes
// lane j's 16 bytes at base + 16 j): piece p of a tile = rows 8p..8p+7, lane j -> row 8p + j / 8, position j % 8, i.e.
// logical chunk (j % 8) ^ img_f(row); lanes whose logical chunk is 6 or 7 are switched off.  No staging registers, no
// ds_write pass, no address arithmetic per tile; rows past the end of the sparse sequence come back as zeros from the
// buffer descriptor's range check (num_records = the sequence's valid bytes), which is what the reference's zero padding
// is (DA:98-101), so ragged tiles need neither clamps nor selects.
constexpr int IMG_ROW = 64;                      // halves per image row
constexpr int IMG_HALVES = 64 * IMG_ROW;         // 8 KB
MT_DEVINL int img_f(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
MT_DEVINL int img_off(int row, int chunk) { return row * IMG_ROW + ((chunk ^ img_f(row)) << 3); }      // in halves

struct DmaLane {      // this thread's two pieces of a tile image: p = 2 * wave + i
  uint32_t voff[2]; bool act[2]; int lds_halves[2];
  MT_DEVINL DmaLane(int tid, int row_stride_bytes) {
    const int wave = tid >> 6, j = tid & 63;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = 2 * wave + i, row = 8 * piece + (j >> 3);
      const int c = (j & 7) ^ img_f(row);
      act[i] = c < 6;
      voff[i] = (uint32_t)(row * row_stride_bytes + min(c, 5) * 16);
      lds_halves[i] = piece * 512;               // 1 KiB per piece
    }
  }
};
// descriptor of the rows [row_first, ...) of one (slab, head) that tile t may touch: base advanced to the tile's first
// row, num_records = what is left of the sequence's valid bytes (0 when the tile lies past the end)
MT_DEVINL __amdgpu_buffer_rsrc_t tile_rsrc(const h16* seq_base, long tile_byte_off, long valid_bytes) {
  const long left = valid_bytes - tile_byte_off;
  const int rec = (int)__builtin_amdgcn_readfirstlane((int)(left > 0 ? (left < 0x7fffffffL ? left : 0x7fffffffL) : 0));
  const char* b = reinterpret_cast<const char*>(seq_base) + tile_byte_off;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), 0, rec, 0x00020000);
}
MT_DEVINL void dma_tile(h16* img, __amdgpu_buffer_rsrc_t rs, const DmaLane& d) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (d.act[i])
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(img + d.lds_halves[i]), 16, d.voff[i], 0, 0, 0);
}
// The two images of a tile (K | V, Q | dO) in one go: piece i of BOTH images under ONE exec mask (two masked blocks per tile
// instead of four), the LDS destinations wave-uniform (`wave_u` = readfirstlane(tid >> 6): the M0 setup of a piece is then a
// scalar add, not a VALU add + v_readfirstlane + wait state).
MT_DEVINL void dma_tile_pair(h16* img_a, __amdgpu_buffer_rsrc_t ra, h16* img_b, __amdgpu_buffer_rsrc_t rb, const DmaLane& d, int wave_u) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (d.act[i]) {
      const int off = (2 * wave_u + i) * 512;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(img_a + off), 16, d.voff[i], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(img_b + off), 16, d.voff[i], 0, 0, 0);
    }
}
MT_DEVINL void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Per-thread staging slots of a 64-row x 48-col fp16 tile = 384 chunks of 16 B.
//
// A ds_write_b128 is served in groups of eight consecutive lanes over 32 banks, so a group is conflict-free when its
// 16-byte pieces sit on distinct 16-byte slots modulo 128 B.  PACKED form (chunk id = thread id, six lanes per row; 1.5
// store instructions per wave and image): a group holds pieces of two rows and pays a 2-way conflict in every image used
// here (tools/lds_bank_check.py; SQ_LDS_BANK_CONFLICT: 23 % of the forward kernel's LDS cycles, 11 % of the dK/dV
// kernel's).  ROW form (eight lanes per row, two of them idle; 2 store instructions per wave and image): one row per
// group, consecutive pieces, conflict-free in every image.  Same-box A/B (tools/ab_lib.sh): the forward kernel times
// the same with either (its LDS array is 39 % busy), the backward kernels run 1 % slower with the ROW form (one more
// store + load instruction per wave and image in kernels that sit on the issue port), so they keep the packed form.
struct StageIdx {       // packed
  int row0, part0, row1, part1; bool has1;
  MT_DEVINL StageIdx(int tid) {
    row0 = tid / 6; part0 = tid - row0 * 6;
    const int c = 256 + tid;
    row1 = c / 6; part1 = c - row1 * 6; has1 = tid < 128;
  }
};
struct StageRow {       // eight lanes per row
  int row0, row1, part; bool act;
  MT_DEVINL StageRow(int tid) {
    row0 = tid >> 3; row1 = 32 + row0;
    act = (tid & 7) < 6;
    part = min(tid & 7, 5);          // idle lanes re-read chunk 5 (same cache line) and store nothing
  }
};

}  // namespace

// The backward's three kernels live in their own translation units (attn_bwd_q.hip, attn_bwd_kv.hip, attn_combine.hip) so that each
// can carry its own LLVM scheduling strategy (a per-module option: __graft_entry__.py, FLAGS_PER_FILE).  Measured after the split
// (round 4): none of the strategies beats the default any more -- the table is empty; the units stay separate (13 s instead of 30 s to
// rebuild one kernel).  mt_dilated_attn_bwd (attn.hip) calls these launchers.
namespace mt_attn {
// (C-ABI types only: the device-side Plan lives in each unit's anonymous namespace and has no linkage)
void launch_bwd_kv(const mt_half* qkv, const mt_half* dmixed, const float* lse_tot, const float* delta_br, const MtDilatedPlan* plan, void* ws, hipStream_t s);
void launch_bwd_q(const mt_half* qkv, const mt_half* dmixed, const float* lse_tot, const float* delta_br, const MtDilatedPlan* plan, void* ws, hipStream_t s);
void launch_bwd_combine(const void* ws, const MtDilatedPlan* plan, mt_half* dqkv, hipStream_t s);
}  // namespace mt_attn
