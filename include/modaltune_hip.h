/* modaltune_hip — C ABI of the MI355X-native Modal-Adapter hot path.
 *
 * The reference (martellab-sri/ModalTune) is pure Python/PyTorch and has no FFI layer; its drop-in
 * boundary is the nn.Module surface.  This header is the native boundary underneath
 * that surface: one `extern "C"` launcher per fused op and direction, plain device pointers and sizes,
 * no torch types.  Each entry cites the reference code whose arithmetic it replaces
 * (paths relative to the reference root; abbreviations as in SURVEY.md).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless stated; the caller owns every buffer (no hidden hipMalloc)
 *   - `mt_half` = IEEE fp16 storage (the reference's AMP dtype, TM:216); accumulation is fp32 everywhere
 *   - activations are row-major [rows, cols]; token buffers are [B*N, C] with pass b at rows [b*N, (b+1)*N)
 *   - every launcher enqueues on `stream` (a hipStream_t) and returns immediately: 0 or a negative MtStatus
 *   - no exception crosses this boundary; nothing here synchronises the device
 */
#ifndef MODALTUNE_HIP_H
#define MODALTUNE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mt_stream_t;     /* hipStream_t */
typedef uint16_t mt_half;      /* IEEE binary16 bit pattern */

typedef enum {
  MT_OK = 0,
  MT_ERR_BAD_ARG = -1,         /* shape/alignment the kernels do not support */
  MT_ERR_LAUNCH = -2,          /* hipGetLastError() != hipSuccess after the launch */
  MT_ERR_UNSUPPORTED = -3
} MtStatus;

/* logical row m -> physical row (m / seg_rows) * seg_stride + row0 + m % seg_rows; seg_rows <= 0: identity.
 * Lets the adapter ops address "the patch rows of a [B, N, C] token buffer" without the torch.cat / slice
 * copies of AM:492,501-510, and broadcast one slide to the 3 task passes (seg_stride = 0). */
typedef struct { int seg_rows, seg_stride, row0; } MtRowMap;

int mt_version(void);
const char* mt_status_string(int status);
/* Build provenance: sha256 (hex) over the kernel sources, this header and the compiler flags the library was built from
 * (modaltune_amd/_build_id.py: tree_build_id); `build()` rebuilds when it differs from the tree's. */
const char* mt_build_id(void);

/* ---------------------------------------------------------------- GEMMs ---------------------------- */
enum { MT_EPI_BIAS = 0,        /* C = acc + bias                                   (nn.Linear)            */
       MT_EPI_BIAS_RESID = 1,  /* C = resid + acc + bias                           (ENC:149-154,169-172)  */
       MT_EPI_INJECT = 2,      /* C = (1+g[n]) * resid + g[n] * (acc + bias)       (AM:231,362; A.1)      */
       MT_EPI_POSEMB = 3,      /* C = acc + bias + sincos(col|row)                 (LVA:232-237)          */
       MT_EPI_QKV_HM = 4 };    /* C = acc + bias, written head-major [N/48][M][48] (fp16): the q|k|v layout the
                                  dilated-attention kernels read (DA:169-175 "b l (h d) -> (b h) l d")          */
enum { MT_OUT_F16 = 0, MT_OUT_F32 = 1 };

/* Train-mode stochastic ops (nn.Dropout, timm DropPath: ENC:149-152,169-170, FFN:142, ENC:339, AM:319-327) as
 * counter-based masks: nothing is stored, the backward regenerates the mask of a site from (seed, step, site, element
 * index) with Philox4x32 (7 rounds).  rng: device uint32[4] = {seed_lo, seed_hi, step, reserved}; mt_rng_advance
 * increments `step` once per train step (inside the captured graph).  Element dropout: element (m, n) of a dense [M, D]
 * activation is kept when its random word >= p * 2^32 and scaled by 1/(1-p).  DropPath: one Bernoulli per (path_site,
 * pass), pass = m / rows_per_pass; a dropped pass is zeroed, a kept one scaled by 1/(1-path_p).  rng == NULL or both
 * probabilities 0: identity. */
typedef struct {
  const unsigned* rng;
  unsigned site;
  float p;
  unsigned path_site;
  float path_p;
  int rows_per_pass;
} MtDropout;
int mt_rng_advance(unsigned* rng, mt_stream_t stream);
/* y(m,:) = drop(x(xmap(m),:)) for a dense fp32 y [M, D] (Encoder.prepare_forward's input dropout, ENC:339: one mask
 * per task pass over the shared patch embedding) */
int mt_dropout_f32(const float* x, long ldx, const MtRowMap* xmap, float* y, int M, int D, const MtDropout* drop,
                   mt_stream_t stream);
/* in place on rows: x(m,:) *= DropPath factor of pass m / rows_per_pass (token-side residual branches) */
int mt_droppath_rows_f32(float* x, int M, int D, const MtDropout* drop, mt_stream_t stream);

typedef struct {
  const float* bias;           /* [N] or NULL */
  const float* resid;          /* fp32 [*, ldr] (BIAS_RESID, INJECT) */
  long ldr;
  MtRowMap rmap;
  const float* colscale;       /* gamma [N] (INJECT) */
  const float* pos_table;      /* [ngrids, N/2] 1-D sin-cos table (POSEMB), pos_embed.py:62-81 */
  const int* pos_row;          /* [M] grid row index  floor(coords[:,0]/256), SE:209-211 */
  const int* pos_col;          /* [M] grid col index */
  MtDropout drop;              /* BIAS_RESID: C = resid + drop(acc + bias)  (ENC:149-154: dropout, DropPath, + residual) */
} MtGemmEpilogue;

/* C[M,N] = epilogue(A[M,K] @ W[N,K]^T); A, W fp16; fp32 accumulate on MFMA.  K % 64 == 0, lda % 8 == 0.
 * Replaces every big-M nn.Linear on the path: PatchEmbed.proj (SE:52-56), q/k/v/out_proj (DA:169-171,260),
 * fc1/fc2 (FFN:134,140), adapter q_proj/output_proj/k|v projections (AM:154-164,221-231), and — with a
 * pre-transposed weight — the activation-gradient (dX) GEMMs of the frozen backbone. */
int mt_gemm_nt_f16(const mt_half* A, long lda, const MtRowMap* amap, const mt_half* W, int M, int N, int K,
                   int epilogue, const MtGemmEpilogue* epi, void* C, long ldc, const MtRowMap* cmap, int out_dtype,
                   mt_stream_t stream);

/* C[N1,N2] (fp32) += sum_m A[m,N1] * B[m,N2] over M rows (split over workgroups, fp32 atomics): the weight
 * gradient of a big-M nn.Linear (autograd of AM:154-164 etc.).  N1 % 64 == 0, N2 % 64 == 0.
 * colsum (fp32 [N1], or NULL): colsum[n] += sum_m A[m,n] on the same pass -- the bias gradient of that nn.Linear. */
int mt_gemm_tn_f16(const mt_half* A, long lda, const MtRowMap* amap, const mt_half* B, long ldb, const MtRowMap* bmap,
                   int M, int N1, int N2, float* C, long ldc, float* colsum, mt_stream_t stream);

/* out[n] (fp32) += sum_m A[m,n]: bias gradient of a big-M nn.Linear.  N % 8 == 0, lda % 8 == 0 (16-byte loads). */
int mt_colsum_f16(const mt_half* A, long lda, const MtRowMap* amap, int M, int N, float* out, mt_stream_t stream);

/* Generic small strided fp32 GEMM for the token-side ops (T <= 66 rows; gene encoder GE:194-223, prompt
 * self-attention projections AM:81-94, extractor FFN AM:284-287, fusion head LVA:343-347), forward and
 * backward:  C(m,n) = act(sum_k A(m,k) B(n,k) + bias[n]) [+ C(m,n) if accumulate].
 * Element (i,j) of X is X[i*xs0 + j*xs1].  batch: pointer offsets a_bs/b_bs/c_bs per batch index.
 * rowsum (fp32 [M], or NULL; batch must be 1): rowsum[m] += sum_k A(m,k) -- the bias gradient riding on the
 * dW = dy^T x product of an nn.Linear backward. */
enum { MT_ACT_NONE = 0, MT_ACT_RELU = 1, MT_ACT_GELU = 2, MT_ACT_ELU = 3 };
int mt_sgemm_small(const float* A, long as0, long as1, long a_bs, const float* B, long bs0, long bs1, long b_bs,
                   const float* bias, int bias_on_m, float* C, long cs0, long cs1, long c_bs, int M, int N, int K,
                   int batch, int act, int accumulate, float* rowsum, mt_stream_t stream);

/* Up to MT_SGEMM_MAX independent small products in ONE launch, each with the nn.Linear forward / backward fused around it
 * (the token side is a chain of ~5 us launches; what it costs is their number):
 *   forward   C = resid + drop_c(act(A B^T + bias)) [+ C];  pre_out (or NULL) receives A B^T + bias, the value the
 *             backward differentiates -- nn.Linear + activation + nn.Dropout + residual add of GE:184-192, AM:284-287;
 *   backward  A'(m,k) = drop_a(A(m,k)) * act'(a_aux(m,k)) is formed as the dy operand is loaded (a_aux = the saved
 *             pre_out, a_drop = the forward's c_drop): dX = A' W and dW = A'^T x (+ rowsum = db) of one nn.Linear share a launch.
 * pre_out / resid are addressed like C, a_aux like A.  The masks are element dropout of a DENSE tensor: the mask index
 * is the element offset from C / from A.  DropPath of the branch (path_p > 0: one Bernoulli per task pass of rows_per_pass
 * rows, AM:319,327) rides on the same specs: in c_drop the pass is (row m of C) / rows_per_pass; in a_drop it is
 * (element offset in A / a_ld) / rows_per_pass with a_ld = the row length of the dense dy tensor (required then).
 * resid_scale: the residual enters as resid_scale * resid (0 is read as 1; `query + (tgt + f(.))` with tgt == query is
 * 2 * query + f(.), AM:231,324).  Products of one launch must not write what another reads or writes.  mt_sgemm_small is the
 * one-product, no-fusion form of this entry. */
#define MT_SGEMM_MAX 3
typedef struct {
  const float* A; long as0, as1, a_bs;
  const float* B; long bs0, bs1, b_bs;
  const float* bias; int bias_on_m;
  float* C; long cs0, cs1, c_bs;
  int M, N, K, batch, act, accumulate;
  float* rowsum;
  float* pre_out;
  const float* resid;
  MtDropout c_drop;
  const float* a_aux; int a_act;
  MtDropout a_drop;
  float resid_scale; int a_ld;
} MtSgemm;
int mt_sgemm_multi(const MtSgemm* probs, int n, mt_stream_t stream);

/* ------------------------------------------------------------- LayerNorm --------------------------- */
/* y = LN(f(x)) * w + b over the last dim D (eps 1e-5), one wave per row; stats (mean, rstd) saved per row.
 * f = identity or erf-GELU computed in fp32 (FFN:136 forces the activation to fp32, then FFN:138 ffn_layernorm).
 * in_dtype/out_dtype: MT_OUT_F16 / MT_OUT_F32.  D in {256, 768, 2304, 3072}.  `add_rows` (fp32 [add_period, D],
 * or NULL) is added AFTER the affine: LN(c) + pe as in AM:218,227 (with_pos_embed on normed memory). */
int mt_layernorm_fwd(const void* x, long ldx, const MtRowMap* xmap, int in_dtype, int gelu_in, const float* w,
                     const float* b, const float* add_rows, int add_period, void* y, long ldy, const MtRowMap* ymap,
                     int out_dtype, float* stats, int M, int D, mt_stream_t stream);
/* h = x + drop(branch) ; y = fp16(LN(h) * w + b): the residual add of a backbone sub-layer (ENC:95-97,121-123 --
 * x = residual stream fp32 [M, D], branch = fp16 output of the out_proj / fc2 GEMM, drop = that branch's Dropout +
 * DropPath, same counters as the BIAS_RESID GEMM epilogue) riding on the LayerNorm that consumes the sum.
 * h (fp32, != x) receives the new residual stream, stats the (mean, rstd) rows.  D = 768. */
int mt_add_layernorm_fwd(const float* x, const mt_half* branch, const MtDropout* drop, const float* w, const float* b,
                         float* h, mt_half* y, float* stats, int M, int D, mt_stream_t stream);

/* The same two launchers with an explicit epsilon (the entries above use the reference's 1e-5, ENC / AM / GE LayerNorms; the
 * TITAN ViT's LayerNorms carry their own `eps`, e.g. timm's 1e-6).  The backward needs no epsilon: it reads the saved rstd. */
int mt_layernorm_fwd_eps(const void* x, long ldx, const MtRowMap* xmap, int in_dtype, int gelu_in, const float* w,
                         const float* b, const float* add_rows, int add_period, void* y, long ldy, const MtRowMap* ymap,
                         int out_dtype, float* stats, int M, int D, float eps, mt_stream_t stream);
int mt_add_layernorm_fwd_eps(const float* x, const mt_half* branch, const MtDropout* drop, const float* w, const float* b,
                             float* h, mt_half* y, float* stats, int M, int D, float eps, mt_stream_t stream);

/* dx (+)= LN backward.  dy fp16 or fp32 [M,D]; x as in forward (gelu_in: also backprop through the GELU).
 * dx_dtype F32 with accumulate=1 adds into the fp32 residual-gradient stream (ENC:137-154 backward);
 * dw/db (fp32 [D], atomically accumulated) may be NULL for frozen norms (selective backward).
 * dx_f16 (dense fp16 [M,D], or NULL): a second, half-precision copy of the final dx -- the operand of the next dX GEMM
 * of the frozen layer below, written here instead of by a separate cast pass over the fp32 stream; dx_f16_drop (or
 * NULL): the dropout / DropPath mask of the residual branch that GEMM differentiates, applied to the copy only. */
int mt_layernorm_bwd(const void* dy, long lddy, const MtRowMap* dymap, int dy_dtype, const void* x, long ldx,
                     const MtRowMap* xmap, int in_dtype, int gelu_in, const float* w, const float* stats, void* dx,
                     long lddx, const MtRowMap* dxmap, int dx_dtype, int accumulate, float* dw, float* db,
                     mt_half* dx_f16, const MtDropout* dx_f16_drop, int M, int D, mt_stream_t stream);

/* ------------------------------------------------------- dilated attention ------------------------- */
#define MT_MAX_BRANCHES 8
#define MT_QK_SCALE_LOG2 0.20823509396846288f     /* 48^-1/2 * log2(e) */
typedef struct {
  int nbranch;
  int N;                              /* tokens per pass (L + 1) */
  int B;                              /* task passes batched */
  int seg[MT_MAX_BRANCHES];           /* s = min(segment_length, N)                  DA:96-98   */
  int ratio[MT_MAX_BRANCHES];         /* dilation r                                  DA:213     */
  int nseg[MT_MAX_BRANCHES];          /* ceil(N / s)                                            */
  int n[MT_MAX_BRANCHES];             /* sparse length ceil(s / r) incl. zero padding DA:22-37  */
  int qlimit[MT_MAX_BRANCHES];        /* 0, or: only the first qlimit sparse entries of every (segment, head) act as QUERIES
                                       * (all n act as keys) -- the sequence-parallel form of DA:61-111, where a rank's
                                       * queries attend over the keys gathered from the ranks of its segment            */
} MtDilatedPlan;

/* qkv: fp16 HEAD-MAJOR [3][16][B*N][48] (q' | k | v; MT_EPI_QKV_HM writes it).  The q slab holds
 * q' = MT_QK_SCALE_LOG2 * q with MT_QK_SCALE_LOG2 = 48^-1/2 * log2(e): the caller bakes the softmax scale (MHA:109-119
 * scaling = head_dim^-0.5; in log2 units because the kernels exponentiate with v_exp_f32 = 2^x) into the q rows of the
 * frozen q_proj weight and bias before rounding them to fp16, so no kernel multiplies by it and q is rounded once.
 * For every branch, every (segment, head): O_b = softmax(Q K^T / sqrt(48)) V over the head's dilated positions, zero-padded rows acting as keys with
 * logit 0 / value 0 (DA:98-101,24-28).  o_br: fp16 [nbranch][B*N, 768]; lse_br: fp32 [nbranch][B*N, 16]
 * (natural log).  (position, head) pairs a branch does not visit are left untouched.  DA:212-253, MHA:109-119. */
int mt_dilated_attn_fwd(const mt_half* qkv, const MtDilatedPlan* plan, mt_half* o_br, float* lse_br,
                        mt_stream_t stream);

/* Branch mix + inner_attn_ln: w_b = softmax_b(lse_b) per (position, head) (constant in backward, DA:132-137),
 * mixed = sum_b w_b O_b, y = LN(mixed) (DA:257-258).  Writes y fp16 [B*N,768], LN stats, and
 * lse_tot = logsumexp_b(lse_b) fp32 [B*N,16] for the backward. */
int mt_dilated_mix_ln_fwd(const mt_half* o_br, const float* lse_br, const MtDilatedPlan* plan, const float* ln_w,
                          const float* ln_b, mt_half* y, float* stats, float* lse_tot, mt_stream_t stream);

/* Backward of mix + inner_attn_ln: given dy (fp16, gradient wrt the LN output) recomputes mixed, writes
 * dmixed fp16 HEAD-MAJOR [16][B*N][48] and delta_br[b][row][head] = sum_d dmixed * O_b (fp32). */
int mt_dilated_mix_ln_bwd(const mt_half* dy, const mt_half* o_br, const float* lse_br, const float* lse_tot,
                          const MtDilatedPlan* plan, const float* ln_w, const float* stats, mt_half* dmixed,
                          float* delta_br, mt_stream_t stream);

/* Flash-style backward of all branches: dqkv fp16 [B*N, 2304] (overwritten) from qkv, dmixed, lse_tot, delta_br.  The q
 * columns of dqkv are the gradient with respect to the PRE-SCALED q' (what the dX GEMM through the scaled q_proj cache needs).
 * P~ = exp(s - lse_tot) (= w_b P_b), dS = P~ (dmixed V^T - delta_b), dQ = dS K, dK = dS^T Q, dV = P~^T dmixed.
 * Two launches cover all branches (a dK/dV kernel with key = lane, a dQ kernel with query = lane); each writes its
 * per-branch result once into `workspace` (fp16, mt_dilated_attn_bwd_workspace_bytes) and a combine kernel sums the
 * branches that visit a (position, head) into the dense fp16 gradient: no atomics, bitwise reproducible. */
long mt_dilated_attn_bwd_workspace_bytes(const MtDilatedPlan* plan);
enum { MT_ATTN_BWD_KV = 1, MT_ATTN_BWD_Q = 2, MT_ATTN_BWD_COMBINE = 4, MT_ATTN_BWD_ALL = 7 };   /* `phases` mask */
int mt_dilated_attn_bwd(const mt_half* qkv, const mt_half* dmixed, const float* lse_tot, const float* delta_br,
                        const MtDilatedPlan* plan, void* workspace, mt_half* dqkv, int phases, mt_stream_t stream);

/* ------------------------------------------------- dense attention with 2-D ALiBi (TITAN blocks) ---- */
/* The attention of the TITAN slide encoder's ViT blocks (titan_adapter.py:253-293 `get_alibi`, :359-361,394
 * `blocks.modules_list[i](x, attn_bias, bg_mask)`; adapter_modules.py:526-558): H heads of 64 over one sequence of N tokens
 * per pass (cls + the slide's foreground cells), softmax(q k^T / 8 + bias) v with bias[h, i, j] = -slope_h * euclidean distance
 * of the tokens' grid cells (0 to and from cls).  The distance is head-independent: ONE fp16 [N, N] table per slide
 * (mt_alibi_dist), stored in the order the kernels' accumulator registers consume it, replaces the reference's [H, N, N] fp32
 * bias; the kernels read it straight into registers and apply the slope with one fused multiply-add per score.
 * qkv: fp16 TOKEN-MAJOR [B*N, 3*H*64] (q' | k | v) as the qkv GEMM writes it, q' = MT_DENSE_QK_SCALE_LOG2 * q (the caller
 * bakes 64^-1/2 log2(e) into the q rows of the frozen qkv weight / bias).  o: fp16 [B*N, H*64]; lse: fp32 [B*N, H] (natural
 * log, bias included). */
#define MT_DENSE_QK_SCALE_LOG2 0.18033688011112042f     /* 64^-1/2 * log2(e) */
typedef struct {
  int N;                 /* tokens per pass */
  int B;                 /* task passes batched */
  int H;                 /* heads (head dim 64) */
  const mt_half* dist;   /* distance table of mt_alibi_dist (mt_alibi_dist_halves(N) halves), or NULL: no bias */
  const float* nslope;   /* [H]: -slope_h * log2(e) (with dist) */
} MtDensePlan;
/* table[...] = euclidean distance between the grid cells of tokens i and j (token 0 = cls: zero row and column), from
 * cells[i - 1] = (row, col) of token i's cell, rounded to fp16 (relative 2^-12: a bias error below 2.5e-4 of the bias), in
 * blocks of 32 x 64 tokens laid out as four 16-byte pieces per lane of a wave (csrc/dense_attn.hip: DistRegs).  Symmetric, so
 * one table serves the query-in-lane and the key-in-lane kernels.  mt_alibi_dist_halves(N) = its size in fp16 elements
 * (N^2 rounded up to whole blocks: 34 MB at N = 4097). */
long mt_alibi_dist_halves(int N);
int mt_alibi_dist(const int* cells, int N, mt_half* table, mt_stream_t stream);
int mt_dense_attn_fwd(const mt_half* qkv, const MtDensePlan* plan, mt_half* o, float* lse, mt_stream_t stream);
/* dqkv fp16 [B*N, 3*H*64] (overwritten; q columns = gradient of the pre-scaled q') from qkv, o, dO (fp16 [B*N, H*64]) and lse.
 * delta: fp32 [B*N, H] workspace (sum_d dO * O, written by the DELTA phase).  The bias carries no gradient.  Every output
 * element is written exactly once: no atomics, no combine pass. */
enum { MT_DENSE_BWD_DELTA = 1, MT_DENSE_BWD_KV = 2, MT_DENSE_BWD_Q = 4, MT_DENSE_BWD_ALL = 7 };
int mt_dense_attn_bwd(const mt_half* qkv, const mt_half* o, const mt_half* d_o, const float* lse, const MtDensePlan* plan,
                      float* delta, mt_half* dqkv, int phases, mt_stream_t stream);

/* y = gelu(x) (erf form, nn.GELU) / dx = dy * gelu'(x) on fp16 vectors, n % 8 == 0: the ViT block's MLP activation */
int mt_gelu_f16_fwd(const mt_half* x, mt_half* y, long n, mt_stream_t stream);
int mt_gelu_f16_bwd(const mt_half* x, const mt_half* dy, mt_half* dx, long n, mt_stream_t stream);

/* Attentional pooling core (TA:401-402 `forward_attn_pool`): nq learned queries attend over the N tokens of every pass, keys
 * split over workgroups (flash-decoding form).  q fp32 [nq, E] (projected; frozen, so no dq); kv fp16 [B*N, 2E] (k | v
 * projected); out fp32 [B, nq, E]; scores fp32 [B, heads, nq, N] (raw scaled logits, saved) and lse fp32 [B, heads, nq]: the
 * backward recomputes p = exp(score - lse); workspace: mt_pool_attn_workspace_floats() floats.
 * backward: dkv fp16 [B*N, 2E] (overwritten) from scores, lse, out and dout (fp32 [B, nq, E]). */
long mt_pool_attn_workspace_floats(int B, int N, int heads, int nq);
int mt_pool_attn_fwd(const float* q, const mt_half* kv, int B, int N, int E, int heads, int nq, float* out, float* scores,
                     float* lse, float* workspace, mt_stream_t stream);
int mt_pool_attn_bwd(const float* q, const mt_half* kv, const float* scores, const float* lse, const float* out,
                     const float* dout, int B, int N, int E, int heads, int nq, mt_half* dkv, mt_stream_t stream);

/* ------------------------------------------------- composite launchers: one backbone layer per call ---- */
/* The launch list of ONE frozen backbone layer behind one entry point per direction (mt_lnqkv_fwd, mt_dilated_attn_*,
 * mt_mix_ln_outproj_*, mt_ffn_* as one sequence).  Exactly the launches listed above, in order, with the same arguments -- results
 * are bit-identical to issuing them one by one; the host makes 2 calls per layer and step instead of ~20.
 * LongNet EncoderLayer (ENC:121-175; DA:146-262; FFN:132-143), D = 768, F = ffn width.  Weights: fp32 LayerNorm affines and
 * biases, fp16 [N, K] weight caches (q rows of w_qkv / b_qkv pre-scaled by MT_QK_SCALE_LOG2) and their transposes for the dX GEMMs.
 * Buffers: saved per layer (hin, hmid fp32 [M, D]; qkv fp16 head-major; o_br fp16 [nbranch, M, D]; lse_br fp32 [nbranch, M, 16];
 * lse_tot fp32 [M, 16]; a1 fp16 [M, F]; st1 / stin / st2 / stf fp32 [M, 2]) and transients shared by all layers.
 * forward: pend_x / pend_branch / pend_drop: the residual stream, fp16 fc2 branch and its dropout of the layer BELOW whose add is
 * still outstanding (hin is then written here) or NULL; defer != 0: leave this layer's fc2 add to the layer above (br16 holds the
 * branch, `out` is not written).  backward: dh (fp32 [M, D]) in / out; dh16_valid: the layer above left fp16(dh) in dh16;
 * feeds_lower: leave fp16(dh) (masked with drop_lower_ffn) for the layer below. */
typedef struct {
  const float *ln1_w, *ln1_b, *inner_ln_w, *inner_ln_b, *ln2_w, *ln2_b, *ffn_ln_w, *ffn_ln_b;
  const float *b_qkv, *b_out, *b_fc1, *b_fc2;
  const mt_half *w_qkv, *w_out, *w_fc1, *w_fc2;
  const mt_half *wt_qkv, *wt_out, *wt_fc1, *wt_fc2;
} MtLongNetLayerWeights;
typedef struct {
  float *hin, *hmid; mt_half *qkv, *o_br; float *lse_br, *lse_tot; mt_half* a1; float *st1, *stin, *st2, *stf;
  mt_half *u16, *br16, *t16;
  float* dh; mt_half *dy16, *dh16, *dt16, *da1, *dmixed, *dqkv16; float* delta; void* attn_ws;
} MtLongNetLayerBuffers;
int mt_longnet_layer_fwd(const MtLongNetLayerWeights* w, const MtLongNetLayerBuffers* b, const MtDilatedPlan* plan, int M, int D, int F,
                         const float* pend_x, const mt_half* pend_branch, const MtDropout* pend_drop, int defer, float* out,
                         const MtDropout* drop_attn, const MtDropout* drop_ffn, mt_stream_t stream);
int mt_longnet_layer_bwd(const MtLongNetLayerWeights* w, const MtLongNetLayerBuffers* b, const MtDilatedPlan* plan, int M, int D, int F,
                         int dh16_valid, int feeds_lower, const MtDropout* drop_attn, const MtDropout* drop_ffn,
                         const MtDropout* drop_lower_ffn, mt_stream_t stream);
/* Dense pre-norm ViT block of the TITAN configuration (TA:359-361: x + proj(attn(LN(x))), then + fc2(gelu(fc1(LN(.)))); layer scale
 * folded into w_proj / w_fc2 by the caller): qkv fp16 TOKEN-major [M, 3D], o16 fp16 [M, D], lse fp32 [M, H], a1 fp16 [M, F]. */
typedef struct {
  const float *n1_w, *n1_b, *n2_w, *n2_b; float n1_eps, n2_eps;
  const float *b_qkv, *b_proj, *b_fc1, *b_fc2;
  const mt_half *w_qkv, *w_proj, *w_fc1, *w_fc2;
  const mt_half *wt_qkv, *wt_proj, *wt_fc1, *wt_fc2;
} MtVitBlockWeights;
typedef struct {
  float *hin, *hmid; mt_half *qkv, *o16; float* lse; mt_half* a1; float *st1, *st2;
  mt_half *u16, *br16, *t16;
  float* dh; mt_half *dy16, *dh16, *dt16, *da1, *dqkv16; float* delta;
} MtVitBlockBuffers;
int mt_vit_block_fwd(const MtVitBlockWeights* w, const MtVitBlockBuffers* b, const MtDensePlan* plan, int M, int D, int F,
                     const float* pend_x, const mt_half* pend_branch, int defer, float* out, mt_stream_t stream);
int mt_vit_block_bwd(const MtVitBlockWeights* w, const MtVitBlockBuffers* b, const MtDensePlan* plan, int M, int D, int F,
                     int dh16_valid, int feeds_lower, mt_stream_t stream);

/* ------------------------------------------------------------ adapter ops -------------------------- */
/* Injector attention core (AM:225-229 inside AM:359-369): for each of M patch rows and 12 heads (dim 16):
 * a = softmax(q k^T / 4) v over the T modal tokens of the row's pass.  q fp16 [M,192]; k,v fp32 [B,T,192];
 * lse fp32 [M,12] (log-sum-exp of the scaled logits, saved for the backward; may be NULL for inference). */
int mt_inject_attn_fwd(const mt_half* q, int M, int rows_per_pass, const float* k, const float* v, int T,
                       mt_half* a, float* lse, mt_stream_t stream);
/* backward (a, lse from the forward): dq fp16 [M,192]; dk, dv fp32 [B,T,192] accumulated with atomics (pre-zeroed
 * by the caller).  The reductions over the patch rows run on MFMA. */
int mt_inject_attn_bwd(const mt_half* q, const mt_half* a, const float* lse, const mt_half* da, int M,
                       int rows_per_pass, const float* k, const float* v, int T, mt_half* dq, float* dk, float* dv,
                       mt_stream_t stream);

/* Extractor attention core (AM:225-229 inside AM:321-335): T token queries attend over the L patch rows of their
 * pass.  q fp32 [B,T,192]; kv fp16 [B*L, 384] (k | v).  Split over L (flash-decoding style): part_* are
 * workspaces of nsplit partials; out fp32 [B,T,192]; lse fp32 [B,T,12]. */
int mt_extract_attn_fwd(const float* q, const mt_half* kv, int B, int T, int L, float* out, float* lse,
                        float* part_acc, float* part_ml, int nsplit, mt_stream_t stream);
/* backward: dq fp32 [B,T,192] (atomics, pre-zeroed), dkv fp16 [B*L,384] */
int mt_extract_attn_bwd(const float* q, const mt_half* kv, const float* out, const float* lse, const float* dout,
                        int B, int T, int L, float* dq, mt_half* dkv, mt_stream_t stream);

/* Pathway networks of the gene encoder, all G pathways in one launch (gene_encoder.py:97-131,194-207: per pathway
 * SNN_Block(n_i -> latent), SNN_Block(latent -> latent), ELU, AlphaDropout off):
 *   z[i] = ELU(W2_i ELU(W1_i g_i + b1_i) + b2_i).
 * params / grads: the flat fp32 parameter / gradient buffers; offs [G][4] = element offsets of (W1_i [latent, n_i],
 * b1_i, W2_i [latent, latent], b2_i) (W2 offsets multiples of 4); sizes [G] = n_i; goff [G] = offset of g_i in the
 * concatenated `genes` vector; latent must be 256.  passes P (1..4): the reference calls the model once per task, so in
 * train mode every task pass draws its own AlphaDropout masks; the weights are streamed once for all passes.
 * z, a2 (second pre-activation, saved) are [G, P, latent] (pathway-major: the mixer's group axis stays outermost); a1 (first
 * pre-activation: the same in every pass) [G, latent].  alpha_drop (or NULL): train-mode nn.AlphaDropout(p) after each ELU, sites
 * alpha_drop->site and site + 1; the mask index of pathway i, pass p, unit j is (i P + p) latent + j. */
int mt_gene_snn_fwd(const float* params, const long* offs, const int* sizes, const long* goff, const float* genes, int G,
                    int latent, int passes, float* a1, float* a2, float* z, const MtDropout* alpha_drop, mt_stream_t stream);
/* backward: grads (+)= dW1, db1, dW2, db2 for every pathway, summed over the passes, given dz [G, P, latent] (no input
 * gradient: genes are data) */
int mt_gene_snn_bwd(const float* params, float* grads, const long* offs, const int* sizes, const long* goff,
                    const float* genes, int G, int latent, int passes, const float* a1, const float* a2, const float* dz,
                    const MtDropout* alpha_drop, mt_stream_t stream);

/* Small dense multi-head attention over tokens (prompt self-attention AM:87): q,k,v fp32 [B,T,E], heads h. */
int mt_token_mha_fwd(const float* q, const float* k, const float* v, int B, int T, int E, int heads, float* out,
                     float* probs, mt_stream_t stream);
int mt_token_mha_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout, int B,
                     int T, int E, int heads, float* dq, float* dk, float* dv, mt_stream_t stream);

/* ------------------------------------------------------------ elementwise -------------------------- */
/* y = fp16(x); with `drop` (rows of D elements): y = fp16(drop(x)) -- the masked gradient of a dropped residual branch */
int mt_cast_f32_to_f16(const float* x, mt_half* y, long n, const MtDropout* drop, int D, mt_stream_t stream);
int mt_cast_f16_to_f32(const mt_half* x, float* y, long n, mt_stream_t stream);
/* Derived fp16 weight cache of an fp32 nn.Linear weight [R, C]: as stored (forward, W[N,K]) or transposed
 * (the dX GEMM's W^T[K,N]); re-run for trainable weights after every optimiser step (SURVEY §8b ownership). */
int mt_pack_weight_f16(const float* src, int R, int C, mt_half* dst, int transpose, mt_stream_t stream);
/* The same for a whole list of weights in ONE launch (the per-step refresh of every trainable big-M weight).
 * `items` is a DEVICE array of n_items records of 8 int64 each:
 *   { src fp32 [rows, cols] , dst fp16 (or 0) , dst_t fp16 (or 0) , rows , cols , row_off , ld , ld_t }
 * dst[(row_off + r) * ld + c] = dst_t[c * ld_t + row_off + r] = fp16(src[r * cols + c]); row_off / ld_t place the rows
 * of one source inside a cache that concatenates several (the fused K|V projection). */
int mt_pack_weights_f16(const long long* items, int n_items, mt_stream_t stream);
/* y = act(x) / dx = dy * act'(x) on fp32 vectors (ELU GE:178, GELU GE:187, ReLU AM:286) */
int mt_act_fwd(const float* x, float* y, long n, int act, mt_stream_t stream);
int mt_act_bwd(const float* x, const float* dy, float* dx, long n, int act, mt_stream_t stream);
/* y[i] = a[i] + alpha * b[i] */
int mt_axpy(const float* a, const float* b, float alpha, float* y, long n, mt_stream_t stream);
/* y[i] = a[i] + alpha * b[i % period]: a [period] row block broadcast over the batch (with_pos_embed, AM:64-65) */
int mt_axpy_bcast(const float* a, const float* b, float alpha, float* y, long n, long period, mt_stream_t stream);
/* out[i] += sum_r x[r * period + i] (r ascending): the gradient of the broadcast operand of mt_axpy_bcast and of the rows
 * mt_layernorm_fwd adds per period (d level / position embeddings summed over the task passes, AM:64-65); period % 4 == 0,
 * 16-byte aligned pointers */
int mt_fold_rows(const float* x, int reps, long period, float* out, mt_stream_t stream);
/* strided row copies between fp32 buffers: dst(map(m), :) (+)= src(map(m), :) */
int mt_copy_rows_f32(const float* src, long lds, const MtRowMap* smap, float* dst, long ldd, const MtRowMap* dmap,
                     int M, int D, int accumulate, mt_stream_t stream);
/* Injector residual-path backward (A.1): dres(m,:) += (1 + g) * dy(m,:) for the patch rows; also reduces
 * dgamma[n] += sum_m dy(m,n) * (x(m,n) + proj(m,n)) where proj = a @ Wo^T + bo is recomputed by the caller as
 * fp16 `proj`.  */
int mt_inject_resid_bwd(const float* dy, long lddy, const MtRowMap* dymap, const float* x, long ldx,
                        const MtRowMap* xmap, const mt_half* proj, const float* gamma, float* dx, long lddx,
                        const MtRowMap* dxmap, int dx_accumulate, mt_half* dproj, float* dgamma, int M, int D,
                        mt_stream_t stream);

/* ---------------------------------------------------------- head, loss, optimiser ------------------ */
/* y[r,:] = x[r,:] / ||x[r,:]||_2 for R contiguous rows of O (projected text rows, TM:213) */
int mt_l2norm_rows(const float* x, float* y, int R, int O, mt_stream_t stream);
/* logits [R,O] -> L2-normalise rows, log_softmax, KL(sum) against softmax(target rows) * 10 (TM:225-233).
 * Writes loss (1 float) and dlogits [R,O] scaled by `loss_scale` (GradScaler semantics, TM:107,235). */
int mt_distill_loss(const float* logits, const float* target, int R, int O, float loss_scale,
                    const float* scale_dev /* device scalar multiplied into loss_scale, or NULL */, float* loss,
                    float* dlogits, mt_stream_t stream);

/* Fused multi-tensor AdamW over one flat fp32 parameter/gradient buffer (torch.optim.AdamW, TM:145-149) with
 * GradScaler.step semantics (TM:235-237): grads are divided by *scale; if any is non-finite the update is
 * skipped and *found_inf is set.  step_count is 1-based. */
int mt_adamw_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                  double eps, double weight_decay /* doubles: torch forms 1 - beta, lr * weight_decay and the bias corrections in
                  double before they meet the fp32 tensors (1 - 0.999f is 4.7e-5 off 0.001) */, int step_count, const int* step_dev /* device count of completed
                  steps (overrides step_count - 1) or NULL */, float grad_mult /* e.g. 1/world_size */,
                  const float* scale, int* found_inf, const float* lr_dev /* device scalar: the schedule's current
                  learning rate (overrides lr; a captured graph replays with whatever it holds -- the reference steps
                  GradualWarmupScheduler + CosineAnnealingLR every epoch, TM:151-154,242) or NULL */, mt_stream_t stream);
/* GradScaler.update on device (TM:237): *found_inf ? scale *= backoff : (every `interval` clean steps scale *= growth) */
int mt_scaler_update(float* scale, int* growth_tracker, int* found_inf, int* step_dev /* ++ on a clean step, or NULL */,
                     float growth, float backoff, int interval, mt_stream_t stream);
int mt_check_finite(const float* g, long n, int* found_inf, mt_stream_t stream);

/* Measurement aid (bench.py, SURVEY §8d "the builder's own MFMA micro-benchmark peak"): `workgroups` x 4 waves each issue
 * iters x 4 independent v_mfma_f32_32x32x16_f16 on non-trivial register operands = workgroups * 4 * iters * 4 * 32768 FLOP. */
int mt_mfma_probe(float* sink, int workgroups, int iters, mt_stream_t stream);

/* ---------------------------------------------------------- module bridge / input boundary --------- */
/* s[0] = target / max|x|, s[1] = 1 / s[0] (both 1 when the maximum is 0 or not finite): device-side rescale of the
 * gradient torch hands to the nn.Module bridge (loss.backward(), TM:235) into the fp16 range of the activation-gradient
 * stream, without a host read-back. */
int mt_absmax_scale(const float* x, long n, float target, float* s, mt_stream_t stream);
/* y[i] = a[i] + (*alpha) * b[i] with alpha a DEVICE scalar (a may be NULL: y = (*alpha) * b) */
int mt_axpy_dev(const float* a, const float* b, const float* alpha, float* y, long n, mt_stream_t stream);
/* LongNetViT.coords_to_pos (SE:198-211) on the device: prow[i] = floor(coords[i][0] / tile), pcol[i] = floor(coords[i][1]
 * / tile) for fp32 coords [L, 2]; cells outside [0, ngrids) are clamped and *err (device int, or NULL) is OR-ed with 1 --
 * the host raises when it next looks (the reference would index out of bounds, SE:116-120,237). */
int mt_coords_to_grid(const float* coords, int L, float tile, int ngrids, int* prow, int* pcol, int* err,
                      mt_stream_t stream);
/* TITAN feature gridding on the device (titan_adapter.py:295-327 `preprocess_features` + the background drop of
 * `prepare_forward_features`, TA:282-291) WITHOUT materialising the H x W grid: the tokens of a slide are its occupied cells
 * in row-major order, each the sum of its patches' features in patch order (what index_add_ produces on the CPU, bitwise
 * reproducible: no atomics, no sort -- O(L^2) integer compares through LDS).
 *   mt_titan_grid         cells[i] = floor((coords[i] - min coords) / patch) (fp32 coords [L, 2]); dims (device int[2]) = {H, W};
 *                         *err |= 1 for non-finite / out-of-range coordinates
 *   mt_titan_cell_sums    first[i] / next[i]: chain of the patches of one cell in patch order; sums[i, :] (fp32 [L, C]) = the cell's
 *                         summed features for its first patch; nz[i] = any(sum != 0) (the reference's bg_mask, TA:326)
 *   mt_titan_token_order  pos[i] = token index (without cls) of the cell patch i owns, or -1; cells_tok[pos] = (row, col);
 *                         *count = number of tokens
 *   mt_titan_gather_tokens x16[pos[i], :] = fp16(sums[i, :]): the A operand of the patch-embedding GEMM */
int mt_titan_grid(const float* coords, int L, float patch, int* cells, int* dims, int* err, mt_stream_t stream);
int mt_titan_cell_sums(const float* feat, long ldf, const int* cells, int L, int C, int* first, int* next, float* sums, int* nz,
                       mt_stream_t stream);
int mt_titan_token_order(const int* cells, const int* first, const int* nz, int L, int* pos, int* cells_tok, int* count,
                         mt_stream_t stream);
int mt_titan_gather_tokens(const float* sums, const int* pos, int L, int C, mt_half* x16, mt_stream_t stream);
/* The same index_add as one strided scatter (host-computed indices): the bring-your-own-backbone path of modaltune_amd/titan.py,
 * which must hand the backbone module the dense [1, C, H, W] grid its own `prepare_forward_features` expects.
 * dst(idx[m], :) (+)= src(src_idx[m], :) for fp32 rows of D (src_idx NULL = identity); the caller launches one pass per
 * occurrence rank of a cell, so the cells of a pass are distinct.
 * mt_row_absmax_f32: out[m] = max_d |x(m, d)| (the background mask is "any feature of the cell != 0", TA:326). */
int mt_scatter_rows_f32(const float* src, const int* src_idx, const int* idx, float* dst, int M, int D, int accumulate,
                        mt_stream_t stream);
int mt_row_absmax_f32(const float* x, float* out, int M, int D, mt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MODALTUNE_HIP_H */
