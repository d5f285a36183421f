/* poet_hip.h -- C ABI of libpoet_hip.so: the MI355X (gfx950) kernels of the PoET encoder-decoder
 * hot path.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Contract for EVERY entry point:
 *   - device pointers are caller-owned (the Python host allocates them as torch tensors and
 *     passes data_ptr()); the library allocates nothing and keeps no mutable global state
 *     apart from a thread-local last-error string;
 *   - work is only ENQUEUED on `stream` (a hipStream_t passed as void*); no device sync;
 *   - returns POET_OK (0) or a negative POET_ERR_* code and never throws;
 *     poet_hip_last_error() describes the last failure on the calling thread;
 *   - dtype codes: POET_F32 = fp32 storage, POET_BF16 = bfloat16 storage (fp32 accumulate);
 *   - host pointers are marked _host.
 *
 * What each function replaces in the reference (paths relative to aau-cns/poet):
 *   poet_msda_fwd / poet_msda_bwd
 *       the un-vendored CUDA extension behind `from deformable_attention import MSDeformAttn`
 *       (models/deformable_transformer.py:24): upstream's pybind pair
 *       ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc,
 *       attn_weight, im2col_step) / ms_deform_attn_backward(..., grad_output).
 *   poet_msda_fused_fwd / _bwd
 *       the same sampling with MSDeformAttn.forward's softmax over L*P logits and the
 *       loc = ref + offset/(W,H) arithmetic folded in (models/deformable_transformer.py:201,283).
 *   poet_gemm, poet_gemm_dw_list, poet_gemm_last_path
 *       every nn.Linear / 1x1 nn.Conv2d on the path (deformable_transformer.py:182,185,258,261,
 *       the 4 Linears of MSDeformAttn, nn.MultiheadAttention's in/out projections :253,
 *       pose_estimation_transformer.py:106-122,684-688) and their backward contractions.
 *   poet_linear_bwd
 *       dW, db and dX of the encoder's 256-wide-output Linears in one pass (deformable_transformer.py:193-197,202-203, backward).
 *   poet_ln_fwd / poet_ln_bwd
 *       `x = norm(x + dropout(y))`  (deformable_transformer.py:202-203,195-196,279-287,271-272).
 *   poet_mha_fwd / poet_mha_bwd
 *       nn.MultiheadAttention core (deformable_transformer.py:277-278): LDS-resident Q x Q scores up to 128 queries at head dim
 *       16 / 32 / 64 (backward at head dim 64: 114), a generic wave-per-row form for any head dim and up to 2048 queries.
 *   poet_pos_sine / poet_bbox_sine
 *       models/position_encoding.py:40-60 and :71-84.
 *   poet_groupnorm_* / poet_im2col3x3s2 / poet_col2im3x3s2_add / poet_nchw_to_tokens / poet_tokens_to_nchw
 *       input_proj and the flatten/transposes (pose_estimation_transformer.py:100-135,313-335;
 *       deformable_transformer.py:128-141).
 *   poet_enc_ref_points        deformable_transformer.py:217-230.
 *   poet_pose_finish_fwd/_bwd  class-slice gather + 6D->R (pose_estimation_transformer.py:354-393,434-451).
 *   poet_pose_loss             translation / rotation losses of all decoder layers + their gradients
 *                              (pose_estimation_transformer.py:635-674).
 *   poet_adamw / poet_sqnorm   optimizer.step + clip_grad_norm_ over the flat arenas (engine.py:75-81).
 *   poet_lsa_boxes / poet_match_gather   PoseMatcher's assignment (models/matcher.py:158-229) + the target gather of
 *                              SetCriterion (pose_estimation_transformer.py:649-668), on the device.
 */
#ifndef POET_HIP_H
#define POET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POET_ABI_VERSION 6
#define POET_SQNORM_SCRATCH 1024

#define POET_F32 0
#define POET_BF16 1
#define POET_F16 2   /* IEEE half STORAGE (11 significant bits in the same 2 bytes as bf16's 8): accepted only as the q_dtype of
                        poet_msda_fused_fwd / _bwd -- the sampling offsets and attention logits, values of a few units that no GEMM
                        ever reads as an operand -- produced by poet_gemm with c_dtype = POET_BF16 and c_f16 = 1.  Gradients of
                        such a buffer are bf16. */

#define POET_OK 0
#define POET_ERR_ARG (-1)
#define POET_ERR_UNSUPPORTED (-2)
#define POET_ERR_LAUNCH (-3)

int poet_hip_version(void);
const char* poet_hip_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM with fused prologue/epilogue:  C = epilogue( alpha * opA(A [+A2]) . opB(B) )
 *   A is [M,K] (a_kmajor=0, row stride lda) or stored [K,M] (a_kmajor=1, row stride lda);
 *   B is [N,K] (b_kmajor=0, i.e. nn.Linear weight layout) or stored [K,N] (b_kmajor=1).
 *   compute = POET_BF16 -> v_mfma_f32_16x16x32_bf16, POET_F32 -> v_mfma_f32_16x16x4_f32.
 *   epilogue order: +bias[col] -> act -> gate (gate_ref[row,col] > 0 ? v*gate_scale : 0)
 *                   -> dropout(drop_p, seed; index = row*N+col) -> +add_src[row,col]
 *                   -> row_mask (row_mask[row] != 0 -> 0) -> store.
 *   out_mode 0: C[row*ldc + col].  out_mode 1 (head-major value maps): row=(n,s), col=(m,d) ->
 *               C[((n*hm_M + m)*hm_S + s)*hm_D + d].
 *   batch > 1: A/B/C advance by strideA/B/C elements per batch index (gate_ref / add_src, laid out like C, by strideC:
 *   the <= 1024-row fp32 kernels and the generic tiled kernel; the streaming / long-K kernels take batch 1 only; bias shared
 *   or strided by stride_bias).  splitk > 1 or atomic != 0: fp32 atomicAdd into C (C must be f32, pre-zeroed
 *   or holding the value to accumulate onto); bias/act/gate are then not allowed -- with one
 *   exception, the WEIGHT-GRADIENT FORM (a_kmajor && b_kmajor && atomic: A = dY stored [rows][M],
 *   B = X stored [rows][N], C = dW): there `bias` is an optional fp32 OUTPUT [M] that receives
 *   += the column sums of A, i.e. the bias gradient of the same nn.Linear, produced in the same
 *   pass over dY whenever the streaming kernel handles the shape (batch must be 1).  In this form C
 *   is ACCUMULATED (fp32 atomicAdd, or a single-owner read-modify-write in the small-row kernel): the
 *   caller zero-fills it once per optimisation step and must not touch it concurrently on another
 *   stream.
 * ---------------------------------------------------------------------------------------------- */
typedef struct PoetGemmDesc {
    const void* A;
    const void* A2;        /* optional, same dtype/layout as A, added on load */
    const void* B;
    void* C;
    const float* bias;     /* optional [N]; weight-gradient form: optional fp32 [M] output (see above) */
    const void* add_src;   /* optional, dtype c_dtype, [M, ld_add] */
    const void* gate_ref;  /* optional, dtype c_dtype, [M, ldc] */
    const uint8_t* row_mask; /* optional [M] */
    int32_t M, N, K;
    int64_t lda, ldb, ldc, ld_add;
    int32_t a_kmajor, b_kmajor;
    int32_t a_dtype, b_dtype, c_dtype, compute;
    int32_t batch;
    int64_t strideA, strideB, strideC, stride_bias;
    int32_t splitk, atomic;
    int32_t act;           /* 0 none, 1 relu */
    float alpha;
    float gate_scale;
    float drop_p;
    uint32_t seed;
    int32_t out_mode, hm_M, hm_S, hm_D;
    const uint32_t* seed_dev; /* optional device word XOR-mixed into `seed` at run time: lets a captured hipGraph draw a
                                 fresh dropout mask on every replay (the host bumps the word between replays) */
    int32_t b_split;       /* 1 (B fp32): B is an fp32 [N,K] weight used as bf16 hi + bf16 lo, lo = bf16(B - float(hi)): two
                              v_mfma_f32_16x16x32_bf16 per fragment pair, so the WEIGHT enters with 16 mantissa bits while
                              the activation operand stays plain bf16 (needs b_dtype f32, compute bf16, b_kmajor 0).  Weight
                              rounding is the same perturbation for every token and does not average out downstream the way
                              per-token activation rounding does: DESIGN.md section 3.
                              2 (experimental, round 6; K = 256, >= 4096 rows, 2-byte C): ONE image of the fp32 weight rounded to
                              IEEE fp16 (2^-12) and one v_mfma_f32_16x16x32_f16 per fragment pair, the bf16 activation fragments
                              converted to fp16 in registers (exact; magnitudes past 65504 saturate).  Kernels without this form
                              treat 2 as 1. */
    int32_t c_f16;         /* 1 with c_dtype = POET_BF16: the 2-byte outputs are written as IEEE fp16 instead of bfloat16 (plain or
                              head-major stores; no add_src / gate_ref).  Was reserved0 (= 0) up to ABI version 2. */
    void* workspace;       /* optional scratch owned by the caller (16-byte aligned); used by the weight-gradient form to merge its
                              partial tiles with plain stores + one reduction launch instead of fp32 atomics.  Contents are
                              undefined afterwards; calls sharing it must be ordered on one stream. */
    int64_t workspace_bytes;
    const void* B_lo;      /* optional, with b_split = 1 and a BF16 B: the second bf16 image of the weight (lo = bf16(W - float(hi)),
                              same layout and ldb as B, which then holds hi) -- the two images the optimiser kernel maintains
                              (poet_adamw p_bf16 / p_bf16_lo); the long-K kernel multiplies every activation fragment with both
                              in one pass over A */
    /* ABI version 4: per-segment column sums of A out of the weight-gradient form's own pass over A (a_kmajor = b_kmajor = 1, atomic):
       seg_sums[s][m] += sum of A[r][m] over the rows r with seg_start[s] <= r % seg_period < seg_start[s + 1], s < seg_n <= 8.
       The encoder's d(offsets | logits) rows feed both the bias gradients and d(level_embed) through the sums over the rows of one
       feature LEVEL (deformable_transformer.py:139-141: pos + level_embed[l]); rows of an image are its levels back to back, so
       seg_period = tokens per image, seg_start = the level starts.  NULL = off.  With seg_sums the plain `bias` sum is not formed. */
    float* seg_sums;
    int64_t ld_seg;        /* row stride of seg_sums (elements) */
    int32_t seg_n, seg_period;
    int32_t seg_start[10]; /* seg_n + 1 entries used */
    /* ABI version 4, weight-gradient form: the rows m >= m_alt of C pair with B_alt (row stride ldb_alt) instead of B -- two Linears
       that share their gradient rows' buffer (A = [dY1 | dY2], column blocks of one row-major buffer) but not their input: the
       encoder's stacked [sampling_offsets ; attention_weights ; value_proj] gradient, whose first two blocks pair with the query
       src + pos and the third with src (deformable_transformer.py:199-201).  One launch of 8 tiles instead of 6 + 2 (the 2-tile
       launch is the slow shape of this product) when m_alt is a multiple of 256 and the shape is one the DMA-ring kernel takes;
       two products otherwise.  NULL = off. */
    const void* B_alt;
    int64_t ldb_alt;
    int32_t m_alt, reserved_alt;
} PoetGemmDesc;
int poet_gemm(const PoetGemmDesc* desc, void* stream);
/* Backward of y = x W^T + b for a Linear with a 256-wide OUTPUT, all three results in one pass over dy and x (ABI v4, gemm_dwr.hip):
 *   dw[256][n2] += dy^T x,   db[256] += column sums of dy (db may be NULL),   dx[rows][n2] = (dy w) [x (x > 0 ? gate_scale : 0) if gate].
 * Reference: autograd of nn.Linear at models/deformable_transformer.py:193-197 (FFN linear2; gate = 1: x is the hidden activation after
 * ReLU + dropout, its zeros ARE the gate of :196 / :194) and :202-203 (MSDeformAttn.output_proj; gate = 0).  dy bf16 [rows][256] (row
 * stride ldy), x bf16 [rows][n2], w bf16 [256][n2] (the weight as stored, [out][in]), dw fp32 (accumulated), dx bf16 (written).
 * rows >= 8192, n2 a multiple of 128, 16-byte aligned operands; `workspace` (caller-owned, like PoetGemmDesc.workspace) takes the
 * partial tiles.  POET_ERR_UNSUPPORTED outside that range: issue the two poet_gemm calls instead. */
int poet_linear_bwd(const void* dy, int64_t ldy, const void* x, int64_t ldx, const void* w, int64_t ldw, float* dw, int64_t lddw,
                    float* db, void* dx, int64_t lddx, int gate, float gate_scale, int64_t rows, int n2, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* Which kernel family the calling thread's last successful poet_gemm launched (profiling aid: lets a caller attribute a
 * launch time to the kernel symbol a rocprofv3 trace shows). */
/* dw[i][n_out, k_in] += dy[i][rows, n_out]^T x[i][rows, k_in] and (db != NULL, db[i] != NULL) db[i][n_out] += column sums of
 * dy[i], for n <= 8 fp32 problems of ONE shape (rows <= 1024) whose operands live at unrelated addresses, in one launch:
 * the weight / bias gradient of the same nn.Linear of every decoder layer (models/deformable_transformer.py:253-292). */
int poet_gemm_dw_list(const float* const* dy, const float* const* x, float* const* dw, float* const* db, int n,
                      int n_out, int k_in, int rows, int64_t ldy, int64_t ldx, int64_t ldw, void* stream);
/* (ABI v6) the same for n_lists <= 8 such lists of DIFFERENT shapes in one launch -- every deferred weight gradient of the decoder stack:
 * list j has n problems dw[j * n + i][n_out[j], k_in[j]] += dy[j * n + i][rows, n_out[j]]^T x[j * n + i][rows, k_in[j]] (row strides ldy[j],
 * ldx[j]; dw contiguous; db as above, entries may be NULL); k_in a multiple of 4, x rows 16-byte aligned, else POET_ERR_UNSUPPORTED. */
int poet_gemm_dw_multi(const float* const* dy, const float* const* x, float* const* dw, float* const* db, int n_lists, int n,
                       const int* n_out, const int* k_in, int rows, const int64_t* ldy, const int64_t* ldx, void* stream);
enum { POET_GEMM_PATH_NONE = 0, POET_GEMM_PATH_TILED = 1, POET_GEMM_PATH_STREAM = 2, POET_GEMM_PATH_DW = 3, POET_GEMM_PATH_SMALL = 4,
       POET_GEMM_PATH_PIPE = 5 /* deep-pipeline kernel: plain bf16 x bf16 -> fp32 (+=) products with N = 256, K >= 512 */ };
int poet_gemm_last_path(void);

/* ------------------------------------------------------------------------------------------------
 * Multi-scale deformable attention core (upstream boundary).  Layouts as upstream:
 *   value (N,S,M,D), sampling_loc (N,Lq,M,L,P,2) in [0,1], attn_weight (N,Lq,M,L,P), out (N,Lq,M*D),
 *   spatial_shapes_host (L,2) int64 = (H_l, W_l), level_start_host (L,) int64.
 *   All device tensors share `dtype`.  grad_value is ALWAYS fp32 (N,S,M,D) and is zero-filled by
 *   the call before accumulation (atomics); grad_loc / grad_attn have `dtype`.
 * ---------------------------------------------------------------------------------------------- */
int poet_msda_fwd(const void* value, const int64_t* spatial_shapes_host, const int64_t* level_start_host,
                  const void* sampling_loc, const void* attn_weight, void* out,
                  int N, int S, int M, int D, int L, int P, int Lq, int dtype, void* stream);
int poet_msda_bwd(const void* value, const int64_t* spatial_shapes_host, const int64_t* level_start_host,
                  const void* sampling_loc, const void* attn_weight, const void* grad_out,
                  float* grad_value, void* grad_loc, void* grad_attn,
                  int N, int S, int M, int D, int L, int P, int Lq, int dtype, void* stream);

/* Fused form: `offattn` (N,Lq,ldq) holds the raw outputs of the sampling_offsets Linear at columns
 * [0, M*L*P*2) and of the attention_weights Linear at columns [logit_col, logit_col + M*L*P).
 * ref (Nref,Lq,L,2) fp32 are the reference points already multiplied by valid ratios; Nref is 1
 * (ref_batch_stride = 0) or N.  value is addressed by element strides (vs_n, vs_s, vs_m), so both
 * the upstream (N,S,M,D) and the head-major (N,M,S,D) layouts are accepted.  out (N,Lq,M*D) q_dtype.
 * Limit (all msda entry points): the gather kernels use 32-bit byte offsets and 24-bit pixel arithmetic, so a value map
 * must span < 4 GiB, S < 2^24 and vs_s * sizeof(element) < 2^24; larger calls return POET_ERR_UNSUPPORTED.
 * Backward: grad_value (gv_dtype POET_F32 or POET_BF16) is accumulated atomically (caller zero-fills), addressed by
 * gv_strides = (n, s, m) element strides, NULL = the value strides -- the decoder passes the strides of token-major rows
 * (N*S, layers*M*D) so that the scatter writes what the value projection's backward GEMMs read, with no transposing pass.
 * bf16: packed bf16x2 atomics (even strides, 4-byte aligned base) -- half the memory-side atomics, half the zero-fill and half
 * the read of the consumer, which rounds the value gradient to bf16 anyway.  Grid queries: the per-tile int32 windows are exact
 * and a pixel on a tile border receives up to 4 rounded partial sums; other queries: one rounding per contribution.
 * grad_offattn (N,Lq,ldq) q_dtype receives d/d(offsets) and d/d(logits) (softmax backward folded). */
int poet_msda_fused_fwd(const void* value, int64_t vs_n, int64_t vs_s, int64_t vs_m,
                        const int64_t* spatial_shapes_host, const int64_t* level_start_host,
                        const void* offattn, int64_t ldq, int logit_col,
                        const float* ref, int64_t ref_batch_stride, void* out,
                        int N, int S, int M, int D, int L, int P, int Lq,
                        int v_dtype, int q_dtype,
                        int grid_queries /* 1: query q is pixel q of the flattened levels (encoder self-attention): enables
                                            the kernels that stage the value maps as per-tile LDS windows */,
                        void* stream);
int poet_msda_fused_bwd(const void* value, int64_t vs_n, int64_t vs_s, int64_t vs_m,
                        const int64_t* spatial_shapes_host, const int64_t* level_start_host,
                        const void* offattn, int64_t ldq, int logit_col,
                        const float* ref, int64_t ref_batch_stride, const void* grad_out,
                        void* grad_value, void* grad_offattn,
                        int64_t ld_grad /* row stride of grad_offattn in elements (0: ldq); wider when the gradient rows are
                                           a column block of a larger buffer, e.g. [d(offsets|logits) | d(value) rows] */,
                        int N, int S, int M, int D, int L, int P, int Lq,
                        int v_dtype, int q_dtype, int gv_dtype,
                        int grid_queries /* 1: query q is pixel q of the flattened levels (encoder self-attention):
                                            enables the LDS-privatised value-gradient scatter */,
                        int parts /* 3 (or 0) = everything; 1 = only d(offsets|logits); 2 = only the d(value) scatter
                                     -- lets a profiler time the two kernels of this call separately */,
                        const int64_t* gv_strides_host /* (n, s, m) element strides of grad_value, or NULL */,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * y = LayerNorm(res + dropout(x)) over the last dim d (d % 4 == 0, d <= 1024), one wave per row.
 * Two storage types: the branch side (x, z_out, dx_out) has dtype_x, the residual-stream side
 * (res, y, dy, dz_out) has dtype_r -- (bf16,bf16), (f32,f32) or the mixed (bf16 branch, f32 stream).
 * z_out (optional) receives the pre-norm sum; mean/rstd (fp32 [rows]) are saved for backward.
 * Backward: dz = dLN(dy); d_res := dz (written to dz_out); dx_out (optional, may alias dz_out when
 * drop_p == 0) := dz * keepmask/(1-p).  dgamma/dbeta (fp32 [d]) are ACCUMULATED atomically.
 * ---------------------------------------------------------------------------------------------- */
int poet_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                void* y, void* z_out, float* mean, float* rstd,
                int64_t rows, int d, float eps, float drop_p, uint32_t seed, int dtype_x, int dtype_r /* of res */,
                int dtype_y /* (ABI v6) of y; -1 = dtype_r.  Behind an fp16 branch (dtype_x = POET_F16) res and y may each be POET_F32 or
                               POET_F16 = the SPLIT stream: bf16 head + IEEE fp16 remainder (2^-20 relative).  As output the head is
                               y_bf16 (required; the operand copy written anyway) and y receives fp16(y - float(y_bf16)); as input the
                               pair is (res_bf16, res).  52 MB less written per launch at 102 080 x 256 than an fp32 stream */,
                int dtype_z /* dtype of z_out; -1 = dtype_x.  (f32 x, f32 stream, bf16 z): the branch arrives as the GEMM's
                               fp32 accumulators, only the copy saved for backward is bf16 */,
                void* y_bf16 /* optional: bf16 copy of y, the MFMA operand of the next GEMM */,
                const void* pos_bf16, void* q_bf16 /* optional pair: q_bf16 = bf16(y + pos_bf16), the query operand of the NEXT
                                                      encoder layer (deformable_transformer.py:201 `src + pos`), same shape as y */,
                const void* res_bf16 /* (ABI v6) the bf16 head of a split input stream (dtype_r = POET_F16), else NULL */,
                const uint32_t* seed_dev /* optional, see PoetGemmDesc.seed_dev */, void* stream);
int poet_ln_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                void* dz_out, void* dx_out, float* dgamma, float* dbeta,
                int64_t rows, int d, float drop_p, uint32_t seed, int dtype_x, int dtype_r,
                int dtype_dz /* (ABI v6) storage of dz_out: = dtype_r, or POET_BF16 with bf16 z and fp32 dy -- the point where the
                                encoder's bf16 gradient stream starts */,
                const uint32_t* seed_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small multi-head self-attention core (nn.MultiheadAttention without the projections):
 * q,k,v (N,Q,M*hd) fp32 with row stride ld (so they may be column slices of one packed buffer);
 * out (N,Q,M*hd).  Fast form: hd in {16, 32, 64} and Q <= 128 (one wave up to 64 queries, two waves above; poet_mha_bwd at
 * hd = 64: Q <= 114 -- 2 Q hd + 2 Q (Q + 1) floats of LDS; Q = 128 at hd = 32 runs with swizzled, unpadded score matrices =
 * exactly 160 KB).  Every other shape (any hd >= 1, Q <= 2048: `--num_queries`, `--hidden_dim`, `--nheads`, main.py:94-98)
 * runs the generic form: one wave per query row / key column, probabilities recomputed in the column phase, no Q x Q matrix
 * and no scratch; Q > 2048 returns POET_ERR_UNSUPPORTED.  softmax(q k^T / sqrt(hd)) with dropout(p) on the probabilities
 * (one counter for all forms), no key-padding mask (the reference passes none).
 * ---------------------------------------------------------------------------------------------- */
int poet_mha_fwd(const float* q, const float* k, const float* v, int64_t ld, float* out, int64_t ld_out,
                 int N, int Q, int M, int hd, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);
int poet_mha_bwd(const float* q, const float* k, const float* v, int64_t ld, const float* dout, int64_t ld_out,
                 float* dq, float* dk, float* dv, int64_t ld_d,
                 int N, int Q, int M, int hd, float drop_p, uint32_t seed, const uint32_t* seed_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Position encodings (fp32 math, full-range sinf/cosf).
 * poet_pos_sine: mask (N,H,W) uint8 (1 = padded) -> out token-major (N, tok_stride rows, 2F) written
 *   at rows [tok_off, tok_off+H*W): first F channels y-features, then F x-features
 *   (position_encoding.py:40-60, normalize=True, scale 2*pi); dim_t (F) fp32 is the host-computed
 *   constant table temperature^(2*(i//2)/F) (a 1-ulp difference in it flips sin() where fully
 *   masked rows/columns push the argument to ~1e6, so it is taken as data, not recomputed); optional
 *   level_embed (2F) fp32 is added (deformable_transformer.py:137).  out dtype = dtype.
 * poet_bbox_sine: boxes (n,4) fp32 -> (n, 8F) fp32, [sin(c*2^k) | cos(c*2^k)] per coordinate; rows with
 *   valid[i] == 0 (optional) are the padded dummy queries and get the constant `fill` (-10,
 *   pose_estimation_transformer.py:225-236).
 * ---------------------------------------------------------------------------------------------- */
int poet_pos_sine(const uint8_t* mask, void* out, const float* level_embed, const float* dim_t,
                  int N, int H, int W, int F, int64_t tok_off, int64_t tok_stride, int dtype, void* stream);
int poet_bbox_sine(const float* boxes, const uint8_t* valid, float* out, int n, int F, float fill, void* stream);

/* encoder reference points (deformable_transformer.py:217-230): valid_ratios (N,L,2) fp32 (w,h)
 * -> ref (N,S,L,2) fp32. */
int poet_enc_ref_points(const float* valid_ratios, const int64_t* spatial_shapes_host, float* ref,
                        int N, int L, int S, void* stream);
/* decoder reference points (deformable_transformer.py:317): ref (N,Q,2) x valid_ratios (N,L,2)
 * -> (N,Q,L,2). */
int poet_dec_ref_points(const float* ref, const float* valid_ratios, float* out, int N, int Q, int L, void* stream);
/* valid ratios of a padding mask (deformable_transformer.py:111-118): mask (N,H,W) uint8 ->
 * out[n*out_stride + 0..1] = (valid_w/W, valid_h/H) fp32. */
int poet_valid_ratio(const uint8_t* mask, float* out, int64_t out_stride, int N, int H, int W, void* stream);
/* nearest-neighbour mask resize == F.interpolate(mask[None].float(), size).bool()
 * (pose_estimation_transformer.py:328-329): src index = floor(dst * in / out). */
int poet_mask_nearest(const uint8_t* src, uint8_t* dst, int N, int H, int W, int Ho, int Wo, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / layout helpers.
 * ---------------------------------------------------------------------------------------------- */
int poet_add(const void* a, const void* b, void* out, int64_t n, int dtype_a, int dtype_b, int dtype_out, void* stream);
/* x[row, :] += vec[:] for rows [row0, row0+rows) of every batch item (batch stride in rows). */
int poet_add_rowvec(void* x, const float* vec, int batch, int64_t batch_stride_rows, int64_t row0, int64_t rows,
                    int cols, int dtype, void* stream);
int poet_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);
/* (ABI v6) a_bf16 = bf16(a), sum_bf16 = bf16(a + b): the operand copy of the fp32 stream and the first encoder layer's query operand
 * `src + pos` (deformable_transformer.py:201) in one pass; a fp32, b bf16. */
int poet_add_cast(const float* a, const void* b_bf16, void* sum_bf16, void* a_bf16, int64_t n, void* stream);
/* (ABI v6) y = dropout(gelu(x)) and dx = dy * mask / (1 - p) * gelu'(x), element-wise, erf form (F.gelu): the FFN with
 * `activation="gelu"` (deformable_transformer.py:347-355,193-197,269-273).  dtype (y, dy, dx): POET_F32 or POET_BF16; dtype_x (the
 * kept pre-activation): = dtype, or POET_F32 under bf16 storage (the Linear's accumulators handed over unrounded).  The dropout mask
 * is a counter function of (seed, element index), redrawn by the backward. */
int poet_gelu_fwd(const void* x, void* y, int64_t n, int dtype_x, int dtype, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                  void* stream);
int poet_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype_x, int dtype, float drop_p, uint32_t seed,
                  const uint32_t* seed_dev, void* stream);
/* (ABI v5) zero-fill of `bytes` bytes (address and size multiples of 4): the value-gradient maps, the gradient arena, per-level sums --
 * every buffer the step accumulates into, so that a captured step holds library kernels only. */
int poet_zero(void* p, int64_t bytes, void* stream);
/* out[seg, col] += sum over rows of segment seg.  x (batch, rows_per_batch, cols) row stride ld;
 * seg_start_host (nseg+1) row boundaries inside one batch item.  out fp32 (nseg, cols) accumulated. */
int poet_colsum(const void* x, int64_t ld, float* out, int batch, int64_t rows_per_batch, int cols,
                const int64_t* seg_start_host, int nseg, int dtype, void* stream);
/* value-gradient maps (gv_dtype fp32 or bf16; addressed by element strides, see fused MSDA) -> (N*S, M*D) row-major `dtype`
 * with row stride ld_out (0: M*D), rows with row_mask != 0 zeroed (masked_fill backward). */
int poet_vgrad_to_rows(const void* gv, int64_t vs_n, int64_t vs_s, int64_t vs_m, const uint8_t* row_mask,
                       void* out, int64_t ld_out, int N, int S, int M, int D, int gv_dtype, int dtype, void* stream);
/* rows with row_mask != 0 of x (rows, cols) `dtype`, row stride ld: zeroed (masked_fill backward for gradient rows that were
 * scattered in place; cols * sizeof(element) % 16 == 0). */
int poet_zero_masked_rows(void* x, int64_t ld, const uint8_t* row_mask, int64_t rows, int cols, int dtype, void* stream);
/* NCHW (N,C,H,W) <-> token-major rows [tok_off, tok_off+H*W) of (N, tok_stride, C). */
int poet_nchw_to_tokens(const void* src, void* dst, int N, int C, int HW, int64_t tok_off, int64_t tok_stride,
                        int src_dtype, int dst_dtype, void* stream);
/* src (N,C,HW) fp32 -> dst (N*HW, 2 C) bf16 rows [hi | lo], hi = bf16(x), lo = bf16(x - hi): the input projection's activation
 * operand with 16 significant bits -- [hi | lo] [W | W]^T = x W^T as one product with K = 2 C (models/pose_estimation_transformer.py
 * :100-135: the 1x1 input_proj convolutions read the backbone's fp32 maps). */
int poet_nchw_to_tokens_split(const float* src, void* dst, int N, int C, int HW, void* stream);
/* the same split of a row-major fp32 matrix: src (rows, K) -> dst (rows, 2 K) bf16 [hi | lo] */
int poet_split_rows(const float* src, void* dst, int64_t rows, int K, void* stream);
int poet_tokens_to_nchw(const void* src, void* dst, int N, int C, int HW, int64_t tok_off, int64_t tok_stride,
                        int src_dtype, int dst_dtype, void* stream);
/* im2col for the 3x3 stride-2 pad-1 conv of the extra level: src NCHW -> (N*Ho*Wo, C*9) with
 * k = c*9 + ky*3 + kx (the flattening of nn.Conv2d's weight). */
int poet_im2col3x3s2(const void* src, void* dst, int N, int C, int H, int W, int Ho, int Wo,
                     int src_dtype, int dst_dtype, void* stream);
/* its adjoint, for the input gradient of a second (third, ...) extra level whose input is the previous PROJECTED level
 * (pose_estimation_transformer.py:327-330): dcol (N*Ho*Wo, C*9) -> ADDED into rows [tok_off, tok_off + H*W) of the
 * token-major stream gradient dst (N, tok_stride, C). */
int poet_col2im3x3s2_add(const void* dcol, void* dst, int N, int C, int H, int W, int Ho, int Wo, int64_t tok_off,
                         int64_t tok_stride, int src_dtype, int dst_dtype, void* stream);

/* GroupNorm over token-major maps.  x (and dx) live at rows [x_off, x_off+HW) of (N, x_stride, C);
 * y (and dy) at rows [y_off, y_off+HW) of (N, y_stride, C) -- so the conv output can stay compact
 * while the normalised map lands directly in the flattened multi-level sequence.  G groups,
 * stats (N,G,2) fp32 = (mean, rstd).  Backward accumulates dgamma/dbeta (fp32 [C]). */
int poet_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                       int N, int HW, int C, int G, int64_t x_off, int64_t x_stride,
                       int64_t y_off, int64_t y_stride, float eps, int dtype_x, int dtype_y,
                       float* scratch, int64_t scratch_floats /* optional caller-owned scratch (contents undefined afterwards):
                           with N * 32 * 64 floats (forward) / N * 32 * 512 floats (backward) and C = 256, G = 32 the
                           whole-row kernels run (two launches, coalesced); without it one workgroup per (image, group) */,
                       void* x_bf16_copy /* (ABI v6) optional, with an fp32 x: bf16(x) in x's own layout -- the operand copy the backward
                           reads -- written in the same pass (was a separate poet_cast over the conv output) */,
                       void* stream);
int poet_groupnorm_bwd(const void* dy, const void* x, const float* stats, const float* gamma,
                       void* dx, float* dgamma, float* dbeta,
                       int N, int HW, int C, int G, int64_t x_off, int64_t x_stride,
                       int64_t y_off, int64_t y_stride, int dtype_x, int dtype_y,
                       float* scratch, int64_t scratch_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pose heads tail: rot_all (R, ncls*6), trans_all (R, ncls*3) fp32, cls (R,) int32 (<=0 -> slot 0)
 * -> rot (R,3,3) via Gram-Schmidt with x,y,z as COLUMNS, trans (R,3).  Backward scatters into the
 * selected class slot (other slots get zero).
 * ---------------------------------------------------------------------------------------------- */
int poet_pose_finish_fwd(const float* rot_all, const float* trans_all, const int32_t* cls,
                         float* rot, float* trans, int R, int ncls, void* stream);
int poet_pose_finish_bwd(const float* rot_all, const int32_t* cls, const float* drot, const float* dtrans,
                         float* drot_all, float* dtrans_all, int R, int ncls, void* stream);

/* Pose losses of ALL decoder layers in one launch (pose_estimation_transformer.py:635-674, the 'translation' and
 * 'rotation' terms of SetCriterion): trans (L, NQ, 3), rot (L, NQ, 3, 3) fp32 predictions; the n_obj matched pairs are
 * (query_idx[i] in [0, NQ) = image*Q + query, tgt_trans[i] (3), tgt_rot[i] (3,3)).  losses (L, 2) =
 * [mean ||t - t_gt||_2, mean acos(clamp((tr(R R_gt^T) - 1)/2, -1+1e-6, 1-1e-6))]; grad_trans / grad_rot (same shapes as
 * the predictions, fully written) = d losses[l][0] / d trans[l] and d losses[l][1] / d rot[l].
 * n_obj_dev (optional, device int32): the pair count is read from device memory instead of n_obj -- the count
 * poet_match_gather leaves behind, so that matcher + loss can sit inside a captured HIP graph (the match arrays then
 * have capacity NQ).  weights (ABI v5, optional, device (L, 2)): grad_trans[l] / grad_rot[l] leave multiplied by weights[l][0] /
 * weights[l][1] (the loss weights of pose_estimation_transformer.py:657-674's weight_dict), and with `total` (optional, one device
 * float) *total = sum_l weights[l] . losses[l] -- the scalar engine.py:66 calls backward on. */
int poet_pose_loss(const float* trans, const float* rot, const int64_t* query_idx, const float* tgt_trans,
                   const float* tgt_rot, int n_obj, int L, int NQ, float* losses, float* grad_trans, float* grad_rot,
                   const int32_t* n_obj_dev, const float* weights, float* total, void* stream);

/* Batched on-device assignment, models/matcher.py:158-229 in 'gt' mode: per image the L1 cost between the first n_pred[i]
 * query boxes (pred_boxes (N,Q,4) fp32) and the image's targets (tgt_boxes rows [tgt_off[i], tgt_off[i+1]), fp32 (T,4)),
 * solved by the algorithm scipy.optimize.linear_sum_assignment uses (same tie-breaking; <= 64 per side).
 * col_out (N,Q) int32: target index matched to query row, -1 = none.  *status != 0 on an oversized / infeasible problem.
 * poet_match_gather turns it into the arrays poet_pose_loss takes (pairs in (image, query) order; N <= 256). */
int poet_lsa_boxes(const float* pred_boxes, const float* tgt_boxes, const int* tgt_off, const int* n_pred, float cost_bbox,
                   int N, int Q, int* col_out, int* status, void* stream);
int poet_match_gather(const int* col, const int* tgt_off, const float* tgt_pos, const float* tgt_rot, int N, int Q,
                      int64_t* query_idx, float* tgt_trans_out, float* tgt_rot_out, int* n_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flat-arena optimizer pieces.  poet_sqnorm: out[0] += sum(g^2) (fp32, caller zero-fills out[0]); `out` must have room
 * for 1 + POET_SQNORM_SCRATCH floats: out[1..] receives per-workgroup partial sums that are then added in a fixed order,
 * so the result is bit-reproducible (data-parallel replicas must derive identical clip factors from identical gradients).
 * poet_adamw: torch.optim.AdamW semantics on a flat fp32 range; grads are first multiplied by
 * clip = min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) when sqnorm != NULL; optionally writes the
 * bf16 shadow copy of the updated parameters, and (p_bf16_lo) the bf16 residual lo = bf16(p - hi) that makes
 * p = hi + lo to 16 mantissa bits (the two operands of a split-weight product run as two plain GEMMs).
 * ---------------------------------------------------------------------------------------------- */
int poet_sqnorm(const float* g, int64_t n, float* out, void* stream);
int poet_adamw(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, uint16_t* p_bf16_lo, int64_t n,
               float lr, float beta1, float beta2, float eps, float weight_decay, int step,
               const float* sqnorm, float max_norm, float grad_scale,
               const uint32_t* step_dev /* optional: the step count is read from this device word instead of `step` */,
               const float* lr_scale /* optional: lr multiplier per block of 64 elements, ceil(n/64) floats (the 0.1x group of
                                        main.py:41,253-271 inside one contiguous range) */,
               void* stream);
/* *word += delta (single-thread kernel): advances the device-side dropout seed / Adam step between graph replays. */
int poet_counter_add(uint32_t* word, uint32_t delta, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POET_HIP_H */
