/* vlmc.h -- C ABI of the MI355X (gfx950) pruning / SparseLoRA kernels.
 *
 * Drop-in boundary for the hot path of Shwai-He/VLM-Compression
 * (lavis/compression/pruners/ and lavis/peft/src/peft/tuners/lora.py).  The
 * reference is pure Python on PyTorch ops, so the "FFI" a maintainer binds is a
 * ctypes stub; every entry point below names the reference
 * op sequence (file:line under /root/reference) it replaces.
 *
 * Conventions
 *  - plain pointers + sizes; no torch types.  All tensor pointers are DEVICE
 *    pointers owned by the caller; the library borrows them for the call,
 *    allocates nothing, frees nothing, and enqueues on `stream` (a hipStream_t
 *    passed as void*; NULL = default stream).  It never synchronises.
 *  - scratch comes from a caller-provided workspace: ask `*_workspace()` for
 *    the size, pass a device buffer of at least that many bytes (256-B aligned).
 *  - every function returns 0 on success or a negative VLMC_E* code;
 *    `vlmc_last_error()` returns a thread-local message for the last failure.
 *  - matrices are row-major `[out, in]` like `nn.Linear.weight`; `ld*` is the
 *    row stride in ELEMENTS.
 *  - masks are `torch.bool` storage: one byte per element, 1 = keep, 0 = pruned
 *    (the reference's `module.mask`, wanda_pruner.py:339).
 */
#ifndef VLMC_H
#define VLMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLMC_ABI_VERSION 20

#define VLMC_OK 0
#define VLMC_EINVAL (-1)     /* bad argument (shape, dtype, alignment, null pointer) */
#define VLMC_EHIP (-2)       /* HIP runtime error (launch failure, no device)        */
#define VLMC_EWORKSPACE (-3) /* workspace too small                                  */
#define VLMC_ENOTPD (-4)     /* Hessian not positive definite after max damping      */

/* element types of weights / activations */
#define VLMC_F32 0
#define VLMC_F16 1
#define VLMC_BF16 2

/* Wanda selection rules */
#define VLMC_SEL_ROW 0    /* per output row, k smallest, stable (wanda_pruner.py:332-337) */
#define VLMC_SEL_MATRIX 1 /* matrix-wide strict threshold (wanda_pruner.py:682-683)      */
#define VLMC_SEL_NM 2     /* n of every m consecutive columns (wanda_pruner.py:326-329)  */

int vlmc_abi_version(void);
const char *vlmc_last_error(void);

/* Timing hook for benchmarks (no reference counterpart): the next statistics or select kernel launched from the
 * calling thread -- vlmc_act_sqnorm[_batch], vlmc_wanda_select[_batch] (first launch of the call), vlmc_linear_fwd[_group],
 * vlmc_attn_matmul, vlmc_sdpa_fwd, vlmc_hessian_accum (its matrix-core kernel) -- records its own
 * begin / end timestamps into the caller's HIP events (hipEvent_t, created by the caller with timing enabled) through
 * hipExtLaunchKernel: hipEventElapsedTime(start, stop) is then the kernel's duration, and no marker packet sits
 * between kernels (hipEventRecord between two kernels idles an MI355X for ~5 us).  Either event may be NULL; the pair
 * is consumed by that one launch. */
void vlmc_set_launch_events(void *start_event, void *stop_event);

/* ---- K1: activation statistics ------------------------------------------------
 * Replaces the body of WrappedGPT.add_batch, wanda_pruner.py:68-81
 *     inp = inp.reshape(-1, in).t().float();  torch.norm(inp, p=2, dim=1) ** 2
 * for `n_calls` hook calls at once.  Call c reads `tokens` rows of `in_features`
 * elements starting at x + c*call_stride, rows `row_stride` elements apart.
 * normsq[c, ch] = square(sqrtf(chain_t fmaf(x, x, acc)))  -- the token reduction
 * is the sequential fused-multiply-add chain torch's CPU kernel performs, so the
 * result is bit-identical to the reference's CPU path.                           */
int vlmc_act_sqnorm(const void *x, int dtype, int64_t n_calls, int64_t tokens, int64_t in_features,
                    int64_t row_stride, int64_t call_stride, float *normsq /* [n_calls, in] */, void *stream);

/* The same for several hook inputs in ONE launch -- the hooks of one transformer block fire
 * in one dense pass (wanda_pruner.py:304-314), so their reductions are independent.  All jobs
 * share dtype and n_calls; output row c of job j starts at normsq + c*normsq_stride.          */
typedef struct vlmc_stat_job {
    const void *x;
    float *normsq;
    int64_t in_features, tokens, row_stride, call_stride, normsq_stride;
    const int32_t *call_tokens;   /* device [n_calls] or NULL: only the first call_tokens[c] <= tokens rows of call c count -- a
                                     group of ragged calibration samples padded to one length (the padding rows are skipped, the
                                     chain over the real rows is the unpadded call's)                                           */
} vlmc_stat_job;
int vlmc_act_sqnorm_batch(const vlmc_stat_job *jobs /* host array */, int n_jobs, int dtype, int64_t n_calls,
                          void *stream);

/* Running mean of wanda_pruner.py:77-81 applied for `n_calls` further calls of
 * `batch` samples each, in call order:
 *     s *= float(n / (n + batch));  n += batch;  s += normsq[c] / float(n)
 * If `sqrt_out` is not NULL it also receives sqrtf(s) -- the `torch.sqrt(scaler_row)`
 * factor of the score (wanda_pruner.py:318) that vlmc_wanda_select consumes.
 * n_calls may be 0 (only the square roots are produced).                           */
int vlmc_wanda_scaler_update(float *scaler_row /* [in], in/out */, int64_t in_features, int64_t nsamples_before,
                             const float *normsq /* [n_calls, in] */, int64_t n_calls, int64_t batch,
                             float *sqrt_out /* [in] or NULL */, void *stream);

/* The same recurrence for several statistics in ONE launch (one per distinct linear input of a
 * block); row c of job j's normsq starts at normsq + c*normsq_stride.                         */
typedef struct vlmc_update_job {
    float *scaler_row;
    const float *normsq;
    float *sqrt_out;
    int64_t in_features, normsq_stride;
} vlmc_update_job;
int vlmc_wanda_scaler_update_batch(const vlmc_update_job *jobs /* host array */, int n_jobs, int64_t nsamples_before,
                                   int64_t n_calls, int64_t batch, void *stream);

/* ---- K2-K7: fused score + select + apply ----------------------------------------
 * Replaces wanda_pruner.py:318-341 (T5/LLM) and :666-687 (ViT) for one linear:
 *     score = |W| * sqrt_scaler                 (fp32, never materialised; sqrt_scaler =
 *                                                sqrtf(scaler_row) from vlmc_wanda_scaler_update)
 *     SEL_ROW:    prune the `k` lowest-score columns of every row, ties -> lowest column
 *     SEL_MATRIX: thr = sort(score.flatten())[k]; prune score < thr   (ties with thr kept)
 *     SEL_NM:     prune the n lowest of every m consecutive columns, ties -> lowest column
 *     mask = keep (1 byte/elt);  if apply_zero: W[pruned] = 0 in place
 *     score_partials[i] = partial sums of score as double; their sum is sum(score)
 *                         (importance_score = sum / (out*in), wanda_pruner.py:320).  The
 *                         caller adds them (fixed count and order => deterministic) -- one
 *                         device reduction per transformer block instead of one launch per
 *                         linear.  `vlmc_wanda_select_partials()` gives the count; may be NULL.
 * `k` is computed by the caller exactly like the reference (int(in*ratio) or
 * int(out*in*ratio)).  Only SEL_MATRIX needs a workspace (else size 0, NULL is fine).  A SEL_MATRIX
 * workspace must be ZERO-FILLED when it is first handed to the library (hipMemset once after allocating
 * it); every call returns it zero-filled again, so it can be reused as is.  (The workgroups of one launch
 * agree on the threshold through counters in it; clearing them inside the call would cost a launch.)
 *
 * vlmc_wanda_select_batch does the same for several linears -- the `for name in subset` loop of
 * wanda_pruner.py:316-341 -- with as few launches as the shapes allow (SEL_ROW: one per distinct
 * (in_features, k), or ONE for up to 12 linears of 16-bit rows of <= 2048 and of 2049..8192 columns;
 * SEL_MATRIX: one kernel per 12 linears; SEL_NM: one launch per 12 linears).  All jobs share
 * dtype, mode, n:m and apply_zero; SEL_MATRIX jobs need DISTINCT workspaces.                   */
typedef struct vlmc_select_job {
    void *W;
    int64_t out_features, in_features, ldw;
    const float *sqrt_scaler;
    int64_t k;
    uint8_t *mask;
    double *score_partials;
    void *workspace;
    size_t workspace_bytes;
} vlmc_select_job;
int vlmc_wanda_select_batch(const vlmc_select_job *jobs /* host array */, int n_jobs, int dtype, int mode, int n, int m,
                            int apply_zero, void *stream);
size_t vlmc_wanda_select_workspace(int mode, int64_t out_features, int64_t in_features);
int64_t vlmc_wanda_select_partials(int mode, int64_t out_features, int64_t in_features);
int vlmc_wanda_select(void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw,
                      const float *sqrt_scaler, int mode, int64_t k, int n, int m, int apply_zero,
                      uint8_t *mask /* [out, in] */, double *score_partials /* device */, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ---- K14-K16: SparseLoRA --------------------------------------------------------------
 * Replaces the tensor algebra of lavis/peft/src/peft/tuners/lora.py:359-394.
 * A = lora_A.weight [r, in] fp32, B = lora_B.weight [out, r] fp32, M = mask, s = alpha/r.
 * The rank-r contraction runs on the matrix cores (f32-input MFMA), the [out,in] delta is
 * never written to memory.
 *
 * vlmc_lora_effective_weight: W_out (may alias W) =
 *   VLMC_LORA_FWD_SPARSE   wd(W + d2) * M,  d2 = wd(wd(B@A) * s)        (forward, sparse=True,  :362-368)
 *   VLMC_LORA_FWD_MASKED   wd(W * M + d2)                               (forward, sparse=False, :369-375)
 *   VLMC_LORA_MERGE_SPARSE wd(W + (s*(B@A)) * M)                        (merge(), sparse=True,  :385-387)
 *   VLMC_LORA_MERGE_MASKED wd(W * M + s*(B@A))                          (merge(), sparse=False, :388-391)
 * wd() = rounding to the weight dtype, i.e. exactly the reference's chain of tensor ops;
 * autocast = dtype code of an active torch.autocast (0 none, VLMC_F16, VLMC_BF16): A, B and
 * B@A are rounded to it like the reference's autocast matmul does.                       */
#define VLMC_LORA_FWD_SPARSE 0
#define VLMC_LORA_FWD_MASKED 1
#define VLMC_LORA_MERGE_SPARSE 2
#define VLMC_LORA_MERGE_MASKED 3
int vlmc_lora_effective_weight(const void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw,
                               const float *A, const float *B, int r, float scaling, const uint8_t *mask, int mode,
                               int autocast, void *W_out, int64_t ldo, void *stream);

/* Gradients of the adapters from G = dL/dW_eff ([out,in], weight dtype; the library GEMM dY^T x):
 *   Gm = wd((sparse ? G . M : G) * s), then rounded to the autocast dtype (as the reference's autograd)
 *   dB[out, r] = Gm @ A^T,  dA[r, in] = B^T @ Gm     (fp32 accumulate; rounded to the autocast dtype)
 * dA or dB may be NULL.  Needs a workspace of vlmc_lora_grad_workspace() bytes (partial sums, combined in
 * a fixed order: deterministic).                                                                 */
size_t vlmc_lora_grad_workspace(int64_t out_features, int64_t in_features, int r);
int vlmc_lora_grad(const void *G, int dtype, int64_t out_features, int64_t in_features, int64_t ldg, const float *A,
                   const float *B, int r, float scaling, const uint8_t *mask, int sparse, int autocast, float *dA,
                   float *dB, void *workspace, size_t workspace_bytes, void *stream);

/* ---- K14 / K15 fused: the layer's three products with the masked low-rank algebra inside the MFMA GEMM -------------------
 * Replaces `F.linear(x, (W + (B@A).to(wd) * s) * M, bias)` (lora.py:362-368; :369-375 for sparse=False) AND its autograd
 * for 16-bit weights when the autocast dtype is the weight dtype: neither W_eff [out,in] nor G = dY^T x [out,in] is ever
 * written to memory.  Restrictions (else the entry points return VLMC_EINVAL and the caller uses
 * vlmc_lora_effective_weight / vlmc_lora_grad with library GEMMs): dtype VLMC_F16 / VLMC_BF16, autocast == dtype,
 * r <= 16, out_features and in_features multiples of 64, 16-byte aligned operands, row strides multiples of 8.
 *
 * vlmc_sparse_lora_prep: 16-bit images of A [r,in] and B [out,r] (rounded to the autocast dtype, rank padded to 16) that
 *   the three kernels read; `prep` holds vlmc_sparse_lora_prep_bytes() bytes and stays valid while A and B do.
 * vlmc_sparse_lora_fwd:        Y [M,out]  = wd(X [M,in] W_eff^T + bias)
 * vlmc_sparse_lora_bwd_input:  dX [M,in]  = wd(dY [M,out] W_eff)
 * vlmc_sparse_lora_bwd_weight: G = wd(dY^T X); Gm = wd((sparse ? G . M : G) * s); dB [out,r] = wd(Gm A16^T),
 *   dA [r,in] = wd(B16^T Gm), fp32 (dA or dB may be NULL); workspace of vlmc_sparse_lora_bwd_weight_workspace() bytes,
 *   256-byte aligned (per-tile partial sums of dA and dB, combined in tile order: deterministic; dY and X are read as
 *   they lie -- the transposed operand comes out of ds_read_b64_tr_b16, nothing is transposed in memory).                 */
size_t vlmc_sparse_lora_prep_bytes(int64_t out_features, int64_t in_features);
int vlmc_sparse_lora_prep(const float *A, const float *B, int64_t out_features, int64_t in_features, int r, int autocast,
                          void *prep, void *stream);
int vlmc_sparse_lora_fwd(const void *X, int64_t M, int64_t ldx, const void *W, int dtype, int64_t out_features,
                         int64_t in_features, int64_t ldw, const uint8_t *mask, const void *prep, int r, float scaling,
                         int sparse, int autocast, const void *bias, void *Y, int64_t ldy, void *stream);
int vlmc_sparse_lora_bwd_input(const void *dY, int64_t M, int64_t lddy, const void *W, int dtype, int64_t out_features,
                               int64_t in_features, int64_t ldw, const uint8_t *mask, const void *prep, int r,
                               float scaling, int sparse, int autocast, void *dX, int64_t lddx, void *stream);
size_t vlmc_sparse_lora_bwd_weight_workspace(int64_t M, int64_t out_features, int64_t in_features);
int vlmc_sparse_lora_bwd_weight(const void *dY, int64_t lddy, const void *X, int64_t ldx, int64_t M, int dtype,
                                int64_t out_features, int64_t in_features, const uint8_t *mask, const void *prep, int r,
                                float scaling, int sparse, int autocast, float *dA, float *dB, void *workspace,
                                size_t workspace_bytes, void *stream);

/* ---- packed 2:4 weights (the optional inference format) -----------------------------------------------------
 * A linear pruned 2:4 by the n:m rule (wanda_pruner.py:326-329; the masks `prune()` leaves on the modules) as
 *     values [out, in / 2]  16-bit, the two kept weights of every group of four input columns, in column order
 *     meta   [out, in / 8]  one byte per two groups: per group the 4-bit code i0 | i1 << 2, i0 < i1 the kept positions
 * (two 2-bit selectors per group, the index form of the structured-sparse MFMA instructions): 9 / 16 of the dense bytes.  The
 * positions are the MASK's (1 = keep), not "the non-zero values": vlmc_unpack_24(vlmc_pack_24(W, mask)) == W . mask bit for
 * bit, and the mask is recovered.  *bad_groups (device, zeroed by the caller) counts the groups that do not keep exactly two
 * (pack) or whose code is not i0 < i1 (unpack); their output is unspecified.  in_features: a multiple of 8.            */
int vlmc_pack_24(const void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw, const uint8_t *mask, int64_t ldm,
                 void *values, uint8_t *meta, unsigned int *bad_groups, void *stream);
int vlmc_unpack_24(const void *values, const uint8_t *meta, int dtype, int64_t out_features, int64_t in_features, void *W, int64_t ldw,
                   uint8_t *mask /* NULL or [out, in] */, int64_t ldm, unsigned int *bad_groups, void *stream);

/* ---- dense calibration forward of one linear (MFMA) ---------------------------------------------
 * Replaces `F.linear(x, weight, bias)` inside the block forwards of the calibration replay,
 * `layer(inps[j], **caches[j])` at wanda_pruner.py:308-311 / :343-346 (sparsegpt_pruner.py:428-431, :452-454;
 * dsnot_pruner.py:349-356, :763-766), for 16-bit weights and activations of the same dtype:
 *     Y[m, n] = wd(sum_k X[m, k] * W[n, k] + bias[n])        fp32 accumulation, ONE rounding to the dtype
 * X [M, K] (row stride ldx), W [N, K] (`nn.Linear.weight`, row stride ldw), bias [N] in the same dtype or NULL,
 * Y [M, N] (row stride ldy).  K, ldx, ldw multiples of 8 elements; X, W 16-byte aligned.
 * Batch-invariant: every output element is ONE accumulator fed the K-steps in ascending order by one
 * matrix-core instruction shape (v_mfma_f32_16x16x32), whatever M is -- replaying 1 or 128 calibration
 * samples per call, or any share of them on another GPU, yields identical rows.
 * dtype VLMC_F32 (round 6: the reference's Q-Former, `ln_vision` and `t5_proj` stay in fp32 outside autocast, blip2_t5_instruct.py:76-95,
 * :143-175): fp32 operands and output on v_mfma_f32_16x16x4_f32, one accumulator per element over k in ascending groups of four -- the same
 * invariance; no alignment requirement beyond 4 bytes.  (vlmc_linear_fwd_group / _rows are 16-bit only.)                    */
int vlmc_linear_fwd(const void *X, const void *W, const void *bias, int dtype, int64_t M, int64_t N, int64_t K, int64_t ldx,
                    int64_t ldw, void *Y, int64_t ldy, void *stream);

/* The same for up to 4 linears fed the SAME activations in ONE launch -- q / k / v of an attention (modeling_t5.py:546-572
 * `self.q(hidden_states)`, `self.k(...)`, `self.v(...)`; modeling_llama.py:204-206), wi_0 / wi_1 of the gated FFN
 * (modeling_t5.py:337-341), k / v of a cross-attention: job g computes Y_g[m, n] = wd(sum_k X[m, k] W_g[n, k] + bias_g[n]).
 * Every output element is accumulated exactly as by vlmc_linear_fwd (same bits); what the launch shares is the chip:
 * the tiles of all jobs are handed out together (three N = 2048 products of the T5 decoder are 3 x 64 tiles of 256 x 256
 * in three launches, or 192 in one).                                                                     */
typedef struct vlmc_linear_job {
    const void *W;      /* [N, K], row stride ldw (multiple of 8 elements), 16-byte aligned */
    const void *bias;   /* [N] in the operand dtype, or NULL                                */
    void *Y;            /* [M, N], row stride ldy                                           */
    int64_t N, ldw, ldy;
} vlmc_linear_job;
int vlmc_linear_fwd_group(const void *X, const vlmc_linear_job *jobs /* host array */, int n_jobs /* 1..4 */, int dtype,
                          int64_t M, int64_t K, int64_t ldx, void *stream);

/* The same product over the rows a ROW MAP names -- the dense calibration forward of a PADDED group of ragged calibration samples
 * (the reference forwards every sample alone, `layer(inps[j], **caches[j])`, wanda_pruner.py:308-311: a sample's rows are all real;
 * stacked as [samples, longest, K] for one forward of the block, the rows behind a sample's own are padding).  X and every Y_g hold
 * M physical rows; `rowmap` (device, int32 [M], a permutation of 0 .. M - 1) lists the n_real rows to compute first and the
 * M - n_real padding rows after them:
 *     Y_g[rowmap[i], :] = wd(X[rowmap[i], :] W_g^T + bias_g)   for i < n_real      (accumulated exactly as by vlmc_linear_fwd: same bits)
 *     Y_g[rowmap[i], :] = 0                                     for i >= n_real     (finite values behind the attention's masks)
 * Padding rows of X are neither loaded nor multiplied; the tiles are cut from the n_real compacted rows.  1 <= n_real <= M. */
int vlmc_linear_fwd_rows(const void *X, const vlmc_linear_job *jobs /* host array */, int n_jobs /* 1..4 */, int dtype, int64_t M,
                         int64_t K, int64_t ldx, const int32_t *rowmap, int64_t n_real, void *stream);

/* The same, fp32 only (the reference's fp32 Q-Former, blip2_t5_instruct.py:143-175), with SEPARATE row lists for X and Y: the input is a
 * token slice of a padded stack -- `attention_output[:, query_length:, :]` of a BERT layer, Qformer.py:434-466 -- read in place through its
 * base rows, the output a compact [samples, slice tokens, N] tensor:
 *     Y[y_rows[i], :] = X[x_rows[i], :] W^T + bias   for i < n_real   (accumulated exactly as by vlmc_linear_fwd with VLMC_F32: same bits)
 *     Y[y_rows[i], :] = 0                             for n_real <= i < n_real + n_zero
 * x_rows: device int32 [n_real] (row stride ldx); y_rows: device int32 [n_real + n_zero] (row stride ldy).  vlmc_linear_fwd_rows takes
 * VLMC_F32 as well (x_rows == y_rows == rowmap, one launch per job). */
int vlmc_linear_fwd_gather(const void *X, const void *W, const void *bias, int dtype /* VLMC_F32 */, int64_t N, int64_t K, int64_t ldx,
                           int64_t ldw, void *Y, int64_t ldy, const int32_t *x_rows, const int32_t *y_rows, int64_t n_real, int64_t n_zero,
                           void *stream);


/* ---- batched attention products of the calibration forward (MFMA) ---------------------------------
 * Replaces the batched matmuls inside the attention of a replayed block -- `attn = q @ k.transpose(-2, -1)` and `attn @ v`
 * (eva_vit.py:147,164), `torch.matmul(query_states, key_states.transpose(3, 2))` and `torch.matmul(attn_weights,
 * value_states)` (modeling_t5.py:590,638; modeling_llama.py likewise) -- for 16-bit operands of one dtype:
 *     C[b0, b1, m, n] = wd(sum_k A[b0, b1, m, k] * B[b0, b1, k, n])        fp32 accumulation, ONE rounding to the dtype
 * Operands are read in place through their ELEMENT strides (sa_* for A's batch dims and rows, A's k stride is 1;
 * sb_* for B: either sb_k == 1 -- B is a transposed view of K-contiguous rows, q @ k^T -- or sb_n == 1 -- attn @ v;
 * sc_* for C, whose n stride is 1).  A batch stride of 0 broadcasts.  Rows need not be 16-byte aligned (2-byte aligned
 * pointers suffice).  Batch-invariant like vlmc_linear_fwd: every output element is ONE accumulator fed the K-steps of 32
 * in ascending order (tail zero-padded) by v_mfma_f32_16x16x32, whatever batch0 x batch1, M and N are -- the products of
 * one calibration sample have the same bits alone, in a group of 128, or on another GPU.
 * dtype VLMC_F32 (the fp32 Q-Former's `torch.matmul(query_layer, key_layer.transpose(-1, -2))` / `torch.matmul(attention_probs,
 * value_layer)`, Qformer.py:201,246): fp32 operands and output on v_mfma_f32_16x16x4_f32, k in ascending groups of four; batch0 * batch1 <= 65535. */
int vlmc_attn_matmul(const void *A, const void *B, void *C, int dtype, int64_t batch0, int64_t batch1, int64_t M, int64_t N,
                     int64_t K, int64_t sa_b0, int64_t sa_b1, int64_t sa_m, int64_t sb_b0, int64_t sb_b1, int64_t sb_k,
                     int64_t sb_n, int64_t sc_b0, int64_t sc_b1, int64_t sc_m, void *stream);

/* ---- fused attention of the calibration forward (MFMA) --------------------------------------------
 * Replaces `F.scaled_dot_product_attention(q, k, v[, is_causal=True])` -- no mask, no dropout -- inside a replayed block: the
 * fused form of the reference models' `softmax(q @ k^T * scale) @ v` (eva_vit.py:129-168; modeling_t5.py:520-640 without
 * position bias), for 16-bit operands of one dtype:
 *     O[b, h, q, :] = sum_k softmax_k(scale * Q[b, h, q, :] . K[b, h, k, :]) V[b, h, k, :]
 * fp32 scores, fp32 softmax (max, exp2, sum, one division), the probabilities rounded to the dtype before the second
 * product (as the unfused form does), fp32 accumulation, one rounding of O.  Q, K, V, O are read / written in place through
 * their ELEMENT strides (batch, head, token; the head_dim stride is 1): `qkv.reshape(B, T, 3, H, d).unbind(2)` views and
 * `[B, T, H, d]` outputs need no copy.  head_dim: a multiple of 8, at most 128; keys per head: at most
 * vlmc_sdpa_max_keys(head_dim) (288; 256 for head_dim > 96) -- a head's K and V live in LDS for the whole head;
 * scale: finite and positive; causal = 1: key j counts for query i iff j <= i (torch's is_causal: aligned to the top left), the
 * self-attention of decoder-only towers (modeling_llama.py).
 * Batch-invariant like vlmc_linear_fwd: an output row depends on its own query row and its head's K and V only, through a
 * fixed order of operations -- a sample's outputs have the same bits alone, in a group of 128, or on another GPU.   */
int vlmc_sdpa_max_keys(int64_t head_dim);
int vlmc_sdpa_fwd(const void *Q, const void *K, const void *V, void *O, int dtype, int64_t batch, int64_t heads, int64_t Tq,
                  int64_t Tk, int64_t head_dim, int64_t sq_b, int64_t sq_h, int64_t sq_t, int64_t sk_b, int64_t sk_h,
                  int64_t sk_t, int64_t sv_b, int64_t sv_h, int64_t sv_t, int64_t so_b, int64_t so_h, int64_t so_t,
                  float scale, int causal, void *stream);

/* ---- batch-invariant mean over the last dimension (the norms of a replayed block) -----------------
 * Replaces `x.mean(-1, keepdim=True)` on the fp32 squares inside the language models' norms --
 * `hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)` (transformers' T5LayerNorm / LlamaRMSNorm, called from
 * the blocks of modeling_t5.py / modeling_llama.py) -- while a block is replayed: torch's reduction kernel chooses how many
 * threads share an output by the NUMBER of outputs, so a 4-token sample alone and inside a group of 128 get other last bits.
 *     out[r] = (sum over c of x[r * ldx + c]) / n        one wave per row, a fixed order that depends on n only
 * x [rows, n] fp32 (row stride ldx elements), out [rows] fp32.                                                 */
int vlmc_row_mean(const float *x, int64_t rows, int64_t n, int64_t ldx, float *out, void *stream);

/* GELU of a replayed block's feed-forward (`self.act(self.fc1(x))`, eva_vit.py:62-64; T5 v1.1's gated GELU, modeling_t5.py:337-346)
 * with one instruction sequence for EVERY element.  torch's elementwise kernels compute a tensor's last partial block with other
 * code than its body (hipcc contracts x/2 * (1 + erf) into an fma there: ~20 % of all 16-bit inputs differ in the last bit), so which
 * rows of a batch get which bits depends on how many samples share the forward.  This kernel uses the body's arithmetic everywhere:
 * equal to torch's body for all 65 536 fp16 / bf16 inputs, batch-invariant.  tanh_approx: 0 = erf form, 1 = `approximate="tanh"`.
 * x, y: n contiguous 16-bit elements (y may be x); dtype VLMC_F32 (the fp32 Q-Former's feed-forward): the same arithmetic on fp32
 * elements, nothing rounded to 16 bits.                                                                                          */
int vlmc_gelu(const void *x, void *y, int64_t n, int dtype, int tanh_approx, void *stream);

/* Row-wise softmax over the last dimension, PADDING-invariant: `F.softmax(scores.float(), dim=-1)` (modeling_t5.py:604-606),
 * `attn.softmax(dim=-1)` on 16-bit scores (eva_vit.py:158), `nn.Softmax(dim=-1)(scores)` (Qformer.py:228) during a replay.
 * fp32 arithmetic in ONE fixed order (csrc/softmax_order.hpp): the maximum; e_j = expf(x_j - max); 16 class sums (class =
 * j mod 16, ascending j); a butterfly over the classes; one IEEE division per element; rounded once if the output is 16-bit.
 * Entries that are masked out (x + finfo.min: expf gives exactly 0) behind a row's real entries add +0 to sums that are
 * otherwise formed in the same order: a sample's row has the same bits alone and padded to a longer group.  vlmc_attn_fwd
 * forms its softmax in the same order.
 * x [rows, n] (row stride ldx), y [rows, n] (row stride ldy); in_dtype / out_dtype: VLMC_F32 -> VLMC_F32, or VLMC_F16 / VLMC_BF16 ->
 * the same dtype or VLMC_F32 (`softmax(x, dim=-1, dtype=torch.float32)`); y may alias x when the dtypes are equal.            */
int vlmc_softmax_rows(const void *x, int in_dtype, int64_t rows, int64_t n, int64_t ldx, void *y, int out_dtype, int64_t ldy,
                      void *stream);

/* ---- the attention of a replayed block as the reference's model files write it, in one launch (MFMA) --------------------
 * eva_vit.py:145-164, modeling_t5.py:588-640, Qformer.py:205-246 spell attention out as tensor ops that each round to the
 * 16-bit dtype:  scores = q @ k^T;  [scores = scores * mul  (`/ sqrt(d)` is torch's multiply by the fp32 reciprocal)];
 * [scores += add0 [+= add1]  (position bias, extended mask: broadcast over batch / heads / queries through stride 0)];
 * probs = softmax in fp32, rounded to the dtype;  out = probs @ v.  This entry point computes exactly that chain -- every
 * intermediate rounded where the tensor op would round it, both products in vlmc_attn_matmul's accumulation order, the
 * softmax in vlmc_softmax_rows' order -- without the [batch, heads, Tq, Tk] scores ever being in HBM: the result has the
 * bits of the unfused sequence on this library's kernels (tests/test_attn_fused_gpu.py).
 * Q [batch, heads, Tq, head_dim], K, V [batch, heads, Tk, head_dim] read in place through ELEMENT strides {batch, head, token}
 * (head_dim stride 1); add0 / add1: NULL or 16-bit tensors of the same dtype with element strides {batch, head, query, key} (any
 * of the first three 0 = broadcast; key stride 1 is the fast path, T5's permuted [Tq, Tk, heads] bias table has stride heads); O is written as a contiguous [batch, Tq, heads, head_dim] tensor (the model's
 * `.transpose(1, 2).reshape(batch, Tq, heads * head_dim)` is then a view).  head_dim: a multiple of 8, at most 128; Tk at most
 * vlmc_attn_max_keys(head_dim) (512 / 352 / 288 for head_dim <= 64 / 96 / 128: a head's K and V live in LDS).
 * Batch- and padding-invariant like vlmc_attn_matmul / vlmc_softmax_rows.                                                    */
int vlmc_attn_max_keys(int64_t head_dim);
int vlmc_attn_fwd(const void *Q, const void *K, const void *V, void *O, int dtype, int64_t batch, int64_t heads, int64_t Tq,
                  int64_t Tk, int64_t head_dim, const int64_t *q_strides, const int64_t *k_strides, const int64_t *v_strides,
                  int has_mul, float mul, const void *add0, const int64_t *add0_strides, const void *add1,
                  const int64_t *add1_strides, void *stream);

/* vlmc_attn_fwd for a PADDED group of ragged calibration samples (the reference forwards every sample alone,
 * wanda_pruner.py:308-311: all its tokens are real).  q_len / k_len: device int32 [batch] or NULL.  k_len[b] = the keys of batch entry b
 * that are real; the caller vouches that an addend masks every key behind them (the dtype's minimum: probability exactly 0), so their
 * tiles are neither staged nor multiplied -- the live rows keep the bits vlmc_attn_fwd gives them (a masked key adds +0 to a sum and 0 . v
 * to a product).  q_len[b] = its real queries; output rows behind them are written as ZEROS.  k_len needs add0.                       */
int vlmc_attn_fwd_lens(const void *Q, const void *K, const void *V, void *O, int dtype, int64_t batch, int64_t heads, int64_t Tq,
                       int64_t Tk, int64_t head_dim, const int64_t *q_strides, const int64_t *k_strides, const int64_t *v_strides,
                       int has_mul, float mul, const void *add0, const int64_t *add0_strides, const void *add1,
                       const int64_t *add1_strides, const int32_t *q_len, const int32_t *k_len, void *stream);

/* ---- the RMS norm of a language-model block in one launch ---------------------------------------------
 * Replaces the op sequence of transformers' T5LayerNorm.forward / LlamaRMSNorm.forward inside a replayed block
 *     variance = x.to(torch.float32).pow(2).mean(-1, keepdim=True);  h = (x * torch.rsqrt(variance + eps)).to(dtype);  y = weight * h
 * (seven launches, ~15 x the activation's bytes in traffic) for 16-bit x [rows, n] (row stride ldx) and weight [n] of one dtype,
 * rounding every intermediate where the op sequence rounds it; the mean is vlmc_row_mean's (same order, same division), so
 * the result equals the op sequence under the replay's patches bit for bit -- provided `rsqrt_mode` (0: 1 / sqrt in double rounded
 * to float -- torch.rsqrt(float) on ROCm, where ATen's `::rsqrt(a)` resolves to the double overload; 1: v_rsq_f32; 2: IEEE fp32
 * 1 / sqrt) is the one that reproduces torch.rsqrt, which the caller establishes once (vlmc/forward.py: against torch.rsqrt itself and
 * against the module's own forward; the kernel is installed only for modules whose own forward it reproduces exactly).  y [rows, n], row stride ldy. */
int vlmc_rms_norm(const void *x, int dtype, int64_t rows, int64_t n, int64_t ldx, const void *weight, float eps, int rsqrt_mode,
                  void *y, int64_t ldy, void *stream);

/* ---- K8: SparseGPT Hessian accumulation (MFMA SYRK) -------------------------------------------------
 * Replaces the arithmetic of SparseGPT.add_batch, sparsegpt_pruner.py:76-79
 *     H *= n / (n + b);  n += b;  inp = sqrt(2 / n) * inp.float();  H += inp @ inp.t()
 * as  H = alpha * H + beta * X^T X  with alpha = n/(n+b), beta = 2/(n+b) chosen by the caller, X [rows, in_features]
 * (row stride ldx) = the tokens of one or more hook calls.  Only the tiles on and below the diagonal of H are
 * computed and updated (H is symmetric): call vlmc_symmetrize_lower once before H is read as a full matrix.
 * fp16 / bf16 X: the products are exact in fp32 (11- / 8-bit significands), fp32 accumulation on the matrix cores;
 * fp32 X is split into three bf16 planes (x = hi + mid + lo exactly) and all nine plane products are accumulated.
 * alpha == 0 does not read H.  Workspace: vlmc_hessian_workspace() bytes (the transposed operand), 16-byte aligned. */
size_t vlmc_hessian_workspace(int dtype, int64_t rows, int64_t in_features);
int vlmc_hessian_accum(const void *X, int dtype, int64_t rows, int64_t in_features, int64_t ldx, float *H, int64_t ldh,
                       float alpha, float beta, void *workspace, size_t workspace_bytes, void *stream);
int vlmc_symmetrize_lower(float *H, int64_t n, int64_t ldh, void *stream);

/* ---- K9: diagonal block of the blocked Cholesky factorization ------------------------------
 * Replaces the unblocked panel step inside `torch.linalg.cholesky(H)` (sparsegpt_pruner.py:116,148):
 * factorizes the nb x nb (nb <= 128) symmetric positive definite block A (lower part read) into its
 * lower Cholesky factor L and also writes inv(L), so that the caller forms the panel below the block
 * and the trailing update with library GEMMs (vlmc/sparsegpt.py: blocked_cholesky).  `*info` (device
 * int, zero-initialised by the caller) receives col0 + j + 1 for the first column j whose pivot is not
 * positive (LAPACK convention); it is left untouched otherwise.                                    */
int vlmc_chol_block(const float *A, int64_t lda, int nb, float *L, int64_t ldl, float *Linv, int64_t ldi, int *info,
                    int col0, void *stream);

/* ---- K9': factor AND inverse factor of a whole Hessian in ONE persistent launch ---------------------------
 * Replaces the chain `torch.linalg.cholesky(H)` -> `torch.cholesky_inverse` -> `torch.linalg.cholesky(., upper=True)`
 * (sparsegpt_pruner.py:112-150) together with vlmc/sparsegpt.py's index reversal: for the symmetric positive definite A
 * (n x n fp32, n a multiple of 128, lower 128 x 128 tiles read) it writes the lower Cholesky factor M (A = M M^T) and
 * X = M^-1 (lower tiles; the tiles above the diagonal of M and X are NOT written).  With A = J H J (rows and columns
 * reversed), U = J X J is the upper factor of H^-1 that the sweep needs.  One grid of persistent workgroups draws 128 x 128
 * tile tasks (left-looking; products on v_mfma_f32_32x32x2_f32) from a ticket counter and hands tiles over through
 * flags -- no host round trip, no launch per 128 columns.  `*info` (device int, zero on entry) receives the 1-based
 * index of the first non-positive pivot (LAPACK convention; the factor is then garbage), or stays 0; -1: a workgroup gave
 * up waiting for a tile (bounded polls; never expected) and every workgroup left.  `workspace`:
 * vlmc_chol_inverse_workspace(n) bytes of device memory, cleared by the call.
 * `max_workgroups` bounds the grid (0: one workgroup per tile task of a column pair, at most n/128 squared) so that
 * several factorizations can share the chip (each workgroup takes a whole CU's LDS).  The summation order of every tile
 * is fixed: the result does not depend on the number of workgroups.                                              */
size_t vlmc_chol_inverse_workspace(int64_t n);
int vlmc_chol_inverse(const float *A, int64_t n, int64_t lda, float *M, int64_t ldm, float *X, int64_t ldx, int *info,
                      void *workspace, size_t workspace_bytes, int max_workgroups, void *stream);

/* ---- K10: SparseGPT blocked OBS sweep ----------------------------------------------------
 * Replaces the per-column Python loop of sparsegpt_pruner.py:186-205 for ONE block of
 * `count` <= 128 columns (fp32 working copy W, pointer at the block's first column):
 *     for i in 0..count-1:
 *         [n:m: if i % m == 0: prune the n smallest w^2/U[j,j]^2 of columns i..i+m-1, ties ->
 *          lowest column, evaluated on the compensated weights]
 *         q = pruned(i) ? 0 : w[:, i];  err = (w[:, i] - q) / U1[i, i];  w[:, i:] -= err (x) U1[i, i:]
 *     W[:, block] = Q;  Err1 = the err columns
 * U1 = the block of the upper Cholesky factor of H^-1 (`Hinv1`), mask1 = the block's
 * unstructured mask (1 = prune; sparsegpt_pruner.py:180-185; ignored for n:m), mask_out
 * (optional) receives the final pruned mask of the block.  Elementwise IEEE fp32 in the
 * reference's operation order: bit-exact given the same factor.  The caller applies the
 * trailing update W[:, i2:] -= Err1 @ U[i1:i2, i2:] (:210) with vlmc_sparsegpt_trailing_update.   */
int vlmc_sparsegpt_sweep(float *W, int64_t out_features, int64_t count, int64_t ldw, const float *U1, int64_t ldu,
                         const uint8_t *mask1, int64_t ldm, int prune_n, int prune_m, float *Err1, int64_t lde,
                         uint8_t *mask_out, int64_t ldmo, void *stream);

/* K10: the trailing update that follows a block's sweep, `W[:, i2:] -= Err1.matmul(Hinv[i1:i2, i2:])` (sparsegpt_pruner.py:210), on
 * fp32 matrix cores (v_mfma_f32_32x32x2_f32) instead of a GEMM library call:
 *     W[r, c] -= sum_{k < count} Err1[r, k] * U[k, c]        r < out_features, c < ncols
 * W: pointer at the first column to update (row stride ldw); Err1 [out_features, count] as vlmc_sparsegpt_sweep wrote it (row stride
 * lde); U: pointer at U[i1, first column] (row stride ldu); count <= 128 (the block).  One accumulator per element over the block's k in
 * ascending pairs, then one subtraction: an element's result does not depend on which columns share the launch.                   */
int vlmc_sparsegpt_trailing_update(float *W, int64_t out_features, int64_t ncols, int64_t ldw, const float *Err1, int64_t lde,
                                   const float *U, int64_t ldu, int64_t count, void *stream);

/* vlmc_sparsegpt_select_sweep: unstructured mode, the block's threshold AND its sweep in one launch
 * (sparsegpt_pruner.py:183-205):
 *     tmp = W1 ** 2 / diag(Hinv1) ** 2;  thresh = sort(tmp.flatten())[rank];  mask1 = tmp <= thresh;  then the sweep above.
 * The rows of W are n_scopes (<= 4) stacked linears that share the factor (q / k / v, wi_0 / wi_1): scope s owns the next
 * scope_rows[s] rows and has its own 0-based scope_ranks[s] = int(tmp.numel() * sparsity) over ITS
 * scope_rows[s] x count scores (NaN scores sort last; a NaN threshold prunes nothing).  scope_rows / scope_ranks are HOST
 * arrays.  `workspace`: vlmc_sparsegpt_select_workspace_bytes() bytes of device memory, ZERO when first handed over; every
 * call returns it zero.  The workgroups of the launch meet at grid barriers, so at most 2 x CUs workgroups of 4 x {1,2,4,8}
 * rows are launched (about 16384 rows on an MI355X; VLMC_EINVAL beyond); if they cannot all be resident (CUs held by another
 * stream) the bounded wait fails for all of them and the last one to finish does the block alone -- same result.        */
int64_t vlmc_sparsegpt_select_workspace_bytes(void);
int vlmc_sparsegpt_select_sweep(float *W, int64_t count, int64_t ldw, const float *U1, int64_t ldu, int n_scopes,
                                const int64_t *scope_rows, const int64_t *scope_ranks, float *Err1, int64_t lde,
                                uint8_t *mask_out, int64_t ldmo, void *workspace, void *stream);

/* The whole block loop of `SparseGPT.fasterprune` (sparsegpt_pruner.py:167-212) from one call: for every block of `blocksize` (<= 128)
 * columns the sweep -- vlmc_sparsegpt_sweep when prune_n != 0, else vlmc_sparsegpt_select_sweep with rank_q = int(rows_q * count *
 * scope_sparsity[q]) (:184; `select_workspace` as that entry point wants it, every scope's rows stacked in W) -- and the trailing update
 * `W[:, i2:] -= Err1 @ U[i1:i2, i2:]` (:210, vlmc_sparsegpt_trailing_update), in order on `stream`.  W [out_features, in_features] fp32
 * working copy (row stride ldw), U the upper factor of H^-1 (row stride ldu), err: scratch of [out_features, blocksize] floats (row
 * stride lde), mask_out optional ([out_features, in_features] bytes, row stride ldmo).  The bits of the entry points called one by one. */
int vlmc_sparsegpt_prune_blocks(float *W, int64_t out_features, int64_t in_features, int64_t ldw, const float *U, int64_t ldu,
                                int64_t blocksize, int prune_n, int prune_m, int n_scopes, const int64_t *scope_rows,
                                const double *scope_sparsity, float *err, int64_t lde, uint8_t *mask_out, int64_t ldmo,
                                void *select_workspace, void *stream);

/* ---- K11-K13: DSnoT --------------------------------------------------------------------
 * vlmc_act_moments: per hook call and channel (layout as vlmc_act_sqnorm) the squared norm, the
 *   plain sum over tokens and the population variance -- the three per-call quantities of the DSnoT
 *   `WrappedGPT.add_batch` (dsnot_pruner.py:79-101).  Any output pointer may be NULL.
 * vlmc_dsnot_stats_update: the three running means in call order (:92-101); tokens_per_call is a
 *   device array (calls may have different token counts); also emits sqrt(scaler_row) if asked.  */
int vlmc_act_moments(const void *x, int dtype, int64_t n_calls, int64_t tokens, int64_t in_features, int64_t row_stride,
                     int64_t call_stride, float *normsq, float *sums, float *vars, void *stream);
int vlmc_dsnot_stats_update(float *scaler_row, float *sum_row, float *var_row, int64_t in_features, int64_t nsamples_before,
                            int64_t ntokens_before, const float *normsq, const float *sums, const float *vars,
                            const int64_t *tokens_per_call, int64_t n_calls, int64_t batch, float *sqrt_out, void *stream);

/* vlmc_dsnot_refine: the prune/regrow cycle loop of dsnot_pruner.py:553-751 (unstructured, prune_n == 0)
 *   or :407-552 (n:m) for every row, starting from keep_mask0 (1 = keep: the initial mask, i.e.
 *   vlmc_wanda_select with k = round(in*ratio) or n:m, apply_zero = 0).  Writes, per row and cycle,
 *   events[row, t] = p | r << 14 | update << 28 for max_cycle cycles, and stop_cycle[row] = first
 *   cycle (1-based) after which the row no longer updates (INT_MAX if it never stops).
 * vlmc_dsnot_apply: replays the first min(*ncycles, max_cycle) events of every row into the mask
 *   (the reference runs all rows for as long as ANY row updates: ncycles = max over rows of
 *   stop_cycle) and zeroes the pruned weights when apply_zero.  in_features <= 16384.          */
int vlmc_dsnot_refine(const void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw,
                      const uint8_t *keep_mask0, const float *sqrt_scaler, const float *sum_row, const float *var_row,
                      int use_wanda_init, int prune_n, int prune_m, int max_cycle, float update_threshold, float pow_of_var,
                      int without_same_sign, uint32_t *events, int32_t *stop_cycle, void *stream);
int vlmc_dsnot_apply(void *W, int dtype, int64_t out_features, int64_t in_features, int64_t ldw, uint8_t *keep_mask,
                     const uint32_t *events, const int32_t *ncycles, int max_cycle, int nm_mode, int apply_zero, void *stream);

/* vlmc_reorder_indices: `return_reorder_indice(input_tensor)` (dsnot_pruner.py:1881-1925), the module-level helper the
 *   unstructured branch orders its prune list with (:623-631).  Per row of x [rows, cols] (row stride ldx; VLMC_F32 / F16 / BF16):
 *   out[row, :] (int64, row stride ldo) = the column indices of the negative entries in ascending order at the head, the indices
 *   of the positive entries in DESCENDING order at the tail, and 0 in every position between them (one per entry that is
 *   neither: zeros, NaN) -- what the reference's two sorts of +inf-padded index matrices, the flip and the sum produce. */
int vlmc_reorder_indices(const void *x, int dtype, int64_t rows, int64_t cols, int64_t ldx, int64_t *out, int64_t ldo,
                         void *stream);

/* ---- K17: one threshold over many score tensors (the global pruners) -----------------------
 * Replaces `get_mask` / `get_layerwise_mask` and the weight update of global_pruner.py:107-148,
 * :166-169, :188-190.  Every job is one parameter tensor (contiguous, numel elements); jobs with
 * the same `scope` share ONE threshold: the scope_k[scope]-th smallest score (1-based, what
 * `torch.topk(all_scores, k, largest=False)[0][-1]` returns) over all their elements; then
 *     keep = score > threshold;   W *= keep   (a pruned weight becomes +-0, as `v.data *= mask`).
 * score (fp32), by `score_mode`:
 *     VLMC_SCORE_W       float(w)            (blipt5_mag_pruner :255 -- signed, as the reference)
 *     VLMC_SCORE_S       S                   (blipt5_rand_pruner :262, or any precomputed score; W may be NULL)
 *     VLMC_SCORE_ABSW_S  |float(w)| * |S|    (blipt5_aobd_pruner :311, S = mean |grad|)
 * multiplied by prev_keep (0/1) when given (iterative pruning, :166-169).  protect_k > 0 is
 * get_mask's per-layer cap (:111-118): scores >= the protect_k-th largest of the job count as FLT_MAX.
 * -0 == +0; NaN scores sort last and are never kept (NaN > t is false).  scope_k[s] must lie in
 * [1, elements of the scope] (k == 0 is an IndexError in the reference).
 * `jobs` and `scope_k` are HOST arrays, read before the call returns (the table travels to the workspace as kernel
 * arguments, 24 jobs per launch: nothing is copied from host memory asynchronously, nothing waits).  Nothing is
 * concatenated or sorted: 3 histogram passes over the operands + one apply pass.                  */
typedef struct {
    void *W;                  /* [numel] weight dtype; rewritten when apply_weights */
    const float *S;           /* [numel] fp32 or NULL (by score_mode) */
    const uint8_t *prev_keep; /* [numel] or NULL */
    uint8_t *keep;            /* [numel] out, 1 = kept */
    int64_t numel;
    int64_t protect_k;
    int32_t scope;
    int32_t dtype;            /* dtype of W (jobs of one call may differ: fp16 vision tower + bf16 language model) */
} vlmc_score_job;
#define VLMC_SCORE_W 0
#define VLMC_SCORE_S 1
#define VLMC_SCORE_ABSW_S 2
size_t vlmc_score_select_workspace(int n_jobs, int n_scopes);
int vlmc_score_select(const vlmc_score_job *jobs, int n_jobs, const int64_t *scope_k, int n_scopes, int score_mode, int apply_weights,
                      void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VLMC_H */
