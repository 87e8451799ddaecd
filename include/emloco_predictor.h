/*
 * emloco_predictor.h -- C ABI of libemloco_hip.so, part 3: the Social-Transmotion + LocoVal kernels (boundary 2).
 *
 * The host side keeps the reference's Python classes (social-transmotion/model_jta.py:130-336 TransMotionJTA,
 * pacer/pacer/learning/value_pose_net.py:10-159 ValuePoseNet) and calls these entry points from
 * torch.autograd.Function wrappers; PyTorch only carries the autograd graph and device memory.
 * All pointers are device memory, fp32, row-major; `stream` is a hipStream_t (0 = default).
 * Return 0 on success, negative on bad arguments / HIP errors (emloco_last_error()).
 */
#ifndef EMLOCO_PREDICTOR_H
#define EMLOCO_PREDICTOR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { EMLOCO_GEMM_BIAS = 1, EMLOCO_GEMM_RELU = 2, EMLOCO_GEMM_ACCUMULATE = 4, EMLOCO_GEMM_DROPOUT = 8,
       /* opt-in reduced precision: operands rounded to bf16 on their way into the matrix cores (fp32 in memory, fp32
        * accumulation, v_mfma_f32_32x32x16_bf16); ~1e-3 relative output error instead of fp32's 1e-6.  Off by default. */
       EMLOCO_GEMM_BF16 = 16,
       /* memory dtypes of the reduced-precision mode (only together with EMLOCO_GEMM_BF16, 16-byte-aligned operands): the A operand /
        * the B operand / the output (for emloco_gemm_relu_bwd: the forward output `y` it masks by) holds bf16 -- 2 bytes per element,
        * leading dimensions and strides still count elements; accumulation, bias and split-K workspaces stay fp32.  This is how the
        * two large activations of an encoder layer (the fused q|k|v projection, the feed-forward hidden layer) live in HBM there. */
       EMLOCO_GEMM_A_BF16MEM = 64, EMLOCO_GEMM_B_BF16MEM = 128, EMLOCO_GEMM_C_BF16MEM = 256, EMLOCO_GEMM_MASK_BF16MEM = 512,
       /* fp32 results on the bf16 matrix rate ("split" mode): every fp32 operand is the exact sum of three bf16 pieces, and the six
        * largest piece products (six v_mfma_f32_32x32x16_bf16 per 16 k, fp32 accumulation) stand where eight
        * v_mfma_f32_32x32x2_f32 stood -- 2.7x less matrix-pipe time (gfx950's fp32 MFMA runs at 1/16 of the bf16 rate, no xf32).
        * Error against float64 is fp32's own class (measured 1e-7 .. 2e-6 of sum |a b|, the same as the fp32 instruction's);
        * results are NOT bit-equal to the fmaf chain of the plain mode.  Served for 16-byte-aligned fp32 operands and n > 32 (others
        * silently take the plain fp32 path); ignored together with EMLOCO_GEMM_BF16.
        * NON-FINITE AND HUGE OPERANDS (defined behaviour, tests/test_emu_kernels.py, tests/test_gpu_predictor.py): a NaN, an Inf, or a
        * finite value whose bf16 rounding overflows (|a| > 3.3895e38 = bf16's largest finite; fp32 reaches 3.4028e38) makes every
        * output element whose reduction it enters NaN -- the remainder a - bf16(a) is inf - inf.  The plain fp32 mode (and torch) give
        * +-Inf for a lone Inf operand; both are non-finite, which is all a caller may rely on (the trainers' NaN handling,
        * train_jta.py:143-165 / :308, treats them alike).  Output elements whose reductions see only finite operands below that
        * bound are unaffected, whatever else is in the matrices. */
       EMLOCO_GEMM_SPLIT = 1024,
       /* (round 5) with EMLOCO_GEMM_SPLIT: B is not the matrix but its PIECE IMAGE made by emloco_gemm_split_pack -- the operand cut into
        * bf16 pieces once, in the order the kernel's LDS stages hold them -- for the products whose B is a weight (y = x W^T: pack with
        * trans = 0; dx = dy W: trans = 1).  A must be row-major fp32, batch 1; ldb / stride_b / trans_b are ignored.  Same pieces, same
        * products, the same bits as the matrix itself; the thousands of workgroups of a tall GEMM no longer cut the same weight tile. */
       EMLOCO_GEMM_B_SPLITIMG = 2048,
       /* (round 6) with EMLOCO_GEMM_SPLIT: TWO bf16 pieces per operand -- the three piece products above 2^-16 of a product instead of the
        * six above 2^-24: half the matrix instructions, a shorter cut.  Meant for the GRADIENT products of the backward pass (dx = dy W,
        * the fused ReLU backward): the trainers' gradient bars (2e-4 of a tensor's scale next to the loss) are met with a margin of
        * ~50 (profiles/r06_ab_bwd_pieces.txt); forward products stay on three pieces.  Ignored with a piece image (three pieces). */
       EMLOCO_GEMM_SPLIT2 = 4096 };

/* Batched strided GEMM on the matrix cores, fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32: exact fp32):
 *   C[b][m][n] (+)= alpha * sum_k A_b(m,k) * B_b(n,k)   [+ bias[n]] [relu]
 *   A_b(m,k) = trans_a ? A[b*sa + k*lda + m] : A[b*sa + m*lda + k]
 *   B_b(n,k) = trans_b ? B[b*sb + k*ldb + n] : B[b*sb + n*ldb + k]
 * i.e. with trans_a = trans_b = 0 this is  C = A . B^T  (nn.Linear: y = x W^T, model_jta.py:145-174 and every
 * nn.TransformerEncoderLayer projection :177-185).  `ksplit` > 1 splits the k range over ksplit partial
 * products written to `workspace` [ksplit][batch][m][n] and summed in a fixed order (deterministic);
 * workspace may be NULL when ksplit == 1. */
int emloco_gemm_f32(int batch, int m, int n, int k, float alpha,
                    const float *A, int lda, int64_t stride_a, int trans_a,
                    const float *B, int ldb, int64_t stride_b, int trans_b,
                    float *C, int ldc, int64_t stride_c,
                    const float *bias, int flags, int ksplit, float *workspace, void *stream);
/* The same with EMLOCO_GEMM_DROPOUT: inverted dropout (nn.Dropout in training mode, model_jta.py:177-178) applied after
 * bias / ReLU; the keep mask is a stateless hash of (drop_seed, flat output index), so nothing is stored for the backward.
 * Needs a dense output (ldc == n, stride_c == m * n). */
int emloco_gemm_f32_ex(int batch, int m, int n, int k, float alpha,
                       const float *A, int lda, int64_t stride_a, int trans_a,
                       const float *B, int ldb, int64_t stride_b, int trans_b,
                       float *C, int ldc, int64_t stride_c,
                       const float *bias, int flags, int ksplit, float *workspace, float drop_p, uint32_t drop_seed, void *stream);
/* Backward of that epilogue in one pass: dz = dy * [relu: y > 0] * [dropout keep / (1 - p)], y = the forward output
 * (with ReLU + dropout a positive output is "active and kept"; without ReLU the mask is recomputed from the seed). */
int emloco_act_bwd(int64_t total, const float *dy, const float *y, int relu, float drop_p, uint32_t drop_seed, float *dz, void *stream);
/* Backward through y = dropout(relu(x W^T + b)) fused into the GEMM that produces the incoming gradient (the feed-forward block
 * of nn.TransformerEncoderLayer, model_jta.py:177: linear2(dropout(relu(linear1(x))))):
 *   C[m][n] = (A[m][k] . B) o [y > 0] * scale      B = B[n][k] (trans_b = 0) or B[k][n] (trans_b = 1), y the forward output [m][n]
 *   colsum[n] = column sums of C  (the bias gradient of the first linear layer)
 * One launch instead of GEMM -> emloco_act_bwd_colsum: the unmasked gradient (m x n) is never written or re-read.  scale =
 * 1 / (1 - p) of the forward's dropout (1 without).  n > 32; A and B 16-byte aligned, lda and ldb multiples of 4.  workspace: emloco_gemm_relu_bwd_workspace(m, n) floats. */
int64_t emloco_gemm_relu_bwd_workspace(int m, int n);
int emloco_gemm_relu_bwd(int m, int n, int k, const float *A, int lda, const float *B, int ldb, int trans_b, float *C,
                         const float *y, float scale, float *colsum, float *workspace, int flags, void *stream);

/* The feed-forward block of nn.TransformerEncoderLayer (social-transmotion/model_jta.py:177-178,311: d_model = 128, dim_feedforward F,
 * relu, dropout p) as two CHAINED matrix products per launch -- the reduced-precision mode's path (bf16 operands into
 * v_mfma_f32_32x32x16_bf16, fp32 accumulation): the hidden tile goes from the product that makes it to the product that consumes it in
 * registers; it is also stored, as bf16, for the two weight-gradient products (ordinary GEMMs with a bf16 operand).  Model width 128,
 * F a multiple of 64; every pointer 16-byte aligned; weights are bf16 copies the caller makes per call (2 x 256 KB at F = 1024).
 *   emloco_ffn_fwd        hidden[M][F] = dropout(relu(x w1^T + b1)) (bf16), out[M][128] = dropout(hidden w2^T + b2) (fp32), mask[M][F / 32]:
 *                         one BIT per hidden unit, "active and kept" (word (row, 64-unit chunk c, h) at [row][2 c + h], bit 16 t + 4 q + e =
 *                         unit 64 c + 32 t + 8 q + 4 h + e: the units one lane of the kernel holds) -- all the input-gradient pass reads of the
 *                         hidden layer; w1_bf16 [F][128], w2_bf16 [128][F].  Hidden dropout mask: emloco_ffn_keep_mask(seed_hidden, ...);
 *                         output mask: the GEMM epilogue's counter hash, emloco_dropout_keep_mask(seed_out, row * 128 + col, ...) -- so
 *                         emloco_act_bwd_colsum(M, 128, dout, NULL, 0, p, seed_out, ...) is the backward of the output dropout.  drop_p = 0:
 *                         no dropout (eval).
 *   emloco_ffn_bwd_input  dz1[M][F] = (dz2 w2) o [mask] / (1 - p) (bf16), dx[M][128] = dz1 w1 (fp32); dz2 [M][128] fp32 is the gradient
 *                         w.r.t. linear2's output (output dropout already applied), w2t_bf16 = w2^T [F][128], w1t_bf16 = w1^T [128][F].
 *                         The weight and bias gradients follow from hidden / dz1: dW2 = dz2^T hidden, dW1 = dz1^T x, db1 = column sums of
 *                         dz1 (emloco_gemm_f32_ex with a bf16 operand, emloco_colsum_ex).
 *   emloco_ffn_keep_mask  the hidden layer's keep mask on the HOST (host_out[r][f] = 1 iff unit f of row first_row + r is kept): one 32-bit
 *                         hash per pair of adjacent units, 16 bits each against p 2^16.  For tests. */
int emloco_ffn_fwd(int M, int F, const float *x, const uint16_t *w1_bf16, const uint16_t *w2_bf16, const float *b1, const float *b2,
                   uint16_t *hidden, uint32_t *mask, float *out, float drop_p, uint32_t seed_hidden, uint32_t seed_out, void *stream);
int emloco_ffn_bwd_input(int M, int F, const float *dz2, const uint16_t *w2t_bf16, const uint16_t *w1t_bf16, const uint32_t *mask,
                         uint16_t *dz1, float *dx, float drop_p, void *stream);
/* (round 6) emloco_ffn_fwd with the post-norm layer's tail in its epilogue (model_jta.py:177: norm2(x1 + dropout(ff(x1)))):
 * xr[M][128] = ff(x) + res, y = LayerNorm(xr) gamma + beta (biased variance, eps as nn.LayerNorm), mean / rstd [M] as
 * emloco_layernorm_fwd_save leaves them (the backward is emloco_layernorm_bwd2 on xr, then emloco_ffn_bwd_input* on its dxr); the
 * feed-forward's own output is never written.  A lane of the kernel holds half a row, its partner the other half. */
int emloco_ffn_fwd_norm(int M, int F, const float *x, const uint16_t *w1_bf16, const uint16_t *w2_bf16, const float *b1, const float *b2,
                        uint16_t *hidden, uint32_t *mask, const float *res, const float *gamma, const float *beta, float eps,
                        float *y, float *xr, float *mean, float *rstd, float drop_p, uint32_t seed_hidden, uint32_t seed_out, void *stream);
/* (round 6) The same, also leaving the column sums of dz1 AS STORED (bf16-rounded) over each wave's 32 rows in colpart
 * [emloco_ffn_bwd_colsum_rows(M)][F] floats (16-byte aligned; rows of waves past the last row are zeroed): linear1's bias gradient is
 * their column sum (emloco_colsum over that matrix) -- a read of M / 32 x F floats where the separate pass read the M x F gradient. */
int emloco_ffn_bwd_input_colsum(int M, int F, const float *dz2, const uint16_t *w2t_bf16, const uint16_t *w1t_bf16, const uint32_t *mask,
                                uint16_t *dz1, float *dx, float drop_p, float *colpart, void *stream);
int64_t emloco_ffn_bwd_colsum_rows(int M);
int emloco_ffn_keep_mask(uint32_t seed_hidden, int64_t first_row, int64_t rows, int F, float p, uint8_t *host_out);

/* the same for a [m][n] gradient plus the bias gradient colsum[n] = sum_m dz (fixed-order folding; workspace as emloco_colsum) */
int emloco_act_bwd_colsum(int m, int n, const float *dy, const float *y, int relu, float drop_p, uint32_t drop_seed, float *dz,
                          float *colsum, float *workspace, void *stream);

/* Row softmax of attention scores with an additive per-key bias (nn.MultiheadAttention's key_padding_mask):
 *   P[r][j] = softmax_j(scale * S[r][j] + key_bias[seq(r)][j]);  a row whose keys are all -inf gives zeros.
 * torch semantics: a BOOL padding mask means -inf on padded keys, a FLOAT mask is ADDED to the scores -- and the
 * reference passes a float 0/1 mask (dataset_jta.py:86 `padding_mask.float()`, model_jta.py:299-300,311,317), so
 * its "padded" keys are biased by +1, not removed.  rows = n_seq * rows_per_seq; key_bias f32 [n_seq][cols] or NULL.
 * In place allowed (P == S). */
int emloco_softmax_fwd(int n_seq, int rows_per_seq, int cols, float scale, const float *S, const float *key_bias,
                       float *P, void *stream);
/* dS = scale * P * (dP - sum_j(dP * P)) ; in place allowed (dS == dP) */
int emloco_softmax_bwd(int rows, int cols, float scale, const float *P, const float *dP, float *dS, void *stream);

/* Fused multi-head self-attention, head dim 32 (d_model = nhead * 32): nn.MultiheadAttention's
 * softmax(q k^T / sqrt(32) + key_bias) v per head (model_jta.py:177-178,311-321), scores never written to memory.
 * qkv [n_seq][S][3 d_model] = in_proj output (q | k | v, heads contiguous inside each); key_bias [n_seq][S] additive or
 * NULL; out [n_seq][S][d_model] (heads concatenated = out_proj input); lse [n_seq * nhead][S] is kept for the backward.
 * n_seq * nhead <= 65535. */
int emloco_attention_fwd(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                         float *out, float *lse, void *stream);
/* Backward: dqkv [n_seq][S][3 d_model] from dout; probabilities are recomputed from q, k and lse.  dsum is a device
 * scratch of n_seq * nhead * S floats.  Deterministic (no atomics): one kernel owns the query rows (dq), one the key
 * rows (dk, dv). */
int emloco_attention_bwd(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                         const float *out, const float *lse, const float *dout, float *dqkv, float *dsum, void *stream);

/* The same with flags: EMLOCO_ATTN_BF16 = the opt-in reduced precision of EMLOCO_GEMM_BF16 for the four (forward) / eight
 * (backward) tile products per step: operands (q, k, v, probabilities, dO, dS) rounded to bf16 into
 * v_mfma_f32_32x32x16_bf16, fp32 accumulation; softmax statistics, log-sum-exp and D stay fp32.  flags = 0 is the call above. */
enum { EMLOCO_ATTN_SPLIT = 64,        /* fp32-class tile products from bf16 pieces (the attention's EMLOCO_GEMM_SPLIT: six v_mfma_f32_32x32x16_bf16
                                       * per 16 reduction entries, fp32 accumulation; error class of the fp32 instruction; ignored with EMLOCO_ATTN_BF16) */
       EMLOCO_ATTN_BF16 = 16,
       EMLOCO_ATTN_QKV_BF16MEM = 32   /* with EMLOCO_ATTN_BF16: qkv (and, in the backward, dqkv) hold bf16 in memory -- 2 bytes per element,
                                       * same shapes; out / dout / lse / dsum stay fp32 */ };
int emloco_attention_fwd_ex(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                            float *out, float *lse, int flags, void *stream);
int emloco_attention_bwd_ex(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                            const float *out, const float *lse, const float *dout, float *dqkv, float *dsum, int flags, void *stream);

/* The same with dropout on the attention probabilities -- nn.MultiheadAttention(dropout = p) inside
 * nn.TransformerEncoderLayer in training mode (model_jta.py:177-178): softmax -> dropout(p) -> . V.  Probability (bh, query,
 * key) of the launch (bh = sequence * nhead + head) is kept iff its BYTE of a 32-bit counter hash of (drop_seed, bh, query, key / 4)
 * clears p 2^8 (emloco_attention_keep_mask evaluates it on the host); the backward recomputes the mask, so pass it the forward's
 * (drop_p, drop_seed).  drop_p = 0 is the call above.  (Round 6) one hash serves four adjacent keys, so the drop probability is REALISED
 * in 1/256ths -- p8 = round(256 p) / 256, 26 / 256 = 0.1016 for the model's 0.1 -- and the kept probabilities are scaled by
 * 1 / (1 - p8), the inverse keep rate of the mask that is drawn (unbiased; rounds 2-5 compared 16 bits per decision: twice the hashes,
 * which were 31 % of the bf16 attention kernels' time). */
int emloco_attention_fwd_dropout(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 float *out, float *lse, int flags, float drop_p, uint32_t drop_seed, void *stream);
int emloco_attention_bwd_dropout(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 const float *out, const float *lse, const float *dout, float *dqkv, float *dsum, int flags,
                                 float drop_p, uint32_t drop_seed, void *stream);

/* The same with only the FIRST n_query <= S rows of every sequence attending (over all S keys): out [n_seq][n_query][d_model],
 * lse / dsum [n_seq * nhead][n_query], dout [n_seq][n_query][d_model]; dqkv stays [n_seq][S][3 d_model] with dQ = 0 on the rows
 * that did not attend.  The last layer of a former only feeds its first 21 tokens on (model_jta.py:316 `out_local[:21]`,
 * :321 the primary agent's rows), so the rows nothing reads are not computed; n_query = S is the call above. */
int emloco_attention_fwd_queries(int n_seq, int S, int n_query, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 float *out, float *lse, int flags, float drop_p, uint32_t drop_seed, void *stream);
int emloco_attention_bwd_queries(int n_seq, int S, int n_query, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 const float *out, const float *lse, const float *dout, float *dqkv, float *dsum, int flags,
                                 float drop_p, uint32_t drop_seed, void *stream);
/* host_out [n_seq_heads][S][S] = 1 where the attention dropout above keeps the probability */
int emloco_attention_keep_mask(uint32_t seed, int n_seq_heads, int S, float p, uint8_t *host_out);
/* The counter-based keep mask of the GEMM-epilogue dropout, evaluated on the
 * HOST: host_out[i] = 1 iff element first_index + i of a launch with `seed` is kept at rate p.  For tests and for callers that
 * need the mask a kernel used. */
int emloco_dropout_keep_mask(uint32_t seed, uint64_t first_index, int64_t n, float p, uint8_t *host_out);

/* y = LayerNorm(x + res) * gamma + beta over the last dim (post-norm encoder layer, d <= 1024);
 * res may be NULL.  Saves mean / rstd [rows] for the backward. */
int emloco_layernorm_fwd(int rows, int d, float eps, const float *x, const float *res, const float *gamma,
                         const float *beta, float *y, float *mean, float *rstd, void *stream);
/* the same, also writing xr [rows][d] = x + res (what emloco_layernorm_bwd wants) when xr is not NULL */
int emloco_layernorm_fwd_save(int rows, int d, float eps, const float *x, const float *res, const float *gamma,
                              const float *beta, float *y, float *mean, float *rstd, float *xr, void *stream);
/* dxr = dL/d(x+res); dgamma/dbeta are reduced over rows in a fixed order.  xr = x + res (the fwd input sum)
 * is recomputed from y:  xhat = (y - beta) / gamma is avoided -- pass the saved sum `xr`. */
int emloco_layernorm_bwd(int rows, int d, const float *xr, const float *gamma, const float *mean, const float *rstd,
                         const float *dy, float *dxr, float *dgamma, float *dbeta, float *workspace, void *stream);
/* the same with TWO incoming gradients, added on load (dy2 may be NULL): a post-norm layer's output feeds the next sublayer AND its
 * residual branch; handing both gradients to the backward saves the add pass autograd would run over two [rows][d] tensors first */
int emloco_layernorm_bwd2(int rows, int d, const float *xr, const float *gamma, const float *mean, const float *rstd,
                          const float *dy, const float *dy2, float *dxr, float *dgamma, float *dbeta, float *workspace, void *stream);
/* floats of device workspace emloco_layernorm_bwd needs (per-block partials + their fold levels) */
int64_t emloco_layernorm_bwd_workspace(int rows, int d);

/* column sums: out[n] = sum_m X[m][n]  (bias gradients), fixed reduction order */
int emloco_colsum(int m, int n, const float *X, float *out, float *workspace, void *stream);
/* The same for a bf16 matrix: flags = EMLOCO_GEMM_A_BF16MEM (the bias gradient of the bf16 q|k|v gradient); sums in fp32. */
int emloco_colsum_ex(int m, int n, const float *X, float *out, float *workspace, int flags, void *stream);
int64_t emloco_colsum_workspace(int m, int n);   /* floats */

/* Policy-input normaliser (frozen policy forward): RunningMeanStd.forward in eval mode,
 * pacer/pacer/utils/running_mean_std.py:81-83:  y = clamp((x - mean) / sqrt(var + eps), -clip, clip), clip = 5.
 * x [rows][ldx]; mean / var are the float32 casts of the float64 running buffers.  Columns [0, split) are written to
 * out0 (leading dimension ld0), columns [split, cols) to out1 (ld1); split == cols writes everything to out0.
 * The MLPs themselves (amp_network_sept_builder.py:50-110: task MLP 1054->512->256, actor MLP 624->2048->1024->69)
 * run on emloco_gemm_f32 with the bias + ReLU epilogue. */
/* AMP style reward from the discriminator's logits [n] (pacer/pacer/learning/amp_continuous.py:675-692 `_calc_disc_rewards`):
 * reward = -log(max(1 - sigmoid(logit), 1e-4)) * scale, one launch */
int emloco_disc_reward(int n, const float *logits, float scale, float *reward, void *stream);

int emloco_obs_normalize(int rows, int cols, const float *x, int ldx, const float *mean, const float *var, float eps,
                         float clip, int split, float *out0, int ld0, float *out1, int ld1, void *stream);

/* Statistics update of the same normaliser (RunningMeanStd.forward in training mode, running_mean_std.py:85-95, called once per
 * rollout step by the PPO / AMP learner on the observation and AMP-observation batches): merges the per-column mean and unbiased
 * variance of x [rows][ldx] into the float64 running moments mean / var [cols] with the parallel-variance rule (count_in [1] rows
 * seen so far), one launch, no intermediate tensors.  Columns below first_col keep their moments (freeze_partial); the new count
 * is written to count_out [1] (a different buffer than count_in). */
int emloco_rms_update(int rows, int cols, const float *x, int ldx, double *mean, double *var, const double *count_in, double *count_out,
                      int first_col, void *stream);
/* The same update for tall batches (the learner's 2 048 .. 25 600-row minibatches, amp_continuous.py:335-346 in training mode): two
 * launches -- per (256-row chunk, column) partial moments taken from registers, then an in-order fold of the chunks -- instead of one
 * serial chain per column; same arithmetic rule, results equal to the single launch to float64 rounding.  `workspace`:
 * emloco_rms_update_workspace(rows, cols) bytes of device memory. */
int64_t emloco_rms_update_workspace(int rows, int cols);
int emloco_rms_update_chunked(int rows, int cols, const float *x, int ldx, double *mean, double *var, const double *count_in, double *count_out,
                              int first_col, void *workspace, void *stream);

/* LocoVal MLP (value_pose_net.py:36-159), fused: yaw normalisation (:73-103) + hidden joints zeroed (:141-144)
 * + 100->49->24->1 MLP with ReLU/ReLU/sigmoid.  traj [B][13][traj_stride>=2], pose [B][24][3], vel [B][2].
 * Outputs value [B]; x100 [B][100] (normalised MLP input) and h1 [B][49], h2 [B][24] are kept for the backward. */
int emloco_locoval_fwd(int B, const float *traj, int traj_stride, const float *pose, const float *vel,
                       const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                       float *value, float *x100, float *h1, float *h2, float *angle, void *stream);
/* Same, evaluating only the rows whose `row_weight` entry is non-zero (NULL: all): the fit of a rollout step needs the value of the
 * episodes that emit a target (amp_continuous_value.py:112-145), a few hundred of the 4096 envs; the other rows of value / x100 / h1 /
 * h2 are left as they are. */
int emloco_locoval_fwd_rows(int B, const float *traj, int traj_stride, const float *pose, const float *vel, const float *w1,
                            const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, float *value,
                            float *x100, float *h1, float *h2, float *angle, const float *row_weight, void *stream);
/* Backward: given dvalue [B] -> gradients of the 6 parameter tensors (summed over the batch, fixed order, written
 * to dparams = [dw1 4900 | db1 49 | dw2 1176 | db2 24 | dw3 24 | db3 1]) and d traj [B][13][traj_stride]
 * (the gradient that reaches the predicted trajectory in the EmLoco loss, train_jta.py:288-308). */
int emloco_locoval_bwd(int B, const float *traj, int traj_stride, const float *pose, const float *vel,
                       const float *w1, const float *w2, const float *w3,
                       const float *value, const float *x100, const float *h1, const float *h2, const float *angle,
                       const float *dvalue, float *dparams, float *dtraj, float *workspace, void *stream);
/* bytes of workspace emloco_locoval_bwd needs for batch B */
int64_t emloco_locoval_bwd_workspace(int B);

/* ------------------------------------------------------------------------------------------------------------
 * LocoVal training step around the MLP: the per-step body of AMPValueAgent.play_steps after env.step
 * (pacer/pacer/learning/amp_continuous_value.py:63-145; optimiser / normalisation common_agent.py:89-97,154-155), as three small
 * launches around emloco_locoval_fwd / _bwd instead of ~90 framework launches:
 *   emloco_locoval_returns   inversion penalty (:63-64), discounted-return bookkeeping with the step_to_pred cut-off
 *                            (:93-118), LocoVal inputs in origin-relative form (vec_task_wrappers.py:50-66, :126-129),
 *                            target (G - min) / (max - min) (:135) and the 0/1 weight of the rows with a finished episode
 *   emloco_locoval_fit_grad  d/dvalue of MSELoss(reduction='sum') over those rows (:137), loss sum and row count
 *   emloco_adamw_gated       AdamW(1e-3, wd 1e-4) step (:139) committed on the device iff the (all-reduced) row count > 0
 * The gradient of the 6 174 parameters comes from emloco_locoval_bwd straight into the flat bucket that is all-reduced. */
typedef struct {
    int32_t n_env, step_to_pred;
    float gamma, inversion_penalty, min_cum_rewards, max_cum_rewards;
    float *current_rewards, *current_lengths, *current_combined_rewards, *discount_coefs;   /* state [n_env] */
    const float *waypoint_traj;     /* [n_env][15][3]  task.waypoint_traj (humanoid_pedestrain_terrain.py:511-516) */
    const float *init_pose;         /* [n_env][24][3]  task.init_pose */
    const float *init_vel;          /* [n_env][2]      task.init_vel */
    float *traj13;                  /* out [n_env][13][3] */
    float *pose;                    /* out [n_env][24][3] */
    float *vel;                     /* out [n_env][2] */
    float *target;                  /* out [n_env] */
    float *weight;                  /* out [n_env] */
    /* Staged mode (both NULL: off).  With an AMP discriminator in the loop the style reward of a step (:90-96) exists three GEMMs
     * after the step, while everything else the bookkeeping reads is overwritten by the resets that follow it.  A step with these two
     * arrays set is STAGED by emloco_locoval_returns / emloco_task_post_physics_returns -- LocoVal inputs copied, the reward after the
     * inversion penalty and the done flag parked here, state untouched -- and finished by emloco_locoval_returns_finish when the
     * discriminator's reward is there: the same operations on the same values, so state / target / weight are the one-call
     * results bit for bit.  The discriminator then runs beside the next step instead of between two steps. */
    float *staged_reward;           /* [n_env] or NULL */
    uint8_t *staged_done;           /* [n_env] or NULL */
} EmlocoLocoValStep;
int emloco_locoval_returns(const EmlocoLocoValStep *s, const float *rewards, const float *amp_rewards /* or NULL = 0 */,
                           const int64_t *dones, const uint8_t *inverted /* or NULL */, void *stream);
/* second half of a staged step: the bookkeeping from s->staged_reward / s->staged_done and the AMP reward (NULL = 0) */
int emloco_locoval_returns_finish(const EmlocoLocoValStep *s, const float *amp_rewards, void *stream);
/* slot (optional, int32 [n]): rank of each valid row among the valid rows, -1 elsewhere -- for emloco_locoval_bwd_rows */
int emloco_locoval_fit_grad(int n, const float *value, const float *target, const float *weight, float *dvalue, float *tail2,
                            int32_t *slot, void *stream);
/* emloco_locoval_bwd over the rows that have a slot only (the others have dvalue = 0 and would add zeros): the workspace
 * holds one 6 174-float row per slot, `count` (device float, = tail2 + 1 BEFORE the all-reduce) is the number of slots.
 * d traj of the skipped rows is left untouched. */
int emloco_locoval_bwd_rows(int B, const float *traj, int traj_stride, const float *pose, const float *vel,
                            const float *w1, const float *w2, const float *w3,
                            const float *value, const float *x100, const float *h1, const float *h2, const float *angle,
                            const float *dvalue, const int32_t *slot, const float *count, float *dparams, float *dtraj,
                            float *workspace, void *stream);
/* tail2 = [loss sum, row count] after the all-reduce (NULL: always step); steps_in / steps_out: device step counters (the caller
 * swaps them after every call); stats: optional device double[5] = [last loss, last count, total loss, total count, fits] */
int emloco_adamw_gated(int n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, const float *steps_in,
                       float *steps_out, const float *tail2, float lr, float beta1, float beta2, float eps, float weight_decay,
                       double *stats, void *stream);

/* torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.Adam.step() (train_jta.py:317-318,411; train_jrdb.py likewise) on ONE
 * flat fp32 buffer of n parameters with their gradients, first and second moments laid out alike -- three launches (block sums of
 * squares, the clip coefficient from them in a fixed order, the update) where the foreach implementations issue ~25.  The gradient is
 * left clipped, as clip_grad_norm_ leaves it; workspace[0] holds the total norm it returns, workspace[1] the coefficient.
 * bias_correction1 = 1 - beta1^t, bias_correction2_sqrt = sqrt(1 - beta2^t) for the step count t AFTER this step (host arithmetic, as
 * torch's non-capturable Adam does it); the betas are doubles because torch rounds 1 - beta from the double.  max_norm <= 0: no clipping (workspace may be NULL).  weight_decay is Adam's L2 term. */
int64_t emloco_adam_clip_flat_workspace(int64_t n);
/* (workspace: [0] norm, [1] clip coefficient, [2] 1 - beta1^t, [3] sqrt(1 - beta2^t) (counted variant), [4 ..] block partials) */
int emloco_adam_clip_flat(int64_t n, float *params, float *grads, float *exp_avg, float *exp_avg_sq, float lr, double beta1, double beta2, float eps,
                          float weight_decay, float bias_correction1, float bias_correction2_sqrt, float max_norm, float *workspace, void *stream);

/* The same with the step count on the DEVICE: step_count[0] (one float, the number of steps taken so far) is incremented and the bias
 * corrections are computed from it by a one-thread launch ahead of the update -- a step captured in a HIP graph (the PPO learner's
 * optimiser step, amp_continuous.py:335-479 through common_agent.py:573-603) replays as the next step.  Four launches with clipping. */
int emloco_adam_clip_flat_counted(int64_t n, float *params, float *grads, float *exp_avg, float *exp_avg_sq, float lr, double beta1, double beta2,
                                  float eps, float weight_decay, float max_norm, float *workspace, float *step_count, void *stream);

/* (round 6) Gradient tensors into their slices of ONE flat buffer: flat[dst_offset[i] .. + numel[i]) = src[i][0 .. numel[i]) for i < n, in
 * one launch per 96 tensors (the table travels in the kernel arguments).  src / numel / dst_offset are HOST arrays of DEVICE pointers /
 * element counts / element offsets.  What it replaces: with every parameter's .grad aliasing the optimiser's flat bucket, autograd's
 * accumulation node ADDS each incoming gradient into its slice -- one launch per parameter and step (132 in the JTA train step,
 * train_jta.py:311-318); with .grad released ahead of the backward pass autograd only keeps the tensors, and this gathers them. */
int emloco_gather_flat(int n, const float *const *src, const int64_t *numel, const int64_t *dst_offset, float *flat, void *stream);

/* ---- PPO loss heads (round 5): the tail between the networks' outputs and the scalar loss of the PPO + AMP update
 * (amp_continuous.py:335-425, common_agent.py:426-468; neglogp / entropy / policy_kl of rl_games 1.1.4 as restated in
 * learning/amp_agent.py) -- ~140 elementwise / reduction launches per optimiser step in torch -- as one launch over the rows, one
 * fixed-order mean and one backward launch per head.  All pointers are device pointers, rows contiguous.
 *
 * Actor head.  mu, logstd, actions, old_mu, old_sigma: [B][A]; old_neglogp, advantages: [B].  out5 = means over the rows of
 * [clipped surrogate max(-adv r, -adv clamp(r, 1 - e, 1 + e)) with r = exp(old_neglogp - neglogp), entropy, bound loss (soft bound 1),
 * fraction of rows with |r - 1| > e, KL(new || old) (old_mu / old_sigma NULL: 0)].  rows: B x 5 floats of workspace.
 * Backward: grad3 = device [d loss / d out5[0], d / d out5[1], d / d out5[2]]; dmu, dlogstd [B][A] (dlogstd may be NULL). */
int emloco_ppo_actor_head_fwd(int B, int A, const float *mu, const float *logstd, const float *actions, const float *old_neglogp,
                              const float *advantages, const float *old_mu, const float *old_sigma, float e_clip, float *rows,
                              float *out5, void *stream);
int emloco_ppo_actor_head_bwd(int B, int A, const float *mu, const float *logstd, const float *actions, const float *old_neglogp,
                              const float *advantages, float e_clip, const float *grad3, float *dmu, float *dlogstd, void *stream);
/* Critic head: out1 = mean(max((v - R)^2, (v_old + clamp(v - v_old, -e, e) - R)^2)), or mean((R - v)^2) with clip_value = 0.
 * rows: B floats of workspace.  Backward: grad1 = device [d loss / d out1]; torch's tie rule (half each) and closed clamp interval. */
int emloco_ppo_critic_head_fwd(int B, const float *values, const float *old_values, const float *returns, float e_clip, int clip_value,
                               float *rows, float *out1, void *stream);
int emloco_ppo_critic_head_bwd(int B, const float *values, const float *old_values, const float *returns, float e_clip, int clip_value,
                               const float *grad1, float *dvalues, void *stream);
/* Discriminator head (amp_continuous.py:515-558): out4 = [mean BCEWithLogits(agent logits, 0), fraction of agent logits < 0,
 * mean BCEWithLogits(demo logits, 1), fraction of demo logits > 0].  rows: (n_agent + n_demo) x 2 floats of workspace.
 * Backward: grad2 = device [d loss / d out4[0], d loss / d out4[2]]. */
int emloco_ppo_disc_head_fwd(int n_agent, int n_demo, const float *agent_logits, const float *demo_logits, float *rows, float *out4, void *stream);
int emloco_ppo_disc_head_bwd(int n_agent, int n_demo, const float *agent_logits, const float *demo_logits, const float *grad2,
                             float *d_agent, float *d_demo, void *stream);

/* The minibatch gather of the learner (AMPDataset._get_item, amp_datasets.py:16-33): dst_t[r][:] = src_t[idx[r]][:] for n_tables
 * (<= 16) fp32 tables in one launch.  idx: n_rows int64 row ids on the device; src / dst / cols: HOST arrays of n_tables device
 * pointers / row widths (copied into the launch's arguments). */
int emloco_ppo_gather_rows(int n_tables, int n_rows, const int64_t *idx, const float *const *src, float *const *dst, const int *cols, void *stream);

/* The piece image of a B operand for EMLOCO_GEMM_B_SPLITIMG: B(n, k) = W[n * ld + k] (trans = 0) or W[k * ld + n] (trans = 1), n x k;
 * image: emloco_gemm_split_image_words(n, k) 32-bit words of device memory, 16-byte aligned (1.5 x the matrix, zero-padded to whole
 * 128 x 16 stages).  One small launch; valid until W changes.  Also accepted by emloco_gemm_relu_bwd (flags). */
int64_t emloco_gemm_split_image_words(int n, int k);
int emloco_gemm_split_pack(const float *W, int n, int k, int ld, int trans, uint32_t *image, void *stream);

/* Tile choice of the split mode (EMLOCO_GEMM_SPLIT): -1 (default) picks the 64 x 64 tile for launches whose 128 x 128 grid would
 * leave CUs idle, 0 / 1 force never / always.  Results do not depend on it (an output element's reduction order is the same in both
 * tiles: bit-equal, tests/test_emu_kernels.py); a tuning / A-B knob, also settable through EMLOCO_GEMM_SMALL. */
int emloco_gemm_set_small_tile(int mode);

/* HIP-event timing of the GEMM launches (same protocol as emloco_sim_timing_stats) */
int emloco_gemm_enable_timing(int on);
int emloco_gemm_timing_stats(int *n_launches, float *total_ms, double *total_flops);
/* algorithmic bytes (each operand and the output once, in their memory dtypes) of the launches the LAST emloco_gemm_timing_stats call
 * summed -- bytes / total_ms is the rate the GEMMs of that window moved their data at (the HBM view of launches whose reduction is short) */
int emloco_gemm_timing_bytes(double *total_bytes);

#ifdef __cplusplus
}
#endif
#endif
