#include <stdint.h>
// TEST INFRASTRUCTURE ONLY: runs emloco_amd/csrc/predictor_kernels.hip on the CPU through tests/emu/hip/.
#include "hip/hip_runtime.h"
#include "../../emloco_amd/csrc/predictor_kernels.hip"
#include "../../emloco_amd/csrc/attention16_kernels.hip"      // (includes attention_kernels.hip)

using namespace emloco;

extern "C" int emu_gemm_f32(int batch, int m, int n, int k, float alpha, const float *A, int lda, long sa, int ta,
                            const float *B, int ldb, long sb, int tb, float *C, int ldc, long sc, const float *bias,
                            int flags, int ksplit, float *ws) {
    GemmArgs g{batch, m, n, k, alpha, A, lda, sa, ta, B, ldb, sb, tb, C, ldc, sc, bias, flags, ksplit, ws, 0, 0};
    g.vec_a = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0) && (sa % 4 == 0);     // as emloco_gemm_f32 decides it
    g.vec_b = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0) && (sb % 4 == 0);
    g.vec_c = ((uintptr_t)C % 16 == 0) && (ldc % 4 == 0) && (sc % 4 == 0);     // the wide-store epilogue, as the launcher decides it
    const bool narrow = n <= 32;
    // flags & (1 << 20): test-only request for the 64 x 64 split tile (the launcher picks it by launch size, gemm_use_small_tile)
    g.small = ((flags & (1 << 20)) && (flags & 1024) && !(flags & 16) && g.vec_a && g.vec_b && n > 32) ? 1 : 0;
    g.flags &= ~(1 << 20);
    if (flags & 2048) {                     // EMLOCO_GEMM_B_SPLITIMG: B is the weight's piece image (emu_gemm_split_pack), as emloco_gemm_f32_ex serves it
        g.bimg = 1; g.vec_b = 1; g.tb = 0;
        g.flags &= ~2048;
    }
    const unsigned bt = g.small ? 64 : 128;
    const unsigned gx = narrow ? (n + 31) / 32 : (n + bt - 1) / bt, gy = (m + bt - 1) / bt, gz = batch * ksplit;
    for (unsigned z = 0; z < gz; ++z)
        for (unsigned y = 0; y < gy; ++y)
            for (unsigned x = 0; x < gx; ++x) {
                emu::launch(1, 256, [&] {
                    blockIdx.x = x; blockIdx.y = y; blockIdx.z = z;
                    const bool deep = (k + ksplit - 1) / ksplit > 256;     // as emloco_gemm_f32 picks the stage depth
                    GemmArgs gl = g;
                    if (!((g.flags & 1024) && !(g.flags & 16) && g.vec_a && g.vec_b && n > 32)) gl.flags &= ~1024;   // as emloco_gemm_f32_ex serves the split mode
                    gemm_pick(gl, deep)(gl);
                });
            }
    blockIdx.y = 0; blockIdx.z = 0;
    if (ksplit > 1) {
        const long total = (long)batch * m * n;
        emu::launch((unsigned)((total + 255) / 256), 256, [&] { gemm_splitk_reduce_kernel(g); });
    }
    return 0;
}

extern "C" long emu_gemm_split_image_words(int n, int k) { return (long)((n + 127) / 128) * ((k + 15) / 16) * SPLIT_IMG_SLOTS * 4; }
extern "C" int emu_gemm_split_pack(const float *W, int n, int k, int ld, int trans, unsigned *image) {
    const unsigned nst = (k + 15) / 16, nt = (n + 127) / 128;
    for (unsigned y = 0; y < nt; ++y)
        for (unsigned x = 0; x < nst; ++x) {
            gridDim.x = nst;
            emu::launch(1, 256, [&] { blockIdx.x = x; blockIdx.y = y; gridDim.x = nst; gemm_split_pack_kernel(W, n, k, ld, trans, (gemm_u32x4 *)image); });
        }
    blockIdx.y = 0;
    return 0;
}

// emloco_gemm_relu_bwd without the final fold: C = (A . B) o [y > 0] * scale, colpart [2 ceil(m / 128)][n] = 64-row column sums
extern "C" int emu_gemm_relu_bwd_ex(int m, int n, int k, const float *A, int lda, const float *B, int ldb, int tb, float *C, const float *y,
                                    float scale, float *colpart, int flags);
extern "C" int emu_gemm_relu_bwd(int m, int n, int k, const float *A, int lda, const float *B, int ldb, int tb, float *C, const float *y,
                                 float scale, float *colpart) {
    return emu_gemm_relu_bwd_ex(m, n, k, A, lda, B, ldb, tb, C, y, scale, colpart, 0);
}
extern "C" int emu_gemm_relu_bwd_ex(int m, int n, int k, const float *A, int lda, const float *B, int ldb, int tb, float *C, const float *y,
                                    float scale, float *colpart, int flags) {
    GemmArgs g{1, m, n, k, 1.0f, A, lda, 0, 0, B, ldb, 0, tb, C, n, 0, nullptr, 32 | (flags & ~2048), 1, nullptr, 0, 0, 0.0f, 0u, y, scale, colpart};
    if (flags & 2048) { g.bimg = 1; g.tb = 0; }
    g.vec_a = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
    g.vec_b = g.bimg ? 1 : ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
    g.vec_c = ((uintptr_t)C % 16 == 0) && ((uintptr_t)y % 16 == 0) && (n % 4 == 0) && !(flags & 256);     // wide epilogue accesses, as the launcher decides them
    const unsigned gx = (n + 127) / 128, gy = (m + 127) / 128;
    for (unsigned y0 = 0; y0 < gy; ++y0)
        for (unsigned x = 0; x < gx; ++x)
            emu::launch(1, 256, [&] { blockIdx.x = x; blockIdx.y = y0; blockIdx.z = 0; gemm_pick(g, k > 256)(g); });
    blockIdx.y = 0;
    return 0;
}

extern "C" int emu_obs_normalize(int rows, int cols, const float *x, int ldx, const float *mean, const float *var, float eps,
                                 float clip, int split, float *out0, int ld0, float *out1, int ld1) {
    for (int r = 0; r < rows; ++r)
        emu::launch((unsigned)((cols + 255) / 256), 256, [&] {
            blockIdx.y = r; gridDim.x = (cols + 255) / 256;
            obs_normalize_kernel(rows, cols, x, ldx, mean, var, eps, clip, split, out0, ld0, out1, ld1);
        });
    blockIdx.y = 0;
    return 0;
}
extern "C" int emu_rms_update(int rows, int cols, const float *x, int ldx, double *mean, double *var, const double *count_in, double *count_out,
                              int first_col) {
    emu::launch((unsigned)((cols + 63) / 64), 256, [&] { rms_update_kernel(rows, cols, x, ldx, mean, var, count_in, count_out, first_col); });
    return 0;
}
extern "C" int emu_rms_update_chunked(int rows, int cols, const float *x, int ldx, double *mean, double *var, const double *count_in,
                                      double *count_out, int first_col, double *ws) {
    const int chunks = (rows + RMS_CHUNK - 1) / RMS_CHUNK;
    for (int c = 0; c < chunks; ++c)
        emu::launch((unsigned)((cols + 63) / 64), 256, [&] { blockIdx.y = c; rms_partial_kernel(rows, cols, x, ldx, ws); });
    blockIdx.y = 0;
    emu::launch((unsigned)((cols + 255) / 256), 256, [&] { rms_merge_kernel(chunks, cols, ws, mean, var, count_in, count_out, first_col); });
    return 0;
}
extern "C" int emu_softmax_fwd(int n_seq, int rps, int cols, float scale, const float *S, const float *kp, float *P) {
    const long rows = (long)n_seq * rps;
    emu::launch((unsigned)((rows + 3) / 4), 256, [&] { softmax_fwd_kernel((int)rows, rps, cols, scale, S, kp, P); });
    return 0;
}
extern "C" int emu_softmax_bwd(int rows, int cols, float scale, const float *P, const float *dP, float *dS) {
    emu::launch((unsigned)((rows + 3) / 4), 256, [&] { softmax_bwd_kernel(rows, cols, scale, P, dP, dS); });
    return 0;
}
extern "C" int emu_layernorm_fwd(int rows, int d, float eps, const float *x, const float *res, const float *g, const float *b,
                                 float *y, float *mean, float *rstd) {
    if (d == 128) emu::launch((unsigned)((rows + 7) / 8), 256, [&] { layernorm_fwd128_kernel(rows, eps, x, res, g, b, y, mean, rstd, nullptr); });
    else emu::launch((unsigned)((rows + 3) / 4), 256, [&] { layernorm_fwd_kernel(rows, d, eps, x, res, g, b, y, mean, rstd, nullptr); });
    return 0;
}
extern "C" int emu_layernorm_bwd2(int rows, int d, const float *xr, const float *g, const float *mean, const float *rstd,
                                  const float *dy, const float *dy2, float *dxr, float *dg, float *db, float *ws);
extern "C" int emu_layernorm_bwd(int rows, int d, const float *xr, const float *g, const float *mean, const float *rstd,
                                 const float *dy, float *dxr, float *dg, float *db, float *ws) {
    return emu_layernorm_bwd2(rows, d, xr, g, mean, rstd, dy, nullptr, dxr, dg, db, ws);
}
extern "C" int emu_layernorm_bwd2(int rows, int d, const float *xr, const float *g, const float *mean, const float *rstd,
                                  const float *dy, const float *dy2, float *dxr, float *dg, float *db, float *ws) {
    const int nb = (rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK;
    if (d == 128) emu::launch((unsigned)nb, 256, [&] { layernorm_bwd128_kernel(rows, xr, g, mean, rstd, dy, dy2, dxr, ws); });       // (the capi's dispatch)
    else emu::launch((unsigned)nb, 256, [&] { layernorm_bwd_kernel(rows, d, xr, g, mean, rstd, dy, dy2, dxr, ws); });
    fold_rows([&](unsigned gx, unsigned gy, int n, int w, const float *in, float *o0, float *o1, int split) {
        for (unsigned y = 0; y < gy; ++y)
            emu::launch(gx, 256, [&] { blockIdx.y = y; rows_fold_kernel(n, w, in, o0, o1, split); });
        blockIdx.y = 0;
    }, nb, 2 * d, ws, dg, db, d);
    return 0;
}
// column sums as emloco_colsum_ex dispatches them: 16-byte rows of fp32 through the quad kernel, anything else through the scalar one
extern "C" long emu_colsum_workspace(int m, int n) { const int cs = cs_rows_for(m); return fold_workspace((m + cs - 1) / cs, n); }
extern "C" int emu_colsum(int m, int n, const float *X, float *out, float *ws, int force_scalar, int x16) {
    const int cs = cs_rows_for(m), np_ = (m + cs - 1) / cs;
    for (unsigned y = 0; y < (unsigned)np_; ++y) {
        if (n % 4 == 0 && !force_scalar) {
            if (x16) emu::launch((unsigned)((n + 255) / 256), 256, [&] { blockIdx.y = y; colsum4_partial_kernel<1>(m, n, X, ws, cs); });
            else emu::launch((unsigned)((n + 255) / 256), 256, [&] { blockIdx.y = y; colsum4_partial_kernel<0>(m, n, X, ws, cs); });
        } else emu::launch((unsigned)((n + 255) / 256), 256, [&] { blockIdx.y = y; colsum_partial_kernel(m, n, X, ws, x16, cs); });
    }
    blockIdx.y = 0;
    fold_rows([&](unsigned gx, unsigned gy, int nn, int w, const float *in, float *o0, float *o1, int split) {
        for (unsigned y = 0; y < gy; ++y)
            emu::launch(gx, 256, [&] { blockIdx.y = y; rows_fold_kernel(nn, w, in, o0, o1, split); });
        blockIdx.y = 0;
    }, np_, n, ws, out, out, n);
    return 0;
}
// step_count non-NULL: the counted variant (emloco_adam_clip_flat_counted) -- the bias corrections come from the device-side counter
extern "C" int emu_adam_clip_flat(long n, float *p, float *g, float *m, float *v, float lr, double b1, double b2, float eps, float wd,
                                  float bc1, float bc2s, float max_norm, float *ws, float *step_count) {
    const float *coef = nullptr;
    if (max_norm > 0.0f) {
        const int np_ = (int)((n + ADAM_BLOCK - 1) / ADAM_BLOCK);
        emu::launch((unsigned)np_, 256, [&] { sumsq_partial_kernel(n, g, ws + 4); });
        emu::launch(1, 256, [&] { clip_coef_kernel(np_, ws + 4, max_norm, ws); });
        coef = ws;
    }
    if (step_count) emu::launch(1, 64, [&] { adam_step_count_kernel(step_count, b1, b2, ws + 2); });
    const float *bc = step_count ? ws + 2 : nullptr;
    emu::launch((unsigned)((n + 255) / 256), 256, [&] { adam_flat_kernel(n, p, g, m, v, coef, lr, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), eps, wd, bc1, bc2s, bc); });
    return 0;
}
extern "C" long emu_layernorm_bwd_workspace(int rows, int d) { return fold_workspace((rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK, 2L * d); }
extern "C" int emu_locoval_fwd(int B, const float *traj, int ts, const float *pose, const float *vel, const float *w1,
                               const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                               float *value, float *x100, float *h1, float *h2, float *angle) {
    emu::launch((unsigned)B, 64, [&] { locoval_fwd_kernel(B, traj, ts, pose, vel, w1, b1, w2, b2, w3, b3, value, x100, h1, h2, angle, (const float *)nullptr); });
    return 0;
}
extern "C" int emu_locoval_bwd(int B, const float *traj, int ts, const float *pose, const float *vel, const float *w1,
                               const float *w2, const float *w3, const float *value, const float *x100, const float *h1,
                               const float *h2, const float *angle, const float *dvalue, float *dparams, float *dtraj, float *ws) {
    emu::launch((unsigned)B, 64, [&] { locoval_bwd_kernel(B, traj, ts, pose, vel, w1, w2, w3, value, x100, h1, h2, angle, dvalue, ws, dtraj, (const int32_t *)nullptr); });
    emu::launch((unsigned)((LV_NPARAM + 255) / 256), 256, [&] { locoval_reduce_kernel(B, ws, dparams, (const float *)nullptr); });
    return 0;
}

static int g_attn_prec = 0;     // 1: bf16 operands (EMLOCO_ATTN_BF16)
static float g_attn_drop_p = 0.0f;   // > 0: dropout on the probabilities with g_attn_drop_seed
static unsigned g_attn_drop_seed = 0;
extern "C" void emu_attention_set_precision(int p) { g_attn_prec = p; }
extern "C" void emu_attention_set_dropout(float p, unsigned seed) { g_attn_drop_p = p; g_attn_drop_seed = seed; }
extern "C" int emu_attn_keep(unsigned seed, unsigned bh, unsigned q, unsigned k, float p) { return emloco::at_keep_bit(emloco::at_head_key(seed, bh), q, k, emloco::at_drop_thr8(p)) ? 1 : 0; }
#define ATTN_DISPATCH(K) do { const bool dr_ = g_attn_drop_p > 0.0f; \
    if (g_attn_prec == 2 && dr_) K<2, 1>(a); else if (g_attn_prec == 2) K<2, 0>(a); else \
    if (g_attn_prec && dr_) K<1, 1>(a); else if (g_attn_prec) K<1, 0>(a); else if (dr_) K<0, 1>(a); else K<0, 0>(a); } while (0)
extern "C" int emu_attention_fwd_queries(int n_seq, int S, int Sq, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                         float *out, float *lse) {
    AttnArgs a{n_seq, S, nhead, d_model, Sq, scale, qkv, key_bias, out, lse, nullptr, nullptr, nullptr, g_attn_drop_p, emloco::at_drop_scale(g_attn_drop_p), g_attn_drop_seed, emloco::at_drop_thr8(g_attn_drop_p)};
    for (int y = 0; y < n_seq * nhead; ++y)
        for (int x = 0; x < (Sq + 127) / 128; ++x)
            emu::launch(1, 256, [&] { blockIdx.x = x; blockIdx.y = y; ATTN_DISPATCH(attn_fwd_kernel); });
    blockIdx.x = 0; blockIdx.y = 0;
    return 0;
}
extern "C" int emu_attention_fwd(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 float *out, float *lse) {
    return emu_attention_fwd_queries(n_seq, S, S, nhead, d_model, scale, qkv, key_bias, out, lse);
}
// as emloco_attention_bwd_queries: the launcher zero-fills the dQ third of the rows that did not attend
extern "C" int emu_attention_bwd_queries(int n_seq, int S, int Sq, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                         float *out, float *lse, const float *dout, float *dqkv, float *dsum) {
    AttnArgs a{n_seq, S, nhead, d_model, Sq, scale, qkv, key_bias, out, lse, dout, dqkv, dsum, g_attn_drop_p, emloco::at_drop_scale(g_attn_drop_p), g_attn_drop_seed, emloco::at_drop_thr8(g_attn_drop_p)};
    if (Sq < S)
        for (long r = 0; r < (long)n_seq * S; ++r)
            for (int c = 0; c < d_model; ++c) dqkv[r * 3 * d_model + c] = 0.0f;
    for (int pass = 0; pass < 2; ++pass)
        for (int y = 0; y < n_seq * nhead; ++y)
            for (int x = 0; x < ((pass == 0 ? Sq : S) + 127) / 128; ++x)
                emu::launch(1, 256, [&] {
                    blockIdx.x = x; blockIdx.y = y;
                    if (pass == 0) { ATTN_DISPATCH(attn_bwd_dq_kernel); }
                    else { ATTN_DISPATCH(attn_bwd_dkv_kernel); }
                });
    blockIdx.x = 0; blockIdx.y = 0;
    return 0;
}
extern "C" int emu_attention_bwd(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 float *out, float *lse, const float *dout, float *dqkv, float *dsum) {
    return emu_attention_bwd_queries(n_seq, S, S, nhead, d_model, scale, qkv, key_bias, out, lse, dout, dqkv, dsum);
}

extern "C" int emu_locoval_returns(const EmlocoLocoValStep *t, const float *rewards, const float *amp, const int64_t *dones, const uint8_t *inv) {
    EmlocoLocoValStep s = *t;
    emu::launch((unsigned)s.n_env, 64, [&] { emloco::locoval_returns_kernel(s, rewards, amp, dones, inv); });
    return 0;
}
extern "C" int emu_locoval_returns_finish(const EmlocoLocoValStep *t, const float *amp) {
    EmlocoLocoValStep s = *t;
    emu::launch((unsigned)((s.n_env + 255) / 256), 256, [&] { emloco::locoval_returns_finish_kernel(s, amp); });
    return 0;
}
extern "C" int emu_locoval_fit_grad(int n, const float *value, const float *target, const float *weight, float *dvalue, float *tail, int32_t *slot) {
    emu::launch(1, 1024, [&] { emloco::locoval_fit_grad_kernel(n, value, target, weight, dvalue, tail, slot); });
    return 0;
}
extern "C" int emu_adamw_gated(int n, float *p, const float *g, float *m, float *v, const float *si, float *so, const float *tail, float lr,
                               float b1, float b2, float eps, float wd, double *stats) {
    emu::launch((unsigned)((n + 255) / 256), 256, [&] { emloco::adamw_gated_kernel(n, p, g, m, v, si, so, tail, lr, b1, b2, eps, wd, stats); });
    return 0;
}

// ---- the chained feed-forward kernels (emloco_amd/csrc/ffn_kernels.hip), launched as emloco_ffn_fwd / emloco_ffn_bwd_input launch them
#include "../../emloco_amd/csrc/ffn_kernels.hip"
static float *g_ffn_colpart = nullptr;       // optional [ceil(M / 32)][F] column partials of the next backward launch (emu_ffn_set_colpart)
extern "C" void emu_ffn_set_colpart(float *p) { g_ffn_colpart = p; }
// the forward's optional residual + LayerNorm epilogue (emloco_ffn_fwd_norm): all NULL = off
static const float *g_ffn_res = nullptr, *g_ffn_gamma = nullptr, *g_ffn_beta = nullptr;
static float *g_ffn_xr = nullptr, *g_ffn_mean = nullptr, *g_ffn_rstd = nullptr, g_ffn_eps = 0.0f;
extern "C" void emu_ffn_set_norm(const float *res, const float *gamma, const float *beta, float eps, float *xr, float *mean, float *rstd) {
    g_ffn_res = res; g_ffn_gamma = gamma; g_ffn_beta = beta; g_ffn_eps = eps; g_ffn_xr = xr; g_ffn_mean = mean; g_ffn_rstd = rstd;
}
extern "C" int emu_ffn_chain(int mode, int M, int F, const float *x, const unsigned short *P, const unsigned short *Q, const float *b1,
                             const float *b2, unsigned short *h, unsigned short *dz1, unsigned *mask, float *out, float drop_p, unsigned seed1, unsigned seed2) {
    FfnArgs a{M, F, x, P, Q, b1, b2, h, dz1, mask, out, drop_p, 1.0f / (1.0f - drop_p), seed1, (unsigned)(drop_p * 65536.0f), seed2, g_ffn_colpart,
              g_ffn_res, g_ffn_gamma, g_ffn_beta, g_ffn_xr, g_ffn_mean, g_ffn_rstd, g_ffn_eps};
    const unsigned grid = (unsigned)((M + FFN_ROWS - 1) / FFN_ROWS);
    if (mode == 0 && drop_p > 0.0f) emu::launch(grid, FFN_THREADS, [&] { ffn_chain_kernel<0, 1>(a); });
    else if (mode == 0) emu::launch(grid, FFN_THREADS, [&] { ffn_chain_kernel<0, 0>(a); });
    else emu::launch(grid, FFN_THREADS, [&] { ffn_chain_kernel<1, 0>(a); });
    return 0;
}
extern "C" int emu_ffn_keep(unsigned seed, unsigned row, unsigned hidden, float p) { return ffn_keep16(seed, row, hidden, (unsigned)(p * 65536.0f)) ? 1 : 0; }
extern "C" int emu_ffn_out_keep(unsigned seed, unsigned long long idx, float p) { return ffn_out_keep(seed, idx, p) ? 1 : 0; }
extern "C" int emu_drop_keep(unsigned seed, unsigned long long idx, float p) { return drop_keep(seed, idx, p) ? 1 : 0; }

// ---- round 5's attention kernels (attention16_kernels.hip), launched as the C ABI launches them.  which = 1: bf16 memory, two blocks per
// wave (qkv / dqkv hold bf16); which = 0: round 4's kernels of that mode (attn_*<1, DROP, 1>); which = 3: the split mode on piece-plane tile
// images (fp32 memory; forward: three pieces, two blocks per wave; backward as the C ABI launches it by default since round 6: TWO pieces,
// dQ two blocks per wave); which = 5: the same with the three-piece backward (EMLOCO_ATTN_BWD_PIECES=3); which = 2: round 4's split-mode
// kernels (attn_*<2, DROP, 0>).
extern "C" int emu_attention16(int which, int n_seq, int S, int Sq, int nhead, int d_model, float scale, const void *qkv, const float *key_bias,
                               float *out, float *lse, const float *dout, void *dqkv, float *dsum, float drop_p, unsigned seed) {
    AttnArgs a{n_seq, S, nhead, d_model, Sq, scale, (const float *)qkv, key_bias, out, lse, dout, (float *)dqkv, dsum, drop_p, emloco::at_drop_scale(drop_p), seed,
               emloco::at_drop_thr8(drop_p)};
    const unsigned rows_per_wg = which == 1 ? 256 : 128;
    const unsigned rows_dq = (which == 1 || which == 3) ? 256 : 128;
    const bool dr = drop_p > 0.0f;
#define A16_RUN(KOLD_BF, KOLD_SP, KNEW) do { \
        if (which == 1) { if (dr) KNEW<1, 2, 1, 1>(a); else KNEW<1, 2, 0, 1>(a); } \
        else if (which == 5) { if (dr) KNEW<3, 1, 1, 0>(a); else KNEW<3, 1, 0, 0>(a); } \
        else if (which == 0) { if (dr) KOLD_BF<1, 1, 1>(a); else KOLD_BF<1, 0, 1>(a); } \
        else { if (dr) KOLD_SP<2, 1, 0>(a); else KOLD_SP<2, 0, 0>(a); } } while (0)
    const bool split_new = which == 3 || which == 5;
    const unsigned rows_fwd = !dout ? (split_new ? 256 : rows_per_wg) : rows_dq;      // the split-mode forward runs two blocks per wave (attention_capi.hip: A16_SPLIT_G_FWD)
    for (unsigned y = 0; y < (unsigned)(n_seq * nhead); ++y)
        for (unsigned x = 0; x < (Sq + rows_fwd - 1) / rows_fwd; ++x)
            emu::launch(1, 256, [&] {
                blockIdx.x = x; blockIdx.y = y;
                if (!dout && split_new) { if (dr) attn16_fwd_kernel<3, 2, 1, 0>(a); else attn16_fwd_kernel<3, 2, 0, 0>(a); }
                else if (dout && which == 3) { if (dr) attn16_bwd_dq_kernel<2, 2, 1, 0>(a); else attn16_bwd_dq_kernel<2, 2, 0, 0>(a); }
                else if (!dout) A16_RUN(attn_fwd_kernel, attn_fwd_kernel, attn16_fwd_kernel);
                else A16_RUN(attn_bwd_dq_kernel, attn_bwd_dq_kernel, attn16_bwd_dq_kernel);
            });
    if (dout)
        for (unsigned y = 0; y < (unsigned)(n_seq * nhead); ++y)
            for (unsigned x = 0; x < (S + rows_per_wg - 1) / rows_per_wg; ++x)
                emu::launch(1, 256, [&] {
                    blockIdx.x = x; blockIdx.y = y;
                    if (which == 3) { if (dr) attn16_bwd_dkv_kernel<2, 1, 1, 0>(a); else attn16_bwd_dkv_kernel<2, 1, 0, 0>(a); }
                    else A16_RUN(attn_bwd_dkv_kernel, attn_bwd_dkv_kernel, attn16_bwd_dkv_kernel);
                });
#undef A16_RUN
    blockIdx.x = 0; blockIdx.y = 0;
    return 0;
}
