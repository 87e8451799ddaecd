// attention_capi.hip -- C ABI of the fused attention kernels (include/emloco_predictor.h), a translation unit of its own: the
// GEMM / LayerNorm unit (predictor_capi.hip) is built with -fno-slp-vectorize (the split-mode GEMMs lose 4 % to the packed-fp32
// pairing of their piece arithmetic), the attention kernels keep the SLP vectoriser (without it the bf16 kernels are 8-18 % slower:
// their softmax / dropout arithmetic pairs well) -- emloco_amd/build.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdlib.h>
#include "attention16_kernels.hip"      // (includes attention_kernels.hip)
#include "../../include/emloco_predictor.h"

namespace {
// EMLOCO_ATTN16_OLD=1: the bf16-in-memory mode on round 4's kernels (one block of 32 rows per wave, fp32 LDS tiles) -- A/B knob
bool attn16_old() { static const bool v = [] { const char *e = getenv("EMLOCO_ATTN16_OLD"); return e && e[0] == '1'; }(); return v; }
// Pieces per operand in the split mode's BACKWARD kernels (round 6).  Default 2: every operand is the sum of two bf16 pieces and a product
// the three piece products above 2^-16 of it -- gradients to ~2^-17 of their magnitude (measured on the shipped-depth model,
// tools/exp/graderr.py: 2-3e-6 of a tensor's scale next to the loss, where three pieces give 6e-7 and the test bar is 2e-4; the early
// layers' 3e-5 .. 1e-4 do not move: that is the ReLU / LayerNorm stack's own fp32 noise), half the matrix instructions and 40 % of the
// piece arithmetic: dQ + dK/dV 5.99 -> 4.09 ms per launch pair at the train step's size.  The FORWARD stays on three pieces (logits to
// 5e-7).  EMLOCO_ATTN_BWD_PIECES=3 restores the three-piece backward.
int attn_bwd_pieces() { static const int v = [] { const char *e = getenv("EMLOCO_ATTN_BWD_PIECES"); return e && e[0] == '3' ? 3 : 2; }(); return v; }
int pfail(int code, const char *what, hipError_t e = hipSuccess) {
    if (e != hipSuccess) fprintf(stderr, "[emloco] %s: %s\n", what, hipGetErrorString(e));
    else fprintf(stderr, "[emloco] %s\n", what);
    return code;
}
}  // namespace

#define PHIPCHK(expr)                                                  \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return pfail(-2, #expr, e_);             \
    } while (0)

extern "C" {

int emloco_attention_fwd(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                         float *out, float *lse, void *stream) {
    return emloco_attention_fwd_ex(n_seq, S, nhead, d_model, scale, qkv, key_bias, out, lse, 0, stream);
}

int emloco_attention_fwd_ex(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                            float *out, float *lse, int flags, void *stream) {
    return emloco_attention_fwd_dropout(n_seq, S, nhead, d_model, scale, qkv, key_bias, out, lse, flags, 0.0f, 0u, stream);
}

int emloco_attention_fwd_dropout(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 float *out, float *lse, int flags, float drop_p, uint32_t drop_seed, void *stream) {
    return emloco_attention_fwd_queries(n_seq, S, S, nhead, d_model, scale, qkv, key_bias, out, lse, flags, drop_p, drop_seed, stream);
}

int emloco_attention_fwd_queries(int n_seq, int S, int n_query, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 float *out, float *lse, int flags, float drop_p, uint32_t drop_seed, void *stream) {
    if (n_query < 1 || n_query > S) return pfail(-1, "emloco_attention_fwd: n_query must be in [1, S]");
    if (n_seq < 1 || S < 1 || nhead < 1 || d_model != nhead * AT_DH || !qkv || !out || !lse || !(drop_p >= 0.0f && drop_p < 1.0f))
        return pfail(-1, "emloco_attention_fwd: bad argument (head dim must be 32, 0 <= drop_p < 1)");
    if ((long)n_seq * nhead > 65535) return pfail(-1, "emloco_attention_fwd: n_seq * nhead exceeds the grid limit");
    emloco::AttnArgs a{n_seq, S, nhead, d_model, n_query, scale, qkv, key_bias, out, lse, nullptr, nullptr, nullptr, drop_p, emloco::at_drop_scale(drop_p), drop_seed, emloco::at_drop_thr8(drop_p)};
    const dim3 grid((unsigned)((n_query + 127) / 128), (unsigned)(n_seq * nhead));
    const bool bf = (flags & EMLOCO_ATTN_BF16) != 0, dr = drop_p > 0.0f;
    hipStream_t st = (hipStream_t)stream;
    if (flags & EMLOCO_ATTN_QKV_BF16MEM) {
        if (!bf) return pfail(-1, "emloco_attention_fwd: a bf16 q|k|v tensor needs EMLOCO_ATTN_BF16");
        if (!attn16_old()) {                                 // round 5: two blocks of 32 queries per wave, bf16 tile images (attention16_kernels.hip)
            const dim3 grid16((unsigned)((n_query + 255) / 256), (unsigned)(n_seq * nhead));
            if (dr) hipLaunchKernelGGL((emloco::attn16_fwd_kernel<1, 2, 1, 1>), grid16, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((emloco::attn16_fwd_kernel<1, 2, 0, 1>), grid16, dim3(256), 0, st, a);
            PHIPCHK(hipGetLastError());
            return 0;
        }
        if (dr) hipLaunchKernelGGL((emloco::attn_fwd_kernel<1, 1, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emloco::attn_fwd_kernel<1, 0, 1>), grid, dim3(256), 0, st, a);
        PHIPCHK(hipGetLastError());
        return 0;
    }
    const bool sp = !bf && (flags & EMLOCO_ATTN_SPLIT) != 0;
    if (sp && !attn16_old()) {                               // round 5: the split mode on the piece-plane tile images (attention16_kernels.hip, NP = 3)
#ifndef A16_SPLIT_G_FWD
#define A16_SPLIT_G_FWD 2                                    /* blocks of 32 queries per wave of the split-mode forward (measured: 2.006 -> 1.855 ms per launch) */
#endif
        const dim3 gridf((unsigned)((n_query + 128 * A16_SPLIT_G_FWD - 1) / (128 * A16_SPLIT_G_FWD)), (unsigned)(n_seq * nhead));
        if (dr) hipLaunchKernelGGL((emloco::attn16_fwd_kernel<3, A16_SPLIT_G_FWD, 1, 0>), gridf, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emloco::attn16_fwd_kernel<3, A16_SPLIT_G_FWD, 0, 0>), gridf, dim3(256), 0, st, a);
        PHIPCHK(hipGetLastError());
        return 0;
    }
    if (sp && dr) hipLaunchKernelGGL((emloco::attn_fwd_kernel<2, 1>), grid, dim3(256), 0, st, a);
    else if (sp) hipLaunchKernelGGL((emloco::attn_fwd_kernel<2, 0>), grid, dim3(256), 0, st, a);
    else if (bf && dr) hipLaunchKernelGGL((emloco::attn_fwd_kernel<1, 1>), grid, dim3(256), 0, st, a);
    else if (bf) hipLaunchKernelGGL((emloco::attn_fwd_kernel<1, 0>), grid, dim3(256), 0, st, a);
    else if (dr) hipLaunchKernelGGL((emloco::attn_fwd_kernel<0, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((emloco::attn_fwd_kernel<0, 0>), grid, dim3(256), 0, st, a);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_attention_bwd(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                         const float *out, const float *lse, const float *dout, float *dqkv, float *dsum, void *stream) {
    return emloco_attention_bwd_ex(n_seq, S, nhead, d_model, scale, qkv, key_bias, out, lse, dout, dqkv, dsum, 0, stream);
}

int emloco_attention_bwd_ex(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                            const float *out, const float *lse, const float *dout, float *dqkv, float *dsum, int flags, void *stream) {
    return emloco_attention_bwd_dropout(n_seq, S, nhead, d_model, scale, qkv, key_bias, out, lse, dout, dqkv, dsum, flags, 0.0f, 0u, stream);
}

int emloco_attention_bwd_dropout(int n_seq, int S, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 const float *out, const float *lse, const float *dout, float *dqkv, float *dsum, int flags,
                                 float drop_p, uint32_t drop_seed, void *stream) {
    return emloco_attention_bwd_queries(n_seq, S, S, nhead, d_model, scale, qkv, key_bias, out, lse, dout, dqkv, dsum, flags, drop_p, drop_seed, stream);
}

int emloco_attention_bwd_queries(int n_seq, int S, int n_query, int nhead, int d_model, float scale, const float *qkv, const float *key_bias,
                                 const float *out, const float *lse, const float *dout, float *dqkv, float *dsum, int flags,
                                 float drop_p, uint32_t drop_seed, void *stream) {
    if (n_query < 1 || n_query > S) return pfail(-1, "emloco_attention_bwd: n_query must be in [1, S]");
    if (n_seq < 1 || S < 1 || nhead < 1 || d_model != nhead * AT_DH || !qkv || !out || !lse || !dout || !dqkv || !dsum ||
        !(drop_p >= 0.0f && drop_p < 1.0f))
        return pfail(-1, "emloco_attention_bwd: bad argument (head dim must be 32; dsum = n_seq * nhead * S floats; 0 <= drop_p < 1)");
    if ((long)n_seq * nhead > 65535) return pfail(-1, "emloco_attention_bwd: n_seq * nhead exceeds the grid limit");
    emloco::AttnArgs a{n_seq, S, nhead, d_model, n_query, scale, qkv, key_bias, const_cast<float *>(out), const_cast<float *>(lse), dout, dqkv, dsum,
                       drop_p, emloco::at_drop_scale(drop_p), drop_seed, emloco::at_drop_thr8(drop_p)};
    const dim3 grid((unsigned)((S + 127) / 128), (unsigned)(n_seq * nhead)), qgrid((unsigned)((n_query + 127) / 128), (unsigned)(n_seq * nhead));
    const bool bf = (flags & EMLOCO_ATTN_BF16) != 0, dr = drop_p > 0.0f;
    hipStream_t st = (hipStream_t)stream;
    const bool q16 = (flags & EMLOCO_ATTN_QKV_BF16MEM) != 0;
    if (q16 && !bf) return pfail(-1, "emloco_attention_bwd: a bf16 q|k|v tensor needs EMLOCO_ATTN_BF16");
    const size_t esz = q16 ? 2 : sizeof(float);
    // rows that do not attend get dQ = 0 (the Q third of every dqkv row; the live rows are overwritten below)
    // (round 5's dK / dV kernels write those zeros themselves: the memset is for round 4's kernels and the modes they still serve)
    const bool new_kernels = !attn16_old() && (q16 || (!bf && (flags & EMLOCO_ATTN_SPLIT) != 0));
    if (n_query < S && !new_kernels) PHIPCHK(hipMemset2DAsync(dqkv, 3 * (size_t)d_model * esz, 0, (size_t)d_model * esz, (size_t)n_seq * S, st));
    if (q16 && !attn16_old()) {
        const dim3 grid16((unsigned)((S + 255) / 256), (unsigned)(n_seq * nhead)), qgrid16((unsigned)((n_query + 255) / 256), (unsigned)(n_seq * nhead));
        if (dr) hipLaunchKernelGGL((emloco::attn16_bwd_dq_kernel<1, 2, 1, 1>), qgrid16, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emloco::attn16_bwd_dq_kernel<1, 2, 0, 1>), qgrid16, dim3(256), 0, st, a);
        PHIPCHK(hipGetLastError());
        if (dr) hipLaunchKernelGGL((emloco::attn16_bwd_dkv_kernel<1, 2, 1, 1>), grid16, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emloco::attn16_bwd_dkv_kernel<1, 2, 0, 1>), grid16, dim3(256), 0, st, a);
        PHIPCHK(hipGetLastError());
        return 0;
    }
    if (q16) {
        if (dr) hipLaunchKernelGGL((emloco::attn_bwd_dq_kernel<1, 1, 1>), qgrid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emloco::attn_bwd_dq_kernel<1, 0, 1>), qgrid, dim3(256), 0, st, a);
        PHIPCHK(hipGetLastError());
        if (dr) hipLaunchKernelGGL((emloco::attn_bwd_dkv_kernel<1, 1, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emloco::attn_bwd_dkv_kernel<1, 0, 1>), grid, dim3(256), 0, st, a);
        PHIPCHK(hipGetLastError());
        return 0;
    }
    // first kernel: dQ, also writes D = rowsum(dO o O); second: dK, dV
    const bool sp = !bf && (flags & EMLOCO_ATTN_SPLIT) != 0;
    if (sp && !attn16_old()) {
        if (attn_bwd_pieces() == 2) {
#ifndef A16_NP2_DQ_G
#define A16_NP2_DQ_G 2                                       /* two blocks of 32 queries per wave fit with two pieces (4.30 -> 4.09 ms); dK/dV: one (two: 4.98) */
#endif
#ifndef A16_NP2_DKV_G
#define A16_NP2_DKV_G 1
#endif
            const dim3 qg2((unsigned)((n_query + 128 * A16_NP2_DQ_G - 1) / (128 * A16_NP2_DQ_G)), (unsigned)(n_seq * nhead));
            const dim3 kg2((unsigned)((S + 128 * A16_NP2_DKV_G - 1) / (128 * A16_NP2_DKV_G)), (unsigned)(n_seq * nhead));
            if (dr) hipLaunchKernelGGL((emloco::attn16_bwd_dq_kernel<2, A16_NP2_DQ_G, 1, 0>), qg2, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((emloco::attn16_bwd_dq_kernel<2, A16_NP2_DQ_G, 0, 0>), qg2, dim3(256), 0, st, a);
            PHIPCHK(hipGetLastError());
            if (dr) hipLaunchKernelGGL((emloco::attn16_bwd_dkv_kernel<2, A16_NP2_DKV_G, 1, 0>), kg2, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((emloco::attn16_bwd_dkv_kernel<2, A16_NP2_DKV_G, 0, 0>), kg2, dim3(256), 0, st, a);
            PHIPCHK(hipGetLastError());
            return 0;
        }
        if (dr) hipLaunchKernelGGL((emloco::attn16_bwd_dq_kernel<3, 1, 1, 0>), qgrid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emloco::attn16_bwd_dq_kernel<3, 1, 0, 0>), qgrid, dim3(256), 0, st, a);
        PHIPCHK(hipGetLastError());
        if (dr) hipLaunchKernelGGL((emloco::attn16_bwd_dkv_kernel<3, 1, 1, 0>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((emloco::attn16_bwd_dkv_kernel<3, 1, 0, 0>), grid, dim3(256), 0, st, a);
        PHIPCHK(hipGetLastError());
        return 0;
    }
    if (sp && dr) hipLaunchKernelGGL((emloco::attn_bwd_dq_kernel<2, 1>), qgrid, dim3(256), 0, st, a);
    else if (sp) hipLaunchKernelGGL((emloco::attn_bwd_dq_kernel<2, 0>), qgrid, dim3(256), 0, st, a);
    else if (bf && dr) hipLaunchKernelGGL((emloco::attn_bwd_dq_kernel<1, 1>), qgrid, dim3(256), 0, st, a);
    else if (bf) hipLaunchKernelGGL((emloco::attn_bwd_dq_kernel<1, 0>), qgrid, dim3(256), 0, st, a);
    else if (dr) hipLaunchKernelGGL((emloco::attn_bwd_dq_kernel<0, 1>), qgrid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((emloco::attn_bwd_dq_kernel<0, 0>), qgrid, dim3(256), 0, st, a);
    PHIPCHK(hipGetLastError());
    if (sp && dr) hipLaunchKernelGGL((emloco::attn_bwd_dkv_kernel<2, 1>), grid, dim3(256), 0, st, a);
    else if (sp) hipLaunchKernelGGL((emloco::attn_bwd_dkv_kernel<2, 0>), grid, dim3(256), 0, st, a);
    else if (bf && dr) hipLaunchKernelGGL((emloco::attn_bwd_dkv_kernel<1, 1>), grid, dim3(256), 0, st, a);
    else if (bf) hipLaunchKernelGGL((emloco::attn_bwd_dkv_kernel<1, 0>), grid, dim3(256), 0, st, a);
    else if (dr) hipLaunchKernelGGL((emloco::attn_bwd_dkv_kernel<0, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((emloco::attn_bwd_dkv_kernel<0, 0>), grid, dim3(256), 0, st, a);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_attention_keep_mask(uint32_t seed, int n_seq_heads, int S, float p, uint8_t *host_out) {
    if (n_seq_heads < 1 || S < 1 || !host_out || !(p >= 0.0f && p < 1.0f)) return pfail(-1, "emloco_attention_keep_mask: bad argument");
    const unsigned thr = emloco::at_drop_thr8(p);
    for (int bh = 0; bh < n_seq_heads; ++bh) {
        const unsigned hk = emloco::at_head_key(seed, (unsigned)bh);
        for (int q = 0; q < S; ++q)
            for (int k = 0; k < S; ++k) host_out[((size_t)bh * S + q) * S + k] = emloco::at_keep_bit(hk, (unsigned)q, (unsigned)k, thr) ? 1 : 0;
    }
    return 0;
}

}  // extern "C"
