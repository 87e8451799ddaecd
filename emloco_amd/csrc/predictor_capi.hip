// predictor_capi.hip -- C ABI (include/emloco_predictor.h) over the predictor / LocoVal kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <stdlib.h>
#include <vector>
#include "predictor_kernels.hip"
#include "../../include/emloco_predictor.h"

namespace {
int pfail(int code, const char *what, hipError_t e = hipSuccess) {
    if (e != hipSuccess) fprintf(stderr, "[emloco] %s: %s\n", what, hipGetErrorString(e));
    else fprintf(stderr, "[emloco] %s\n", what);
    return code;
}
constexpr int kRing = 4096;
std::vector<hipEvent_t> g_e0, g_e1;
std::vector<double> g_fl, g_by;
double g_last_bytes = 0.0;        // algorithmic bytes of the launches the last emloco_gemm_timing_stats call summed
int g_head = 0, g_count = 0;
bool g_timing = false;
}  // namespace

#define PHIPCHK(expr)                                                  \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return pfail(-2, #expr, e_);             \
    } while (0)

extern "C" {

int emloco_gemm_enable_timing(int on) {
    if (on && g_e0.empty()) {
        g_e0.resize(kRing); g_e1.resize(kRing); g_fl.resize(kRing); g_by.resize(kRing);
        for (int i = 0; i < kRing; ++i) { PHIPCHK(hipEventCreate(&g_e0[i])); PHIPCHK(hipEventCreate(&g_e1[i])); }
    }
    g_timing = on != 0; g_head = 0; g_count = 0;
    return 0;
}

int emloco_gemm_timing_stats(int *n_launches, float *total_ms, double *total_flops) {
    if (!n_launches || !total_ms || !total_flops) return pfail(-1, "emloco_gemm_timing_stats: null argument");
    *n_launches = 0; *total_ms = 0.0f; *total_flops = 0.0; g_last_bytes = 0.0;
    for (int k = 0; k < g_count; ++k) {
        const int slot = (g_head - 1 - k + 2 * kRing) % kRing;
        PHIPCHK(hipEventSynchronize(g_e1[slot]));
        float ms = 0.0f;
        PHIPCHK(hipEventElapsedTime(&ms, g_e0[slot], g_e1[slot]));
        *total_ms += ms; *total_flops += g_fl[slot]; g_last_bytes += g_by[slot]; ++*n_launches;
    }
    g_count = 0;
    return 0;
}

int emloco_gemm_timing_bytes(double *total_bytes) {
    if (!total_bytes) return pfail(-1, "emloco_gemm_timing_bytes: null argument");
    *total_bytes = g_last_bytes;
    return 0;
}

int emloco_gemm_f32(int batch, int m, int n, int k, float alpha, const float *A, int lda, int64_t stride_a, int trans_a,
                    const float *B, int ldb, int64_t stride_b, int trans_b, float *C, int ldc, int64_t stride_c,
                    const float *bias, int flags, int ksplit, float *workspace, void *stream) {
    if (flags & EMLOCO_GEMM_DROPOUT) return pfail(-1, "emloco_gemm_f32: the dropout epilogue needs emloco_gemm_f32_ex");
    return emloco_gemm_f32_ex(batch, m, n, k, alpha, A, lda, stride_a, trans_a, B, ldb, stride_b, trans_b, C, ldc, stride_c, bias, flags,
                              ksplit, workspace, 0.0f, 0u, stream);
}

int emloco_act_bwd(int64_t total, const float *dy, const float *y, int relu, float drop_p, uint32_t drop_seed, float *dz, void *stream) {
    if (total < 1 || !dy || !dz || (relu && !y) || !(drop_p >= 0.0f && drop_p < 1.0f)) return pfail(-1, "emloco_act_bwd: bad argument");
    hipLaunchKernelGGL(emloco::act_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (long)total, dy, y, relu, drop_p, drop_seed, dz);
    PHIPCHK(hipGetLastError());
    return 0;
}

// tile choice of the split mode: -1 = by launch size (gemm_use_small_tile), 0 / 1 = never / always the 64 x 64 tile
static int g_force_small = getenv("EMLOCO_GEMM_SMALL") ? atoi(getenv("EMLOCO_GEMM_SMALL")) : -1;
int emloco_gemm_set_small_tile(int mode) {
    if (mode < -1 || mode > 1) return pfail(-1, "emloco_gemm_set_small_tile: mode -1 (by launch size), 0 (never) or 1 (always)");
    g_force_small = mode;
    return 0;
}

int emloco_gemm_f32_ex(int batch, int m, int n, int k, float alpha, const float *A, int lda, int64_t stride_a, int trans_a,
                       const float *B, int ldb, int64_t stride_b, int trans_b, float *C, int ldc, int64_t stride_c,
                       const float *bias, int flags, int ksplit, float *workspace, float drop_p, uint32_t drop_seed, void *stream) {
    if ((flags & EMLOCO_GEMM_DROPOUT) && (!(drop_p > 0.0f && drop_p < 1.0f) || ldc != n || (batch > 1 && stride_c != (int64_t)m * n)))
        return pfail(-1, "emloco_gemm_f32_ex: dropout needs 0 < p < 1 and a dense output (the mask is indexed by the flat element)");
    if (batch < 1 || m < 1 || n < 1 || k < 1 || !A || !B || !C) return pfail(-1, "emloco_gemm_f32: bad argument");
    if ((flags & EMLOCO_GEMM_BIAS) && !bias) return pfail(-1, "emloco_gemm_f32: bias flag without bias");
    if (ksplit < 1) ksplit = 1;
    if (ksplit > 1 && !workspace) return pfail(-1, "emloco_gemm_f32: ksplit > 1 needs a workspace");
    if ((long)batch * ksplit > 65535) return pfail(-1, "emloco_gemm_f32: batch * ksplit exceeds the grid z limit");
    emloco::GemmArgs g{batch, m, n, k, alpha, A, lda, (long)stride_a, trans_a, B, ldb, (long)stride_b, trans_b,
                       C, ldc, (long)stride_c, bias, flags, ksplit, workspace, 0, 0, drop_p, drop_seed, nullptr, 1.0f, nullptr};
    // 16-byte global loads need the base, the leading dimension and the batch stride 16 B aligned
    g.vec_a = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0) && (stride_a % 4 == 0);
    g.vec_b = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0) && (stride_b % 4 == 0);
    g.a16 = (flags & EMLOCO_GEMM_A_BF16MEM) ? 1 : 0; g.b16 = (flags & EMLOCO_GEMM_B_BF16MEM) ? 1 : 0;
    g.c16 = (flags & EMLOCO_GEMM_C_BF16MEM) ? 1 : 0; g.m16 = 0;
    if (flags & EMLOCO_GEMM_B_SPLITIMG) {                     // B = the piece image of a weight (emloco_gemm_split_pack), A row-major fp32
        if (!(flags & EMLOCO_GEMM_SPLIT) || (flags & EMLOCO_GEMM_BF16) || batch != 1 || trans_a || !g.vec_a || n <= 32 || (uintptr_t)B % 16)
            return pfail(-1, "emloco_gemm_f32: a piece image serves the split mode with a row-major, 16-byte-aligned A, batch 1, n > 32");
        g.bimg = 1; g.vec_b = 1; g.tb = 0;
        g.flags &= ~EMLOCO_GEMM_B_SPLITIMG;
    }
    // 16-byte stores of whole output lines (the LDS-transposed epilogue): EMLOCO_GEMM_WIDE_STORES=0 keeps the per-register stores (A/B knob)
    static const bool wide = !(getenv("EMLOCO_GEMM_WIDE_STORES") && getenv("EMLOCO_GEMM_WIDE_STORES")[0] == '0');
    g.vec_c = (wide && (uintptr_t)C % 16 == 0 && ldc % (g.c16 ? 8 : 4) == 0 && stride_c % (g.c16 ? 8 : 4) == 0) ? 1 : 0;
    if (g.a16 || g.b16 || g.c16) {
        if (!(flags & EMLOCO_GEMM_BF16) || !g.vec_a || !g.vec_b || n <= 32)
            return pfail(-1, "emloco_gemm_f32: bf16 memory operands need EMLOCO_GEMM_BF16, 16-byte-aligned operands and n > 32");
        if (g.c16 && ksplit > 1) return pfail(-1, "emloco_gemm_f32: a bf16 output cannot be split along k (the partial sums are fp32)");
        if ((g.b16) && !(trans_a && trans_b)) return pfail(-1, "emloco_gemm_f32: a bf16 B operand is served for the weight-gradient layout only (trans_a and trans_b)");
    }
    hipStream_t st = (hipStream_t)stream;
    const int slot = g_head;
    if (g_timing) PHIPCHK(hipEventRecord(g_e0[slot], st));
    // stage depth: 32 for long reductions (covers the prefetch latency), 16 for short ones (less LDS, more workgroups / CU)
    static const int force_bk = getenv("EMLOCO_GEMM_BK") ? atoi(getenv("EMLOCO_GEMM_BK")) : 0;
    const bool split_ok = (flags & EMLOCO_GEMM_SPLIT) && !(flags & EMLOCO_GEMM_BF16) && g.vec_a && g.vec_b && n > 32;
    g.small = split_ok && (g_force_small >= 0 ? g_force_small == 1 : emloco::gemm_use_small_tile(batch, m, n, ksplit)) ? 1 : 0;
    const unsigned bn = n <= 32 ? 32 : (g.small ? 64 : 128), bm = g.small ? 64 : 128;
    dim3 grid((unsigned)((n + bn - 1) / bn), (unsigned)((m + bm - 1) / bm), (unsigned)(batch * ksplit));
    // measured (tools/exp/probe_jta_gemm.py, tools/probe_gemm.py): the 32-deep stage only pays while the launch is at most ~2
    // workgroups per CU (4096 x 1024 x 2048: 87 vs 85 TFLOP/s); with many waves of workgroups the 16-deep stage's third resident
    // workgroup per CU wins (927744 x 128 x 1024: 111 vs 102), and so it does for the k-major (transposed) operands of the
    // weight gradients (1024 x 128 x 927744 split 64: 114 vs 107)
    const long n_wg = (long)grid.x * grid.y * grid.z;
    const bool long_k = (k + ksplit - 1) / ksplit > 256;
    bool deep = force_bk ? force_bk == 32 : (long_k && (n <= 32 || (n_wg <= 1024 && !(trans_a && trans_b))));
    if ((flags & EMLOCO_GEMM_SPLIT) && !(flags & EMLOCO_GEMM_BF16) && g.vec_a && g.vec_b && n > 32) deep = false;   // split stages are 16 deep
    else g.flags &= ~EMLOCO_GEMM_SPLIT;
    const emloco::GemmKernel kern = emloco::gemm_pick(g, deep);
    if (!kern) return pfail(-1, "emloco_gemm_f32: this combination of bf16 memory operands and layouts is not served");
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, g);
    PHIPCHK(hipGetLastError());
    if (ksplit > 1) {
        const long total = (long)batch * m * n;
        hipLaunchKernelGGL(emloco::gemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g);
        PHIPCHK(hipGetLastError());
    }
    if (g_timing) {
        PHIPCHK(hipEventRecord(g_e1[slot], st));
        g_fl[slot] = 2.0 * batch * (double)m * n * k;
        // algorithmic bytes: each operand once, the output once (split-K partial slabs and re-reads through the caches are not algorithmic)
        g_by[slot] = (double)batch * ((double)m * k * (g.a16 ? 2 : 4) + (double)n * k * (g.b16 ? 2 : 4) + (double)m * n * (g.c16 ? 2 : 4));
        g_head = (slot + 1) % kRing;
        if (g_count < kRing) ++g_count;
    }
    return 0;
}

int64_t emloco_gemm_split_image_words(int n, int k) {
    if (n < 1 || k < 1) return 0;
    return (int64_t)((n + 127) / 128) * ((k + 15) / 16) * SPLIT_IMG_SLOTS * 4;
}

int emloco_gemm_split_pack(const float *W, int n, int k, int ld, int trans, uint32_t *image, void *stream) {
    if (!W || !image || n < 1 || k < 1 || ld < (trans ? n : k) || (uintptr_t)image % 16)
        return pfail(-1, "emloco_gemm_split_pack: bad argument (image = emloco_gemm_split_image_words(n, k) 32-bit words, 16-byte aligned)");
    hipLaunchKernelGGL(emloco::gemm_split_pack_kernel, dim3((unsigned)((k + 15) / 16), (unsigned)((n + 127) / 128)), dim3(256), 0, (hipStream_t)stream,
                       W, n, k, ld, trans, (emloco::gemm_u32x4 *)image);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_softmax_fwd(int n_seq, int rows_per_seq, int cols, float scale, const float *S, const float *key_pad, float *P, void *stream) {
    if (n_seq < 1 || rows_per_seq < 1 || cols < 1 || !S || !P) return pfail(-1, "emloco_softmax_fwd: bad argument");
    const long rows = (long)n_seq * rows_per_seq;
    hipLaunchKernelGGL(emloco::softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (int)rows, rows_per_seq, cols, scale, S, key_pad, P);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_softmax_bwd(int rows, int cols, float scale, const float *P, const float *dP, float *dS, void *stream) {
    if (rows < 1 || cols < 1 || !P || !dP || !dS) return pfail(-1, "emloco_softmax_bwd: bad argument");
    hipLaunchKernelGGL(emloco::softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rows, cols, scale, P, dP, dS);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_layernorm_fwd(int rows, int d, float eps, const float *x, const float *res, const float *gamma, const float *beta,
                         float *y, float *mean, float *rstd, void *stream) {
    return emloco_layernorm_fwd_save(rows, d, eps, x, res, gamma, beta, y, mean, rstd, nullptr, stream);
}

int emloco_layernorm_fwd_save(int rows, int d, float eps, const float *x, const float *res, const float *gamma, const float *beta,
                              float *y, float *mean, float *rstd, float *xr, void *stream) {
    if (rows < 1 || d < 1 || d > 1024 || !x || !gamma || !beta || !y || !mean || !rstd) return pfail(-1, "emloco_layernorm_fwd: bad argument");
    const bool al16 = (((uintptr_t)x | (uintptr_t)res | (uintptr_t)y | (uintptr_t)xr | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
    if (d == 128 && al16)            // the predictor's width: two rows per wave, 16-byte accesses (predictor_kernels.hip)
        hipLaunchKernelGGL(emloco::layernorm_fwd128_kernel, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                           rows, eps, x, res, gamma, beta, y, mean, rstd, xr);
    else
        hipLaunchKernelGGL(emloco::layernorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           rows, d, eps, x, res, gamma, beta, y, mean, rstd, xr);
    PHIPCHK(hipGetLastError());
    return 0;
}

namespace {
struct FoldLaunch {
    hipStream_t st;
    void operator()(unsigned gx, unsigned gy, int n, int w, const float *in, float *o0, float *o1, int split) const {
        hipLaunchKernelGGL(emloco::rows_fold_kernel, dim3(gx, gy), dim3(256), 0, st, n, w, in, o0, o1, split);
    }
};
}  // namespace

int64_t emloco_layernorm_bwd_workspace(int rows, int d) {
    return emloco::fold_workspace((rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK, 2L * d);
}

int emloco_layernorm_bwd(int rows, int d, const float *xr, const float *gamma, const float *mean, const float *rstd,
                         const float *dy, float *dxr, float *dgamma, float *dbeta, float *workspace, void *stream) {
    return emloco_layernorm_bwd2(rows, d, xr, gamma, mean, rstd, dy, nullptr, dxr, dgamma, dbeta, workspace, stream);
}

int emloco_layernorm_bwd2(int rows, int d, const float *xr, const float *gamma, const float *mean, const float *rstd,
                          const float *dy, const float *dy2, float *dxr, float *dgamma, float *dbeta, float *workspace, void *stream) {
    if (rows < 1 || d < 1 || d > 1024 || !xr || !gamma || !mean || !rstd || !dy || !dxr || !dgamma || !dbeta || !workspace)
        return pfail(-1, "emloco_layernorm_bwd: bad argument (workspace = emloco_layernorm_bwd_workspace(rows, d) floats)");
    const int nblocks = (rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK;
    const bool al16 = (((uintptr_t)xr | (uintptr_t)dy | (uintptr_t)dy2 | (uintptr_t)dxr | (uintptr_t)gamma) & 15) == 0;
    if (d == 128 && al16)
        hipLaunchKernelGGL(emloco::layernorm_bwd128_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                           rows, xr, gamma, mean, rstd, dy, dy2, dxr, workspace);
    else
        hipLaunchKernelGGL(emloco::layernorm_bwd_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                           rows, d, xr, gamma, mean, rstd, dy, dy2, dxr, workspace);
    PHIPCHK(hipGetLastError());
    emloco::fold_rows(FoldLaunch{(hipStream_t)stream}, nblocks, 2 * d, workspace, dgamma, dbeta, d);
    PHIPCHK(hipGetLastError());
    return 0;
}

int64_t emloco_colsum_workspace(int m, int n) { const int cs = emloco::cs_rows_for(m); return emloco::fold_workspace((m + cs - 1) / cs, n); }

int emloco_colsum(int m, int n, const float *X, float *out, float *workspace, void *stream) {
    return emloco_colsum_ex(m, n, X, out, workspace, 0, stream);
}

int emloco_colsum_ex(int m, int n, const float *X, float *out, float *workspace, int flags, void *stream) {
    if (m < 1 || n < 1 || !X || !out || !workspace)
        return pfail(-1, "emloco_colsum: bad argument (workspace = emloco_colsum_workspace(m, n) floats)");
    const int cs = emloco::cs_rows_for(m), nparts = (m + cs - 1) / cs;
    const bool x16 = (flags & EMLOCO_GEMM_A_BF16MEM) != 0;
    if (n % 4 == 0 && (((uintptr_t)X | (uintptr_t)workspace) & 15) == 0) {
        if (x16) hipLaunchKernelGGL(emloco::colsum4_partial_kernel<1>, dim3((unsigned)((n + 255) / 256), (unsigned)nparts), dim3(256), 0, (hipStream_t)stream, m, n, X, workspace, cs);
        else hipLaunchKernelGGL(emloco::colsum4_partial_kernel<0>, dim3((unsigned)((n + 255) / 256), (unsigned)nparts), dim3(256), 0, (hipStream_t)stream, m, n, X, workspace, cs);
    } else
        hipLaunchKernelGGL(emloco::colsum_partial_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nparts), dim3(256), 0, (hipStream_t)stream, m, n, X, workspace,
                           (flags & EMLOCO_GEMM_A_BF16MEM) ? 1 : 0, cs);
    PHIPCHK(hipGetLastError());
    emloco::fold_rows(FoldLaunch{(hipStream_t)stream}, nparts, n, workspace, out, out, n);
    PHIPCHK(hipGetLastError());
    return 0;
}

int64_t emloco_gemm_relu_bwd_workspace(int m, int n) { return emloco::fold_workspace(2L * ((m + 127) / 128), n); }

int emloco_gemm_relu_bwd(int m, int n, int k, const float *A, int lda, const float *B, int ldb, int trans_b, float *C,
                         const float *y, float scale, float *colsum, float *workspace, int flags, void *stream) {
    if (m < 1 || n < 33 || k < 1 || !A || !B || !C || !y || !colsum || !workspace)
        return pfail(-1, "emloco_gemm_relu_bwd: bad argument (n > 32; workspace = emloco_gemm_relu_bwd_workspace(m, n) floats)");
    emloco::GemmArgs g{1, m, n, k, 1.0f, A, lda, 0, 0, B, ldb, 0, trans_b, C, n, 0, nullptr,
                       32 | (flags & EMLOCO_GEMM_BF16) | ((flags & EMLOCO_GEMM_BF16) ? 0 : (flags & (EMLOCO_GEMM_SPLIT | EMLOCO_GEMM_SPLIT2))), 1, nullptr,
                       0, 0, 0.0f, 0u, y, scale, workspace};
    g.vec_a = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
    g.vec_b = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
    if (flags & EMLOCO_GEMM_B_SPLITIMG) {                     // B = the weight's piece image (emloco_gemm_split_pack)
        if (!(g.flags & EMLOCO_GEMM_SPLIT) || (uintptr_t)B % 16) return pfail(-1, "emloco_gemm_relu_bwd: a piece image needs EMLOCO_GEMM_SPLIT (without EMLOCO_GEMM_BF16) and 16-byte alignment");
        g.bimg = 1; g.vec_b = 1; g.tb = 0;
    }
    g.a16 = g.b16 = 0;                                        // the incoming gradient and the weight are fp32
    g.c16 = (flags & EMLOCO_GEMM_C_BF16MEM) ? 1 : 0; g.m16 = (flags & EMLOCO_GEMM_MASK_BF16MEM) ? 1 : 0;
    if ((g.c16 || g.m16) && (!(flags & EMLOCO_GEMM_BF16) || g.c16 != g.m16))
        return pfail(-1, "emloco_gemm_relu_bwd: a bf16 hidden layer needs EMLOCO_GEMM_BF16 and comes with a bf16 gradient (C and MASK flags together)");
    if (!g.vec_a || !g.vec_b) return pfail(-1, "emloco_gemm_relu_bwd: A and B must be 16-byte aligned with leading dimensions that are multiples of 4");
    {   // wide accesses of the fp32 epilogue (mask in, gradient out as whole lines); EMLOCO_GEMM_WIDE_STORES=0: per-register accesses
        static const bool wide = !(getenv("EMLOCO_GEMM_WIDE_STORES") && getenv("EMLOCO_GEMM_WIDE_STORES")[0] == '0');
        g.vec_c = (wide && !g.c16 && (uintptr_t)C % 16 == 0 && (uintptr_t)y % 16 == 0 && n % 4 == 0) ? 1 : 0;
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((n + 127) / 128), (unsigned)((m + 127) / 128), 1);
    const bool deep = k > 256 && (long)grid.x * grid.y <= 1024;
    hipLaunchKernelGGL(emloco::gemm_pick(g, deep), grid, dim3(256), 0, st, g);
    PHIPCHK(hipGetLastError());
    emloco::fold_rows(FoldLaunch{st}, 2 * (int)grid.y, n, workspace, colsum, colsum, n);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_act_bwd_colsum(int m, int n, const float *dy, const float *y, int relu, float drop_p, uint32_t drop_seed, float *dz,
                          float *colsum, float *workspace, void *stream) {
    if (m < 1 || n < 1 || !dy || !dz || !colsum || !workspace || (relu && !y) || !(drop_p >= 0.0f && drop_p < 1.0f))
        return pfail(-1, "emloco_act_bwd_colsum: bad argument (workspace = emloco_colsum_workspace(m, n) floats)");
    const int cs = emloco::cs_rows_for(m), nparts = (m + cs - 1) / cs;
    hipLaunchKernelGGL(emloco::act_bwd_colsum_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)nparts), dim3(256), 0, (hipStream_t)stream,
                       m, n, dy, y, relu, drop_p, drop_seed, dz, workspace, cs);
    PHIPCHK(hipGetLastError());
    emloco::fold_rows(FoldLaunch{(hipStream_t)stream}, nparts, n, workspace, colsum, colsum, n);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_gather_flat(int n, const float *const *src, const int64_t *numel, const int64_t *dst_offset, float *flat, void *stream) {
    if (n < 0 || (n > 0 && (!src || !numel || !dst_offset || !flat))) return pfail(-1, "emloco_gather_flat: bad argument");
    for (int i0 = 0; i0 < n; i0 += GATHER_MAX) {
        emloco::GatherArgs a;
        const int m = n - i0 < GATHER_MAX ? n - i0 : GATHER_MAX;
        long big = 0;
        for (int i = 0; i < m; ++i) {
            if (!src[i0 + i] || numel[i0 + i] < 0 || dst_offset[i0 + i] < 0) return pfail(-1, "emloco_gather_flat: null source or negative size / offset");
            a.src[i] = src[i0 + i]; a.numel[i] = (long)numel[i0 + i]; a.off[i] = (long)dst_offset[i0 + i];
            if (a.numel[i] > big) big = a.numel[i];
        }
        a.flat = flat;
        long gx = (big / 4 + 255) / 256;                       // workgroups along the largest tensor: one 16-byte copy per thread, at most 64
        gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
        hipLaunchKernelGGL(emloco::gather_flat_kernel, dim3((unsigned)gx, (unsigned)m), dim3(256), 0, (hipStream_t)stream, a);
        PHIPCHK(hipGetLastError());
    }
    return 0;
}

int emloco_disc_reward(int n, const float *logits, float scale, float *reward, void *stream) {
    if (n < 1 || !logits || !reward) return pfail(-1, "emloco_disc_reward: bad argument");
    hipLaunchKernelGGL(emloco::disc_reward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, logits, scale, reward);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_obs_normalize(int rows, int cols, const float *x, int ldx, const float *mean, const float *var, float eps,
                         float clip, int split, float *out0, int ld0, float *out1, int ld1, void *stream) {
    if (rows < 1 || cols < 1 || !x || !mean || !var || !out0 || split < 0 || split > cols || (split < cols && !out1) ||
        ldx < cols || ld0 < split || (split < cols && ld1 < cols - split))
        return pfail(-1, "emloco_obs_normalize: bad argument");
    hipLaunchKernelGGL(emloco::obs_normalize_kernel, dim3((unsigned)((cols + 255) / 256), (unsigned)rows), dim3(256), 0,
                       (hipStream_t)stream, rows, cols, x, ldx, mean, var, eps, clip, split, out0, ld0, out1 ? out1 : out0, ld1);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_rms_update(int rows, int cols, const float *x, int ldx, double *mean, double *var, const double *count_in, double *count_out,
                      int first_col, void *stream) {
    if (rows < 1 || cols < 1 || !x || !mean || !var || !count_in || !count_out || count_in == count_out || ldx < cols || first_col < 0 || first_col > cols)
        return pfail(-1, "emloco_rms_update: bad argument");
    hipLaunchKernelGGL(emloco::rms_update_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(256), 0, (hipStream_t)stream, rows, cols, x, ldx,
                       mean, var, count_in, count_out, first_col);
    PHIPCHK(hipGetLastError());
    return 0;
}

int64_t emloco_rms_update_workspace(int rows, int cols) {
    const int64_t chunks = (rows + RMS_CHUNK - 1) / RMS_CHUNK;
    return chunks * 3 * (int64_t)cols * (int64_t)sizeof(double);
}

int emloco_rms_update_chunked(int rows, int cols, const float *x, int ldx, double *mean, double *var, const double *count_in, double *count_out,
                              int first_col, void *workspace, void *stream) {
    if (rows < 1 || cols < 1 || !x || !mean || !var || !count_in || !count_out || count_in == count_out || ldx < cols || first_col < 0 ||
        first_col > cols || !workspace)
        return pfail(-1, "emloco_rms_update_chunked: bad argument");
    const int chunks = (rows + RMS_CHUNK - 1) / RMS_CHUNK;
    hipLaunchKernelGGL(emloco::rms_partial_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, rows,
                       cols, x, ldx, (double *)workspace);
    PHIPCHK(hipGetLastError());
    hipLaunchKernelGGL(emloco::rms_merge_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, chunks, cols,
                       (const double *)workspace, mean, var, count_in, count_out, first_col);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_locoval_fwd_rows(int B, const float *traj, int traj_stride, const float *pose, const float *vel, const float *w1,
                            const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, float *value,
                            float *x100, float *h1, float *h2, float *angle, const float *row_weight, void *stream) {
    if (B < 1 || traj_stride < 2 || !traj || !pose || !vel || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !value || !x100 || !h1 || !h2)
        return pfail(-1, "emloco_locoval_fwd: bad argument");
    hipLaunchKernelGGL(emloco::locoval_fwd_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, B, traj, traj_stride, pose, vel,
                       w1, b1, w2, b2, w3, b3, value, x100, h1, h2, angle, row_weight);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_locoval_fwd(int B, const float *traj, int traj_stride, const float *pose, const float *vel, const float *w1,
                       const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, float *value,
                       float *x100, float *h1, float *h2, float *angle, void *stream) {
    return emloco_locoval_fwd_rows(B, traj, traj_stride, pose, vel, w1, b1, w2, b2, w3, b3, value, x100, h1, h2, angle, nullptr, stream);
}

int64_t emloco_locoval_bwd_workspace(int B) { return (int64_t)B * LV_NPARAM * (int64_t)sizeof(float); }

int emloco_locoval_bwd(int B, const float *traj, int traj_stride, const float *pose, const float *vel, const float *w1,
                       const float *w2, const float *w3, const float *value, const float *x100, const float *h1, const float *h2,
                       const float *angle, const float *dvalue, float *dparams, float *dtraj, float *workspace, void *stream) {
    if (B < 1 || traj_stride < 2 || !traj || !pose || !vel || !w1 || !w2 || !w3 || !value || !x100 || !h1 || !h2 || !angle || !dvalue ||
        !dparams || !dtraj || !workspace)
        return pfail(-1, "emloco_locoval_bwd: bad argument");
    hipLaunchKernelGGL(emloco::locoval_bwd_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, B, traj, traj_stride, pose, vel,
                       w1, w2, w3, value, x100, h1, h2, angle, dvalue, workspace, dtraj, (const int32_t *)nullptr);
    PHIPCHK(hipGetLastError());
    hipLaunchKernelGGL(emloco::locoval_reduce_kernel, dim3((unsigned)((LV_NPARAM + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, workspace,
                       dparams, (const float *)nullptr);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_locoval_bwd_rows(int B, const float *traj, int traj_stride, const float *pose, const float *vel, const float *w1,
                            const float *w2, const float *w3, const float *value, const float *x100, const float *h1, const float *h2,
                            const float *angle, const float *dvalue, const int32_t *slot, const float *count, float *dparams, float *dtraj,
                            float *workspace, void *stream) {
    if (B < 1 || traj_stride < 2 || !traj || !pose || !vel || !w1 || !w2 || !w3 || !value || !x100 || !h1 || !h2 || !angle || !dvalue ||
        !slot || !count || !dparams || !dtraj || !workspace)
        return pfail(-1, "emloco_locoval_bwd_rows: bad argument");
    hipLaunchKernelGGL(emloco::locoval_bwd_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, B, traj, traj_stride, pose, vel,
                       w1, w2, w3, value, x100, h1, h2, angle, dvalue, workspace, dtraj, slot);
    PHIPCHK(hipGetLastError());
    hipLaunchKernelGGL(emloco::locoval_reduce_kernel, dim3((unsigned)((LV_NPARAM + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, workspace,
                       dparams, count);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_locoval_returns(const EmlocoLocoValStep *t, const float *rewards, const float *amp_rewards, const int64_t *dones,
                           const uint8_t *inverted, void *stream) {
    if (!t || !rewards || !dones || t->n_env < 1 || !t->current_rewards || !t->current_lengths || !t->current_combined_rewards ||
        !t->discount_coefs || !t->waypoint_traj || !t->init_pose || !t->init_vel || !t->traj13 || !t->pose || !t->vel || !t->target || !t->weight)
        return pfail(-1, "emloco_locoval_returns: bad argument");
    if ((t->staged_reward == nullptr) != (t->staged_done == nullptr))
        return pfail(-1, "emloco_locoval_returns: staged_reward and staged_done go together");
    hipLaunchKernelGGL(emloco::locoval_returns_kernel, dim3((unsigned)t->n_env), dim3(64), 0, (hipStream_t)stream, *t, rewards, amp_rewards,
                       dones, inverted);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_locoval_returns_finish(const EmlocoLocoValStep *t, const float *amp_rewards, void *stream) {
    if (!t || t->n_env < 1 || !t->current_rewards || !t->current_lengths || !t->current_combined_rewards || !t->discount_coefs ||
        !t->target || !t->weight)
        return pfail(-1, "emloco_locoval_returns_finish: bad argument");
    if (!t->staged_reward || !t->staged_done) return pfail(-1, "emloco_locoval_returns_finish: the step has no staging arrays (nothing was staged)");
    hipLaunchKernelGGL(emloco::locoval_returns_finish_kernel, dim3((unsigned)((t->n_env + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *t,
                       amp_rewards);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_locoval_fit_grad(int n, const float *value, const float *target, const float *weight, float *dvalue, float *tail2, int32_t *slot,
                            void *stream) {
    if (n < 1 || !value || !target || !weight || !dvalue || !tail2) return pfail(-1, "emloco_locoval_fit_grad: bad argument");
    hipLaunchKernelGGL(emloco::locoval_fit_grad_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, value, target, weight, dvalue, tail2, slot);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_adamw_gated(int n, float *params, const float *grads, float *exp_avg, float *exp_avg_sq, const float *steps_in, float *steps_out,
                       const float *tail2, float lr, float beta1, float beta2, float eps, float weight_decay, double *stats, void *stream) {
    if (n < 1 || !params || !grads || !exp_avg || !exp_avg_sq || !steps_in || !steps_out || steps_in == steps_out)
        return pfail(-1, "emloco_adamw_gated: bad argument");
    hipLaunchKernelGGL(emloco::adamw_gated_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, params, grads,
                       exp_avg, exp_avg_sq, steps_in, steps_out, tail2, lr, beta1, beta2, eps, weight_decay, stats);
    PHIPCHK(hipGetLastError());
    return 0;
}

int64_t emloco_adam_clip_flat_workspace(int64_t n) { return (n + ADAM_BLOCK - 1) / ADAM_BLOCK + 4; }

// workspace: [0] norm, [1] clip coefficient, [2] 1 - beta1^t, [3] sqrt(1 - beta2^t) (counted variant), [4..] block partials
static int adam_clip_flat(int64_t n, float *params, float *grads, float *exp_avg, float *exp_avg_sq, float lr, double beta1, double beta2, float eps,
                          float weight_decay, float bc1, float bc2_sqrt, float max_norm, float *workspace, float *step_count, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const float *coef = nullptr;
    if (max_norm > 0.0f) {
        const int nparts = (int)((n + ADAM_BLOCK - 1) / ADAM_BLOCK);
        hipLaunchKernelGGL(emloco::sumsq_partial_kernel, dim3((unsigned)nparts), dim3(256), 0, st, (long)n, grads, workspace + 4);
        hipLaunchKernelGGL(emloco::clip_coef_kernel, dim3(1), dim3(256), 0, st, nparts, workspace + 4, max_norm, workspace);
        coef = workspace;
    }
    if (step_count) hipLaunchKernelGGL(emloco::adam_step_count_kernel, dim3(1), dim3(64), 0, st, step_count, beta1, beta2, workspace + 2);
    hipLaunchKernelGGL(emloco::adam_flat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (long)n, params, grads, exp_avg, exp_avg_sq,
                       coef, lr, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, weight_decay, bc1, bc2_sqrt,
                       step_count ? (const float *)(workspace + 2) : (const float *)nullptr);
    PHIPCHK(hipGetLastError());
    return 0;
}

int emloco_adam_clip_flat(int64_t n, float *params, float *grads, float *exp_avg, float *exp_avg_sq, float lr, double beta1, double beta2, float eps,
                          float weight_decay, float bias_correction1, float bias_correction2_sqrt, float max_norm, float *workspace, void *stream) {
    if (n < 1 || !params || !grads || !exp_avg || !exp_avg_sq || !(bias_correction1 > 0.0f) || !(bias_correction2_sqrt > 0.0f)
        || (max_norm > 0.0f && !workspace))
        return pfail(-1, "emloco_adam_clip_flat: bad argument (workspace = emloco_adam_clip_flat_workspace(n) floats when max_norm > 0)");
    return adam_clip_flat(n, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt,
                          max_norm, workspace, nullptr, stream);
}

int emloco_adam_clip_flat_counted(int64_t n, float *params, float *grads, float *exp_avg, float *exp_avg_sq, float lr, double beta1, double beta2,
                                  float eps, float weight_decay, float max_norm, float *workspace, float *step_count, void *stream) {
    if (n < 1 || !params || !grads || !exp_avg || !exp_avg_sq || !workspace || !step_count)
        return pfail(-1, "emloco_adam_clip_flat_counted: bad argument (workspace = emloco_adam_clip_flat_workspace(n) floats, step_count = 1 device float)");
    return adam_clip_flat(n, params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, 1.0f, 1.0f, max_norm, workspace, step_count, stream);
}

int emloco_dropout_keep_mask(uint32_t seed, uint64_t first_index, int64_t n, float p, uint8_t *host_out) {
    if (n < 0 || !host_out || !(p >= 0.0f && p < 1.0f)) return pfail(-1, "emloco_dropout_keep_mask: bad argument");
    for (int64_t i = 0; i < n; ++i) host_out[i] = emloco::drop_keep(seed, first_index + (uint64_t)i, p) ? 1 : 0;
    return 0;
}

}  // extern "C"
