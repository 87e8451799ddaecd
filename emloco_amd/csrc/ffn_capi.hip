// ffn_capi.hip -- C ABI of the chained feed-forward kernels (include/emloco_predictor.h), a translation unit of its own so that its
// compiler switches can be chosen apart from the GEMM unit's (emloco_amd/build.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdlib.h>
#include "ffn_kernels.hip"
#include "../../include/emloco_predictor.h"

namespace {
int ffail(int code, const char *what, hipError_t e = hipSuccess) {
    if (e != hipSuccess) fprintf(stderr, "[emloco] %s: %s\n", what, hipGetErrorString(e));
    else fprintf(stderr, "[emloco] %s\n", what);
    return code;
}
bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }
}  // namespace

extern "C" {

int emloco_ffn_fwd(int M, int F, const float *x, const uint16_t *w1_bf16, const uint16_t *w2_bf16, const float *b1, const float *b2,
                   uint16_t *hidden, uint32_t *mask, float *out, float drop_p, uint32_t seed_hidden, uint32_t seed_out, void *stream) {
    if (M < 1 || F < FFN_CH || F % FFN_CH || F > 2048 || !x || !w1_bf16 || !w2_bf16 || !b1 || !b2 || !hidden || !mask || !out || !(drop_p >= 0.0f && drop_p < 1.0f))
        return ffail(-1, "emloco_ffn_fwd: bad argument (hidden width must be a multiple of 64, at most 2048; 0 <= drop_p < 1)");
    if (!aligned16(x) || !aligned16(w1_bf16) || !aligned16(w2_bf16) || !aligned16(b1) || !aligned16(b2) || !aligned16(hidden) || !aligned16(mask) || !aligned16(out))
        return ffail(-1, "emloco_ffn_fwd: operands must be 16-byte aligned");
    emloco::FfnArgs a{M, F, x, w1_bf16, w2_bf16, b1, b2, hidden, nullptr, mask, out, drop_p, 1.0f / (1.0f - drop_p), seed_hidden,
                      (unsigned)(drop_p * 65536.0f), seed_out, nullptr};
    const dim3 grid((unsigned)((M + FFN_ROWS - 1) / FFN_ROWS));
    if (drop_p > 0.0f) hipLaunchKernelGGL((emloco::ffn_chain_kernel<0, 1>), grid, dim3(FFN_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((emloco::ffn_chain_kernel<0, 0>), grid, dim3(FFN_THREADS), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffail(-2, "emloco_ffn_fwd launch", e);
}

int emloco_ffn_fwd_norm(int M, int F, const float *x, const uint16_t *w1_bf16, const uint16_t *w2_bf16, const float *b1, const float *b2,
                        uint16_t *hidden, uint32_t *mask, const float *res, const float *gamma, const float *beta, float eps,
                        float *y, float *xr, float *mean, float *rstd, float drop_p, uint32_t seed_hidden, uint32_t seed_out, void *stream) {
    if (M < 1 || F < FFN_CH || F % FFN_CH || F > 2048 || !x || !w1_bf16 || !w2_bf16 || !b1 || !b2 || !hidden || !mask || !res || !gamma || !beta ||
        !y || !xr || !mean || !rstd || !(drop_p >= 0.0f && drop_p < 1.0f) || !(eps >= 0.0f))
        return ffail(-1, "emloco_ffn_fwd_norm: bad argument (hidden width must be a multiple of 64, at most 2048; 0 <= drop_p < 1)");
    if (!aligned16(x) || !aligned16(w1_bf16) || !aligned16(w2_bf16) || !aligned16(b1) || !aligned16(b2) || !aligned16(hidden) || !aligned16(mask) ||
        !aligned16(res) || !aligned16(gamma) || !aligned16(beta) || !aligned16(y) || !aligned16(xr))
        return ffail(-1, "emloco_ffn_fwd_norm: operands must be 16-byte aligned");
    emloco::FfnArgs a{M, F, x, w1_bf16, w2_bf16, b1, b2, hidden, nullptr, mask, y, drop_p, 1.0f / (1.0f - drop_p), seed_hidden,
                      (unsigned)(drop_p * 65536.0f), seed_out, nullptr, res, gamma, beta, xr, mean, rstd, eps};
    const dim3 grid((unsigned)((M + FFN_ROWS - 1) / FFN_ROWS));
    if (drop_p > 0.0f) hipLaunchKernelGGL((emloco::ffn_chain_kernel<0, 1>), grid, dim3(FFN_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((emloco::ffn_chain_kernel<0, 0>), grid, dim3(FFN_THREADS), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffail(-2, "emloco_ffn_fwd_norm launch", e);
}

int emloco_ffn_bwd_input(int M, int F, const float *dz2, const uint16_t *w2t_bf16, const uint16_t *w1t_bf16, const uint32_t *mask,
                         uint16_t *dz1, float *dx, float drop_p, void *stream) {
    return emloco_ffn_bwd_input_colsum(M, F, dz2, w2t_bf16, w1t_bf16, mask, dz1, dx, drop_p, nullptr, stream);
}

int64_t emloco_ffn_bwd_colsum_rows(int M) { return ((int64_t)M + FFN_ROWS - 1) / FFN_ROWS * (FFN_THREADS / 64); }

int emloco_ffn_bwd_input_colsum(int M, int F, const float *dz2, const uint16_t *w2t_bf16, const uint16_t *w1t_bf16, const uint32_t *mask,
                                uint16_t *dz1, float *dx, float drop_p, float *colpart, void *stream) {
    if (M < 1 || F < FFN_CH || F % FFN_CH || !dz2 || !w2t_bf16 || !w1t_bf16 || !mask || !dz1 || !dx || !(drop_p >= 0.0f && drop_p < 1.0f))
        return ffail(-1, "emloco_ffn_bwd_input: bad argument (hidden width must be a multiple of 64, 0 <= drop_p < 1)");
    if (!aligned16(dz2) || !aligned16(w2t_bf16) || !aligned16(w1t_bf16) || !aligned16(mask) || !aligned16(dz1) || !aligned16(dx))
        return ffail(-1, "emloco_ffn_bwd_input: operands must be 16-byte aligned");
    emloco::FfnArgs a{M, F, dz2, w2t_bf16, w1t_bf16, nullptr, nullptr, nullptr, dz1, const_cast<uint32_t *>(mask), dx, drop_p, 1.0f / (1.0f - drop_p), 0u, 0u, 0u, colpart};
    if (colpart && !aligned16(colpart)) return ffail(-1, "emloco_ffn_bwd_input_colsum: colpart must be 16-byte aligned");
    // (waves wholly past the last row write nothing: their rows of colpart must read as zero)
    if (colpart && M % FFN_ROWS) {
        const int64_t rows = emloco_ffn_bwd_colsum_rows(M), live = ((int64_t)M + 31) / 32;
        if (rows > live && hipMemsetAsync(colpart + live * F, 0, (size_t)(rows - live) * F * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return ffail(-2, "emloco_ffn_bwd_input_colsum: memset");
    }
    const dim3 grid((unsigned)((M + FFN_ROWS - 1) / FFN_ROWS));
    hipLaunchKernelGGL((emloco::ffn_chain_kernel<1, 0>), grid, dim3(FFN_THREADS), 0, (hipStream_t)stream, a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ffail(-2, "emloco_ffn_bwd_input launch", e);
}

int emloco_ffn_keep_mask(uint32_t seed_hidden, int64_t first_row, int64_t rows, int F, float p, uint8_t *host_out) {
    if (rows < 0 || F < 1 || !host_out || !(p >= 0.0f && p < 1.0f)) return ffail(-1, "emloco_ffn_keep_mask: bad argument");
    const unsigned thr = (unsigned)(p * 65536.0f);
    for (int64_t r = 0; r < rows; ++r)
        for (int f = 0; f < F; ++f) host_out[r * F + f] = emloco::ffn_keep16(seed_hidden, (unsigned)(first_row + r), (unsigned)f, thr) ? 1 : 0;
    return 0;
}

}  // extern "C"
