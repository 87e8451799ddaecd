// Is v_mfma_f32_32x32x2_f32 bit-equal to a chain of fmaf in k order?  (and how about k = 0 vs k = 1 order, zero C)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef float f32x16 __attribute__((vector_size(64)));
__global__ void k(const float *A, const float *B, const float *C0, float *D, int steps) {
    // A[32][2*steps], B[32][2*steps] row-major; D[32][32] = C0 + A B^T accumulated over `steps` MFMAs
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C0[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i];
    for (int s = 0; s < steps; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 2 * steps + 2 * s + h], B[i * 2 * steps + 2 * s + h], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
int main() {
    const int steps = 19, K = 2 * steps;
    float *hA = (float *)malloc(32 * K * 4), *hB = (float *)malloc(32 * K * 4), *hC = (float *)malloc(4096), *hD = (float *)malloc(4096);
    srand(1);
    for (int t = 0; t < 32 * K; ++t) { hA[t] = (rand() / (float)RAND_MAX - 0.5f) * expf((rand() % 16) - 8.0f); hB[t] = (rand() / (float)RAND_MAX - 0.5f) * expf((rand() % 16) - 8.0f); }
    for (int t = 0; t < 1024; ++t) hC[t] = (rand() / (float)RAND_MAX - 0.5f);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, 32 * K * 4); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dD, steps);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad_fma = 0, bad_fma_rev = 0, bad_muladd = 0, bad_pair = 0;
    for (int r = 0; r < 32; ++r)
        for (int c = 0; c < 32; ++c) {
            float f1 = hC[r * 32 + c], f2 = f1, f3 = f1, f4 = f1;
            for (int kk = 0; kk < K; kk += 2) {
                float a0 = hA[r * K + kk], a1 = hA[r * K + kk + 1], b0 = hB[c * K + kk], b1 = hB[c * K + kk + 1];
                f1 = fmaf(a1, b1, fmaf(a0, b0, f1));                 // k ascending fma chain
                f2 = fmaf(a0, b0, fmaf(a1, b1, f2));                 // k descending within the pair
                f3 = (f3 + a0 * b0) + a1 * b1;                       // separate multiply and add
                f4 = f4 + (float)((double)a0 * b0 + (double)a1 * b1);   // exact pair dot, one rounding, then add
            }
            float d = hD[r * 32 + c];
            bad_fma += memcmp(&d, &f1, 4) != 0; bad_fma_rev += memcmp(&d, &f2, 4) != 0; bad_muladd += memcmp(&d, &f3, 4) != 0; bad_pair += memcmp(&d, &f4, 4) != 0;
        }
    printf("mismatches of 1024: fma chain k-ascending %d | pair reversed %d | mul+add %d | exact pair dot then add %d\n", bad_fma, bad_fma_rev, bad_muladd, bad_pair);
    return 0;
}
