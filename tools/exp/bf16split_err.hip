// bf16split_err.hip -- how exact is an fp32 product rebuilt from bf16 pieces on v_mfma_f32_32x32x16_bf16?
//   a = a1 + a2 + a3 (bf16 each, 8 significant bits per piece), same for b;
//   x3: a1b1 + a1b2 + a2b1            (error ~2^-16 per product)
//   x6: + a2b2 + a1b3 + a3b1          (error ~2^-24: the fp32 class, if the matrix core's own accumulation is that good)
//   x9: all nine
// against a float64 reference, beside v_mfma_f32_32x32x2_f32 (an fmaf chain).  One wave, one 32x32 tile, K swept.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/bf16split_err.hip -o /tmp/bf16split_err && /tmp/bf16split_err
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x16 __attribute__((vector_size(64)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float vf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &p1, bf16x8 &p2, bf16x8 &p3, int trunc) {
    vf8 v, r1, r2;
    for (int i = 0; i < 8; ++i) v[i] = x[i];
    if (trunc) {
        for (int i = 0; i < 8; ++i) {
            const float h1 = __uint_as_float(__float_as_uint(v[i]) & 0xffff0000u);
            r1[i] = v[i] - h1;
            const float h2 = __uint_as_float(__float_as_uint(r1[i]) & 0xffff0000u);
            r2[i] = r1[i] - h2;
            p1[i] = (__bf16)h1; p2[i] = (__bf16)h2; p3[i] = (__bf16)r2[i];
        }
        return;
    }
    p1 = __builtin_convertvector(v, bf16x8);
    for (int i = 0; i < 8; ++i) r1[i] = v[i] - (float)p1[i];
    p2 = __builtin_convertvector(r1, bf16x8);
    for (int i = 0; i < 8; ++i) r2[i] = r1[i] - (float)p2[i];
    p3 = __builtin_convertvector(r2, bf16x8);
}

// mode 0: fp32 MFMA; 3/6/9: split, one accumulator; 16: x6 with the small terms in their own accumulator; 26: x6, truncating split
__global__ void tile_kernel(const float *A, const float *B, float *C, int K, int mode) {
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    f32x16 acc = {0}, acc2 = {0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + h], B[l31 * K + k + h], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            float xa[8], xb[8];
            for (int i = 0; i < 8; ++i) { xa[i] = A[l31 * K + k + 8 * h + i]; xb[i] = B[l31 * K + k + 8 * h + i]; }
            bf16x8 a1, a2, a3, b1, b2, b3;
            split3(xa, a1, a2, a3, mode == 26);
            split3(xb, b1, b2, b3, mode == 26);
#define MM(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0)
            if (mode == 16) {
                MM(a3, b1, acc2); MM(a1, b3, acc2); MM(a2, b2, acc2); MM(a2, b1, acc2); MM(a1, b2, acc2); MM(a1, b1, acc);
            } else {
                // small terms first
                if (mode >= 9 && mode != 26) { MM(a3, b3, acc); MM(a3, b2, acc); MM(a2, b3, acc); }
                if (mode >= 6) { MM(a3, b1, acc); MM(a1, b3, acc); MM(a2, b2, acc); }
                MM(a2, b1, acc); MM(a1, b2, acc); MM(a1, b1, acc);
            }
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        C[row * 32 + l31] = acc[r] + acc2[r];
    }
}

int main() {
    const int Ks[] = {128, 1024, 4096};
    const int modes[] = {0, 3, 6, 16, 26, 9};
    for (int dist = 0; dist < 3; ++dist)
        for (int K : Ks) {
            std::vector<float> A(32 * K), B(32 * K);
            srand(1234 + K + dist);
            for (auto *v : {&A, &B})
                for (auto &x : *v) {
                    const double u = (rand() + 1.0) / (RAND_MAX + 2.0), w = (rand() + 1.0) / (RAND_MAX + 2.0);
                    const double g = sqrt(-2 * log(u)) * cos(6.283185307179586 * w);
                    x = dist == 0 ? (float)g : dist == 1 ? (float)fabs(g) : (float)(g * exp(4 * (u - 0.5)));   // signed / all positive / wide range
                }
            std::vector<double> ref(1024), mag(1024);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0, m = 0;
                    for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[j * K + k]; m += fabs((double)A[i * K + k] * B[j * K + k]); }
                    ref[i * 32 + j] = s; mag[i * 32 + j] = m;
                }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            printf("dist %d K %5d:", dist, K);
            for (int mode : modes) {
                tile_kernel<<<1, 64>>>(dA, dB, dC, K, mode);
                std::vector<float> C(1024);
                hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
                double worst = 0, mean = 0, bias = 0;
                for (int i = 0; i < 1024; ++i) {
                    const double e = ((double)C[i] - ref[i]) / mag[i];
                    worst = fmax(worst, fabs(e)); mean += fabs(e) / 1024; bias += e / 1024;
                }
                printf("  m%-2d max %.2e mean %.2e bias %+.1e |", mode, worst, mean, bias);
            }
            printf("\n");
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    printf("(errors relative to sum |a b| of the element)\n");
    return 0;
}
