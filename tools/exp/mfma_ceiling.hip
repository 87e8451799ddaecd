// Micro-experiments: what bounds a 128x128x16 fp32 MFMA stage on gfx950?  hipcc --offload-arch=gfx950 -O3 mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((vector_size(64)));
struct __attribute__((aligned(16))) f32x4 { float x, y, z, w; };
#define LDK 20
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int stages, const float *A = nullptr, const float *B = nullptr, int ld = 0) {
    __shared__ __attribute__((aligned(16))) float As[2][128 * LDK], Bs[2][128 * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 128 * LDK; i += 256) { (&As[0][0])[i] = 1.0f + i * 1e-6f; (&Bs[0][0])[i] = 0.5f; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    f32x4 fa[2][2], fb[2][2];
    const int half8 = 8 * (lane >> 5), l31 = lane & 31;
    for (int i = 0; i < 2; ++i) for (int f = 0; f < 2; ++f) { fa[i][f] = f32x4{1, 2, 3, 4}; fb[i][f] = f32x4{1, 1, 1, 1}; }
    int buf = 0;
    f32x4 ra[2], rb[2];
    const int m0 = (blockIdx.x / 32) * 128, n0 = (blockIdx.x % 32) * 128;
    for (int s = 0; s < stages; ++s) {
        if (MODE >= 3) {
            for (int e = 0; e < 2; ++e) {
                const int idx = tid + 256 * e, r = idx >> 2, q = idx & 3;
                ra[e] = *(const f32x4 *)(A + (long)(m0 + r) * ld + ((s * 16) & (ld - 1)) + 4 * q);
                rb[e] = *(const f32x4 *)(B + (long)(n0 + r) * ld + ((s * 16) & (ld - 1)) + 4 * q);
            }
        }
        if (MODE >= 1) {
            for (int i = 0; i < 2; ++i) {
                const float *src = &As[buf][((wm * 2 + i) * 32 + l31) * LDK + half8];
                fa[i][0] = *(const f32x4 *)src; fa[i][1] = *(const f32x4 *)(src + 4);
                const float *sb = &Bs[buf][((wn * 2 + i) * 32 + l31) * LDK + half8];
                fb[i][0] = *(const f32x4 *)sb; fb[i][1] = *(const f32x4 *)(sb + 4);
            }
        }
#define STEP(H, C) for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][H].C, fb[j][H].C, acc[i][j], 0, 0, 0);
        STEP(0, x) STEP(0, y) STEP(0, z) STEP(0, w) STEP(1, x) STEP(1, y) STEP(1, z) STEP(1, w)
        if (MODE >= 3) {
            for (int e = 0; e < 2; ++e) {
                const int idx = tid + 256 * e, r = idx >> 2, q = idx & 3;
                *(f32x4 *)&As[buf ^ 1][r * LDK + 4 * q] = ra[e];
                *(f32x4 *)&Bs[buf ^ 1][r * LDK + 4 * q] = rb[e];
            }
        }
        if (MODE >= 2) { __syncthreads(); buf ^= 1; }
    }
    float t = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = t;
}
template <int MODE> void run(const char *name, int wgs, int stages) {
    float *out; hipMalloc(&out, wgs * 256 * 4);
    static float *A = nullptr, *B = nullptr;
    if (!A) { hipMalloc(&A, 4096L * 4096 * 4); hipMalloc(&B, 4096L * 4096 * 4); hipMemset(A, 0, 4096L * 4096 * 4); hipMemset(B, 0, 4096L * 4096 * 4); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<wgs, 256>>>(out, stages, A, B, 4096); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) k<MODE><<<wgs, 256>>>(out, stages, A, B, 4096); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double fl = (double)wgs * 4 * stages * 32 * 4096.0;
    printf("%-28s wgs %5d: %.3f ms  %.1f TFLOP/s\n", name, wgs, ms, fl / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int wgs : {1024}) {
        run<0>("mfma only", wgs, 256);
        run<2>("+ frags + barrier per stage", wgs, 256);
        run<3>("+ global fetch + stash", wgs, 256);
    }
    return 0;
}
