// Latency probe for a single wave on gfx950: cycles per dependent step of a few instruction patterns (s_memtime).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/exp/lat_probe.hip -o /tmp/lat_probe && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
__device__ __forceinline__ float lane_bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__global__ void probe(float *out, long long *t, int nrt) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x;
    float w = out[lane], a = 1.0001f + lane * 1e-6f, lam = 0.5f;
    for (int i = lane; i < 4096; i += 64) lds[i] = 1.0f + i * 1e-5f;
    __syncthreads();
    long long t0, t1;
    // 0: dependent fma chain
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) w = fmaf(w, a, 0.25f);
    t1 = clock64(); if (lane == 0) t[0] = t1 - t0;
    // 1: fma -> readlane(runtime lane) -> fma
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { const float s = lane_bcast(w, (i + nrt) & 63); w = fmaf(a, s, 0.25f); }
    t1 = clock64(); if (lane == 0) t[1] = t1 - t0;
    // 2: fma -> readlane(constant lane) -> fma
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { const float s = lane_bcast(w, 5); w = fmaf(a, s, 0.25f); }
    t1 = clock64(); if (lane == 0) t[2] = t1 - t0;
    // 3: masked update + readlane: the GS row step
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        const int rr = (i + nrt) & 63;
        float nl = fmaf(-w, a, lam); if (nl < 0.0f) nl = 0.0f;
        const float dl = nl - lam;
        w = fmaf(a, lane_bcast(dl, rr), w);
        lam = (lane == rr) ? nl : lam;
    }
    t1 = clock64(); if (lane == 0) t[3] = t1 - t0;
    // 4: dependent LDS round trip (write own, read neighbour)
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { lds[lane] = w; w = lds[(lane + 1) & 63] + 0.25f; }
    t1 = clock64(); if (lane == 0) t[4] = t1 - t0;
    // 5: dependent ds_bpermute
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { w = __shfl(w, (lane + 1) & 63) + 0.25f; }
    t1 = clock64(); if (lane == 0) t[5] = t1 - t0;
    // 6: DPP dependent chain (row_shr 1)
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { w = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(w), 0x111, 0xF, 0xF, false)) + 0.25f; }
    t1 = clock64(); if (lane == 0) t[6] = t1 - t0;
    // 7: independent fmas (8 chains)
    float v[8]; for (int k = 0; k < 8; ++k) v[k] = w + k;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], a, 0.25f); }
    t1 = clock64(); if (lane == 0) t[7] = t1 - t0;
    for (int k = 0; k < 8; ++k) w += v[k];
    // 8: dependent sqrt (correctly rounded) ; 9: dependent division
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) w = sqrtf(w + 2.0f);
    t1 = clock64(); if (lane == 0) t[8] = t1 - t0;
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) w = 1.0f / (w + 2.0f);
    t1 = clock64(); if (lane == 0) t[9] = t1 - t0;
    // 10: uniform branch taken per iteration on a VALU compare
    t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; ++i) { const float s = lane_bcast(w, 3); if (s > 1e30f) w += 1.0f; w = fmaf(w, a, 0.25f); }
    t1 = clock64(); if (lane == 0) t[10] = t1 - t0;
    out[lane] = w + lam;
}
int main() {
    float *o; long long *t; hipMalloc(&o, 256); hipMalloc(&t, 16 * 8);
    hipMemset(o, 0, 256);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o, t, rep); hipDeviceSynchronize(); }
    long long h[16]; hipMemcpy(h, t, 16 * 8, hipMemcpyDeviceToHost);
    const char *nm[] = {"dependent fma", "fma->readlane(sgpr lane)->fma", "fma->readlane(const lane)->fma", "GS row step (4 VALU + readlane + select)",
                        "LDS write+read round trip", "ds_bpermute", "DPP row_shr + add", "8 independent fmas (per 8)", "dependent sqrtf", "dependent 1/x",
                        "readlane + uniform branch + fma"};
    for (int i = 0; i < 11; ++i) printf("%-44s %7.1f ticks(100MHz)/iter = ~%6.1f cycles @2.4GHz\n", nm[i], (double)h[i] / N, (double)h[i] / N * 24.0);
    return 0;
}
