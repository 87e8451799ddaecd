// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  hipcc --offload-arch=gfx950 -O2 anyorder_probe.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out) out[0] = 1;
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int *d; hipMalloc(&d, 4);
    for (int flags = 0; flags < 2; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 10000LL, d);                 // 100 us at 100 MHz
            hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, flags, 10000LL, d);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("flags=%d rep=%d: two 100 us kernels back to back took %.1f us\n", flags, rep, ms * 1e3f);
        }
    }
    return 0;
}
