// pmc_calib.hip -- known-bytes kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section):
// each kernel moves exactly `n` bytes in and `n` bytes out (buffers far larger than the caches, touched once), with the access
// patterns sim_step_kernel uses: plain 16-byte accesses, 16-byte `sc1` (write-through / system-coherent) accesses of the split
// launch's hand-over, and 4-byte accesses.  Build: hipcc --offload-arch=gfx950 -O3 tools/exp/pmc_calib.hip -o tools/exp/_build/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void copy16_plain(const f4 *a, f4 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy16_sc1(const f4 *a, f4 *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a + i) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(b + i), "v"(v) : "memory");
    }
}
__global__ void copy4_plain(const float *a, float *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
    const size_t bytes = 1ull << 30;          // 1 GiB in, 1 GiB out per launch
    void *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(copy16_plain, dim3(4096), dim3(256), 0, 0, (const f4 *)a, (f4 *)b, bytes / 16);
        hipLaunchKernelGGL(copy16_sc1, dim3(4096), dim3(256), 0, 0, (const f4 *)a, (f4 *)b, bytes / 16);
        hipLaunchKernelGGL(copy4_plain, dim3(4096), dim3(256), 0, 0, (const float *)a, (float *)b, bytes / 4);
    }
    hipDeviceSynchronize();
    printf("moved %zu bytes in and out per launch\n", bytes);
    return 0;
}
