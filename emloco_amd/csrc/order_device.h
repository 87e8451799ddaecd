// order_device.h -- the counting sort behind the cost-ordered dispatch of the rigid-body launch, as a device function of a
// 1024-thread workgroup: sim_order_kernel (sim_kernels.hip) and the second workgroup of compact_order_kernel (chain_kernels.hip:
// one launch for the finished-env compaction and the next step's dispatch order) run the same body.
#pragma once
#include <hip/hip_runtime.h>

namespace emloco {

#define EMLOCO_ORDER_BUCKETS 128
#define EMLOCO_ORDER_LDS_ENVS 16384

__device__ __forceinline__ void order_sort(const unsigned *ticks, int n, int *order, unsigned char *bucket_ws) {
    __shared__ int sh_cnt[EMLOCO_ORDER_BUCKETS];
    // every key is read ONCE (a launch that still writes keys beside this one cannot make the two passes disagree) and its
    // bucket kept in LDS -- in the global workspace `bucket_ws` [n] beyond EMLOCO_ORDER_LDS_ENVS envs
    __shared__ unsigned char sh_b[EMLOCO_ORDER_LDS_ENVS];
    unsigned char *bk = n <= EMLOCO_ORDER_LDS_ENVS ? sh_b : bucket_ws;
    const int tid = threadIdx.x;
    if (tid < EMLOCO_ORDER_BUCKETS) sh_cnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        unsigned b = ticks[i];
        b = b > EMLOCO_ORDER_BUCKETS - 1 ? EMLOCO_ORDER_BUCKETS - 1 : b;
        bk[i] = (unsigned char)b;
        atomicAdd(&sh_cnt[b], 1);
    }
    __syncthreads();
    if (tid == 0) {                                               // descending: start offset of bucket b = envs in buckets above it
        int run = 0;
        for (int b = EMLOCO_ORDER_BUCKETS - 1; b >= 0; --b) { const int c = sh_cnt[b]; sh_cnt[b] = run; run += c; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&sh_cnt[bk[i]], 1)] = i;      // a thread re-reads only what it wrote itself
}

}  // namespace emloco
