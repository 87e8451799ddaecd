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
    if (tid < 64) {                                               // descending: start offset of bucket b = envs in buckets above it
        // one wave scans the 128 counts (a lane takes two neighbouring buckets, highest first): an exclusive prefix sum over the
        // lanes in six shuffle steps -- the single-thread loop this replaces was 128 dependent LDS round trips, ~5 us of the 6.5 us
        // launch that sits between the flags launch and the reset / observation launch of every rollout step
        static_assert(EMLOCO_ORDER_BUCKETS == 128, "two buckets per lane of one wave");
        const int b0 = EMLOCO_ORDER_BUCKETS - 1 - 2 * tid, b1 = b0 - 1;
        const int c0 = sh_cnt[b0], c1 = sh_cnt[b1];
        int inc = c0 + c1;                                        // inclusive scan of the pair sums
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl(inc, tid - d < 0 ? 0 : tid - d);
            if (tid >= d) inc += up;
        }
        const int excl = inc - (c0 + c1);
        sh_cnt[b0] = excl;
        sh_cnt[b1] = excl + c0;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&sh_cnt[bk[i]], 1)] = i;      // a thread re-reads only what it wrote itself
}

}  // namespace emloco
