// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny host-side stand-in for the parts of the HIP device language the kernels in
// emloco_amd/csrc/*_kernels.hip use, so the SAME kernel source can be compiled with g++ and
// executed on the CPU by the `-m "not gpu"` tests (there is no GPU in the build container).
// One workgroup = blockDim.x fibers; __syncthreads() yields to a round-robin scheduler; wave shuffles /
// ballots exchange through a per-block buffer.  Blocks run one after another.  This is NOT a product path:
// emloco_amd/ never includes it, and the product fails loudly without the gfx950 library.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct emu_dim3 { unsigned x = 1, y = 1, z = 1; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
extern emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

// One workgroup = blockDim.x fibers (ucontext) run round-robin by a scheduler: a fiber runs until its
// next barrier, then the next fiber runs; when all have arrived the round starts again.  That is the
// lock-step-between-barriers semantics the kernels rely on, at ~0.1 us per switch.
namespace emu {
void barrier();
void wave_barrier();
void launch_impl(unsigned grid, unsigned block, const std::function<void()> &body);
extern uint64_t g_xbuf[1024];
template <class F> void launch(unsigned grid, unsigned block, F body) { launch_impl(grid, block, std::function<void()>(body)); }
}  // namespace emu

static inline void __syncthreads() { emu::barrier(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}

template <class T> static inline T emu_exchange(T v, int src) {
    static_assert(sizeof(T) <= 8, "");
    uint64_t raw = 0; std::memcpy(&raw, &v, sizeof(T));
    emu::g_xbuf[threadIdx.x] = raw;
    emu::wave_barrier();
    uint64_t got = emu::g_xbuf[(threadIdx.x & ~63u) | (unsigned)(src & 63)];
    emu::wave_barrier();
    T out; std::memcpy(&out, &got, sizeof(T));
    return out;
}
template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu_exchange(v, (int)(threadIdx.x & 63) ^ mask); }
template <class T> static inline T __shfl(T v, int src, int = 64) { return emu_exchange(v, src); }
static inline unsigned long long __ballot(int pred) {
    emu::g_xbuf[threadIdx.x] = pred ? 1 : 0;
    emu::wave_barrier();
    unsigned long long m = 0;
    unsigned base = threadIdx.x & ~63u;
    for (int i = 0; i < 64; ++i) if (base + i < blockDim.x && emu::g_xbuf[base + i]) m |= 1ull << i;
    emu::wave_barrier();
    return m;
}
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline unsigned long long wall_clock64() { return 0ull; }
static inline void __threadfence() {}
static inline void __builtin_amdgcn_s_sleep(int) {}
// lock-step fibers run one at a time: a plain read-modify-write is atomic
static inline int atomicAdd(int *p, int v) { const int o = *p; *p = o + v; return o; }
static inline float __int_as_float(int x) { float f; std::memcpy(&f, &x, 4); return f; }
static inline int __float_as_int(float f) { int x; std::memcpy(&x, &f, 4); return x; }
static inline float __fdividef(float a, float b) { return a / b; }
#define AT_EXP(x) std::exp(x)   // the attention kernels use v_exp_f32 (__expf) on the device
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
using std::fabs; using std::sqrt; using std::floor; using std::ceil;

// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; result fragment
// col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5); exact fp32 fmaf chain in k order (cdna_hip_programming.md section 3).
namespace emu { extern uint64_t g_ybuf[1024]; }
typedef float emu_f32x16 __attribute__((vector_size(64)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
    const unsigned tid = threadIdx.x, base = tid & ~63u, lane = tid & 63u;
    uint64_t ra = 0, rb = 0;
    std::memcpy(&ra, &a, 4); std::memcpy(&rb, &b, 4);
    emu::g_xbuf[tid] = ra; emu::g_ybuf[tid] = rb;
    emu::wave_barrier();
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (int)(lane >> 5), col = (int)(lane & 31);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            std::memcpy(&av, &emu::g_xbuf[base + row + 32 * k], 4);
            std::memcpy(&bv, &emu::g_ybuf[base + col + 32 * k], 4);
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
// v_mfma_f32_32x32x16_bf16 behind the kernel's gemm_* wrappers: lane l supplies 8 consecutive k = 8 * (l >> 5) + e of row /
// column l & 31, operands rounded to bf16 (nearest even), products and sums in fp32 (ascending k: the hardware's internal
// order is not specified, the test compares with a tolerance).
#define EMLOCO_EMU 1
struct gemm_bf16x8 { unsigned short v[8]; };
struct f32x4;
static inline unsigned short emu_f32_to_bf16(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);     // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float emu_bf16_to_f32(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
template <class F4> static inline gemm_bf16x8 gemm_pack_bf16(const F4 &lo, const F4 &hi) {
    gemm_bf16x8 r;
    const float in[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    for (int e = 0; e < 8; ++e) r.v[e] = emu_f32_to_bf16(in[e]);
    return r;
}
static inline emu_f32x16 gemm_mfma_bf16(gemm_bf16x8 a, gemm_bf16x8 b, emu_f32x16 c) {
    static gemm_bf16x8 bufa[1024], bufb[1024];
    const unsigned tid = threadIdx.x, base = tid & ~63u, lane = tid & 63u;
    bufa[tid] = a; bufb[tid] = b;
    emu::wave_barrier();
    emu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (int)(lane >> 5), col = (int)(lane & 31);
        float acc = c[r];
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e)
                acc = std::fmaf(emu_bf16_to_f32(bufa[base + row + 32 * h].v[e]), emu_bf16_to_f32(bufb[base + col + 32 * h].v[e]), acc);
        d[r] = acc;
    }
    emu::wave_barrier();
    return d;
}
struct alignas(16) bf16w4 { unsigned x, y, z, w; };
static inline unsigned gemm_pack2_bf16(float a, float b) { return (unsigned)emu_f32_to_bf16(a) | ((unsigned)emu_f32_to_bf16(b) << 16); }
static inline emu_f32x16 gemm_mfma_bf16_w(const bf16w4 &a, const bf16w4 &b, emu_f32x16 c) {
    gemm_bf16x8 pa, pb;
    std::memcpy(pa.v, &a, 16); std::memcpy(pb.v, &b, 16);
    return gemm_mfma_bf16(pa, pb, c);
}
using std::exp; using std::cos; using std::sin; using std::atan2;

// DPP lane permutations (the gfx9 dpp_ctrl codes the kernels use) and v_readlane
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)old; (void)row_mask; (void)bank_mask; (void)bound_ctrl;
    const unsigned lane = threadIdx.x & 63u;
    unsigned from;
    if (ctrl == 0x140) from = (lane & ~15u) | (15u - (lane & 15u));                 // row_mirror
    else if (ctrl == 0x141) from = (lane & ~7u) | (7u - (lane & 7u));               // row_half_mirror
    else if (ctrl >= 0 && ctrl <= 0xFF) from = (lane & ~3u) | ((unsigned)(ctrl >> (2 * (lane & 3u))) & 3u);   // quad_perm
    else from = lane;
    return emu_exchange(src, (int)from);
}
static inline int __builtin_amdgcn_readlane(int v, int src_lane) { return emu_exchange(v, src_lane); }
// v_permlane32_swap_b32: lanes 32-63 of the first operand change places with lanes 0-31 of the second
typedef unsigned emu_u32x2 __attribute__((vector_size(8)));
static inline emu_u32x2 __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned pa = emu_exchange(a, (int)(lane ^ 32u)), pb = emu_exchange(b, (int)(lane ^ 32u));
    emu_u32x2 r;
    r[0] = lane < 32u ? a : pb;
    r[1] = lane < 32u ? pa : b;
    return r;
}
static inline unsigned __float_as_uint(float f) { unsigned x; std::memcpy(&x, &f, 4); return x; }
static inline float __uint_as_float(unsigned x) { float f; std::memcpy(&f, &x, 4); return f; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return emu_exchange(v, 0); }
