// predictor_kernels.hip -- Social-Transmotion + LocoVal kernels for gfx950 (MI355X).
//
// Reference model: /root/reference/social-transmotion/model_jta.py:130-336 (TransMotionJTA: 6 local + 3 global
// post-norm encoder layers, d=128, 4 heads, ff=1024) and /root/reference/pacer/pacer/learning/value_pose_net.py
// (LocoVal MLP 100->49->24->1).  Every dense contraction (QKV / out / FFN projections, Q.K^T, P.V and all their
// backward products) runs on the matrix cores through ONE batched GEMM kernel built on
// v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, bit-equal to an fmaf chain, 157 TFLOP/s peak on MI355X
// (cdna_hip_programming.md section 3).  fp32 keeps the predictor inside the 1e-4 parity bar of the north star;
// a bf16 operand path (v_mfma_f32_32x32x16_bf16, 16x the rate) is the next step.
//
// Tiling: 128x128 output tile per 256-thread workgroup = 4 waves of 64x64 (2x2 MFMA tiles, 64 accumulator
// registers per lane); K is staged through LDS 16 deep, operands stored k-major so a wave reads its A / B
// fragments as 32 consecutive floats (conflict-free ds_read_b32).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/emloco_predictor.h"
#include "mfma_bf16.h"
#include "drop_hash.h"
#include "dev_math.h"
#include "locoval_returns_device.h"

namespace emloco {

typedef float f32x16 __attribute__((vector_size(64)));

// K depth of one LDS stage is a template parameter GBK (16 or 32); LDS rows are [row][GBK + 4] floats: the 80 B / 144 B
// row stride keeps the ds_read_b128 of 16 consecutive rows on distinct banks.

struct GemmArgs {
    int batch, m, n, k;
    float alpha;
    const float *A; int lda; long sa; int ta;
    const float *B; int ldb; long sb; int tb;
    float *C; int ldc; long sc;
    const float *bias; int flags; int ksplit; float *ws;
    int vec_a, vec_b;   // operand may be read with 16-byte loads (base, leading dimension and batch stride 16 B aligned)
    float drop_p; unsigned drop_seed;   // flags & 8: inverted dropout in the epilogue (after bias / ReLU)
    // flags & 32 (backward through dropout(relu(.)) fused into the GEMM that produces the incoming gradient): the value is
    // multiplied by mask_scale where mask[row][col] > 0 (the forward OUTPUT, laid out like C) and zeroed elsewhere, and the
    // column sums of what was written over the 64 rows of this wave go to colpart[(tile_row * WM + wm)][col] (bias gradient)
    const float *mask; float mask_scale; float *colpart;
    // memory dtypes of the reduced-precision mode (flags & 16 only): an operand / the output / the mask holds bf16 (2 bytes per
    // element; leading dimensions and strides still count elements).  The large activations of an encoder layer -- the fused
    // q|k|v projection and the feed-forward hidden layer -- live in HBM as bf16 there; accumulation stays fp32.
    int a16, b16, c16, m16;
    int bimg;           // split mode: B points at the piece image of a weight (gemm_split_pack_kernel), not at the matrix
    int small;          // split mode on the 64 x 64 tile (gemm_split_small_kernel): the launcher sets it and sizes the grid for it
    int vec_c;          // the output may be written with 16-byte stores (base, leading dimension and batch stride 16 B aligned): the
                        // epilogue then passes each 32 x 32 accumulator tile through LDS and stores whole lines (round 5)
};

// the 64 x 64 split tile serves a launch whose 128 x 128 grid (split-K included) would be at most this many workgroups.  Measured on
// MI355X (tools/exp/probe_small_gemm.py, M = 2 048 / 4 096 layers of the policy, the discriminator and the PPO update): with the
// split-K factor chosen for ~512 workgroups the 128 x 128 tile wins everywhere except the narrowest layers (4096 x 256 x 512: 19.8
// vs 22.5 us, the 69-wide action head: 18.2 vs 20.3 us) -- the small tile halves the matrix work per split piece; it is kept for
// launches that stay under half a workgroup per CU even after splitting k.
constexpr long GEMM_SMALL_MAX_WG = 128;
inline bool gemm_use_small_tile(int batch, int m, int n, int ksplit) {
    return (long)((m + 127) / 128) * ((n + 127) / 128) * batch * ksplit <= GEMM_SMALL_MAX_WG;
}

// bf16 <-> fp32 on the way to / from memory: widening is a shift, narrowing rounds to nearest even
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {                         // branch-free (selects): unrolled epilogues stay unrolled
    const unsigned u = __float_as_uint(f);
    const unsigned r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    const unsigned q = (u >> 16) | 0x40u;                                                  // NaN stays NaN
    return (unsigned short)(((u & 0x7fffffffu) > 0x7f800000u) ? q : r);
}
__device__ __forceinline__ float ld_act(const float *p, long i, int is16) {
    return is16 ? bf16_lo(((const unsigned short *)p)[i]) : p[i];
}
__device__ __forceinline__ void st_act(float *p, long i, float v, int is16) {
    if (is16) ((unsigned short *)p)[i] = f32_to_bf16(v); else p[i] = v;
}

// Counter-based dropout mask of the GEMM epilogues: keep element `idx` of a launch with seed `seed` iff drop_keep(seed, idx, p)
// (drop_hash.h: a stateless hash of the element's flat index, so the backward recomputes the mask instead of storing it).
struct __attribute__((aligned(16))) f32x4 { float x, y, z, w; };
#ifdef EMLOCO_EMU
__device__ __forceinline__ void gemm_wave_sync() { emu::wave_barrier(); }           // lock-step fibers: the wave's lanes meet between a scratch write and its read
#else
__device__ __forceinline__ void gemm_wave_sync() { __builtin_amdgcn_wave_barrier(); }  // DS instructions of one wave complete in issue order: a scheduling fence only
#endif

// One operand stage: a ROWS x GBK slab of X (rows = output rows for A, output columns for B) fetched with 16-byte
// global loads into registers (`gemm_fetch`), later stored to LDS as [row][k] (`gemm_stash`).  Splitting fetch
// from stash lets the loads of stage t+1 fly while the MFMAs of stage t run (register-prefetch double buffering).
// The fetch is BRANCH-FREE: out-of-range rows / k are clamped to a valid address and the value is zeroed with a
// select, so the compiler keeps every load asynchronous (a branchy version made it wait after each load).
// TRANS: 0 = X[row][k] (k contiguous), 1 = X[k][row] (row contiguous).  VEC: 16-byte loads allowed (base, leading
// dimension and batch stride 16 B aligned; then a clamped quad never leaves its row: ld % 4 == 0 and rows, k <= ld).
// IO = 1: the operand is bf16 in memory (VEC only): a quad is one 8-byte load, widened on arrival.
// MASKED = 0 (round 5): the tile lies inside the operand and its k range is whole stages -- no row clamps, no masks (mk = 15 is a
// compile-time constant: the selects of the stash fold away); only k is still clamped, for the one stage fetched past the end.
template <int ROWS, int GBK, int TRANS, int VEC, int IO = 0, int MASKED = 1>
__device__ __forceinline__ void gemm_fetch(f32x4 (&v)[(ROWS * GBK / 4 + 255) / 256], unsigned (&mk)[(ROWS * GBK / 4 + 255) / 256],
                                           const float *X, int ld, int row0, int rows, int k0, int ke, int tid) {
    static_assert(IO == 0 || VEC == 1, "bf16 operands take the vector-load path");
    static_assert(MASKED == 1 || VEC == 1, "the unmasked fetch is the vector-load path's");
    constexpr int NQ = ROWS * GBK / 4, QR = GBK / 4;
    for (int e = 0; e < (NQ + 255) / 256; ++e) {
        int idx = tid + 256 * e;
        if (NQ % 256 != 0 && idx >= NQ) idx = NQ - 1;          // surplus threads re-read the last quad (never stashed)
        f32x4 t;
        if constexpr (!MASKED) {
            // the stage's k origin is workgroup-uniform (a scalar), and so is its clamp for the stage past the end: the per-thread part of
            // the address does not depend on the stage
            const int kbase = k0 < ke ? k0 : ke - GBK;
            if (!TRANS) {
                const int r = idx / QR, q = idx - r * QR;
                const long off = (long)(row0 + r) * ld + 4 * q;
                if (IO) {
                    const uint2 u = *(const uint2 *)((const unsigned short *)X + kbase + off);
                    t.x = bf16_lo(u.x); t.y = bf16_hi(u.x); t.z = bf16_lo(u.y); t.w = bf16_hi(u.y);
                } else t = *(const f32x4 *)(X + kbase + off);
            } else {
                const int kk = idx / (ROWS / 4), rq = idx - kk * (ROWS / 4);
                const long off = (long)kk * ld + row0 + 4 * rq;
                const long kb_off = (long)kbase * ld;
                if (IO) {
                    const uint2 u = *(const uint2 *)((const unsigned short *)X + kb_off + off);
                    t.x = bf16_lo(u.x); t.y = bf16_hi(u.x); t.z = bf16_lo(u.y); t.w = bf16_hi(u.y);
                } else t = *(const f32x4 *)(X + kb_off + off);
            }
            mk[e] = 15u;
            v[e] = t;
            continue;
        }
        bool ok0, ok1, ok2, ok3;
        if (!TRANS) {                                          // 4 consecutive k of one row
            const int r = idx / QR, q = idx - r * QR;
            const int row = row0 + r, kk = k0 + 4 * q;
            const int rowc = row < rows ? row : rows - 1;
            const bool rok = row < rows;
            ok0 = rok && kk < ke; ok1 = rok && kk + 1 < ke; ok2 = rok && kk + 2 < ke; ok3 = rok && kk + 3 < ke;
            const float *base = X + (long)rowc * ld;
            if (IO) {
                const int kq = (ke - 1) & ~3;
                const uint2 u = *(const uint2 *)((const unsigned short *)X + (long)rowc * ld + (kk < kq ? kk : kq));
                t.x = bf16_lo(u.x); t.y = bf16_hi(u.x); t.z = bf16_lo(u.y); t.w = bf16_hi(u.y);
            } else if (VEC) {
                const int kq = (ke - 1) & ~3;
                t = *(const f32x4 *)(base + (kk < kq ? kk : kq));
            } else {
                const int kl = ke - 1;
                t.x = base[kk < kl ? kk : kl]; t.y = base[kk + 1 < kl ? kk + 1 : kl];
                t.z = base[kk + 2 < kl ? kk + 2 : kl]; t.w = base[kk + 3 < kl ? kk + 3 : kl];
            }
        } else {                                               // 4 consecutive rows at one k
            const int kk = idx / (ROWS / 4), rq = idx - kk * (ROWS / 4);
            const int row = row0 + 4 * rq, kg = k0 + kk;
            const int kc = kg < ke ? kg : ke - 1;
            const bool kok = kg < ke;
            ok0 = kok && row < rows; ok1 = kok && row + 1 < rows; ok2 = kok && row + 2 < rows; ok3 = kok && row + 3 < rows;
            const float *base = X + (long)kc * ld;
            if (IO) {
                const int rq4 = (rows - 1) & ~3;
                const uint2 u = *(const uint2 *)((const unsigned short *)X + (long)kc * ld + (row < rq4 ? row : rq4));
                t.x = bf16_lo(u.x); t.y = bf16_hi(u.x); t.z = bf16_lo(u.y); t.w = bf16_hi(u.y);
            } else if (VEC) {
                const int rq4 = (rows - 1) & ~3;
                t = *(const f32x4 *)(base + (row < rq4 ? row : rq4));
            } else {
                const int rl = rows - 1;
                t.x = base[row < rl ? row : rl]; t.y = base[row + 1 < rl ? row + 1 : rl];
                t.z = base[row + 2 < rl ? row + 2 : rl]; t.w = base[row + 3 < rl ? row + 3 : rl];
            }
        }
        // the loaded value is not touched here (the mask is applied by the stash): a select now would wait for the load
        mk[e] = (ok0 ? 1u : 0u) | (ok1 ? 2u : 0u) | (ok2 ? 4u : 0u) | (ok3 ? 8u : 0u);
        v[e] = t;
    }
}

template <int ROWS, int GBK, int TRANS>
__device__ __forceinline__ void gemm_stash(const f32x4 (&v)[(ROWS * GBK / 4 + 255) / 256], const unsigned (&mk)[(ROWS * GBK / 4 + 255) / 256],
                                           float *S, int tid) {
    constexpr int NQ = ROWS * GBK / 4, QR = GBK / 4, GLDK = GBK + 4;
    for (int e = 0; e < (NQ + 255) / 256; ++e) {
        const int idx = tid + 256 * e;
        if (NQ % 256 == 0 || idx < NQ) {
            f32x4 t = v[e];
            t.x = (mk[e] & 1u) ? t.x : 0.0f; t.y = (mk[e] & 2u) ? t.y : 0.0f; t.z = (mk[e] & 4u) ? t.z : 0.0f; t.w = (mk[e] & 8u) ? t.w : 0.0f;
            if (!TRANS) {
                const int r = idx / QR, q = idx - r * QR;
                *(f32x4 *)&S[r * GLDK + 4 * q] = t;
            } else {                                           // row-contiguous operand: LDS stage is k-major [k][ROWS + 4]
                const int kk = idx / (ROWS / 4), rq = idx - kk * (ROWS / 4);
                *(f32x4 *)&S[kk * (ROWS + 4) + 4 * rq] = t;
            }
        }
    }
}

// A lane's fragment values of one 32-row block for a stage: k in [half8, half8 + GBK/2).  [row][k] stages give them as
// GBK/8 ds_read_b128 of one LDS row; k-major stages (TRANS operands, stored with one ds_write_b128 per fetched quad instead
// of four scattered ds_write_b32) as conflict-free ds_read_b32 of 32 consecutive rows.
template <int ROWS, int GBK, int TRANS>
__device__ __forceinline__ void gemm_frag(const float *S, int row, int half8, f32x4 (&out)[GBK / 8]) {
    if (!TRANS) {
        const float *src = S + row * (GBK + 4) + half8;
        for (int f = 0; f < GBK / 8; ++f) out[f] = *(const f32x4 *)(src + 4 * f);
    } else {
        const float *src = S + half8 * (ROWS + 4) + row;
        for (int f = 0; f < GBK / 8; ++f) {
            out[f].x = src[(4 * f) * (ROWS + 4)]; out[f].y = src[(4 * f + 1) * (ROWS + 4)];
            out[f].z = src[(4 * f + 2) * (ROWS + 4)]; out[f].w = src[(4 * f + 3) * (ROWS + 4)];
        }
    }
}

// ---- PREC = 2, the split mode: fp32 products rebuilt from bf16 pieces -------------------------------------------------------
// gfx950 runs v_mfma_f32_32x32x2_f32 at 1/16 of the bf16 rate (157 vs 2500 TFLOP/s) and has no xf32.  An fp32 value is the exact
// sum of three bf16 pieces a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2); each remainder is exact in
// fp32 and round-to-nearest leaves it at most half an ulp of the piece above, so three 8-bit significands cover the 24), and
//   a b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + O(2^-24 a b)
// -- six v_mfma_f32_32x32x16_bf16 per 16 k (192 matrix-pipe cycles) where the fp32 instruction needs eight x 64 = 512.  Every
// bf16 x bf16 product is exact in fp32 and the accumulation is fp32, so the result is in fp32's error class: measured against
// float64 (tools/exp/bf16split_err.hip, K = 128 .. 4096, signed / all-positive / wide-range data) max 1.2e-7 .. 2.4e-6 of
// sum |a b| for the six-term form vs 1.2e-7 .. 2.8e-6 for the fp32 instruction; nine terms change nothing, three terms give
// 2e-6 (not fp32 class).  It is NOT bit-equal to the fmaf chain.  Inf operands come out NaN (inf - inf in the remainder).
//
// The pieces are made ONCE per element, when a stage goes to LDS (not per consuming wave).  LDS image of one operand stage
// (16 k): three planes, each [2 halves][ROWS + 4 slots], a slot = the 8 bf16 (16 B) of one row and one half of the stage -- what
// a lane feeds one matrix instruction, read with a single ds_read_b128; 16 consecutive rows = 256 consecutive bytes = all 64
// banks.  The +4 slots put the two halves on different banks for the 8-byte stores of a [row][k] operand (thread = 4 k of a row).
// A row-contiguous (TRANS) operand arrives as 4 rows at one k per thread; a thread takes TWO adjacent k there and stores one
// packed word per row and plane, the rows of a 16-row group bit-permuted (and crossed with the next row bit) so that the 32
// lanes of a store group hit 16 banks (2-way: free on a 4-byte store) instead of 2; reads stay conflict-free because the
// permutation keeps every aligned 16-row group inside its 16 slots.
constexpr int split_half_words(int rows) { return (rows + 4) * 4; }
constexpr int split_plane_words(int rows) { return 2 * split_half_words(rows); }
constexpr int split_stage_words(int rows) { return 3 * split_plane_words(rows); }
template <int TRANS>
__device__ __forceinline__ int split_slot(int row) {
    if (!TRANS) return row;
    const int b0 = row & 1, b1 = (row >> 1) & 1, b2 = (row >> 2) & 1, b3 = (row >> 3) & 1, b4 = (row >> 4) & 1;
    return (row & ~15) | (b0 << 3) | ((b1 ^ b4) << 2) | (b3 << 1) | b2;
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = gemm_pack2_bf16(a, b);
    float ra = a - bf16_lo(p1), rb = b - bf16_hi(p1);
    p2 = gemm_pack2_bf16(ra, rb);
    ra -= bf16_lo(p2); rb -= bf16_hi(p2);
    p3 = gemm_pack2_bf16(ra, rb);
}
// thread -> (row quad, k pair) of a TRANS operand stage: 16-lane runs of row quads (256 contiguous bytes per run and k), the
// two runs of a 32-lane store group on adjacent k pairs.  A 64-row stage (the small tile) has 16 row quads x 8 k pairs: the first
// two waves carry it (returns false for the idle threads, which fetch a valid duplicate and stash nothing).
template <int ROWS>
__device__ __forceinline__ bool split_trans_map(int tid, int &rq, int &kp) {
    static_assert(ROWS == 128 || ROWS == 64, "split stages are 128 or 64 rows x 16 k on 256 threads");
    if (ROWS == 128) {
        rq = (tid & 15) + 16 * ((tid >> 5) & 1);
        kp = ((tid >> 4) & 1) + 2 * (tid >> 6);
        return true;
    }
    rq = tid & 15;
    kp = (tid >> 4) & 7;
    return tid < 128;
}
// TRANS operand, ROWS rows x 16 k, 256 threads: the quads (4 rows) at k = 2 kp and 2 kp + 1
template <int ROWS, int MASKED = 1>
__device__ __forceinline__ void split_fetch_trans(f32x4 (&v)[2], unsigned (&mk)[2], const float *X, int ld, int row0, int rows, int k0, int ke, int tid) {
    int rq, kp;
    split_trans_map<ROWS>(tid, rq, kp);
    if constexpr (!MASKED) {
        const long kb_off = (long)(k0 < ke ? k0 : ke - 16) * ld;          // workgroup-uniform (scalar): the stage, clamped past the end
        for (int e = 0; e < 2; ++e) {
            const long off = (long)(2 * kp + e) * ld + row0 + 4 * rq;     // per thread, the same for every stage
            v[e] = *(const f32x4 *)(X + kb_off + off);
            mk[e] = 15u;
        }
        return;
    }
    const int row = row0 + 4 * rq, rq4 = (rows - 1) & ~3;
    const int rowc = row < rq4 ? row : rq4;
    const unsigned rm = (row < rows ? 1u : 0u) | (row + 1 < rows ? 2u : 0u) | (row + 2 < rows ? 4u : 0u) | (row + 3 < rows ? 8u : 0u);
    for (int e = 0; e < 2; ++e) {
        const int kg = k0 + 2 * kp + e;
        const int kc = kg < ke ? kg : ke - 1;
        v[e] = *(const f32x4 *)(X + (long)kc * ld + rowc);
        mk[e] = kg < ke ? rm : 0u;
    }
}
// quads a thread holds of one operand stage: 2 adjacent k of a row quad (TRANS), or ROWS / 64 quads of 4 consecutive k
constexpr int split_nv(int rows, int trans) { return trans ? 2 : rows / 64; }
template <int ROWS, int TRANS, int NPC = 3>      // NPC = 2: the third plane is neither cut nor written (two-piece products, round 6)
__device__ __forceinline__ void split_stash(const f32x4 (&v)[split_nv(ROWS, TRANS)], const unsigned (&mk)[split_nv(ROWS, TRANS)], unsigned *S, int tid) {
    static_assert(ROWS == 128 || ROWS == 64, "split stages are 128 or 64 rows x 16 k on 256 threads");
    constexpr int HW = split_half_words(ROWS), PW = split_plane_words(ROWS), NV = split_nv(ROWS, TRANS);
    f32x4 t[NV];
    for (int e = 0; e < NV; ++e) {
        t[e] = v[e];
        t[e].x = (mk[e] & 1u) ? t[e].x : 0.0f; t[e].y = (mk[e] & 2u) ? t[e].y : 0.0f;
        t[e].z = (mk[e] & 4u) ? t[e].z : 0.0f; t[e].w = (mk[e] & 8u) ? t[e].w : 0.0f;
    }
    if constexpr (!TRANS) {
        for (int e = 0; e < NV; ++e) {                         // quad = 4 consecutive k of one row: 8 bytes per plane
            const int idx = tid + 256 * e, r = idx >> 2, q = idx & 3;
            unsigned a1, a2, a3, b1, b2, b3;
            split_pair(t[e].x, t[e].y, a1, a2, a3);
            split_pair(t[e].z, t[e].w, b1, b2, b3);
            unsigned *dst = S + (q >> 1) * HW + r * 4 + (q & 1) * 2;
            *(uint2 *)(dst) = uint2{a1, b1};
            *(uint2 *)(dst + PW) = uint2{a2, b2};
            if constexpr (NPC == 3) *(uint2 *)(dst + 2 * PW) = uint2{a3, b3};
        }
    } else {                                                   // 4 rows x 2 adjacent k: one word per row and plane
        int rq, kp;
        if (!split_trans_map<ROWS>(tid, rq, kp)) return;
        const float lo[4] = {t[0].x, t[0].y, t[0].z, t[0].w}, hi[4] = {t[1].x, t[1].y, t[1].z, t[1].w};
        unsigned *base = S + (kp >> 2) * HW + (kp & 3);
        for (int j = 0; j < 4; ++j) {
            unsigned p1, p2, p3;
            split_pair(lo[j], hi[j], p1, p2, p3);
            unsigned *dst = base + split_slot<1>(4 * rq + j) * 4;
            dst[0] = p1; dst[PW] = p2;
            if constexpr (NPC == 3) dst[2 * PW] = p3;
        }
    }
}
// the three pieces of a lane's 8 reduction entries (its half of the stage) of one row
template <int ROWS, int TRANS, int NPC = 3>
__device__ __forceinline__ void split_frag(const unsigned *S, int row, int half, bf16w4 (&out)[3]) {
    const unsigned *src = S + half * split_half_words(ROWS) + split_slot<TRANS>(row) * 4;
    for (int p = 0; p < NPC; ++p) out[p] = *(const bf16w4 *)(src + p * split_plane_words(ROWS));
}

// ---- a B operand that was cut ONCE (round 5): the piece image of a weight matrix -------------------------------------------
// In y = x W^T and dx = dy W the B operand is a WEIGHT: every one of the thousands of workgroups of a tall GEMM cuts the same 128 x 16
// stage of it into pieces again (44 of a stage's ~105 vector instructions, beside 24 matrix instructions whose pipe takes turns with
// the vector pipe).  gemm_split_pack_kernel cuts it once into the stages' own LDS image -- per (128-column tile, 16-k stage) three
// planes x two halves x 128 slots of 16 bytes = 768 slots, zeros beyond the matrix -- and the GEMM (IOB = 2) copies a stage with three
// 16-byte loads and three 16-byte LDS stores per thread, no vector arithmetic, no masks.  Same pieces, same products: the same bits.
#define SPLIT_IMG_SLOTS 768            /* 16-byte slots per (tile, stage) */
typedef unsigned gemm_u32x4 __attribute__((vector_size(16)));
__global__ void __launch_bounds__(256)
gemm_split_pack_kernel(const float *W, int n, int k, int ld, int trans, gemm_u32x4 *image) {
    const int st = blockIdx.x, jt = blockIdx.y, nst = gridDim.x;
    const int h = threadIdx.x >> 7, slot = threadIdx.x & 127;
    const int row = jt * 128 + slot, k8 = st * 16 + 8 * h;
    float v[8];
    for (int i = 0; i < 8; ++i) {
        const int kk = k8 + i;
        const bool ok = row < n && kk < k;
        const long idx = trans ? (long)(ok ? kk : 0) * ld + (ok ? row : 0) : (long)(ok ? row : 0) * ld + (ok ? kk : 0);
        const float x = W[idx];
        v[i] = ok ? x : 0.0f;
    }
    unsigned p1[4], p2[4], p3[4];
    for (int w = 0; w < 4; ++w) split_pair(v[2 * w], v[2 * w + 1], p1[w], p2[w], p3[w]);
    gemm_u32x4 *dst = image + ((long)jt * nst + st) * SPLIT_IMG_SLOTS + h * 128 + slot;
    dst[0] = gemm_u32x4{p1[0], p1[1], p1[2], p1[3]};
    dst[256] = gemm_u32x4{p2[0], p2[1], p2[2], p2[3]};
    dst[512] = gemm_u32x4{p3[0], p3[1], p3[2], p3[3]};
}
// a stage of the image for a ROWS-wide tile (ROWS = 64: one half of the 128 slots): slots per thread
constexpr int split_img_nv(int rows) { return (6 * rows + 255) / 256; }
template <int ROWS>
__device__ __forceinline__ void split_img_fetch(gemm_u32x4 (&v)[split_img_nv(ROWS)], const gemm_u32x4 *image, int n0, int nst, int k0, int tid) {
    int st = k0 >> 4;
    st = st < nst ? st : nst - 1;                              // (the stage fetched past the end: any valid one, it is never multiplied)
    const gemm_u32x4 *src = image + ((long)(n0 >> 7) * nst + st) * SPLIT_IMG_SLOTS + (ROWS == 64 ? (n0 & 64) : 0);
#pragma unroll
    for (int e = 0; e < split_img_nv(ROWS); ++e) {
        int q = tid + 256 * e;
        if (6 * ROWS % 256 != 0 && q >= 6 * ROWS) q = 6 * ROWS - 1;
        const int ph = q / ROWS, sl = q - ph * ROWS;           // ph = plane * 2 + half
        v[e] = src[ph * 128 + sl];
    }
}
template <int ROWS>
__device__ __forceinline__ void split_img_stash(const gemm_u32x4 (&v)[split_img_nv(ROWS)], unsigned *S, int tid) {
    constexpr int HW = split_half_words(ROWS), PW = split_plane_words(ROWS);
#pragma unroll
    for (int e = 0; e < split_img_nv(ROWS); ++e) {
        const int q = tid + 256 * e;
        if (6 * ROWS % 256 == 0 || q < 6 * ROWS) {
            const int ph = q / ROWS, sl = q - ph * ROWS;
            *(gemm_u32x4 *)(S + (ph >> 1) * PW + (ph & 1) * HW + sl * 4) = v[e];
        }
    }
}

// Tile shapes: <2,2,2,2> = 128x128 (4 waves as 2x2, each 2x2 MFMA tiles) for the projections / FFN / score products;
// <4,1,1,1> = 128x32 (4 waves stacked in M, one MFMA tile each) for the products whose N is the head dim (32):
// P.V, dQ, dK, dV -- a 128-wide tile would waste 3/4 of its MFMAs there.
//
// v_mfma_f32_32x32x2_f32 wants A[i = lane & 31][k = lane >> 5]: the two lane halves consume different k.  The k order
// inside a stage is free (both operands agree on it), so half h takes k in [h GBK/2, (h+1) GBK/2): a lane's fragment
// values per 32-row block are GBK/8 ds_read_b128 of one LDS row, and step j multiplies k = j (low half) with
// k = GBK/2 + j (high half).  GBK = 32 (one stage = 64 MFMAs per wave, ~4 k cycles) covers the global-load latency
// of the register prefetch for long reductions; GBK = 16 keeps LDS small (more workgroups per CU) for K <= 256.
// PREC = 1 (opt-in, EMLOCO_GEMM_BF16): the same tiles, loads and LDS stages, but the fp32 fragments are rounded to bf16
// (v_cvt_pk_bf16_f32, round to nearest even) on their way into v_mfma_f32_32x32x16_bf16 -- one matrix instruction per 16 k
// instead of eight, fp32 accumulation.  Lane (l & 31, h = l >> 5) supplies 8 consecutive k of its half, which is what
// two adjacent fp32 fragments already hold; A and B use the same k permutation, so no layout changes.

// EPI = 1: the fused backward epilogue (flags & 32) -- its own instantiations, so that its registers (a tile of mask values in
// flight) do not lower the occupancy of the plain variants.
// EPIO (EPI = 1 only): the mask and the output are bf16 in memory -- a template parameter, not a flag: a run-time dtype test inside
// the mask prefetch turns sixteen loads in flight into sixteen round trips (measured: 1.7 -> 5.4 ms per launch)
// (Measured and dropped in round 5, profiles/r05_ab_gemm_pf2.txt: the split mode with the global loads TWO stages ahead of the matrix
// instructions -- two register sets, the stash woven between the matrix instructions with sched_group_barrier -- costs 70 registers, i.e.
// two workgroups per CU instead of three, and is 5 % slower on the tall GEMMs, 15 % on the M = 2 048 ones.)
template <int V> struct gemm_int_c { static constexpr int value = V; };
// NPC (split mode, round 6): pieces per operand.  3: the six piece products above 2^-24 of a product (fp32 class).  2: the three above
// 2^-16 -- for the GRADIENT products of the backward pass (EMLOCO_GEMM_SPLIT2): half the matrix instructions, a shorter cut.
template <int WM, int WN, int TI, int TJ, int GBK, int TA, int TB, int VEC, int PREC, int EPI, int IOA = 0, int IOB = 0, int EPIO = 0, int NPC = 3>
__device__ __forceinline__ void gemm_body(const GemmArgs &g) {
    constexpr int BM = WM * TI * 32, BN = WN * TJ * 32, GLDK = GBK + 4, NF = GBK / 8;
    constexpr int QA = (PREC == 2 && TA) ? 2 : (BM * GBK / 4 + 255) / 256, QB = (PREC == 2 && TB) ? 2 : (BN * GBK / 4 + 255) / 256;
    static_assert(PREC != 2 || (GBK == 16 && (BM == 128 || BM == 64) && (BN == 128 || BN == 64) && VEC == 1 && IOA == 0 && (IOB == 0 || IOB == 2)),
                  "split mode: 128 / 64-row x 16 stages, fp32 operands (or B as its piece image), 16-byte loads");
    static_assert(IOB != 2 || PREC == 2, "the piece image is the split mode's");
#ifndef GEMM_SPLIT2_PLANES
#define GEMM_SPLIT2_PLANES 3                                  /* LDS planes allocated per stage by the two-piece kernels (2: a fourth workgroup per CU fits --
                                                                 measured with GEMM_SPLIT2_WAVES = 4, 128 registers: JTA step 117 ms against 110, HISTORY.md) */
#endif
    constexpr int NPL = NPC == 2 ? GEMM_SPLIT2_PLANES : 3;
    constexpr int SA = PREC == 2 ? NPL * split_plane_words(BM) : BM * GLDK, SB = PREC == 2 ? NPL * split_plane_words(BN) : BN * GLDK;
    __shared__ __attribute__((aligned(16))) float As[2][SA], Bs[2][SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    // (An XCD-aware tile order -- XCD c walks the column tiles of the rows = c (mod 8) back to back so that a slab of A is
    // fetched by one L2 only -- was measured and is SLOWER here (927744 x 1024 x 128: 2.80 -> 3.13 ms): in dispatch order every
    // XCD keeps ONE 64 KB column tile of B hot and the eight readers of a slab of A arrive together and hit the shared L3.)
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int b = bz / g.ksplit, split = bz - b * g.ksplit;
    const int m0 = by * BM, n0 = bx * BN;
    int kchunk = (g.k + g.ksplit - 1) / g.ksplit;
    kchunk = ((kchunk + GBK - 1) / GBK) * GBK;
    const int kb = split * kchunk;
    const int ke = (kb + kchunk < g.k) ? kb + kchunk : g.k;
    // (a bf16 operand is addressed in 2-byte elements: the batch stride is applied on that type)
    const float *A = IOA ? (const float *)((const unsigned short *)g.A + (long)b * g.sa) : g.A + (long)b * g.sa;
    const float *B = IOB == 2 ? g.B : (IOB ? (const float *)((const unsigned short *)g.B + (long)b * g.sb) : g.B + (long)b * g.sb);
    const int img_nst = (g.k + 15) >> 4;                       // stages of the piece image (IOB = 2)

    f32x16 acc[TI][TJ];
    for (int i = 0; i < TI; ++i)
        for (int j = 0; j < TJ; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // The main loop twice (round 5): MK = 0 for a tile that lies inside both operands with a k range of whole stages (every tile of
    // the predictor's tall GEMMs but the last row of tiles) -- no clamps, no masks, no selects: 107 instead of 132 vector instructions
    // per stage of the split mode beside its 24 matrix instructions (the two pipes take turns on a SIMD: profiles/r05_gemm_split_counters.txt).
    const bool inside = VEC && m0 + BM <= g.m && (IOB == 2 || n0 + BN <= g.n) && ke > kb && (ke - kb) % GBK == 0;
    auto main_loop = [&](auto mk_c) {
    constexpr int MK = decltype(mk_c)::value;
    f32x4 ra[QA], rb[QB];
    unsigned ma[QA], mb[QB];
    gemm_u32x4 rbi[split_img_nv(BN)];                              // (IOB = 2: a stage of B's piece image)
    // fetch / stash of one stage in the mode's own LDS image
#define GEMM_FETCH(K0)                                                                                          \
    if constexpr (PREC == 2 && TA) split_fetch_trans<BM, MK>(ra, ma, A, g.lda, m0, g.m, (K0), ke, tid);         \
    else gemm_fetch<BM, GBK, TA, VEC, IOA, MK>(ra, ma, A, g.lda, m0, g.m, (K0), ke, tid);                       \
    if constexpr (IOB == 2) split_img_fetch<BN>(rbi, (const gemm_u32x4 *)B, n0, img_nst, (K0), tid);                 \
    else if constexpr (PREC == 2 && TB) split_fetch_trans<BN, MK>(rb, mb, B, g.ldb, n0, g.n, (K0), ke, tid);    \
    else gemm_fetch<BN, GBK, TB, VEC, (IOB == 2 ? 0 : IOB), MK>(rb, mb, B, g.ldb, n0, g.n, (K0), ke, tid);
#define GEMM_STASH(BUF)                                                                                         \
    if constexpr (PREC == 2) {                                                                                  \
        split_stash<BM, TA, NPC>(ra, ma, (unsigned *)As[BUF], tid);                                             \
        if constexpr (IOB == 2) split_img_stash<BN>(rbi, (unsigned *)Bs[BUF], tid);                             \
        else split_stash<BN, TB, NPC>(rb, mb, (unsigned *)Bs[BUF], tid);                                        \
    } else {                                                                                                    \
        gemm_stash<BM, GBK, TA>(ra, ma, As[BUF], tid);                                                          \
        gemm_stash<BN, GBK, TB>(rb, mb, Bs[BUF], tid);                                                          \
    }
    const int half8 = (GBK / 2) * (lane >> 5), l31 = lane & 31;
    if (kb < ke) {
        GEMM_FETCH(kb)
        GEMM_STASH(0)
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = kb; k0 < ke; k0 += GBK) {
        // next stage's global loads are in flight during the MFMAs (past the end they fetch zeros: clamped + masked)
        GEMM_FETCH(k0 + GBK)
        // keep the loads HERE: left free, the scheduler sinks them below most of the stage's MFMAs (they are only consumed by the
        // stash) and the s_waitcnt in front of the stash then exposes the whole memory latency every stage
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PREC == 2) {
            bf16w4 sa[TI][3], sb[TJ][3];
            static_assert(NPC == 3 || IOB != 2, "the piece image serves the three-piece products");
            for (int i = 0; i < TI; ++i) split_frag<BM, TA, NPC>((const unsigned *)As[buf], (wm * TI + i) * 32 + l31, lane >> 5, sa[i]);
            for (int j = 0; j < TJ; ++j) split_frag<BN, (IOB == 2 ? 0 : TB), NPC>((const unsigned *)Bs[buf], (wn * TJ + j) * 32 + l31, lane >> 5, sb[j]);
            // the six products, smallest first; term-outer so that consecutive matrix instructions go to different accumulators
#define SPLIT_TERM(PA, PB)                                                                                      \
            for (int i = 0; i < TI; ++i)                                                                        \
                for (int j = 0; j < TJ; ++j) acc[i][j] = gemm_mfma_bf16_w(sa[i][PA], sb[j][PB], acc[i][j]);
            if constexpr (NPC == 3) { SPLIT_TERM(2, 0) SPLIT_TERM(0, 2) SPLIT_TERM(1, 1) }
            SPLIT_TERM(1, 0) SPLIT_TERM(0, 1) SPLIT_TERM(0, 0)
#undef SPLIT_TERM
        } else {
        f32x4 fa[TI][NF], fb[TJ][NF];
        for (int i = 0; i < TI; ++i) gemm_frag<BM, GBK, TA>(As[buf], (wm * TI + i) * 32 + l31, half8, fa[i]);
        for (int j = 0; j < TJ; ++j) gemm_frag<BN, GBK, TB>(Bs[buf], (wn * TJ + j) * 32 + l31, half8, fb[j]);
#define GEMM_STEP(H, C)                                                                                          \
        for (int i = 0; i < TI; ++i)                                                                             \
            for (int j = 0; j < TJ; ++j)                                                                         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][H].C, fb[j][H].C, acc[i][j], 0, 0, 0);
        if constexpr (PREC == 0) {
            for (int f = 0; f < NF; ++f) { GEMM_STEP(f, x) GEMM_STEP(f, y) GEMM_STEP(f, z) GEMM_STEP(f, w) }
        } else {
            static_assert(PREC != 1 || NF % 2 == 0, "bf16 operands take k in groups of 16");
            for (int f = 0; f + 1 < NF; f += 2) {
                gemm_bf16x8 pa[TI], pb[TJ];
                for (int i = 0; i < TI; ++i) pa[i] = gemm_pack_bf16(fa[i][f], fa[i][f + 1]);
                for (int j = 0; j < TJ; ++j) pb[j] = gemm_pack_bf16(fb[j][f], fb[j][f + 1]);
                for (int i = 0; i < TI; ++i)
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = gemm_mfma_bf16(pa[i], pb[j], acc[i][j]);
            }
        }
#undef GEMM_STEP
        }
        __builtin_amdgcn_sched_barrier(0);
        GEMM_STASH(buf ^ 1)
        __syncthreads();
        buf ^= 1;
    }
#undef GEMM_FETCH
#undef GEMM_STASH
    };
    if constexpr (VEC) {
        if (inside) main_loop(gemm_int_c<0>{});
        else main_loop(gemm_int_c<1>{});
    } else main_loop(gemm_int_c<1>{});
    // epilogue: C/D fragment layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  A lane owns ONE column per (j): its
    // bias is read once (left inside the row loop, every store to C makes the compiler reload it: C may alias the bias).
    // Addresses are a workgroup-uniform tile base (scalar registers) + a 32-bit lane offset: one VGPR per access instead of a
    // 64-bit multiply-add and a register pair each.
    const bool has_bias = (g.flags & 1) != 0, relu = (g.flags & 2) != 0, accum = (g.flags & 4) != 0, drop = (g.flags & 8) != 0;
    const float keep_scale = drop ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    const int mrem = g.m - m0, nrem = g.n - n0;
    if (g.ksplit > 1) {
        float *wt = g.ws + (((long)split * g.batch + b) * g.m + m0) * g.n + n0;
        for (int j = 0; j < TJ; ++j) {
            const int cl = (wn * TJ + j) * 32 + (lane & 31);
            for (int i = 0; i < TI; ++i)
                for (int r = 0; r < 16; ++r) {
                    const int rl = (wm * TI + i) * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    if (rl < mrem && cl < nrem) wt[rl * g.n + cl] = g.alpha * acc[i][j][r];
                }
        }
    } else if (EPI == 0 && g.vec_c && 2 * SA >= 4 * 1024 && mrem >= BM && nrem >= BN) {
        // Wide stores (round 5).  A lane of the accumulator layout owns ONE column and 16 scattered rows of a 32 x 32 tile: written
        // straight from the registers a tile costs 16 four-byte store instructions per lane (two 128-byte row segments each) -- 64 per
        // lane for the workgroup's 128 x 128 tile, and with K = 128 (eight stages) that store tail was most of a tile's life: the
        // feed-forward's first layer (927 744 x 1024 x 128) ran at 82 TFLOP/s, 1.3 TB/s of output (profiles/r05_jta_launches_*).  Here
        // the epilogue's arithmetic stays on the accumulator registers (bias per lane = per column), the tile then crosses 4 KB of the
        // wave's own LDS (the operand stages are dead) and leaves as four 16-byte stores per lane (two for a bf16 output), whole lines.
        __syncthreads();                                     // (every wave is past its last fragment read of the stages)
        const int c16 = g.c16;
        float *scr = &As[0][0] + wave * 1024;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int cl = (wn * TJ + j) * 32 + (lane & 31);
            const float bj = has_bias ? g.bias[n0 + cl] : 0.0f;
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int r0 = (wm * TI + i) * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rt = 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    float v = g.alpha * acc[i][j][r] + bj;
                    if (relu) v = v > 0.0f ? v : 0.0f;
                    scr[rt * 32 + (lane & 31)] = v;
                }
                gemm_wave_sync();
                const int c0 = (wn * TJ + j) * 32;
                // The dropout mask is applied BEHIND the transposition, where a lane holds 4 (8) consecutive elements of a row: one hash
                // per PAIR of elements, as the mask is defined, and one index per lane and store -- in the accumulator layout the two
                // elements of a pair sit in neighbouring lanes and each lane evaluated the pair's hash for its own half (measured:
                // the mask was 0.37 of the 2.11 ms of the feed-forward's first layer, profiles/r05_gemm_k128_probe.txt).  Same mask.
                const unsigned dthr = (unsigned)(g.drop_p * 65536.0f);
                if (c16) {                                       // 32 bf16 = 64 bytes per row: 4 lanes x 16 bytes, 16 rows per instruction
                    unsigned short *cb = (unsigned short *)g.C + (long)b * g.sc + (long)(m0 + r0) * g.ldc + n0 + c0;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int rt = 16 * p + (lane >> 2), q = lane & 3;
                        const f32x4 t0 = *(const f32x4 *)(scr + rt * 32 + 8 * q), t1 = *(const f32x4 *)(scr + rt * 32 + 8 * q + 4);
                        unsigned short *dst = cb + (long)rt * g.ldc + 8 * q;
                        float e[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                        if (drop) {
                            const unsigned long long pr = ((((unsigned long long)b * g.m + m0 + r0 + rt) * g.n + n0 + c0) >> 1) + 4 * q;
#pragma unroll
                            for (int h2 = 0; h2 < 4; ++h2) {
                                const unsigned x = drop_pair_hash(g.drop_seed, pr + h2);
                                e[2 * h2] = (x & 0xffffu) >= dthr ? e[2 * h2] * keep_scale : 0.0f;
                                e[2 * h2 + 1] = (x >> 16) >= dthr ? e[2 * h2 + 1] * keep_scale : 0.0f;
                            }
                        }
                        if (accum) { const uint4 o = *(const uint4 *)dst; e[0] += bf16_lo(o.x); e[1] += bf16_hi(o.x); e[2] += bf16_lo(o.y); e[3] += bf16_hi(o.y);
                                     e[4] += bf16_lo(o.z); e[5] += bf16_hi(o.z); e[6] += bf16_lo(o.w); e[7] += bf16_hi(o.w); }
                        uint4 w;
                        w.x = (unsigned)f32_to_bf16(e[0]) | ((unsigned)f32_to_bf16(e[1]) << 16); w.y = (unsigned)f32_to_bf16(e[2]) | ((unsigned)f32_to_bf16(e[3]) << 16);
                        w.z = (unsigned)f32_to_bf16(e[4]) | ((unsigned)f32_to_bf16(e[5]) << 16); w.w = (unsigned)f32_to_bf16(e[6]) | ((unsigned)f32_to_bf16(e[7]) << 16);
                        *(uint4 *)dst = w;
                    }
                } else {                                         // 32 floats = 128 bytes per row: 8 lanes x 16 bytes, 8 rows per instruction
                    float *cf = g.C + (long)b * g.sc + (long)(m0 + r0) * g.ldc + n0 + c0;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int rt = 8 * p + (lane >> 3), q = lane & 7;
                        f32x4 t = *(const f32x4 *)(scr + rt * 32 + 4 * q);
                        float *dst = cf + (long)rt * g.ldc + 4 * q;
                        if (drop) {
                            const unsigned long long pr = ((((unsigned long long)b * g.m + m0 + r0 + rt) * g.n + n0 + c0) >> 1) + 2 * q;
                            const unsigned x0 = drop_pair_hash(g.drop_seed, pr), x1 = drop_pair_hash(g.drop_seed, pr + 1);
                            t.x = (x0 & 0xffffu) >= dthr ? t.x * keep_scale : 0.0f;
                            t.y = (x0 >> 16) >= dthr ? t.y * keep_scale : 0.0f;
                            t.z = (x1 & 0xffffu) >= dthr ? t.z * keep_scale : 0.0f;
                            t.w = (x1 >> 16) >= dthr ? t.w * keep_scale : 0.0f;
                        }
                        if (accum) { const f32x4 o = *(const f32x4 *)dst; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
                        *(f32x4 *)dst = t;
                    }
                }
                gemm_wave_sync();
            }
        }
    } else if constexpr (EPI == 0) {
        const int c16 = g.c16;
        float *ct = c16 ? (float *)((unsigned short *)g.C + (long)b * g.sc + (long)m0 * g.ldc + n0) : g.C + (long)b * g.sc + (long)m0 * g.ldc + n0;
        for (int j = 0; j < TJ; ++j) {
            const int cl = (wn * TJ + j) * 32 + (lane & 31);
            if (cl >= nrem) continue;
            const float bj = has_bias ? g.bias[n0 + cl] : 0.0f;
            for (int i = 0; i < TI; ++i)
                for (int r = 0; r < 16; ++r) {
                    const int rl = (wm * TI + i) * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    if (rl < mrem) {
                        float v = g.alpha * acc[i][j][r] + bj;
                        if (relu) v = v > 0.0f ? v : 0.0f;
                        if (drop) v = drop_keep(g.drop_seed, ((unsigned long long)b * g.m + m0 + rl) * g.n + n0 + cl, g.drop_p) ? v * keep_scale : 0.0f;
                        const int ci = rl * g.ldc + cl;
                        if (accum) v += ld_act(ct, ci, c16);
                        st_act(ct, ci, v, c16);
                    }
                }
        }
    } else if (EPIO == 0 && g.vec_c && 2 * SA >= 8 * 1024 && mrem >= BM && nrem >= BN) {
        // the fused backward epilogue with wide accesses (round 5, fp32 mask and output): the forward output's 32 x 32 tile comes in as
        // whole lines (four 16-byte loads per lane), crosses the wave's LDS into the accumulator layout (where the column sums are a
        // lane's own registers), and the masked gradient leaves like the plain epilogue's tile, as whole lines
        __syncthreads();
        float *scr = &As[0][0] + wave * 2048, *msk = scr + 1024;
        for (int j = 0; j < TJ; ++j) {
            const int c0 = (wn * TJ + j) * 32;
            float csum = 0.0f;
            for (int i = 0; i < TI; ++i) {
                const int r0 = (wm * TI + i) * 32;
                const long tile = (long)b * g.sc + (long)(m0 + r0) * g.ldc + n0 + c0;
                for (int p = 0; p < 4; ++p) {
                    const int rt = 8 * p + (lane >> 3), q = lane & 7;
                    *(f32x4 *)(msk + rt * 32 + 4 * q) = *(const f32x4 *)(g.mask + tile + (long)rt * g.ldc + 4 * q);
                }
                gemm_wave_sync();
                for (int r = 0; r < 16; ++r) {
                    const int rt = 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    const float v = msk[rt * 32 + (lane & 31)] > 0.0f ? g.alpha * acc[i][j][r] * g.mask_scale : 0.0f;
                    scr[rt * 32 + (lane & 31)] = v;
                    csum += v;
                }
                gemm_wave_sync();
                for (int p = 0; p < 4; ++p) {
                    const int rt = 8 * p + (lane >> 3), q = lane & 7;
                    *(f32x4 *)(g.C + tile + (long)rt * g.ldc + 4 * q) = *(const f32x4 *)(scr + rt * 32 + 4 * q);
                }
                gemm_wave_sync();
            }
            csum += __shfl_xor(csum, 32);                    // the other 32 rows of the wave's 64
            if (lane < 32) g.colpart[((long)by * WM + wm) * g.n + n0 + c0 + lane] = csum;
        }
    } else {
        constexpr int c16 = EPIO, m16 = EPIO;
        const float *mt = m16 ? (const float *)((const unsigned short *)g.mask + (long)b * g.sc + (long)m0 * g.ldc + n0) : g.mask + (long)b * g.sc + (long)m0 * g.ldc + n0;
        float *ct = c16 ? (float *)((unsigned short *)g.C + (long)b * g.sc + (long)m0 * g.ldc + n0) : g.C + (long)b * g.sc + (long)m0 * g.ldc + n0;
        for (int j = 0; j < TJ; ++j) {
            const int cl = (wn * TJ + j) * 32 + (lane & 31);
            const bool colok = cl < nrem;
            const int clc = colok ? cl : nrem - 1;
            float csum = 0.0f;
            for (int i = 0; i < TI; ++i) {
                // the 16 mask values of the fragment are requested before the first is looked at (one at a time the epilogue
                // is a chain of memory latencies)
                float mv[16];
                for (int r = 0; r < 16; ++r) {
                    const int rl = (wm * TI + i) * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    if constexpr (m16) mv[r] = bf16_lo(((const unsigned short *)mt)[(rl < mrem ? rl : mrem - 1) * g.ldc + clc]);
                    else mv[r] = mt[(rl < mrem ? rl : mrem - 1) * g.ldc + clc];
                }
                for (int r = 0; r < 16; ++r) {
                    const int rl = (wm * TI + i) * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                    if (rl < mrem && colok) {
                        float v = mv[r] > 0.0f ? g.alpha * acc[i][j][r] * g.mask_scale : 0.0f;
                        if constexpr (c16) {                               // rounded once; the bias gradient sums what the next GEMMs will read
                            const unsigned short hb = f32_to_bf16(v);
                            ((unsigned short *)ct)[rl * g.ldc + cl] = hb;
                            v = bf16_lo(hb);
                        } else {
                            ct[rl * g.ldc + cl] = v;
                        }
                        csum += v;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);           // one fragment's mask values in flight, not the whole tile's
            }
            csum += __shfl_xor(csum, 32);                    // the other 32 rows of the wave's 64
            if (lane < 32 && colok) g.colpart[((long)by * WM + wm) * g.n + n0 + cl] = csum;
        }
    }
}

template <int WM, int WN, int TI, int TJ, int GBK, int TA, int TB, int VEC, int PREC = 0, int IOA = 0, int IOB = 0>
__global__ void __launch_bounds__(256)
gemm_f32_kernel(GemmArgs g) { gemm_body<WM, WN, TI, TJ, GBK, TA, TB, VEC, PREC, 0, IOA, IOB>(g); }

// the fused-backward-epilogue variants: held to 3 waves per SIMD (168 registers; left alone the compiler keeps the whole
// tile's mask values and offsets live and falls to 2)
template <int GBK, int TB, int PREC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
gemm_f32_relu_bwd_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, GBK, 0, TB, 1, PREC, 1, 0, 0, 0>(g); }
// bf16 hidden layer and gradient
template <int GBK, int TB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
gemm_bf16_relu_bwd_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, GBK, 0, TB, 1, 1, 1, 0, 0, 1>(g); }
// the 32-deep stages of the two (long reductions on small grids): their LDS stages hold them to 2 waves per SIMD whatever is asked for
template <int TB, int PREC>
__global__ void __launch_bounds__(256)
gemm_f32_relu_bwd_deep_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, 32, 0, TB, 1, PREC, 1, 0, 0, 0>(g); }
template <int TB>
__global__ void __launch_bounds__(256)
gemm_bf16_relu_bwd_deep_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, 32, 0, TB, 1, 1, 1, 0, 0, 1>(g); }

// split mode: three workgroups per CU (50.7 KB of LDS each)
template <int TA, int TB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
gemm_split_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, 16, TA, TB, 1, 2, 0>(g); }
template <int TB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
gemm_split_relu_bwd_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, 16, 0, TB, 1, 2, 1, 0, 0, 0>(g); }
// two pieces per operand (EMLOCO_GEMM_SPLIT2, round 6): the gradient products of the backward pass
#ifndef GEMM_SPLIT2_WAVES
#define GEMM_SPLIT2_WAVES 3
#endif
template <int TA, int TB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GEMM_SPLIT2_WAVES, GEMM_SPLIT2_WAVES)))
gemm_split2_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, 16, TA, TB, 1, 2, 0, 0, 0, 0, 2>(g); }
template <int TB>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GEMM_SPLIT2_WAVES, GEMM_SPLIT2_WAVES)))
gemm_split2_relu_bwd_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, 16, 0, TB, 1, 2, 1, 0, 0, 0, 2>(g); }
template <int TA, int TB>
__global__ void __launch_bounds__(256)
gemm_split2_small_kernel(GemmArgs g) { gemm_body<2, 2, 1, 1, 16, TA, TB, 1, 2, 0, 0, 0, 0, 2>(g); }
// B as the piece image of a weight (IOB = 2; A row-major): the forward and input-gradient GEMMs of the linear layers
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
gemm_split_img_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, 16, 0, 0, 1, 2, 0, 0, 2>(g); }
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
gemm_split_relu_bwd_img_kernel(GemmArgs g) { gemm_body<2, 2, 2, 2, 16, 0, 0, 1, 2, 1, 0, 2, 0>(g); }
__global__ void __launch_bounds__(256)
gemm_split_small_img_kernel(GemmArgs g) { gemm_body<2, 2, 1, 1, 16, 0, 0, 1, 2, 0, 0, 2>(g); }

// split mode on a 64 x 64 tile (four waves, one matrix tile each; 26 KB of LDS: up to six workgroups per CU) for launches that
// are too small to fill the chip with 128 x 128 tiles -- the learners' and the policy's M = 2 048 .. 4 096 layers: 128 .. 512
// workgroups on 256 CUs leave every SIMD with one wave and nothing to hide a load, a barrier or a matrix chain behind.
template <int TA, int TB>
__global__ void __launch_bounds__(256)
gemm_split_small_kernel(GemmArgs g) { gemm_body<2, 2, 1, 1, 16, TA, TB, 1, 2, 0>(g); }

// kernel variant for a problem: tile shape (narrow: n <= 32), stage depth, operand layouts, 16-byte loads
typedef void (*GemmKernel)(GemmArgs);
template <int WM, int WN, int TI, int TJ, int GBK>
inline GemmKernel gemm_pick_layout(int ta, int tb, int vec, int bf16 = 0, int a16 = 0, int b16 = 0) {
    if (vec && bf16 && a16 && !b16) {       // A (an activation) is bf16 in memory
        if (!ta && !tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 0, 0, 1, 1, 1, 0>;
        if (!ta && tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 0, 1, 1, 1, 1, 0>;
        if (ta && !tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 0, 1, 1, 1, 0>;
        return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 1, 1, 1, 1, 0>;
    }
    if (vec && bf16 && !a16 && b16) {       // B is bf16 in memory (weight gradient against a bf16 activation)
        if (ta && tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 1, 1, 1, 0, 1>;
        return nullptr;
    }
    if (vec && bf16 && a16 && b16) {
        if (ta && tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 1, 1, 1, 1, 1>;
        return nullptr;
    }
    if (vec && bf16) {       // bf16 operands: the 16-byte-load variants only (every hot shape qualifies; others stay fp32)
        if (!ta && !tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 0, 0, 1, 1>;
        if (!ta && tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 0, 1, 1, 1>;
        if (ta && !tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 0, 1, 1>;
        return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 1, 1, 1>;
    }
    if (vec) {
        if (!ta && !tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 0, 0, 1>;
        if (!ta && tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 0, 1, 1>;
        if (ta && !tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 0, 1>;
        return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 1, 1>;
    }
    if (!ta && !tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 0, 0, 0>;
    if (!ta && tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 0, 1, 0>;
    if (ta && !tb) return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 0, 0>;
    return gemm_f32_kernel<WM, WN, TI, TJ, GBK, 1, 1, 0>;
}
inline GemmKernel gemm_pick(const GemmArgs &g, bool deep) {
    const int vec = g.vec_a && g.vec_b;
    const int bf16 = (g.flags & 16) ? 1 : 0;
    const bool two = (g.flags & 4096) && !g.bimg;              // EMLOCO_GEMM_SPLIT2 (a piece image holds three pieces: it wins)
    if ((g.flags & 32) && (g.flags & 1024) && !bf16 && !g.c16 && !g.m16 && two) return g.tb ? gemm_split2_relu_bwd_kernel<1> : gemm_split2_relu_bwd_kernel<0>;
    if ((g.flags & 32) && (g.flags & 1024) && !bf16 && !g.c16 && !g.m16 && g.bimg) return gemm_split_relu_bwd_img_kernel;
    if ((g.flags & 32) && (g.flags & 1024) && !bf16 && !g.c16 && !g.m16) return g.tb ? gemm_split_relu_bwd_kernel<1> : gemm_split_relu_bwd_kernel<0>;
    if (g.flags & 32) {           // fused backward epilogue: A row-major, 16-byte loads (the launcher checks)
        if (g.c16 || g.m16) {     // bf16 hidden layer and gradient (both or neither: the launcher checks)
            if (deep) return g.tb ? gemm_bf16_relu_bwd_deep_kernel<1> : gemm_bf16_relu_bwd_deep_kernel<0>;
            return g.tb ? gemm_bf16_relu_bwd_kernel<16, 1> : gemm_bf16_relu_bwd_kernel<16, 0>;
        }
        if (deep) {
            if (bf16) return g.tb ? gemm_f32_relu_bwd_deep_kernel<1, 1> : gemm_f32_relu_bwd_deep_kernel<0, 1>;
            return g.tb ? gemm_f32_relu_bwd_deep_kernel<1, 0> : gemm_f32_relu_bwd_deep_kernel<0, 0>;
        }
        if (bf16) return g.tb ? gemm_f32_relu_bwd_kernel<16, 1, 1> : gemm_f32_relu_bwd_kernel<16, 0, 1>;
        return g.tb ? gemm_f32_relu_bwd_kernel<16, 1, 0> : gemm_f32_relu_bwd_kernel<16, 0, 0>;
    }
    if (g.n <= 32) return deep ? gemm_pick_layout<4, 1, 1, 1, 32>(g.ta, g.tb, vec) : gemm_pick_layout<4, 1, 1, 1, 16>(g.ta, g.tb, vec);
    if ((g.flags & 1024) && !bf16 && g.bimg) return g.small ? gemm_split_small_img_kernel : gemm_split_img_kernel;
    if ((g.flags & 1024) && vec && !bf16 && two) {
        if (g.small) {
            if (!g.ta && !g.tb) return gemm_split2_small_kernel<0, 0>;
            if (!g.ta && g.tb) return gemm_split2_small_kernel<0, 1>;
            if (g.ta && !g.tb) return gemm_split2_small_kernel<1, 0>;
            return gemm_split2_small_kernel<1, 1>;
        }
        if (!g.ta && !g.tb) return gemm_split2_kernel<0, 0>;
        if (!g.ta && g.tb) return gemm_split2_kernel<0, 1>;
        if (g.ta && !g.tb) return gemm_split2_kernel<1, 0>;
        return gemm_split2_kernel<1, 1>;
    }
    if ((g.flags & 1024) && vec && !bf16 && g.small) {
        if (!g.ta && !g.tb) return gemm_split_small_kernel<0, 0>;
        if (!g.ta && g.tb) return gemm_split_small_kernel<0, 1>;
        if (g.ta && !g.tb) return gemm_split_small_kernel<1, 0>;
        return gemm_split_small_kernel<1, 1>;
    }
    if ((g.flags & 1024) && vec && !bf16) {        // split mode (fp32 class on the bf16 matrix rate): 16-byte-aligned operands, 128-wide tiles
        if (!g.ta && !g.tb) return gemm_split_kernel<0, 0>;
        if (!g.ta && g.tb) return gemm_split_kernel<0, 1>;
        if (g.ta && !g.tb) return gemm_split_kernel<1, 0>;
        return gemm_split_kernel<1, 1>;
    }
    return deep ? gemm_pick_layout<2, 2, 2, 2, 32>(g.ta, g.tb, vec, bf16, g.a16, g.b16) : gemm_pick_layout<2, 2, 2, 2, 16>(g.ta, g.tb, vec, bf16, g.a16, g.b16);
}

// sum of the ksplit partial products in a fixed order, then the epilogue
__global__ void gemm_splitk_reduce_kernel(GemmArgs g) {
    const long total = (long)g.batch * g.m * g.n;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // (round 6: eight partials in flight per thread -- one load per iteration made a launch a chain of ksplit memory latencies, 41 us for
    // the 25 MB of a weight gradient's 48 partial slabs; the additions keep their order, the bits are the same)
    float v = 0.0f;
    int s = 0;
    for (; s + 8 <= g.ksplit; s += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = g.ws[(long)(s + u) * total + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < g.ksplit; ++s) v += g.ws[(long)s * total + i];
    const int col = (int)(i % g.n);
    const long rowb = i / g.n;
    const int row = (int)(rowb % g.m);
    const int b = (int)(rowb / g.m);
    float *c = g.C + (long)b * g.sc + (long)row * g.ldc + col;
    if (g.flags & 1) v += g.bias[col];
    if (g.flags & 2) v = v > 0.0f ? v : 0.0f;
    if (g.flags & 8) v = drop_keep(g.drop_seed, (unsigned long long)i, g.drop_p) ? v * (1.0f / (1.0f - g.drop_p)) : 0.0f;
    if (g.flags & 4) v += *c;
    *c = v;
}

// Backward of the fused epilogue: dz = dy * [relu: y > 0] * [dropout: keep / (1 - p)] in one pass (y is the forward OUTPUT:
// after ReLU + dropout a positive output means "active and kept", so the mask is only recomputed when there is no ReLU).
__global__ void act_bwd_kernel(long total, const float *dy, const float *y, int relu, float drop_p, unsigned drop_seed, float *dz) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float v = dy[i];
    if (relu) v = y[i] > 0.0f ? v : 0.0f;
    if (drop_p > 0.0f) {
        const bool keep = relu ? true : drop_keep(drop_seed, (unsigned long long)i, drop_p);
        v = keep ? v * (1.0f / (1.0f - drop_p)) : 0.0f;
    }
    dz[i] = v;
}

// ------------------------------------------------------------------ softmax (one wave per row)
// A row of <= 1024 scores lives in 16 registers per lane: one read of S (+ key bias), one write of P -- the row is
// never re-read from HBM (the three-pass version moved 5 row-lengths; the score matrices are the largest tensors of
// the step).  Longer rows take the streaming path.
#define SM_REG 16
__global__ void __launch_bounds__(256)
softmax_fwd_kernel(int rows, int rows_per_seq, int cols, float scale, const float *S, const float *key_bias, float *P) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *s = S + row * cols;
    float *p = P + row * cols;
    const float *kb = key_bias ? key_bias + (row / rows_per_seq) * cols : nullptr;
    const float NEG = -3.0e38f;
    if (cols <= 64 * SM_REG) {
        float v[SM_REG];
        float mx = NEG;
        for (int t = 0; t < SM_REG; ++t) {
            const int j = lane + 64 * t;
            v[t] = NEG;
            if (j < cols) v[t] = s[j] * scale + (kb ? kb[j] : 0.0f);   // kb = -inf masks the key, a finite value biases it
            mx = v[t] > mx ? v[t] : mx;
        }
        for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
        float sum = 0.0f;
        for (int t = 0; t < SM_REG; ++t) {
            v[t] = (v[t] > NEG) ? expf(v[t] - mx) : 0.0f;
            sum += v[t];
        }
        sum = wave_sum(sum);
        const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;   // fully masked row -> zeros ("safe softmax")
        for (int t = 0; t < SM_REG; ++t) { const int j = lane + 64 * t; if (j < cols) p[j] = v[t] * inv; }
        return;
    }
    float mx = NEG;
    for (int j = lane; j < cols; j += 64) {
        const float v = s[j] * scale + (kb ? kb[j] : 0.0f);
        mx = v > mx ? v : mx;
    }
    for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
    float sum = 0.0f;
    for (int j = lane; j < cols; j += 64) {
        const float v = s[j] * scale + (kb ? kb[j] : 0.0f);
        const float e = (v > NEG) ? expf(v - mx) : 0.0f;
        p[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
    for (int j = lane; j < cols; j += 64) p[j] *= inv;
}

__global__ void __launch_bounds__(256)
softmax_bwd_kernel(int rows, int cols, float scale, const float *P, const float *dP, float *dS) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = P + row * cols, *dp = dP + row * cols;
    float *ds = dS + row * cols;
    if (cols <= 64 * SM_REG) {
        float pv[SM_REG], dv[SM_REG], dot = 0.0f;
        for (int t = 0; t < SM_REG; ++t) {
            const int j = lane + 64 * t;
            pv[t] = 0.0f; dv[t] = 0.0f;
            if (j < cols) { pv[t] = p[j]; dv[t] = dp[j]; }
            dot += dv[t] * pv[t];
        }
        dot = wave_sum(dot);
        for (int t = 0; t < SM_REG; ++t) { const int j = lane + 64 * t; if (j < cols) ds[j] = scale * pv[t] * (dv[t] - dot); }
        return;
    }
    float dot = 0.0f;
    for (int j = lane; j < cols; j += 64) dot += dp[j] * p[j];
    dot = wave_sum(dot);
    for (int j = lane; j < cols; j += 64) ds[j] = scale * p[j] * (dp[j] - dot);
}

// ------------------------------------------------------------------ layer norm (one wave per row, d <= 1024)
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(int rows, int d, float eps, const float *x, const float *res, const float *gamma, const float *beta,
                     float *y, float *mean, float *rstd, float *xr_out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[16];
    float s = 0.0f;
    for (int t = 0; t < 16; ++t) {
        const int j = lane + 64 * t;
        v[t] = 0.0f;
        if (j < d) {
            v[t] = x[row * d + j] + (res ? res[row * d + j] : 0.0f); s += v[t];
            if (xr_out) xr_out[row * d + j] = v[t];          // the sum the backward needs (saves a separate add pass)
        }
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.0f;
    for (int t = 0; t < 16; ++t) { const int j = lane + 64 * t; if (j < d) { const float c = v[t] - mu; q += c * c; } }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    for (int t = 0; t < 16; ++t) { const int j = lane + 64 * t; if (j < d) y[row * d + j] = (v[t] - mu) * rs * gamma[j] + beta[j]; }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dxr = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ; per-block partial dgamma / dbeta
#define LN_ROWS_PER_BLOCK 64
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(int rows, int d, const float *xr, const float *gamma, const float *mean, const float *rstd,
                     const float *dy, const float *dy2 /* optional: a second incoming gradient, added on load */, float *dxr,
                     float *part /* [nblocks][2][d] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ float sh_g[4][1024], sh_b[4][1024];
    for (int j = lane; j < d; j += 64) { sh_g[w][j] = 0.0f; sh_b[w][j] = 0.0f; }
    const long r0 = (long)blockIdx.x * LN_ROWS_PER_BLOCK;
    for (int rr = w; rr < LN_ROWS_PER_BLOCK; rr += 4) {
        const long row = r0 + rr;
        if (row >= rows) break;
        const float mu = mean[row], rs = rstd[row];
        float xh[16], gg[16], s1 = 0.0f, s2 = 0.0f;
        for (int t = 0; t < 16; ++t) {
            const int j = lane + 64 * t;
            xh[t] = 0.0f; gg[t] = 0.0f;
            if (j < d) {
                const float dyv = dy2 ? dy[row * d + j] + dy2[row * d + j] : dy[row * d + j];
                xh[t] = (xr[row * d + j] - mu) * rs;
                gg[t] = dyv * gamma[j];
                s1 += gg[t]; s2 += gg[t] * xh[t];
                sh_g[w][j] += dyv * xh[t];
                sh_b[w][j] += dyv;
            }
        }
        const float m1 = wave_sum(s1) / (float)d, m2 = wave_sum(s2) / (float)d;
        for (int t = 0; t < 16; ++t) { const int j = lane + 64 * t; if (j < d) dxr[row * d + j] = rs * (gg[t] - m1 - xh[t] * m2); }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += 256) {
        part[((long)blockIdx.x * 2 + 0) * d + j] = (sh_g[0][j] + sh_g[1][j]) + (sh_g[2][j] + sh_g[3][j]);
        part[((long)blockIdx.x * 2 + 1) * d + j] = (sh_b[0][j] + sh_b[1][j]) + (sh_b[2][j] + sh_b[3][j]);
    }
}

// The predictor's width (d = 128): a row is 32 lanes x 16 bytes, so a wave takes TWO rows at a time (one per 32-lane half) with one
// dwordx4 load per operand, the row sums stay inside a half, and the dgamma / dbeta partials are carried in registers over the wave's
// rows (the generic kernel above walks a row with 4-byte loads and keeps the partials in LDS: 2 read-modify-writes per element;
// measured on MI355X at 470 k rows: 263 -> see profiles/r04_jta_step_kernels.txt).  Same partial layout, same fold.
__device__ __forceinline__ float half_sum(float v) {          // sum over the lane's 32-lane half, in every lane of it
    v += dpp_mov<0x140>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v);      // the 16-lane row (as wave_sum)
    return v + __shfl_xor(v, 16);
}
__global__ void __launch_bounds__(256)
layernorm_fwd128_kernel(int rows, float eps, const float *x, const float *res, const float *gamma, const float *beta,
                        float *y, float *mean, float *rstd, float *xr_out) {
    const int lane = threadIdx.x & 63, l = lane & 31;
    const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool on = row < rows;
    const long rowc = on ? row : (long)rows - 1;
    f32x4 v = *(const f32x4 *)(x + rowc * 128 + 4 * l);
    if (res) { const f32x4 r = *(const f32x4 *)(res + rowc * 128 + 4 * l); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (xr_out && on) *(f32x4 *)(xr_out + row * 128 + 4 * l) = v;
    const float mu = half_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 128.0f);
    const float cx = v.x - mu, cy = v.y - mu, cz = v.z - mu, cw = v.w - mu;
    const float rs = 1.0f / sqrtf(half_sum((cx * cx + cy * cy) + (cz * cz + cw * cw)) * (1.0f / 128.0f) + eps);
    if (!on) return;
    const f32x4 g = *(const f32x4 *)(gamma + 4 * l), b = *(const f32x4 *)(beta + 4 * l);
    *(f32x4 *)(y + row * 128 + 4 * l) = f32x4{cx * rs * g.x + b.x, cy * rs * g.y + b.y, cz * rs * g.z + b.z, cw * rs * g.w + b.w};
    if (l == 0) { mean[row] = mu; rstd[row] = rs; }
}
__global__ void __launch_bounds__(256)
layernorm_bwd128_kernel(int rows, const float *xr, const float *gamma, const float *mean, const float *rstd,
                        const float *dy, const float *dy2 /* optional: added on load */, float *dxr, float *part /* [nblocks][2][128] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane >> 5, l = lane & 31;
    __shared__ float sh_g[8][128], sh_b[8][128];
    const f32x4 gm = *(const f32x4 *)(gamma + 4 * l);
    float ag[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ab[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const long r0 = (long)blockIdx.x * LN_ROWS_PER_BLOCK;
    for (int rr = 2 * w + half; rr < LN_ROWS_PER_BLOCK; rr += 8) {
        const long row = r0 + rr;
        const bool on = row < rows;                           // (both halves run the exchanges; a half past the end carries zeros)
        const long rowc = on ? row : (long)rows - 1;
        const float mu = mean[rowc], rs = rstd[rowc];
        f32x4 dv4 = *(const f32x4 *)(dy + rowc * 128 + 4 * l);
        const f32x4 xv4 = *(const f32x4 *)(xr + rowc * 128 + 4 * l);
        if (dy2) { const f32x4 e = *(const f32x4 *)(dy2 + rowc * 128 + 4 * l); dv4.x += e.x; dv4.y += e.y; dv4.z += e.z; dv4.w += e.w; }
        const float dv[4] = {dv4.x, dv4.y, dv4.z, dv4.w}, xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, g4[4] = {gm.x, gm.y, gm.z, gm.w};
        float xh[4], gg[4], s1 = 0.0f, s2 = 0.0f;
        for (int c = 0; c < 4; ++c) {
            xh[c] = (xv[c] - mu) * rs;
            gg[c] = dv[c] * g4[c];
            s1 += gg[c]; s2 += gg[c] * xh[c];
            if (on) { ag[c] += dv[c] * xh[c]; ab[c] += dv[c]; }
        }
        const float m1 = half_sum(s1) * (1.0f / 128.0f), m2 = half_sum(s2) * (1.0f / 128.0f);
        if (on) *(f32x4 *)(dxr + row * 128 + 4 * l) = f32x4{rs * (gg[0] - m1 - xh[0] * m2), rs * (gg[1] - m1 - xh[1] * m2),
                                                               rs * (gg[2] - m1 - xh[2] * m2), rs * (gg[3] - m1 - xh[3] * m2)};
    }
    for (int c = 0; c < 4; ++c) { sh_g[2 * w + half][4 * l + c] = ag[c]; sh_b[2 * w + half][4 * l + c] = ab[c]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int j = threadIdx.x;
        float sg = 0.0f, sb = 0.0f;
        for (int k = 0; k < 8; ++k) { sg += sh_g[k][j]; sb += sh_b[k][j]; }
        part[((long)blockIdx.x * 2 + 0) * 128 + j] = sg;
        part[((long)blockIdx.x * 2 + 1) * 128 + j] = sb;
    }
}

// Row folding: out[r][j] = sum of rows [32 r, 32 r + 32) of in[n][w] in ascending order.  Applied level by level
// (n -> n/32 -> ... -> 1) it reduces per-block partials with a fixed association order and full-chip parallelism
// (the earlier single-pass reduce walked 14 k partial rows with 128 threads: 3.5 ms).  The last level can split its
// row into two outputs (dgamma | dbeta).
#define FOLD 32      // (64 -- one level less on most reductions -- measured the same: tools/exp/ab_fold.sh)
__global__ void rows_fold_kernel(int n, int w, const float *in, float *out0, float *out1, int split) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= w) return;
    const long r = blockIdx.y, i0 = r * FOLD;
    // (round 6: all FOLD loads in flight, rows past the end read the last row and are not added -- the loop with its early exit was a chain of
    // 32 dependent latencies; same order of additions)
    float t[FOLD];
#pragma unroll
    for (int i = 0; i < FOLD; ++i) {
        const long row = i0 + i < n ? i0 + i : (long)n - 1;
        t[i] = in[row * w + j];
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < FOLD; ++i) s = (i0 + i < n) ? s + t[i] : s;
    if (j < split) out0[r * split + j] = s; else out1[r * (w - split) + (j - split)] = s;
}
inline long fold_workspace(long n, long w) { return (n + (n + FOLD - 1) / FOLD) * w; }
// LAUNCH(grid_x, grid_y, n, w, in, out0, out1, split) runs rows_fold_kernel; buf holds [n][w] and fold_workspace(n, w) floats
template <class LAUNCH>
inline void fold_rows(LAUNCH launch, int n, int w, float *buf, float *out0, float *out1, int split) {
    float *cur = buf, *alt = buf + (long)n * w;
    for (;;) {
        const int no = (n + FOLD - 1) / FOLD;
        const bool fin = no == 1;
        float *dst = cur == buf ? alt : buf;
        launch((unsigned)((w + 255) / 256), (unsigned)no, n, w, cur, fin ? out0 : dst, fin ? out1 : dst, fin ? split : w);
        if (fin) break;
        cur = dst; n = no;
    }
}

// column sums: cs_rows-row partials, then folded.  256 rows per partial for the predictor's tall activations (M ~ 10^6); 32 for the
// learners' minibatches (M = 2 048 .. 25 600), where a 256-row walk is a chain of 64 dependent load latencies per thread and 8
// workgroup rows leave most of the chip idle (measured: 27.6 us for a 16 MB pass).
#define CS_ROWS 256
__host__ __device__ inline int cs_rows_for(int m) { return m >= 32768 ? CS_ROWS : 32; }
__global__ void colsum_partial_kernel(int m, int n, const float *X, float *part, int x16, int cs_rows) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long r0 = (long)blockIdx.y * cs_rows;
    float s = 0.0f;
    for (int r = 0; r < cs_rows && r0 + r < m; ++r) s += ld_act(X, (r0 + r) * n + j, x16);
    part[(long)blockIdx.y * n + j] = s;
}

// the same for fp32 rows of a multiple of 4 columns (16-byte aligned): 256 threads = 64 column quads x 4 row phases, one dwordx4 load per
// thread and row; the four phase sums are added in a fixed order through LDS (the q|k|v gradient of the predictor, 470 k x 384: 200 us ->
// profiles/r04_jta_step_kernels.txt)
template <int X16>      // X16: the rows are bf16 in memory (8-byte quads)
__global__ void __launch_bounds__(256)
colsum4_partial_kernel(int m, int n, const float *X, float *part, int cs_rows) {
    __shared__ f32x4 sh[4][64];
    const int ql = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int j = (blockIdx.x * 64 + ql) * 4;
    const long r0 = (long)blockIdx.y * cs_rows;
    f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    if (j < n)
        for (int r = ph; r < cs_rows && r0 + r < m; r += 4) {
            f32x4 v;
            if constexpr (X16) {
                const uint2 u = *(const uint2 *)((const unsigned short *)X + (r0 + r) * n + j);
                v = f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
            } else v = *(const f32x4 *)(X + (r0 + r) * n + j);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    sh[ph][ql] = s;
    __syncthreads();
    if (ph == 0 && j < n) {
        const f32x4 a = sh[0][ql], b = sh[1][ql], c = sh[2][ql], d = sh[3][ql];
        *(f32x4 *)(part + (long)blockIdx.y * n + j) = f32x4{(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)};
    }
}

// act_bwd + bias gradient in one pass: thread j walks 256 rows of column j (coalesced across j), writes dz and leaves the
// column partial sum for the fixed-order row folding (replaces act_bwd_kernel + colsum_partial_kernel: one read of dz less).
__global__ void __launch_bounds__(256)
act_bwd_colsum_kernel(int m, int n, const float *dy, const float *y, int relu, float drop_p, unsigned drop_seed,
                      float *dz, float *part, int cs_rows) {
    // 256 threads = 64 columns x 4 row phases (a wave reads 64 consecutive floats of a row); the four phase sums of a column
    // are added in a fixed order through LDS, so narrow matrices (n = 128) still fill the chip.
    __shared__ float sh[4][64];
    const int jl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    const long r0 = (long)blockIdx.y * cs_rows;
    const float inv = drop_p > 0.0f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float s = 0.0f;
    if (j < n)
        for (int r = ph; r < cs_rows && r0 + r < m; r += 4) {
            const long i = (r0 + r) * n + j;
            float v = dy[i];
            if (relu) v = y[i] > 0.0f ? v : 0.0f;
            if (drop_p > 0.0f) v = (relu || drop_keep(drop_seed, (unsigned long long)i, drop_p)) ? v * inv : 0.0f;
            dz[i] = v;
            s += v;
        }
    sh[ph][jl] = s;
    __syncthreads();
    if (ph == 0 && j < n) part[(long)blockIdx.y * n + j] = (sh[0][jl] + sh[1][jl]) + (sh[2][jl] + sh[3][jl]);
}

// ------------------------------------------------------------------ gradients into the flat bucket (round 6)
// flat[off[t] ..] = src[t][..] for the tensors of one table (blockIdx.y = tensor); 16-byte copies where both ends are aligned
#define GATHER_MAX 96
struct GatherArgs { const float *src[GATHER_MAX]; long numel[GATHER_MAX]; long off[GATHER_MAX]; float *flat; };
__global__ void __launch_bounds__(256)
gather_flat_kernel(GatherArgs a) {
    const int t = blockIdx.y;
    const long n = a.numel[t];
    const float *s = a.src[t];
    float *d = a.flat + a.off[t];
    const bool v4 = ((((unsigned long long)s) | ((unsigned long long)d)) & 15ull) == 0ull;
    const long n4 = v4 ? (n >> 2) : 0;
    const long step = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) ((f32x4 *)d)[i] = ((const f32x4 *)s)[i];
    for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) d[i] = s[i];
}

// ------------------------------------------------------------------ observation normaliser (policy input, A19)
// RunningMeanStd.forward in eval mode (pacer/pacer/utils/running_mean_std.py:81-83):
//   y = clamp((x - float(mean)) / sqrt(float(var) + eps), -5, 5)
// Columns [0, split) go to out0 (leading dimension ld0), columns [split, cols) to out1 (ld1): the policy wants the
// self observation in the first 368 columns of the actor-MLP input and the task observation as a separate,
// 16-byte-aligned GEMM operand, so the split costs nothing here and saves a torch.cat and two copies.
__global__ void __launch_bounds__(256)
obs_normalize_kernel(int rows, int cols, const float *x, int ldx, const float *mean, const float *var, float eps, float clip,
                     int split, float *out0, int ld0, float *out1, int ld1) {
    const long row = blockIdx.y;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < cols; j += gridDim.x * 256) {
        float y = (x[row * ldx + j] - mean[j]) / sqrtf(var[j] + eps);
        y = fminf(fmaxf(y, -clip), clip);
        if (j < split) out0[row * ld0 + j] = y; else out1[row * ld1 + (j - split)] = y;
    }
}

// The AMP style reward of a step from the discriminator's logits (amp_continuous.py:675-692):
//   prob = 1 / (1 + exp(-logit)),  r = -log(max(1 - prob, 1e-4)) * disc_reward_scale
// one launch for the scalar chain the reference writes as seven tensor operations (the rollout issues it every step).
__global__ void __launch_bounds__(256)
disc_reward_kernel(int n, const float *logits, float scale, float *reward) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float prob = 1.0f / (1.0f + expf(-logits[i]));
    const float om = 1.0f - prob;
    reward[i] = -logf(om > 0.0001f ? om : 0.0001f) * scale;
}

// ------------------------------------------------------------------ observation normaliser: statistics update
// RunningMeanStd.forward in training mode (pacer/pacer/utils/running_mean_std.py:85-95): the batch's per-column mean and
// unbiased variance (torch.var) are merged into the float64 running moments with the parallel-variance rule,
//   n = count + rows, d = mean_b - mean, mean <- mean + d rows / n, var <- (var count + var_b rows + d^2 count rows / n) / n,
// in ONE launch with no intermediate tensors.  A workgroup owns 64 columns; its 4 waves take every fourth row each (coalesced
// 256-byte row segments), keep a Welford pair (mean, M2) per column in float64 and are folded with the same rule in LDS.
// Columns below first_col keep their moments (freeze_partial: only the last `diff` columns learn); the new count goes to
// count_out (every workgroup reads count_in, so it is not updated in place).
__global__ void __launch_bounds__(256)
rms_update_kernel(int rows, int cols, const float *x, int ldx, double *mean, double *var, const double *count_in, double *count_out, int first_col) {
    __shared__ double sh_m[4][64], sh_s[4][64];
    __shared__ int sh_n[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    double m = 0.0, s2 = 0.0;
    int n = 0;
    if (j < cols)
        for (int r = w; r < rows; r += 4) {
            const double v = (double)x[(long)r * ldx + j];
            n += 1;
            const double d = v - m;
            m += d / (double)n;
            s2 += d * (v - m);
        }
    else
        for (int r = w; r < rows; r += 4) n += 1;
    sh_m[w][lane] = m; sh_s[w][lane] = s2;
    if (lane == 0) sh_n[w] = n;
    __syncthreads();
    if (w == 0 && j < cols) {
        double nb = (double)sh_n[0];
        for (int k = 1; k < 4; ++k) {
            const double nk = (double)sh_n[k];
            if (nk > 0.0) {
                const double d = sh_m[k][lane] - m, nt = nb + nk;
                m += d * nk / nt;
                s2 += sh_s[k][lane] + d * d * nb * nk / nt;
                nb = nt;
            }
        }
        const double cnt = count_in[0], bvar = s2 / (nb - 1.0);        // torch.var: unbiased (NaN for a one-row batch, as in torch)
        if (j >= first_col) {
            const double d = m - mean[j], tot = cnt + nb;
            const double M2 = var[j] * cnt + bvar * nb + d * d * cnt * nb / tot;
            mean[j] += d * nb / tot;
            var[j] = M2 / tot;
        }
        if (j == 0) count_out[0] = cnt + nb;
    }
}

// The same update for tall batches in two launches (the PPO learner folds a 2 048 x 3 090 minibatch into the moments on every
// optimiser step: as one 512-deep serial Welford chain per wave with a float64 division per row the launch above takes 167 us).
// rms_partial_kernel: workgroup = 64 columns x RMS_CHUNK rows; a wave holds its 64 rows of a column in registers, takes their
// mean and then the squared deviations about it (two passes over registers, no division in the loop), the four waves fold with the
// parallel-variance rule; partial (n, mean, M2) per (chunk, column).  rms_merge_kernel folds the chunks in order and applies the
// running-moment update of rms_update_kernel.
#define RMS_CHUNK 256
__global__ void __launch_bounds__(256)
rms_partial_kernel(int rows, int cols, const float *x, int ldx, double *part /* [chunks][3][cols] */) {
    __shared__ double sh_m[4][64], sh_s[4][64];
    __shared__ int sh_n[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane, chunk = blockIdx.y;
    const int r0 = chunk * RMS_CHUNK + w * (RMS_CHUNK / 4);
    int n = rows - r0;
    n = n < 0 ? 0 : (n > RMS_CHUNK / 4 ? RMS_CHUNK / 4 : n);
    float v[RMS_CHUNK / 4];
    const int jc = j < cols ? j : cols - 1;
    for (int i = 0; i < RMS_CHUNK / 4; ++i) v[i] = x[(long)(r0 + (i < n ? i : 0) < rows ? r0 + (i < n ? i : 0) : 0) * ldx + jc];
    double sum = 0.0;
    for (int i = 0; i < RMS_CHUNK / 4; ++i) sum += i < n ? (double)v[i] : 0.0;
    double m = n > 0 ? sum / (double)n : 0.0, s2 = 0.0;
    for (int i = 0; i < RMS_CHUNK / 4; ++i) { const double d = (double)v[i] - m; s2 += i < n ? d * d : 0.0; }
    sh_m[w][lane] = m; sh_s[w][lane] = s2;
    if (lane == 0) sh_n[w] = n;
    __syncthreads();
    if (w == 0 && j < cols) {
        double nb = (double)sh_n[0];
        for (int k = 1; k < 4; ++k) {
            const double nk = (double)sh_n[k];
            if (nk > 0.0) {
                const double d = sh_m[k][lane] - m, nt = nb + nk;
                m += d * nk / nt;
                s2 += sh_s[k][lane] + d * d * nb * nk / nt;
                nb = nt;
            }
        }
        double *p = part + (long)chunk * 3 * cols;
        p[j] = nb; p[cols + j] = m; p[2 * cols + j] = s2;
    }
}

__global__ void __launch_bounds__(256)
rms_merge_kernel(int chunks, int cols, const double *part, double *mean, double *var, const double *count_in, double *count_out, int first_col) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    double nb = part[j], m = part[cols + j], s2 = part[2 * cols + j];
    for (int c = 1; c < chunks; ++c) {
        const double *p = part + (long)c * 3 * cols;
        const double nk = p[j];
        if (nk > 0.0) {
            const double d = p[cols + j] - m, nt = nb + nk;
            m += d * nk / nt;
            s2 += p[2 * cols + j] + d * d * nb * nk / nt;
            nb = nt;
        }
    }
    const double cnt = count_in[0], bvar = s2 / (nb - 1.0);
    if (j >= first_col) {
        const double d = m - mean[j], tot = cnt + nb;
        const double M2 = var[j] * cnt + bvar * nb + d * d * cnt * nb / tot;
        mean[j] += d * nb / tot;
        var[j] = M2 / tot;
    }
    if (j == 0) count_out[0] = cnt + nb;
}

// ------------------------------------------------------------------ LocoVal (one wave per sample)
#define LV_IN 100
#define LV_H1 49
#define LV_H2 24

// value_pose_net.py:73-103 _rotate_normalization + :141-147 forward_full input assembly
__device__ __forceinline__ void locoval_input(int lane, const float *traj, int ts, const float *pose, const float *vel,
                                              float *x /* LDS [100] */, float *ang_out) {
    float xv = traj[ts + 0], yv = traj[ts + 1];
    if (fabsf(xv) < 1e-10f) xv = 1e-10f;                 // epsilon guard on x (:79-83)
    const float ang = atan2f(yv, xv);
    const float c = cosf(ang), s = sinf(ang);
    if (lane == 0 && ang_out) *ang_out = ang;
    // bmm(v, R) with R = [[c, -s], [s, c]]:  (x, y) -> (x c + y s, -x s + y c)
    if (lane < 13) {
        const float px = traj[lane * ts], py = traj[lane * ts + 1];
        x[2 * lane] = px * c + py * s;
        x[2 * lane + 1] = -px * s + py * c;
    }
    if (lane < 24) {
        const bool hidden = lane == 4 || lane == 8 || lane == 9 || lane == 10 || lane == 11;
        const float px = pose[lane * 3], py = pose[lane * 3 + 1], pz = pose[lane * 3 + 2];
        x[26 + lane * 3] = hidden ? 0.0f : px * c + py * s;
        x[26 + lane * 3 + 1] = hidden ? 0.0f : -px * s + py * c;
        x[26 + lane * 3 + 2] = hidden ? 0.0f : pz;
    }
    if (lane == 0) {
        x[98] = vel[0] * c + vel[1] * s;
        x[99] = -vel[0] * s + vel[1] * c;
    }
}

__global__ void __launch_bounds__(64)
locoval_fwd_kernel(int B, const float *traj, int ts, const float *pose, const float *vel, const float *w1, const float *b1,
                   const float *w2, const float *b2, const float *w3, const float *b3, float *value, float *x100,
                   float *h1o, float *h2o, float *angle, const float *row_weight) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= B) return;
    // fit of a rollout step: only the rows that carry a target are evaluated (a few hundred of 4096); the others leave at once and
    // keep whatever value / activations they held -- their weight is 0 in the loss and its gradient
    if (row_weight && row_weight[i] == 0.0f) return;
    __shared__ float x[LV_IN], h1[LV_H1], h2[LV_H2];
    locoval_input(lane, traj + (long)i * 13 * ts, ts, pose + (long)i * 72, vel + (long)i * 2, x, angle ? angle + i : nullptr);
    __syncthreads();
    for (int k = lane; k < LV_IN; k += 64) x100[(long)i * LV_IN + k] = x[k];
    if (lane < LV_H1) {
        float a = b1[lane];
        for (int k = 0; k < LV_IN; ++k) a += w1[lane * LV_IN + k] * x[k];
        a = a > 0.0f ? a : 0.0f;
        h1[lane] = a; h1o[(long)i * LV_H1 + lane] = a;
    }
    __syncthreads();
    if (lane < LV_H2) {
        float a = b2[lane];
        for (int k = 0; k < LV_H1; ++k) a += w2[lane * LV_H1 + k] * h1[k];
        a = a > 0.0f ? a : 0.0f;
        h2[lane] = a; h2o[(long)i * LV_H2 + lane] = a;
    }
    __syncthreads();
    float p = lane < LV_H2 ? w3[lane] * h2[lane] : 0.0f;
    p = wave_sum(p);
    if (lane == 0) value[i] = 1.0f / (1.0f + expf(-(p + b3[0])));
}

// per-sample backward: writes this sample's parameter-gradient contribution to ws[i][6174] and d traj
#define LV_NPARAM (LV_H1 * LV_IN + LV_H1 + LV_H2 * LV_H1 + LV_H2 + LV_H2 + 1)
__global__ void __launch_bounds__(64)
locoval_bwd_kernel(int B, const float *traj, int ts, const float *pose, const float *vel, const float *w1, const float *w2,
                   const float *w3, const float *value, const float *x100, const float *h1, const float *h2,
                   const float *angle, const float *dvalue, float *ws, float *dtraj, const int32_t *slot) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= B) return;
    // sparse mode (slot != NULL): only the rows with a slot contribute, row i's parameter-gradient share goes to ws[slot[i]]
    const int row = slot ? slot[i] : i;
    if (row < 0) return;
    __shared__ float x[LV_IN], d1[LV_H1], d2[LV_H2], dx[LV_IN];
    for (int k = lane; k < LV_IN; k += 64) x[k] = x100[(long)i * LV_IN + k];
    const float v = value[i];
    const float dz3 = dvalue[i] * v * (1.0f - v);
    float *g = ws + (long)row * LV_NPARAM;
    float *gw1 = g, *gb1 = g + LV_H1 * LV_IN, *gw2 = gb1 + LV_H1, *gb2 = gw2 + LV_H2 * LV_H1, *gw3 = gb2 + LV_H2, *gb3 = gw3 + LV_H2;
    if (lane < LV_H2) {
        const float hv = h2[(long)i * LV_H2 + lane];
        gw3[lane] = dz3 * hv;
        const float dd = hv > 0.0f ? dz3 * w3[lane] : 0.0f;
        d2[lane] = dd; gb2[lane] = dd;
    }
    if (lane == 0) gb3[0] = dz3;
    __syncthreads();
    if (lane < LV_H1) {
        const float hv = h1[(long)i * LV_H1 + lane];
        float a = 0.0f;
        for (int j = 0; j < LV_H2; ++j) a += w2[j * LV_H1 + lane] * d2[j];
        const float dd = hv > 0.0f ? a : 0.0f;
        d1[lane] = dd; gb1[lane] = dd;
        for (int j = 0; j < LV_H2; ++j) gw2[j * LV_H1 + lane] = d2[j] * hv;
    }
    __syncthreads();
    for (int e = lane; e < LV_H1 * LV_IN; e += 64) { const int j = e / LV_IN, k = e - j * LV_IN; gw1[e] = d1[j] * x[k]; }
    for (int k = lane; k < LV_IN; k += 64) {
        float a = 0.0f;
        for (int j = 0; j < LV_H1; ++j) a += w1[j * LV_IN + k] * d1[j];
        dx[k] = a;
    }
    __syncthreads();
    // back through the yaw normalisation to the trajectory (pose / velocity inputs carry no gradient in the loss)
    const float *tr = traj + (long)i * 13 * ts, *po = pose + (long)i * 72, *ve = vel + (long)i * 2;
    float *dt = dtraj + (long)i * 13 * ts;
    const float ang = angle[i];
    const float c = cosf(ang), s = sinf(ang);
    float dth = 0.0f;   // d loss / d angle
    if (lane < 13) {
        const float px = tr[lane * ts], py = tr[lane * ts + 1], gx = dx[2 * lane], gy = dx[2 * lane + 1];
        for (int k = 0; k < ts; ++k) dt[lane * ts + k] = 0.0f;
        dt[lane * ts] = gx * c - gy * s;
        dt[lane * ts + 1] = gx * s + gy * c;
        dth += gx * (-px * s + py * c) + gy * (-px * c - py * s);
    }
    if (lane < 24) {
        const bool hidden = lane == 4 || lane == 8 || lane == 9 || lane == 10 || lane == 11;
        if (!hidden) {
            const float px = po[lane * 3], py = po[lane * 3 + 1], gx = dx[26 + lane * 3], gy = dx[26 + lane * 3 + 1];
            dth += gx * (-px * s + py * c) + gy * (-px * c - py * s);
        }
    }
    if (lane == 0) dth += dx[98] * (-ve[0] * s + ve[1] * c) + dx[99] * (-ve[0] * c - ve[1] * s);
    dth = wave_sum(dth);
    __syncthreads();
    if (lane == 0) {    // angle = atan2(y1, x1'), x1' = guarded x of waypoint 1
        float xv = tr[ts], yv = tr[ts + 1];
        const bool guarded = fabsf(xv) < 1e-10f;
        if (guarded) xv = 1e-10f;
        const float r2 = xv * xv + yv * yv;
        if (!guarded) dt[ts] += dth * (-yv / r2);
        dt[ts + 1] += dth * (xv / r2);
    }
}

__global__ void locoval_reduce_kernel(int B, const float *ws, float *dparams, const float *count) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= LV_NPARAM) return;
    const int rows = count ? (int)count[0] : B;         // sparse mode: the number of slots in use
    float s = 0.0f;
    int i = 0;
    for (; i + 8 <= rows; i += 8) {             // eight rows requested at once, added in row order (one at a time the sum is a
        float v[8];                              // chain of memory latencies: 38 us for the ~14 finished episodes of a step)
        for (int u = 0; u < 8; ++u) v[u] = ws[(long)(i + u) * LV_NPARAM + p];
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < rows; ++i) s += ws[(long)i * LV_NPARAM + p];
    dparams[p] = s;
}

// ---- LocoVal training step around the MLP (amp_continuous_value.py:63-145, common_agent.py:89-97,154-155) ----------------
// the return bookkeeping of one env: locoval_returns_device.h (shared with the task's flags launch)
__global__ void __launch_bounds__(64)
locoval_returns_kernel(EmlocoLocoValStep t, const float *rewards, const float *amp_rewards, const int64_t *dones, const uint8_t *inverted) {
    const int e = blockIdx.x, lane = threadIdx.x;
    if (e >= t.n_env) return;
    locoval_returns_env(t, e, lane, rewards[e], amp_rewards ? amp_rewards[e] : 0.0f, dones[e] != 0, inverted && inverted[e]);
}

// second half of a staged step (locoval_returns_device.h): one thread per env
__global__ void __launch_bounds__(256)
locoval_returns_finish_kernel(EmlocoLocoValStep t, const float *amp_rewards) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= t.n_env) return;
    locoval_advance_env(t, e, t.staged_reward[e], amp_rewards ? amp_rewards[e] : 0.0f, t.staged_done[e] != 0);
}

// d/dvalue of sum_e w_e (value_e - target_e)^2 (MSELoss(reduction='sum') over the valid rows, common_agent.py:96) and the two
// scalars that travel with the gradient bucket: tail = [loss sum, number of valid rows].  Also ranks the valid rows:
// slot[e] = number of valid rows before e (or -1), so that the backward pass touches only those rows (a few dozen of 4096
// per step).  One workgroup, thread t owns the contiguous rows [t c, (t + 1) c): fixed order, no atomics.
__global__ void __launch_bounds__(1024)
locoval_fit_grad_kernel(int n, const float *value, const float *target, const float *weight, float *dvalue, float *tail, int32_t *slot) {
    __shared__ float sl[1024];
    __shared__ int sc[1024];
    const int tid = threadIdx.x;
    const int c = (n + 1023) / 1024;
    const int lo = tid * c, hi = (lo + c < n) ? lo + c : n;
    float l = 0.0f;
    int cnt = 0;
    for (int i = lo; i < hi; ++i) {
        // rows without a target (w == 0) are not evaluated by the rows-only forward: whatever value[] still holds for them (a NaN
        // of a diverged episode included, 0 * NaN = NaN) must not reach the loss sum or the gradient
        const float w = weight[i];
        const float d = w != 0.0f ? value[i] - target[i] : 0.0f;
        dvalue[i] = 2.0f * w * d;
        l += w * d * d;
        cnt += w != 0.0f ? 1 : 0;
    }
    sl[tid] = l; sc[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // inclusive scan of the counts (Hillis-Steele)
        const int add = tid >= off ? sc[tid - off] : 0;
        __syncthreads();
        sc[tid] += add;
        __syncthreads();
    }
    if (slot) {
        int k = sc[tid] - cnt;
        for (int i = lo; i < hi; ++i) slot[i] = weight[i] != 0.0f ? k++ : -1;
    }
    const int total = sc[1023];
    __syncthreads();
    for (int off = 512; off >= 1; off >>= 1) {
        if (tid < off) sl[tid] += sl[tid + off];
        __syncthreads();
    }
    if (tid == 0) { tail[0] = sl[0]; tail[1] = (float)total; }
}

// torch.optim.AdamW's update of a flat parameter buffer, committed only when the gate is open: tail[1] (the all-reduced number
// of finished episodes) > 0.  The step count lives on the device, double-buffered so that every thread reads the old one
// (steps_in) while thread 0 writes the new one (steps_out).  stats (optional, double[5]) accumulates
// [last loss sum, last count, total loss sum, total count, number of fits].
__global__ void adamw_gated_kernel(int n, float *p, const float *g, float *m, float *v, const float *steps_in, float *steps_out,
                                   const float *tail, float lr, float b1, float b2, float eps, float wd, double *stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool open = tail ? tail[1] > 0.5f : true;
    const float st = steps_in[0] + (open ? 1.0f : 0.0f);
    if (i == 0) {
        steps_out[0] = st;
        if (stats && open && tail) { stats[0] = tail[0]; stats[1] = tail[1]; stats[2] += tail[0]; stats[3] += tail[1]; stats[4] += 1.0; }
    }
    if (i >= n || !open) return;
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = m[i] + (1.0f - b1) * (gi - m[i]);                    // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    const double bc1 = 1.0 - pow((double)b1, (double)st), bc2 = 1.0 - pow((double)b2, (double)st);
    const float step_size = (float)((double)lr / bc1);
    const float denom = sqrtf(vi) / (float)sqrt(bc2) + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
}

// clip_grad_norm_ + torch.optim.Adam on ONE flat parameter / gradient buffer in three launches (the foreach implementations issue ~25
// over the predictor's 118 tensors): sum of squares per 4096-element block, the clip coefficient from the partials in a fixed order, the
// update.  Adam as torch writes it (adam.py, _single_tensor_adam): g += wd * p; m.lerp_(g, 1 - b1); v = b2 v + (1 - b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps); the clipped gradient is written back as clip_grad_norm_ leaves it.
#define ADAM_BLOCK 4096
__global__ void __launch_bounds__(256)
sumsq_partial_kernel(long n, const float *g, float *part) {
    __shared__ float sh[4];
    const long b0 = (long)blockIdx.x * ADAM_BLOCK;
    float s = 0.0f;
    for (int k = 0; k < ADAM_BLOCK / 256; ++k) {
        const long i = b0 + k * 256 + threadIdx.x;
        const float x = i < n ? g[i] : 0.0f;
        s += x * x;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// out[0] = total norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (clip_grad_norm_'s); one workgroup, fixed order
__global__ void __launch_bounds__(256)
clip_coef_kernel(int nparts, const float *part, float max_norm, float *out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(sh[0]);
        const float c = max_norm / (norm + 1e-6f);
        // torch.clamp(c, max=1): a NaN norm gives a NaN coefficient and with it NaN in EVERY gradient -- loud, like clip_grad_norm_
        // (c < 1 ? c : 1 alone would turn it into 1 and let the finite slices train on a corrupt step)
        out[0] = norm; out[1] = (c < 1.0f || c != c) ? c : 1.0f;
    }
}
// The step counter on the device (a captured optimiser step replays with the same kernel arguments): count[0] += 1, then
// bc[0] = 1 - beta1^t, bc[1] = sqrt(1 - beta2^t) for the update launch behind it.  One thread.
__global__ void adam_step_count_kernel(float *count, double beta1, double beta2, float *bc) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float t = count[0] + 1.0f;
    count[0] = t;
    bc[0] = (float)(1.0 - pow(beta1, (double)t));
    bc[1] = (float)sqrt(1.0 - pow(beta2, (double)t));
}
__global__ void adam_flat_kernel(long n, float *p, float *g, float *m, float *v, const float *coef, float lr, float omb1, float b2, float omb2,
                                 float eps, float wd, float bc1, float bc2_sqrt, const float *bc_dev) {      // omb = 1 - beta, rounded from the double (as torch passes it)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (bc_dev) { bc1 = bc_dev[0]; bc2_sqrt = bc_dev[1]; }          // (uniform: two scalar loads)
    float gi = g[i];
    if (coef) { gi *= coef[1]; g[i] = gi; }
    float pi = p[i];
    if (wd != 0.0f) gi += wd * pi;
    const float mi = m[i] + omb1 * (gi - m[i]);
    const float vi = v[i] * b2 + omb2 * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
    m[i] = mi; v[i] = vi;
}

}  // namespace emloco
