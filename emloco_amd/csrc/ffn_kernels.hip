// ffn_kernels.hip -- the feed-forward block of nn.TransformerEncoderLayer as two CHAINED matrix products per launch, gfx950 (MI355X).
//
// Reference: /root/reference/social-transmotion/model_jta.py:177-178,311 -- nn.TransformerEncoderLayer(d_model = 128,
// dim_feedforward = 1024, dropout = 0.1, activation relu, post-norm): f = dropout(linear2(dropout(relu(linear1(x))))).
// At batch 256 the hidden layer is M x 1024 with M = 927 744 token rows: as separate GEMMs (predictor_kernels.hip) it crosses HBM
// seven times per layer (written by linear1, read by linear2 and by three backward products, its gradient written once and read
// twice), and every one of those launches is bound by that traffic, not by the matrix cores (profiles/r05_jta_hbm_*.txt).  The two
// kernels below keep the hidden tile in REGISTERS between the product that makes it and the product that consumes it:
//
//   ffn_chain_kernel<0> (forward):         h = dropout(relu(x W1^T + b1))  ->  f = dropout(h W2^T + b2);   h is also stored (bf16)
//   ffn_chain_kernel<1> (input gradient):  dz1 = (dz2 W2) o [h > 0] / (1 - p)  ->  dx = dz1 W1;             dz1 is also stored (bf16)
//
// (the stored h / dz1 feed the two weight-gradient products, which stay ordinary GEMMs: their reduction runs over the M rows).
// Per layer the hidden layer and its gradient cross HBM four times instead of seven, as bf16.  This is the reduced-precision mode's
// path (ops.set_matmul_precision("bf16")): operands rounded to bf16 into v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//
// Mapping.  A 512-thread workgroup owns 256 rows; a wave owns 32 of them and -- as in the fused attention (attention_kernels.hip)
// -- everything is computed TRANSPOSED, so that the accumulator layout of the matrix instruction (lane = column) makes the wave's
// own ROW the lane index: T^T (hidden x rows) = P_chunk (hidden x 128) . X^T, and the 16 accumulator registers of a lane -- hidden
// units kappa(r, h) = (r & 3) + 8 (r >> 2) + 4 h of the tile -- ARE the B operand of the second product, whose reduction runs over
// the hidden units: out^T (128 x rows) += Q_chunk (128 x hidden) . T^T.  The reduction order inside a matrix instruction is free,
// so Q's LDS image simply stores each 16-hidden block in the order [0..3, 8..11 | 4..7, 12..15] the two lane halves supply.  A
// lane's row of X (128 values) stays in registers as bf16 for the whole launch; the hidden width is walked in chunks of 64 whose
// weight tiles (16 KB + 16 KB) are staged through LDS, double buffered, fetched one chunk ahead into registers.
//
// Memory access (measured on MI355X, tools/exp/ffn_probe.py, one launch at M = 927 744): with lane = row a lane's eight 8-byte pieces of
// the stored tile lie 2 KB apart from its neighbours' -- written straight from the registers the stores touched 32 lines per
// instruction and cost 0.31 ms (forward) / 0.72 ms (backward) of a launch, and the backward's mask read straight off the stored hidden
// layer another 0.77 ms.  So (a) a wave passes its 32 x 64 tile through 4 KB of LDS of its own and stores it as whole 128-byte lines
// (four 16-byte stores per lane and chunk), and (b) the forward leaves the mask as BITS (one word per lane and chunk), which is all
// the backward reads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mfma_bf16.h"
#include "drop_hash.h"

namespace emloco {

typedef float ffn_f32x16 __attribute__((vector_size(64)));
struct __attribute__((aligned(16))) ffn_f32x4 { float x, y, z, w; };
struct __attribute__((aligned(16))) ffn_u32x4 { unsigned x, y, z, w; };
struct __attribute__((aligned(8))) ffn_u32x2 { unsigned x, y; };

#define FFN_D 128                 // model width (rows of X / out)
#define FFN_CH 64                 // hidden units per chunk
#define FFN_ROWS 256              // rows per workgroup (8 waves x 32)
#define FFN_THREADS 512

struct FfnArgs {
    int M, F;                       // rows; hidden width (a multiple of FFN_CH)
    const float *x;                 // [M][128] fp32: forward input x / backward dz2 (gradient w.r.t. linear2's output, output dropout applied)
    const unsigned short *P;        // [F][128] bf16: first product's weight tile source (forward W1; backward W2^T)
    const unsigned short *Q;        // [128][F] bf16: second product's                 (forward W2; backward W1^T)
    const float *b1;                // forward: [F] bias of linear1
    const float *b2;                // forward: [128] bias of linear2
    unsigned short *h;              // forward OUT [M][F] bf16: hidden layer after ReLU + dropout (operand of the weight gradient dW2)
    unsigned short *dz1;            // backward OUT [M][F] bf16: gradient w.r.t. linear1's output
    unsigned *mask;                 // [M][F / 32] words: bit = "hidden unit active and kept" (forward OUT, backward IN); word (row, chunk c, lane half hi)
                                    // at [row][2 c + hi], bit 16 t + 4 q + e = unit c 64 + t 32 + 8 q + 4 hi + e -- what one lane holds of a chunk
    float *out;                     // [M][128] fp32: forward f / backward dx
    float drop_p, keep_scale;       // dropout of the block (both nn.Dropout(p)): keep_scale = 1 / (1 - p)
    unsigned seed1, thr16;          // hidden-layer mask: ffn_keep16 below, thr16 = (unsigned)(p 65536)
    unsigned seed2;                 // output mask: the GEMM epilogue's drop_keep(seed2, row * 128 + col, p) (predictor_kernels.hip)
    float *colpart;                 // backward, optional OUT [ceil(M / 32)][F]: column sums of dz1 AS STORED (bf16-rounded) over each wave's 32 rows --
                                    // linear1's bias gradient after a fixed-order fold (round 6: was a pass of its own over the M x F gradient)
    // forward, optional (round 6): the post-norm layer's residual add + LayerNorm in the epilogue -- a lane holds half a row, its partner
    // half the other: xr = f + res, y = (xr - mean) rstd gamma + beta.  With xr set, `out` receives y and f itself is never written.
    const float *res, *gamma, *beta;
    float *xr, *mean, *rstd;
    float eps;
};

// Keep mask of the hidden layer's dropout: one 32-bit hash per PAIR of adjacent hidden units (its low / high 16 bits against p 2^16),
// keyed by (seed, row): x = drop_hash(fmix32(seed ^ row c0), pair).  Stateless, so nothing is stored: the backward reads the mask
// off the stored hidden layer (after ReLU + dropout a positive value means "active and kept").
__host__ __device__ __forceinline__ unsigned ffn_fmix32(unsigned x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ unsigned ffn_row_key(unsigned seed, unsigned row) { return ffn_fmix32(seed ^ (row * 0x9E3779B1u)); }
__host__ __device__ __forceinline__ unsigned ffn_pair_hash(unsigned row_key, unsigned pair) { return drop_hash(row_key, pair); }   // full-rate 24-bit multiplies (drop_hash.h)
__host__ __device__ __forceinline__ bool ffn_keep16(unsigned seed, unsigned row, unsigned hidden, unsigned thr16) {
    const unsigned x = ffn_pair_hash(ffn_row_key(seed, row), hidden >> 1);
    return ((hidden & 1u) ? (x >> 16) : (x & 0xffffu)) >= thr16;
}
// the output dropout is the GEMM epilogue's mask (drop_hash.h: drop_keep over the flat element index row * 128 + col), so that the
// existing backward pass over the output gradient (emloco_act_bwd_colsum) recomputes it
__host__ __device__ __forceinline__ bool ffn_out_keep(unsigned seed, unsigned long long idx, float p) { return drop_keep(seed, idx, p); }

__device__ __forceinline__ float ffn_bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float ffn_bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
struct __attribute__((aligned(8))) ffn_f32x2 { float x, y; };
// the lane's value plus its partner's in the other 32-lane half (v_permlane32_swap: a vector instruction; which of the two results is the
// lane's own differs between the halves, the sum does not)
__device__ __forceinline__ float ffn_both_halves(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LDS images of one chunk (bytes).  P tile [64 hidden][128 k] bf16: 256-byte rows, the 16-byte slot c of row r at slot c ^ (r & 15)
// (a lane group of a ds_read_b128 -- 16 rows at one column -- then covers all 64 banks).  Q tile [128 rows][64 hidden] bf16: 128-byte
// rows of eight 16-byte slots s = 2 (16-hidden block) + lane half, slot s of row r at s ^ ((r >> 1) & 7) (two rows fill a bank row).
#define FFN_P_BYTES (FFN_CH * FFN_D * 2)
#define FFN_Q_BYTES (FFN_D * FFN_CH * 2)
#define FFN_T_BYTES (32 * FFN_CH * 2)       /* a wave's 32 x 64 bf16 tile on its way out: 128-byte rows, 16-byte slot s of row r at s ^ (r & 7) */
__device__ __forceinline__ int ffn_p_off(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 4); }
__device__ __forceinline__ int ffn_q_off(int r, int s) { return r * 128 + ((s ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ int ffn_t_off(int r, int s) { return r * 128 + ((s ^ (r & 7)) << 4); }
#ifdef EMLOCO_EMU
__device__ __forceinline__ void ffn_wave_sync() { emu::wave_barrier(); }        // lock-step fibers: the wave's lanes meet between the tile's write and read
#else
__device__ __forceinline__ void ffn_wave_sync() { __builtin_amdgcn_wave_barrier(); }   // DS instructions of one wave complete in issue order: a scheduling fence only
#endif

// one thread's share of a chunk's weight tiles: two 16-byte pieces of P (the chunk's 64 rows are one contiguous 16 KB run) and two of Q
// (rows 2 F bytes apart, 128 bytes each)
struct FfnStage { ffn_u32x4 p[2], q[2]; };
__device__ __forceinline__ void ffn_fetch(const FfnArgs &a, int chunk, int tid, FfnStage &s) {
    const int nch = a.F / FFN_CH;
    const int c = chunk < nch ? chunk : nch - 1;                       // past the end: a valid duplicate, never consumed
    const unsigned short *pp = a.P + (long)c * FFN_CH * FFN_D;
    #pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + FFN_THREADS * e;
        s.p[e] = *(const ffn_u32x4 *)(pp + (long)idx * 8);
        const int r = idx >> 3, cc = idx & 7;
        s.q[e] = *(const ffn_u32x4 *)(a.Q + (long)r * a.F + (long)c * FFN_CH + cc * 8);
    }
}
__device__ __forceinline__ void ffn_stash(char *lp, char *lq, int tid, const FfnStage &s) {
    #pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + FFN_THREADS * e;
        *(ffn_u32x4 *)(lp + ffn_p_off(idx >> 4, idx & 15)) = s.p[e];
        // eight consecutive hidden units of row r: units 0..3 go to the first (second, for the odd piece of a 16-block) half of lane half 0's
        // slot, units 4..7 to the same half of lane half 1's slot
        const int r = idx >> 3, cc = idx & 7, blk = cc >> 1, half = (cc & 1) * 8;
        *(ffn_u32x2 *)(lq + ffn_q_off(r, 2 * blk) + half) = ffn_u32x2{s.q[e].x, s.q[e].y};
        *(ffn_u32x2 *)(lq + ffn_q_off(r, 2 * blk + 1) + half) = ffn_u32x2{s.q[e].z, s.q[e].w};
    }
}

// MODE 0: forward, MODE 1: input gradient.  DROP: the block's dropout is on (training)
template <int MODE, int DROP>
__global__ void __launch_bounds__(FFN_THREADS)
ffn_chain_kernel(FfnArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[2][FFN_P_BYTES + FFN_Q_BYTES];
    __shared__ __attribute__((aligned(16))) char lds_t[FFN_THREADS / 64][FFN_T_BYTES];
    // linear1's bias in LDS (forward): read from memory inside the chunk loop its loads queue behind the previous chunk's tile stores
    // (the vector-memory counter retires in order) and every chunk waited for those stores to land -- 56 % of the forward's wave
    // cycles were such waits (profiles/r05_attn_counters_bf16.txt)
    __shared__ __attribute__((aligned(16))) float lds_b1[MODE == 0 ? 2048 : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    // rows past the end compute -- and store -- the LAST row again (identical values to the same addresses): no lane-dependent branch
    // splits the chunk loop into basic blocks, the matrix instructions schedule across the stores
    const int row_raw = blockIdx.x * FFN_ROWS + wave * 32 + l31;
    const long rowc = row_raw < a.M ? row_raw : a.M - 1;
    const int row = (int)rowc;
    const int nch = a.F / FFN_CH;

    // this lane's row of X as the B operand of the first product: step s covers k = 16 s + 8 hi + 0..7
    bf16w4 xb[8];
    {
        const float *xr = a.x + rowc * FFN_D + 8 * hi;
        #pragma unroll
        for (int s = 0; s < 8; ++s) {
            const ffn_f32x4 v0 = *(const ffn_f32x4 *)(xr + 16 * s), v1 = *(const ffn_f32x4 *)(xr + 16 * s + 4);
            xb[s] = bf16w4{gemm_pack2_bf16(v0.x, v0.y), gemm_pack2_bf16(v0.z, v0.w), gemm_pack2_bf16(v1.x, v1.y), gemm_pack2_bf16(v1.z, v1.w)};
        }
    }
    ffn_f32x16 acc[4];
    #pragma unroll
    for (int n = 0; n < 4; ++n)
        #pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    const unsigned rkey = DROP ? ffn_row_key(a.seed1, (unsigned)row) : 0u;

    if (MODE == 0)
        for (int i = tid; i < a.F && i < 2048; i += FFN_THREADS) lds_b1[i] = a.b1[i];
    FfnStage st;
    ffn_fetch(a, 0, tid, st);
    ffn_stash(lds[0], lds[0] + FFN_P_BYTES, tid, st);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        const char *lp = lds[buf], *lq = lds[buf] + FFN_P_BYTES;
        ffn_fetch(a, c + 1, tid, st);                                  // the next chunk's weights fly during this chunk's products
        unsigned mword = 0u;
        if (MODE == 1) mword = a.mask[rowc * (a.F / 32) + 2 * c + hi];  // this lane's 32 units of the chunk: requested ahead of the first product
        __builtin_amdgcn_sched_barrier(0);
        // ---- first product: T^T[t] (32 hidden x 32 rows) = P[t] . X^T, reduction over the 128 model columns
        ffn_f32x16 T[2];
        #pragma unroll
        for (int t = 0; t < 2; ++t) {
            #pragma unroll
            for (int r = 0; r < 16; ++r) T[t][r] = 0.0f;
            #pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bf16w4 pa = *(const bf16w4 *)(lp + ffn_p_off(t * 32 + l31, 2 * s + hi));
                T[t] = gemm_mfma_bf16_w(pa, xb[s], T[t]);
            }
        }
        // ---- between the products: register r of tile t is hidden unit c 64 + t 32 + (r & 3) + 8 (r >> 2) + 4 hi of this lane's row
        bf16w4 tb[2][2];
        #pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned w[8];
            #pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hid = c * FFN_CH + t * 32 + 8 * q + 4 * hi;      // four consecutive hidden units
                float v[4] = {T[t][4 * q], T[t][4 * q + 1], T[t][4 * q + 2], T[t][4 * q + 3]};
                if (MODE == 0) {
                    const ffn_f32x4 b = *(const ffn_f32x4 *)(lds_b1 + hid);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    #pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                    if (DROP) {
                        const unsigned h0 = ffn_pair_hash(rkey, (unsigned)(hid >> 1)), h1 = ffn_pair_hash(rkey, (unsigned)(hid >> 1) + 1u);
                        v[0] = (h0 & 0xffffu) >= a.thr16 ? v[0] * a.keep_scale : 0.0f;
                        v[1] = (h0 >> 16) >= a.thr16 ? v[1] * a.keep_scale : 0.0f;
                        v[2] = (h1 & 0xffffu) >= a.thr16 ? v[2] * a.keep_scale : 0.0f;
                        v[3] = (h1 >> 16) >= a.thr16 ? v[3] * a.keep_scale : 0.0f;
                    }
                    #pragma unroll
                    for (int e = 0; e < 4; ++e) mword |= (v[e] > 0.0f ? 1u : 0u) << (16 * t + 4 * q + e);
                } else {
                    #pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ((mword >> (16 * t + 4 * q + e)) & 1u) ? v[e] * a.keep_scale : 0.0f;
                }
                const unsigned w0 = gemm_pack2_bf16(v[0], v[1]), w1 = gemm_pack2_bf16(v[2], v[3]);      // what the second product consumes, rounded once
                w[2 * q] = w0; w[2 * q + 1] = w1;
                *(ffn_u32x2 *)(lds_t[wave] + ffn_t_off(l31, 4 * t + q) + 8 * hi) = ffn_u32x2{w0, w1};
            }
            tb[t][0] = bf16w4{w[0], w[1], w[2], w[3]};                  // hidden 16-block 0 of the tile: units 4 hi + 0..3, 8 + 4 hi + 0..3
            tb[t][1] = bf16w4{w[4], w[5], w[6], w[7]};
        }
        // the tile leaves as whole lines: 8 lanes x 16 bytes = the 128 bytes a row holds of this chunk, 8 rows per store instruction
        // (rows past the end store the last row's values again, as everywhere in this kernel)
        if (MODE == 0) a.mask[rowc * (a.F / 32) + 2 * c + hi] = mword;
        ffn_wave_sync();
        if (MODE == 1 && a.colpart) {
            // column sums of the wave's tile, read back as it was rounded: lane half h takes rows h, h + 2, ... (128 bytes apart: the two
            // halves hit different banks), lane (h, j) columns 2 j, 2 j + 1 -- sixteen 4-byte reads; rows past the end (duplicates of the
            // last row) are left out; the halves meet through v_permlane32_swap and lanes 0 .. 31 store the 64 sums as one 256-byte line
            const int r0w = blockIdx.x * FFN_ROWS + wave * 32;
            float s0 = 0.0f, s1 = 0.0f;
            #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = 2 * r + hi;
                const unsigned w = *(const unsigned *)(lds_t[wave] + ffn_t_off(rl, l31 >> 2) + 4 * (l31 & 3));
                const bool ok = r0w + rl < a.M;
                s0 += ok ? ffn_bf16_lo(w) : 0.0f;
                s1 += ok ? ffn_bf16_hi(w) : 0.0f;
            }
            s0 = ffn_both_halves(s0); s1 = ffn_both_halves(s1);
            if (hi == 0 && r0w < a.M) *(ffn_f32x2 *)(a.colpart + ((long)blockIdx.x * (FFN_THREADS / 64) + wave) * a.F + (long)c * FFN_CH + 2 * l31) = ffn_f32x2{s0, s1};
        }
        {
            unsigned short *dst = (MODE == 0 ? a.h : a.dz1) + (long)c * FFN_CH + 8 * (lane & 7);
            const int r0w = blockIdx.x * FFN_ROWS + wave * 32;
            #pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = 8 * i + (lane >> 3);
                const long rg = r0w + rl < a.M ? r0w + rl : a.M - 1;
                const ffn_u32x4 piece = *(const ffn_u32x4 *)(lds_t[wave] + ffn_t_off(rl, lane & 7));
                *(ffn_u32x4 *)(dst + rg * a.F) = piece;
            }
        }
        ffn_wave_sync();
        // ---- second product: out^T[n] (32 model columns x 32 rows) += Q[n] . T^T, reduction over the chunk's 64 hidden units
        #pragma unroll
        for (int n = 0; n < 4; ++n)
            #pragma unroll
            for (int t = 0; t < 2; ++t)
                #pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const bf16w4 qa = *(const bf16w4 *)(lq + ffn_q_off(n * 32 + l31, 2 * (2 * t + g) + hi));
                    acc[n] = gemm_mfma_bf16_w(qa, tb[t][g], acc[n]);
                }
        __builtin_amdgcn_sched_barrier(0);
        ffn_stash(lds[buf ^ 1], lds[buf ^ 1] + FFN_P_BYTES, tid, st);
        __syncthreads();
    }
    // ---- epilogue: register r of acc[n] is model column n 32 + (r & 3) + 8 (r >> 2) + 4 hi of this lane's row
    if (MODE == 0 && a.xr) {
        // residual add + LayerNorm of the row (model_jta.py:177: norm2(x + dropout(ff(x)))): the 128 values of a row sit in this lane and
        // its partner of the other half -- sums in a fixed order, the halves meet through v_permlane32_swap (biased variance, as nn.LayerNorm)
        const float *rr = a.res + rowc * FFN_D;
        float *xo = a.xr + rowc * FFN_D;
        float s1 = 0.0f;
        #pragma unroll
        for (int n = 0; n < 4; ++n)
            #pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n * 32 + 8 * q + 4 * hi;
                float v[4] = {acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]};
                const ffn_f32x4 b = *(const ffn_f32x4 *)(a.b2 + col);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                if (DROP)
                    #pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = ffn_out_keep(a.seed2, (unsigned long long)row * FFN_D + col + e, a.drop_p) ? v[e] * a.keep_scale : 0.0f;
                const ffn_f32x4 r4 = *(const ffn_f32x4 *)(rr + col);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                *(ffn_f32x4 *)(xo + col) = ffn_f32x4{v[0], v[1], v[2], v[3]};
                #pragma unroll
                for (int e = 0; e < 4; ++e) acc[n][4 * q + e] = v[e];
                s1 += (v[0] + v[1]) + (v[2] + v[3]);
            }
        const float mu = ffn_both_halves(s1) * (1.0f / 128.0f);
        float s2 = 0.0f;
        #pragma unroll
        for (int n = 0; n < 4; ++n)
            #pragma unroll
            for (int q = 0; q < 4; ++q) {
                float c[4];
                #pragma unroll
                for (int e = 0; e < 4; ++e) { c[e] = acc[n][4 * q + e] - mu; acc[n][4 * q + e] = c[e]; }
                s2 += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
            }
        const float rs = 1.0f / sqrtf(ffn_both_halves(s2) * (1.0f / 128.0f) + a.eps);
        float *o = a.out + rowc * FFN_D;
        #pragma unroll
        for (int n = 0; n < 4; ++n)
            #pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n * 32 + 8 * q + 4 * hi;
                const ffn_f32x4 g = *(const ffn_f32x4 *)(a.gamma + col), be = *(const ffn_f32x4 *)(a.beta + col);
                *(ffn_f32x4 *)(o + col) = ffn_f32x4{acc[n][4 * q] * rs * g.x + be.x, acc[n][4 * q + 1] * rs * g.y + be.y,
                                                    acc[n][4 * q + 2] * rs * g.z + be.z, acc[n][4 * q + 3] * rs * g.w + be.w};
            }
        if (hi == 0) { a.mean[rowc] = mu; a.rstd[rowc] = rs; }
        return;
    }
    {
        float *o = a.out + rowc * FFN_D;
        #pragma unroll
        for (int n = 0; n < 4; ++n)
            #pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n * 32 + 8 * q + 4 * hi;
                float v[4] = {acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]};
                if (MODE == 0) {
                    const ffn_f32x4 b = *(const ffn_f32x4 *)(a.b2 + col);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    if (DROP)
                        #pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = ffn_out_keep(a.seed2, (unsigned long long)row * FFN_D + col + e, a.drop_p) ? v[e] * a.keep_scale : 0.0f;
                }
                *(ffn_f32x4 *)(o + col) = ffn_f32x4{v[0], v[1], v[2], v[3]};
            }
    }
}

}  // namespace emloco
