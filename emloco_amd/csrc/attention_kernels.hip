// attention_kernels.hip -- fused multi-head self-attention for head dim 32 on gfx950 (MI355X), fp32 MFMA.
//
// Reference: nn.MultiheadAttention inside nn.TransformerEncoderLayer as used by
// /root/reference/social-transmotion/model_jta.py:177-178,311-321 (local former: S = 453 tokens per person, d = 128,
// 4 heads -> head dim 32; global former: S = 21 N).  The score matrix (S x S per head) is never written to HBM: the
// forward keeps an online softmax per query, the backward recomputes the probabilities from Q, K and the saved
// log-sum-exp.  The unfused path (GEMM -> softmax -> GEMM, predictor_kernels.hip) moved ~27 GB of scores per layer
// at batch 256; this one reads Q, K, V once per 128-query block.
//
// Mapping (all three kernels): a 256-thread workgroup = 4 waves, a wave owns 32 rows (queries, or keys in the dK/dV
// kernel) and walks the other sequence in tiles of 32 staged through LDS (double buffered, register prefetch).
// Everything is computed TRANSPOSED so that the v_mfma_f32_32x32x2_f32 accumulator layout (lane = column) makes the
// wave's own row index the lane index: a lane then owns one query (key) and
//   * softmax statistics (running max / sum, log-sum-exp, D = rowsum(dO o O)) are per-lane scalars,
//   * the probabilities P^T it just computed ARE the B operand of the next MFMA (P.V, dS.K, ...) -- no LDS round trip:
//     accumulator register s of lane half h holds tile row kappa(s, h) = (s & 3) + 8 (s >> 2) + 4 h, and since the
//     reduction index order of an MFMA chain is free, step s simply pairs it with row kappa(s, h) of the LDS operand.
// Reductions over d = 32 use dk(s, h) = 16 h + s, so a lane's 16 operand values are four ds_read_b128 of one LDS row.
// The exponentials are `__expf` (v_exp_f32 of x log2 e, ~2 ulp): the kernels' time follows their VALU instruction count (the
// 16 exponentials of a tile were a third of it with the range-checked expf), the 1e-4 parity bar is four orders above that error.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_math.h"
#include "mfma_bf16.h"
#include "drop_hash.h"

namespace emloco {

typedef float at_f32x16 __attribute__((vector_size(64)));
struct __attribute__((aligned(16))) at_f32x4 { float x, y, z, w; };

#define AT_DH 32                 // head dim
#define AT_T 32                  // tile of the walked sequence
#define AT_LD 36                 // LDS row stride (floats): 144 B keeps 16 consecutive rows' b128 reads on distinct banks
#ifndef AT_EXP
#define AT_EXP(x) __expf(x)
#endif
#define AT_NEG (-3.0e38f)        // "masked" sentinel: scores at or below it get probability 0

struct AttnArgs {
    int n_seq, S, nhead, d_model;   // qkv [n_seq][S][3 d_model], d_model = nhead * 32
    int Sq;                         // live queries: the first Sq <= S rows of every sequence attend (over all S keys); out, lse, dout,
                                    // dsum have Sq rows per sequence.  The last layer of a former only feeds its first 21 tokens on.
    float scale;
    const float *qkv;
    const float *key_bias;          // [n_seq][S] additive (-inf masks a key) or NULL
    float *out;                     // [n_seq][Sq][d_model]
    float *lse;                     // [n_seq * nhead][Sq] log-sum-exp of the scaled, biased scores
    const float *dout;              // backward: [n_seq][Sq][d_model]
    float *dqkv;                    // backward: [n_seq][S][3 d_model]  (dQ of rows >= Sq: zero-filled by the launcher)
    float *dsum;                    // backward: [n_seq * nhead][Sq]  D = rowsum(dO o O)
    // dropout on the attention probabilities (nn.MultiheadAttention(dropout=p) in training mode): probability (head-sequence bh,
    // query, key) is kept iff at_keep_bit(...) below (a 32-bit counter hash shared by two adjacent keys: ~5 integer operations per element)
    // and scaled by 1 / (1 - p).  The softmax normaliser uses the undropped probabilities; the backward recomputes the mask.
    float drop_p, drop_scale;       // drop_scale = 256 / (256 - drop_thr): the scale of the probability the mask REALISES (at_drop_thr8)
    unsigned drop_seed, drop_thr;   // drop_thr = at_drop_thr8(p): p in 1/256ths
};
// (round 6) A keep decision is one BYTE of a hash against p 2^8 -- one hash serves FOUR adjacent keys -- so the drop probability is
// realised in 1/256ths: p = 0.1 -> 26/256 = 0.1016, and the kept probabilities are scaled by 256 / (256 - 26), the inverse keep rate of
// the mask that is actually drawn (unbiased; nn.MultiheadAttention(dropout=0.1) with p off by 1.6 % of itself).  Half the hashes of the
// 16-bit decisions: the hash was 31 % of the bf16 attention kernels' time (profiles/r06_attn_dq_ablation.txt).  0 < thr8 < 256.
__host__ __device__ __forceinline__ unsigned at_drop_thr8(float p) {
    const unsigned t = (unsigned)(p * 256.0f + 0.5f);
    return t < 1u ? (p > 0.0f ? 1u : 0u) : (t > 255u ? 255u : t);
}
__host__ __device__ __forceinline__ float at_drop_scale(float p) { return 256.0f / (256.0f - (float)at_drop_thr8(p)); }

// keep mask of the attention dropout: x = drop_hash(fmix32(seed ^ bh c0) + query c1, key >> 2), key kept iff its byte of x >= p 2^8
__host__ __device__ __forceinline__ unsigned at_fmix32(unsigned x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ unsigned at_head_key(unsigned seed, unsigned bh) { return at_fmix32(seed ^ (bh * 0x9E3779B1u)); }
// One hash serves the four keys 4j .. 4j + 3 (its bytes against p 2^8): a lane's 16 keys of a tile come in aligned quads, so the
// unrolled loops evaluate 4 hashes per 16 probabilities.  thr8 = at_drop_thr8(p).
// (round 5: the per-probability mixer is drop_hash -- full-rate 24-bit multiplies, drop_hash.h; the head key stays murmur-mixed)
__host__ __device__ __forceinline__ bool at_keep_bit(unsigned head_key, unsigned query, unsigned key, unsigned thr8) {
    const unsigned x = drop_hash(head_key + query * 0x85EBCA6Bu, key >> 2);
    return ((x >> (8u * (key & 3u))) & 0xffu) >= thr8;
}
__device__ __forceinline__ float at_keep(const AttnArgs &a, unsigned head_key, int query, int key) {
    return at_keep_bit(head_key, (unsigned)query, (unsigned)key, a.drop_thr) ? a.drop_scale : 0.0f;
}

__device__ __forceinline__ int at_kappa(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

// 16 operand values of row `row` (columns 16 h .. 16 h + 15) of an LDS tile
__device__ __forceinline__ void at_row16(const float *tile, int row, int h, at_f32x4 (&f)[4]) {
    const float *src = tile + row * AT_LD + 16 * h;
    f[0] = *(const at_f32x4 *)src; f[1] = *(const at_f32x4 *)(src + 4);
    f[2] = *(const at_f32x4 *)(src + 8); f[3] = *(const at_f32x4 *)(src + 12);
}
// IO = 1: the tensor behind `base` holds bf16 (the reduced-precision mode keeps q|k|v and its gradient in HBM as bf16): element
// offsets are applied on the 2-byte type, four values arrive in one 8-byte load and are widened, results are rounded on the way out
template <int IO>
__device__ __forceinline__ const float *at_off(const float *base, long elems) {
    return IO ? (const float *)((const unsigned short *)base + elems) : base + elems;
}
template <int IO>
__device__ __forceinline__ at_f32x4 at_ld4(const float *base, long elems) {
    if (IO) {
        const uint2 u = *(const uint2 *)((const unsigned short *)base + elems);
        return at_f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    }
    return *(const at_f32x4 *)(base + elems);
}
__device__ __forceinline__ unsigned at_bf16_pair(float lo, float hi) {          // two values rounded to nearest even, packed
    unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
    a += 0x7fffu + ((a >> 16) & 1u); b += 0x7fffu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xffff0000u);
}
// the same from global memory (a row of Q / K / V / dO / O of one head), zero for rows past the sequence
template <int IO = 0>
__device__ __forceinline__ void at_grow16(const float *base, long ld, int row, int S, int h, at_f32x4 (&f)[4]) {
    const int rc = row < S ? row : S - 1;
    const long src = (long)rc * ld + 16 * h;
    for (int i = 0; i < 4; ++i) {
        at_f32x4 t = at_ld4<IO>(base, src + 4 * i);
        if (row >= S) t = at_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        f[i] = t;
    }
}
// ---- PREC = 2, the split mode (the default precision class of the product, ops.DEFAULT_PRECISION): fp32 products rebuilt from bf16
// pieces, as in the GEMMs (predictor_kernels.hip).  v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate; an fp32 value is the exact
// sum of three bf16 pieces and the six piece products above 2^-24 of the result on v_mfma_f32_32x32x16_bf16 (fp32 accumulation) give
// fp32-class products: 12 matrix instructions (384 matrix cycles) per 32 x 32 x 32 tile product where the fp32 instruction needs 16
// (1 024).  The price is vector work: ~7 instructions per pair of values cut into pieces -- the loop-invariant operand of a kernel
// (its own q / k / v / dO rows) is cut once, the walked tile's rows and the probabilities per tile.
struct AtPieces { bf16w4 g[2][3]; };          // 16 reduction entries of a lane: two groups of 8, three pieces each
__device__ __forceinline__ void at_split8(const float (&v)[8], bf16w4 (&pc)[3]) {
    unsigned w[3][4];
    for (int i = 0; i < 4; ++i) {
        float a = v[2 * i], b = v[2 * i + 1];
        const unsigned p1 = gemm_pack2_bf16(a, b);
        a -= __uint_as_float(p1 << 16); b -= __uint_as_float(p1 & 0xffff0000u);
        const unsigned p2 = gemm_pack2_bf16(a, b);
        a -= __uint_as_float(p2 << 16); b -= __uint_as_float(p2 & 0xffff0000u);
        w[0][i] = p1; w[1][i] = p2; w[2][i] = gemm_pack2_bf16(a, b);
    }
    for (int q = 0; q < 3; ++q) pc[q] = bf16w4{w[q][0], w[q][1], w[q][2], w[q][3]};
}
__device__ __forceinline__ void at_split16(const at_f32x4 (&f)[4], AtPieces &o) {
    const float v0[8] = {f[0].x, f[0].y, f[0].z, f[0].w, f[1].x, f[1].y, f[1].z, f[1].w};
    const float v1[8] = {f[2].x, f[2].y, f[2].z, f[2].w, f[3].x, f[3].y, f[3].z, f[3].w};
    at_split8(v0, o.g[0]); at_split8(v1, o.g[1]);
}
// the six products of two cut operands, smallest first
__device__ __forceinline__ at_f32x16 at_mfma_split(const bf16w4 (&x)[3], const bf16w4 (&y)[3], at_f32x16 acc) {
    acc = gemm_mfma_bf16_w(x[2], y[0], acc); acc = gemm_mfma_bf16_w(x[0], y[2], acc); acc = gemm_mfma_bf16_w(x[1], y[1], acc);
    acc = gemm_mfma_bf16_w(x[1], y[0], acc); acc = gemm_mfma_bf16_w(x[0], y[1], acc); acc = gemm_mfma_bf16_w(x[0], y[0], acc);
    return acc;
}
// a kernel's own row operand (16 values of q / k / v / dO per lane): fp32 fragments, and in the split mode their pieces (cut once)
template <int PREC> struct AtRow { at_f32x4 f[4]; };
template <> struct AtRow<2> { at_f32x4 f[4]; AtPieces s; };
template <int PREC> __device__ __forceinline__ void at_row_prep(AtRow<PREC> &r) { if constexpr (PREC == 2) at_split16(r.f, r.s); }

// C^T tile (32 x 32) = X_tile (rows from LDS) . Yfrag^T : acc[r] of lane (j, h) = sum_dk X[kappa(r,h)][dk] Y[j][dk]
// PREC = 1 (opt-in, EMLOCO_ATTN_BF16): the same operands rounded to bf16 into v_mfma_f32_32x32x16_bf16 -- the lane's 16
// values dk = 16 h + 0..15 are two groups of 8 consecutive reduction entries, A and B alike: 2 instructions instead of 16.
template <int PREC>
__device__ __forceinline__ at_f32x16 at_xyT(const float *tile, int l31, int h, const AtRow<PREC> &y) {
    const at_f32x4 (&yf)[4] = y.f;
    at_f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    at_f32x4 xf[4];
    at_row16(tile, l31, h, xf);
    if constexpr (PREC == 2) {
        AtPieces xs;
        at_split16(xf, xs);
        acc = at_mfma_split(xs.g[0], y.s.g[0], acc);
        acc = at_mfma_split(xs.g[1], y.s.g[1], acc);
        return acc;
    }
    if constexpr (PREC == 1) {
        acc = gemm_mfma_bf16(gemm_pack_bf16(xf[0], xf[1]), gemm_pack_bf16(yf[0], yf[1]), acc);
        acc = gemm_mfma_bf16(gemm_pack_bf16(xf[2], xf[3]), gemm_pack_bf16(yf[2], yf[3]), acc);
        return acc;
    }
    for (int f = 0; f < 4; ++f) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[f].x, yf[f].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[f].y, yf[f].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[f].z, yf[f].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[f].w, yf[f].w, acc, 0, 0, 0);
    }
    return acc;
}
// acc^T (32 d x 32 own rows) += X_tile^T . p : sum over tile rows kappa(s, h) of X[kappa][d = lane & 31] * p[s]
// PREC = 1: reduction entries s = 8 g + e (g = 0, 1) of the lane's half form the two groups of 8.
template <int PREC>
__device__ __forceinline__ at_f32x16 at_xTp(const float *tile, int l31, int h, const float (&p)[16], at_f32x16 acc) {
    if constexpr (PREC == 2) {
        for (int g = 0; g < 2; ++g) {
            float x[8], pv[8];
            for (int e = 0; e < 8; ++e) { x[e] = tile[at_kappa(8 * g + e, h) * AT_LD + l31]; pv[e] = p[8 * g + e]; }
            bf16w4 xs[3], ps[3];
            at_split8(x, xs); at_split8(pv, ps);
            acc = at_mfma_split(xs, ps, acc);
        }
        return acc;
    }
    if constexpr (PREC == 1) {
        for (int g = 0; g < 2; ++g) {
            float x[8];
            for (int e = 0; e < 8; ++e) x[e] = tile[at_kappa(8 * g + e, h) * AT_LD + l31];
            const at_f32x4 x0{x[0], x[1], x[2], x[3]}, x1{x[4], x[5], x[6], x[7]};
            const at_f32x4 p0{p[8 * g], p[8 * g + 1], p[8 * g + 2], p[8 * g + 3]}, p1{p[8 * g + 4], p[8 * g + 5], p[8 * g + 6], p[8 * g + 7]};
            acc = gemm_mfma_bf16(gemm_pack_bf16(x0, x1), gemm_pack_bf16(p0, p1), acc);
        }
        return acc;
    }
    for (int s = 0; s < 16; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[at_kappa(s, h) * AT_LD + l31], p[s], acc, 0, 0, 0);
    return acc;
}
// one thread's share of a 32 x 32 tile fetch: rows row0.., 8 float4 per row
struct AtFetch { at_f32x4 a, b; float c, d; };
template <int IO = 0>
__device__ __forceinline__ at_f32x4 at_fetch4(const float *base, long ld, int row0, int S, int tid) {
    const int r = tid >> 3, q = tid & 7;
    const int row = row0 + r, rc = row < S ? row : S - 1;
    at_f32x4 t = at_ld4<IO>(base, (long)rc * ld + 4 * q);
    if (row >= S) t = at_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    return t;
}
__device__ __forceinline__ void at_stash4(float *tile, int tid, const at_f32x4 &v) {
    *(at_f32x4 *)&tile[(tid >> 3) * AT_LD + 4 * (tid & 7)] = v;
}
// the 16 per-row scalars kappa(r, h), r = 0..15, of a 32-entry LDS vector (4 b128 reads)
__device__ __forceinline__ void at_vec16(const float *vec, int h, float (&o)[16]) {
    for (int g = 0; g < 4; ++g) {
        const at_f32x4 t = *(const at_f32x4 *)&vec[8 * g + 4 * h];
        o[4 * g] = t.x; o[4 * g + 1] = t.y; o[4 * g + 2] = t.z; o[4 * g + 3] = t.w;
    }
}
// write the transposed accumulator (rows = d, column = own row) as 4 float4 of the row's 32 floats
template <int IO = 0>
__device__ __forceinline__ void at_store_rowT(float *dst, int h, const at_f32x16 &acc, float mul) {
    for (int g = 0; g < 4; ++g) {
        if (IO) {
            uint2 u;
            u.x = at_bf16_pair(acc[4 * g] * mul, acc[4 * g + 1] * mul); u.y = at_bf16_pair(acc[4 * g + 2] * mul, acc[4 * g + 3] * mul);
            *(uint2 *)((unsigned short *)dst + 8 * g + 4 * h) = u;
        } else {
            *(at_f32x4 *)&dst[8 * g + 4 * h] = at_f32x4{acc[4 * g] * mul, acc[4 * g + 1] * mul, acc[4 * g + 2] * mul, acc[4 * g + 3] * mul};
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <int PREC, int DROP, int IO = 0>
__global__ void __launch_bounds__(256)
attn_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[2][AT_T * AT_LD], Vs[2][AT_T * AT_LD], Bs[2][AT_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.nhead, hd = bh - b * a.nhead;
    const unsigned hkey = at_head_key(a.drop_seed, (unsigned)bh);
    const long ld = 3L * a.d_model;
    const float *Q = at_off<IO>(a.qkv, (long)b * a.S * ld + hd * AT_DH), *K = at_off<IO>(Q, a.d_model), *V = at_off<IO>(Q, 2 * a.d_model);
    const float *kb = a.key_bias ? a.key_bias + (long)b * a.S : nullptr;
    const int query = blockIdx.x * 128 + wave * 32 + l31;
    AtRow<PREC> qf;
    at_grow16<IO>(Q, ld, query, a.Sq, h, qf.f);
    at_row_prep(qf);
    const bool live = blockIdx.x * 128 + wave * 32 < a.Sq;        // wave-uniform: waves past the live queries only help staging
    at_f32x16 acc_o;
    for (int r = 0; r < 16; ++r) acc_o[r] = 0.0f;
    float m = AT_NEG, lsum = 0.0f;
    const int ntiles = (a.S + AT_T - 1) / AT_T;

    at_f32x4 rk = at_fetch4<IO>(K, ld, 0, a.S, tid), rv = at_fetch4<IO>(V, ld, 0, a.S, tid);
    // key bias of the tile: every thread loads entry (tid & 31) from a clamped address and the value is only looked at by the
    // stash (a branch around the load, or a select right behind it, makes wave 0 wait for ALL its loads before the tile's MFMAs)
    const float *kbp = kb ? kb : a.qkv;
    const int t31 = tid & 31;
    float rb = kbp[t31 < a.S ? t31 : a.S - 1];
    at_stash4(Ks[0], tid, rk); at_stash4(Vs[0], tid, rv);
    if (tid < AT_T) Bs[0][tid] = tid < a.S ? (kb ? rb : 0.0f) : -INFINITY;
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0n = (t + 1) * AT_T;
        rk = at_fetch4<IO>(K, ld, k0n, a.S, tid); rv = at_fetch4<IO>(V, ld, k0n, a.S, tid);      // past the end: zeros, never used
        rb = kbp[k0n + t31 < a.S ? k0n + t31 : a.S - 1];
        if (live) {
        // scores^T tile: keys (rows) x own queries (lanes)
        at_f32x16 st = at_xyT<PREC>(Ks[buf], l31, h, qf);
        float bias[16], p[16];
        at_vec16(Bs[buf], h, bias);
        float tmax = AT_NEG;
        for (int r = 0; r < 16; ++r) { p[r] = st[r] * a.scale + bias[r]; tmax = p[r] > tmax ? p[r] : tmax; }
        { const float o = __shfl_xor(tmax, 32); tmax = o > tmax ? o : tmax; }
        const float m_new = tmax > m ? tmax : m;
        const float alpha = AT_EXP(m - m_new);            // m = AT_NEG on the first live tile -> 0 (or 1 while nothing is live)
        float tsum = 0.0f;
        for (int r = 0; r < 16; ++r) { p[r] = p[r] > AT_NEG ? AT_EXP(p[r] - m_new) : 0.0f; tsum += p[r]; }
        tsum += __shfl_xor(tsum, 32);
        lsum = lsum * alpha + tsum;
        m = m_new;
        for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;
        if constexpr (DROP == 1) {                         // the output sums the dropped probabilities, the normaliser does not
            const int k0 = t * AT_T;
            for (int r = 0; r < 16; ++r) p[r] *= at_keep(a, hkey, query, k0 + at_kappa(r, h));
        }
        acc_o = at_xTp<PREC>(Vs[buf], l31, h, p, acc_o);     // O^T += V^T P^T
        }
        at_stash4(Ks[buf ^ 1], tid, rk); at_stash4(Vs[buf ^ 1], tid, rv);
        if (tid < AT_T) Bs[buf ^ 1][tid] = k0n + tid < a.S ? (kb ? rb : 0.0f) : -INFINITY;
        __syncthreads();
    }
    if (query < a.Sq) {
        const float inv = lsum > 0.0f ? 1.0f / lsum : 0.0f;            // fully masked row -> zeros ("safe softmax")
        at_store_rowT(a.out + ((long)b * a.Sq + query) * a.d_model + hd * AT_DH, h, acc_o, inv);
        if (h == 0 && a.lse) a.lse[(long)bh * a.Sq + query] = lsum > 0.0f ? m + logf(lsum) : 3.0e38f;
    }
}

// ------------------------------------------------------------------------------------------------ backward 1: dQ (and D)
template <int PREC, int DROP, int IO = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
attn_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[2][AT_T * AT_LD], Vs[2][AT_T * AT_LD], Bs[2][AT_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.nhead, hd = bh - b * a.nhead;
    const unsigned hkey = at_head_key(a.drop_seed, (unsigned)bh);
    const long ld = 3L * a.d_model;
    const float *Q = at_off<IO>(a.qkv, (long)b * a.S * ld + hd * AT_DH), *K = at_off<IO>(Q, a.d_model), *V = at_off<IO>(Q, 2 * a.d_model);
    const float *dO = a.dout + (long)b * a.Sq * a.d_model + hd * AT_DH, *O = a.out + (long)b * a.Sq * a.d_model + hd * AT_DH;
    const float *kb = a.key_bias ? a.key_bias + (long)b * a.S : nullptr;
    const int query = blockIdx.x * 128 + wave * 32 + l31;
    AtRow<PREC> qf, dof;
    at_f32x4 of[4];
    at_grow16<IO>(Q, ld, query, a.Sq, h, qf.f);
    at_grow16(dO, a.d_model, query, a.Sq, h, dof.f);
    at_grow16(O, a.d_model, query, a.Sq, h, of);
    const bool live = blockIdx.x * 128 + wave * 32 < a.Sq;
    float dsum = 0.0f;                                 // D = sum_d dO O (the lane pair holds the two halves of d)
    for (int f = 0; f < 4; ++f) dsum += (dof.f[f].x * of[f].x + dof.f[f].y * of[f].y) + (dof.f[f].z * of[f].z + dof.f[f].w * of[f].w);
    at_row_prep(qf); at_row_prep(dof);
    dsum += __shfl_xor(dsum, 32);
    const float lse = query < a.Sq ? a.lse[(long)bh * a.Sq + query] : 3.0e38f;
    if (query < a.Sq && h == 0) a.dsum[(long)bh * a.Sq + query] = dsum;
    at_f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int ntiles = (a.S + AT_T - 1) / AT_T;

    at_f32x4 rk = at_fetch4<IO>(K, ld, 0, a.S, tid), rv = at_fetch4<IO>(V, ld, 0, a.S, tid);
    // key bias of the tile: every thread loads entry (tid & 31) from a clamped address and the value is only looked at by the
    // stash (a branch around the load, or a select right behind it, makes wave 0 wait for ALL its loads before the tile's MFMAs)
    const float *kbp = kb ? kb : a.qkv;
    const int t31 = tid & 31;
    float rb = kbp[t31 < a.S ? t31 : a.S - 1];
    at_stash4(Ks[0], tid, rk); at_stash4(Vs[0], tid, rv);
    if (tid < AT_T) Bs[0][tid] = tid < a.S ? (kb ? rb : 0.0f) : -INFINITY;
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0n = (t + 1) * AT_T;
        rk = at_fetch4<IO>(K, ld, k0n, a.S, tid); rv = at_fetch4<IO>(V, ld, k0n, a.S, tid);
        rb = kbp[k0n + t31 < a.S ? k0n + t31 : a.S - 1];
        if (live) {
        const at_f32x16 st = at_xyT<PREC>(Ks[buf], l31, h, qf);           // S^T
        const at_f32x16 dpt = at_xyT<PREC>(Vs[buf], l31, h, dof);         // dP^T = V dO^T
        float bias[16], ds[16];
        at_vec16(Bs[buf], h, bias);
        for (int r = 0; r < 16; ++r) {
            const float v = st[r] * a.scale + bias[r];
            const float p = v > AT_NEG ? AT_EXP(v - lse) : 0.0f;
            float dpr = dpt[r];                                           // d loss / d (dropped probability)
            if constexpr (DROP == 1) dpr *= at_keep(a, hkey, query, t * AT_T + at_kappa(r, h));
            ds[r] = a.scale * p * (dpr - dsum);
        }
        acc = at_xTp<PREC>(Ks[buf], l31, h, ds, acc);                     // dQ^T += K^T dS^T
        }
        at_stash4(Ks[buf ^ 1], tid, rk); at_stash4(Vs[buf ^ 1], tid, rv);
        if (tid < AT_T) Bs[buf ^ 1][tid] = k0n + tid < a.S ? (kb ? rb : 0.0f) : -INFINITY;
        __syncthreads();
    }
    if (query < a.Sq) at_store_rowT<IO>((float *)at_off<IO>(a.dqkv, ((long)b * a.S + query) * ld + hd * AT_DH), h, acc, 1.0f);
}

// ------------------------------------------------------------------------------------------------ backward 2: dK, dV
// (the split mode holds two cut row operands, 48 registers: 184 registers unconstrained.  Held at 3 waves per SIMD it spills 68 B and
// the train step takes 160.6 ms; at 2 waves per SIMD without the spill 189.7 ms -- measured, one box)
template <int PREC, int DROP, int IO = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
attn_bwd_dkv_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Qs[2][AT_T * AT_LD], Os[2][AT_T * AT_LD], Ls[2][AT_T], Ds[2][AT_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.nhead, hd = bh - b * a.nhead;
    const unsigned hkey = at_head_key(a.drop_seed, (unsigned)bh);
    const long ld = 3L * a.d_model;
    const float *Q = at_off<IO>(a.qkv, (long)b * a.S * ld + hd * AT_DH), *K = at_off<IO>(Q, a.d_model), *V = at_off<IO>(Q, 2 * a.d_model);
    const float *dO = a.dout + (long)b * a.Sq * a.d_model + hd * AT_DH;
    const float *lse = a.lse + (long)bh * a.Sq, *dsm = a.dsum + (long)bh * a.Sq;
    const int key = blockIdx.x * 128 + wave * 32 + l31;
    AtRow<PREC> kf, vf;
    at_grow16<IO>(K, ld, key, a.S, h, kf.f);
    at_grow16<IO>(V, ld, key, a.S, h, vf.f);
    at_row_prep(kf); at_row_prep(vf);
    const bool live = blockIdx.x * 128 + wave * 32 < a.S;         // wave-uniform: a wave wholly past the keys only helps staging
    float bias = -INFINITY;
    if (key < a.S) bias = a.key_bias ? a.key_bias[(long)b * a.S + key] : 0.0f;
    at_f32x16 acc_k, acc_v;
    for (int r = 0; r < 16; ++r) { acc_k[r] = 0.0f; acc_v[r] = 0.0f; }
    const int ntiles = (a.Sq + AT_T - 1) / AT_T;      // walks the live queries

    at_f32x4 rq = at_fetch4<IO>(Q, ld, 0, a.Sq, tid), ro = at_fetch4(dO, a.d_model, 0, a.Sq, tid);
    const int t31 = tid & 31;                        // per-query log-sum-exp and D: loaded branch-free, masked by the stash
    float rl = lse[t31 < a.Sq ? t31 : a.Sq - 1], rd = dsm[t31 < a.Sq ? t31 : a.Sq - 1];
    at_stash4(Qs[0], tid, rq); at_stash4(Os[0], tid, ro);
    if (tid < AT_T) { Ls[0][tid] = tid < a.Sq ? rl : 3.0e38f; Ds[0][tid] = tid < a.Sq ? rd : 0.0f; }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, q0n = (t + 1) * AT_T;
        rq = at_fetch4<IO>(Q, ld, q0n, a.Sq, tid); ro = at_fetch4(dO, a.d_model, q0n, a.Sq, tid);
        { const int qc = q0n + t31 < a.Sq ? q0n + t31 : a.Sq - 1; rl = lse[qc]; rd = dsm[qc]; }
        if (live) {
        const at_f32x16 s = at_xyT<PREC>(Qs[buf], l31, h, kf);            // S: queries (rows) x own keys (lanes)
        const at_f32x16 dp = at_xyT<PREC>(Os[buf], l31, h, vf);           // dP = dO V^T
        float lrow[16], drow[16], p[16], ds[16];
        at_vec16(Ls[buf], h, lrow);
        at_vec16(Ds[buf], h, drow);
        for (int r = 0; r < 16; ++r) {
            const float v = s[r] * a.scale + bias;
            p[r] = v > AT_NEG ? AT_EXP(v - lrow[r]) : 0.0f;           // rows past the sequence carry lse = 3e38 -> 0
            float dpr = dp[r];
            if constexpr (DROP == 1) {
                const float kp = at_keep(a, hkey, t * AT_T + at_kappa(r, h), key);
                dpr *= kp;
                ds[r] = a.scale * p[r] * (dpr - drow[r]);
                p[r] *= kp;                                                // dV sums the dropped probabilities
            } else {
                ds[r] = a.scale * p[r] * (dpr - drow[r]);
            }
        }
        acc_v = at_xTp<PREC>(Os[buf], l31, h, p, acc_v);                  // dV^T += dO^T P
        acc_k = at_xTp<PREC>(Qs[buf], l31, h, ds, acc_k);                 // dK^T += Q^T dS
        }
        at_stash4(Qs[buf ^ 1], tid, rq); at_stash4(Os[buf ^ 1], tid, ro);
        if (tid < AT_T) { Ls[buf ^ 1][tid] = q0n + tid < a.Sq ? rl : 3.0e38f; Ds[buf ^ 1][tid] = q0n + tid < a.Sq ? rd : 0.0f; }
        __syncthreads();
    }
    if (key < a.S) {
        const float *dst = at_off<IO>(a.dqkv, ((long)b * a.S + key) * ld + hd * AT_DH);
        at_store_rowT<IO>((float *)at_off<IO>(dst, a.d_model), h, acc_k, 1.0f);
        at_store_rowT<IO>((float *)at_off<IO>(dst, 2 * a.d_model), h, acc_v, 1.0f);
    }
}

}  // namespace emloco
