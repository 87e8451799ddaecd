// attention16_kernels.hip -- the fused attention of the reduced-precision mode (q|k|v held in HBM as bf16, ops.set_matmul_precision("bf16"):
// BASELINE configs[3] "bf16 MFMA attention"), head dim 32, gfx950 (MI355X).  Round 5.
//
// Reference: nn.MultiheadAttention inside nn.TransformerEncoderLayer, /root/reference/social-transmotion/model_jta.py:177-178,311-321.
// Same mathematics, arguments, dropout mask (at_keep_bit) and memory formats as attn_fwd / attn_bwd_dq / attn_bwd_dkv<1, DROP, 1>
// (attention_kernels.hip), which these kernels replace for that mode.  What changed, and why (profiles/r04_mfma_utilisation.txt: the
// matrix pipe of those kernels was 8.5-10.9 % busy; their time was their vector-instruction count and the latency of one dependent
// chain per wave):
//   * a wave owns TWO blocks of 32 rows (64 queries, or keys in the dK/dV kernel): every operand fragment read from LDS feeds two matrix
//     instructions, a workgroup covers 256 rows (half the barriers and tile fetches per row), and the two blocks' softmax chains are
//     independent instruction streams the scheduler interleaves with the other block's matrix instructions;
//   * the walked tiles live in LDS as bf16 in the layout the matrix instruction consumes -- a [32][32] row image whose 16-byte slots
//     are the 8 consecutive reduction entries a lane feeds (one ds_read_b128 per instruction operand, no widening to fp32 on the way in
//     and no per-wave re-packing on the way out), and for the products that reduce over the tile's rows a TRANSPOSED image whose slots
//     hold the rows in the order the accumulator registers of the producing instruction have them (kappa order);
//   * softmax in base 2 (one v_exp_f32 per probability, scale and bias folded into one fma), no select around the exponential (a masked
//     key carries -inf), the dropout scale folded into the final normalisation.
#include "attention_kernels.hip"

namespace emloco {

#define A16_LOG2E 1.4426950408889634f
#define A16_LN2 0.6931471805599453f
#define A16_TILE_BYTES 2048                 // 32 x 32 bf16
// 2^x as ONE v_exp_f32: exp2f() wraps the instruction in a range-scaling sequence (compare, select, add, v_ldexp: five instructions per
// probability) so that results below 2^-126 come out as denormals -- here they may flush to zero, like the probabilities of masked keys
#ifdef EMLOCO_EMU
#define A16_EXP2(x) exp2f(x)
#else
#define A16_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
struct __attribute__((aligned(8))) a16_u32x2 { unsigned x, y; };

// row image: row r (64 bytes), 16-byte slot s (8 consecutive columns) at s ^ ((r >> 2) & 3): the 16 rows of a ds_read_b128 lane group
// cover all 64 banks.  Transposed image: row = column c of the tile, slot s = 2 (16-row block of the tile) + lane half, holding the
// block's rows [4 h .. 4 h + 3, 8 + 4 h .. 8 + 4 h + 3] -- what registers 8 g .. 8 g + 7 of lane half h hold of an accumulator.
__device__ __forceinline__ int a16_slot_off(int r, int s) { return r * 64 + ((s ^ ((r >> 2) & 3)) << 4); }
__device__ __forceinline__ bf16w4 a16_frag(const char *tile, int row, int slot) { return *(const bf16w4 *)(tile + a16_slot_off(row, slot)); }
// one thread's 4 consecutive columns (c0 = 4 piece) of tile row `r`, as two packed words
__device__ __forceinline__ void a16_stash_rows(char *tile, int r, int piece, a16_u32x2 v) {
    *(a16_u32x2 *)(tile + a16_slot_off(r, piece >> 1) + (piece & 1) * 8) = v;
}
__device__ __forceinline__ void a16_stash_cols(char *tile, int r, int piece, a16_u32x2 v) {
    const int slot = 2 * (r >> 4) + ((r >> 2) & 1), idx = (r & 3) + 4 * ((r >> 3) & 1);
    unsigned short *t = (unsigned short *)tile;
    const unsigned short e[4] = {(unsigned short)(v.x & 0xffffu), (unsigned short)(v.x >> 16), (unsigned short)(v.y & 0xffffu), (unsigned short)(v.y >> 16)};
    #pragma unroll
    for (int k = 0; k < 4; ++k) t[(a16_slot_off(4 * piece + k, slot) >> 1) + idx] = e[k];
}
__device__ __forceinline__ a16_u32x2 a16_ld_bf16x4(const unsigned short *base, long elems, bool ok) {
    a16_u32x2 v = *(const a16_u32x2 *)(base + elems);
    if (!ok) v = a16_u32x2{0u, 0u};
    return v;
}
__device__ __forceinline__ a16_u32x2 a16_ld_f32x4(const float *base, long elems, bool ok) {
    const at_f32x4 t = *(const at_f32x4 *)(base + elems);
    a16_u32x2 v{gemm_pack2_bf16(t.x, t.y), gemm_pack2_bf16(t.z, t.w)};
    if (!ok) v = a16_u32x2{0u, 0u};
    return v;
}
// ---- operands in NP pieces.  NP = 1: bf16 operands (the reduced-precision mode).  NP = 3: the split mode -- every fp32 value is the exact
// sum of three bf16 pieces and a product is the six piece products above 2^-24 of it (attention_kernels.hip: at_mfma_split); a tile image
// is then three planes (one per piece), a fragment three 16-byte reads, a tile product six matrix instructions per 16 reduction entries.
// The pieces of a WALKED tile are cut ONCE, by the thread that stages the element (round 4 cut them per consuming wave: 630 vector
// instructions per wave and 32 x 32 tile in the forward, profiles/r05_attn_counters_fp32_split.txt).
template <int NP> struct A16Frag { bf16w4 p[NP]; };
__device__ __forceinline__ void a16_split_pair(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = gemm_pack2_bf16(a, b);
    a -= __uint_as_float(p1 << 16); b -= __uint_as_float(p1 & 0xffff0000u);
    p2 = gemm_pack2_bf16(a, b);
    a -= __uint_as_float(p2 << 16); b -= __uint_as_float(p2 & 0xffff0000u);
    p3 = gemm_pack2_bf16(a, b);
}
// four consecutive values -> NP packed pairs of words
template <int NP>
__device__ __forceinline__ void a16_cut4(float v0, float v1, float v2, float v3, a16_u32x2 (&o)[NP]) {
    if constexpr (NP == 1) {
        o[0] = a16_u32x2{gemm_pack2_bf16(v0, v1), gemm_pack2_bf16(v2, v3)};
    } else if constexpr (NP == 2) {                           // two pieces: the value to 2^-17 of its magnitude
        const unsigned a1 = gemm_pack2_bf16(v0, v1), b1 = gemm_pack2_bf16(v2, v3);
        v0 -= __uint_as_float(a1 << 16); v1 -= __uint_as_float(a1 & 0xffff0000u);
        v2 -= __uint_as_float(b1 << 16); v3 -= __uint_as_float(b1 & 0xffff0000u);
        o[0] = a16_u32x2{a1, b1};
        o[1] = a16_u32x2{gemm_pack2_bf16(v0, v1), gemm_pack2_bf16(v2, v3)};
    } else {
        unsigned a[3], b[3];
        a16_split_pair(v0, v1, a[0], a[1], a[2]);
        a16_split_pair(v2, v3, b[0], b[1], b[2]);
        #pragma unroll
        for (int q = 0; q < 3; ++q) o[q] = a16_u32x2{a[q], b[q]};
    }
}
// eight consecutive values (the reduction entries a lane feeds one matrix instruction) -> a fragment
template <int NP>
__device__ __forceinline__ A16Frag<NP> a16_cut8(const float (&v)[8]) {
    a16_u32x2 lo[NP], hi[NP];
    a16_cut4<NP>(v[0], v[1], v[2], v[3], lo);
    a16_cut4<NP>(v[4], v[5], v[6], v[7], hi);
    A16Frag<NP> f;
    #pragma unroll
    for (int q = 0; q < NP; ++q) f.p[q] = bf16w4{lo[q].x, lo[q].y, hi[q].x, hi[q].y};
    return f;
}
template <int NP>
__device__ __forceinline__ at_f32x16 a16_mma(const A16Frag<NP> &x, const A16Frag<NP> &y, at_f32x16 acc) {
    if constexpr (NP == 1) {
        return gemm_mfma_bf16_w(x.p[0], y.p[0], acc);
    } else if constexpr (NP == 2) {                           // the three products above 2^-16 of the result
        acc = gemm_mfma_bf16_w(x.p[1], y.p[0], acc); acc = gemm_mfma_bf16_w(x.p[0], y.p[1], acc); acc = gemm_mfma_bf16_w(x.p[0], y.p[0], acc);
        return acc;
    } else {                                                  // the six products, smallest first
        acc = gemm_mfma_bf16_w(x.p[2], y.p[0], acc); acc = gemm_mfma_bf16_w(x.p[0], y.p[2], acc); acc = gemm_mfma_bf16_w(x.p[1], y.p[1], acc);
        acc = gemm_mfma_bf16_w(x.p[1], y.p[0], acc); acc = gemm_mfma_bf16_w(x.p[0], y.p[1], acc); acc = gemm_mfma_bf16_w(x.p[0], y.p[0], acc);
        return acc;
    }
}
template <int NP>
__device__ __forceinline__ A16Frag<NP> a16_frag_n(const char *tile, int row, int slot) {
    A16Frag<NP> f;
    #pragma unroll
    for (int q = 0; q < NP; ++q) f.p[q] = a16_frag(tile + q * A16_TILE_BYTES, row, slot);
    return f;
}
// one thread's 4 consecutive columns of a tile row, from memory (MEM16: bf16, else fp32), cut and stored into the row and / or the
// transposed image (NP planes each)
template <int NP, int MEM16>
__device__ __forceinline__ void a16_fetch4(const void *base, long elems, bool ok, a16_u32x2 (&o)[NP]) {
    if constexpr (MEM16) {
        static_assert(NP == 1 || !MEM16, "bf16 memory holds one piece");
        o[0] = a16_ld_bf16x4((const unsigned short *)base, elems, ok);
    } else {
        at_f32x4 t = *(const at_f32x4 *)((const float *)base + elems);
        if (!ok) t = at_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        a16_cut4<NP>(t.x, t.y, t.z, t.w, o);
    }
}
template <int NP>
__device__ __forceinline__ void a16_put_rows(char *tile, int r, int piece, const a16_u32x2 (&o)[NP]) {
    #pragma unroll
    for (int q = 0; q < NP; ++q) a16_stash_rows(tile + q * A16_TILE_BYTES, r, piece, o[q]);
}
template <int NP>
__device__ __forceinline__ void a16_put_cols(char *tile, int r, int piece, const a16_u32x2 (&o)[NP]) {
    #pragma unroll
    for (int q = 0; q < NP; ++q) a16_stash_cols(tile + q * A16_TILE_BYTES, r, piece, o[q]);
}
// a lane's own row as the B operand of a product that reduces over the head dim: step ks covers d = 16 ks + 8 hi + 0..7
template <int NP, int MEM16>
__device__ __forceinline__ void a16_own_row(const void *base, long ld, int row, int rows, int hi, A16Frag<NP> (&f)[2], float *raw /* 16 values or NULL */) {
    const long rc = row < rows ? row : rows - 1;
    #pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if constexpr (MEM16) {
            f[ks].p[0] = *(const bf16w4 *)((const unsigned short *)base + rc * ld + 16 * ks + 8 * hi);
        } else {
            const float *src = (const float *)base + rc * ld + 16 * ks + 8 * hi;
            const at_f32x4 v0 = *(const at_f32x4 *)src, v1 = *(const at_f32x4 *)(src + 4);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            f[ks] = a16_cut8<NP>(v);
            if (raw) {
                #pragma unroll
                for (int e = 0; e < 8; ++e) raw[8 * ks + e] = v[e];
            }
        }
    }
}
// 16 accumulator-layout values -> the two B operands (reduction over the tile's rows, kappa order) of the next product
template <int NP>
__device__ __forceinline__ void a16_pack16(const float (&p)[16], A16Frag<NP> (&o)[2]) {
    #pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float v[8] = {p[8 * g], p[8 * g + 1], p[8 * g + 2], p[8 * g + 3], p[8 * g + 4], p[8 * g + 5], p[8 * g + 6], p[8 * g + 7]};
        o[g] = a16_cut8<NP>(v);
    }
}
#ifndef A16_TREE
#define A16_TREE 1
#endif
// max / sum of a lane's 16 tile entries: pairwise trees (depth 4) instead of 16-deep dependent chains
__device__ __forceinline__ float a16_max16(const float (&p)[16]) {
#if A16_TREE
    float a[8];
    #pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = fmaxf(p[2 * i], p[2 * i + 1]);
    #pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = fmaxf(a[2 * i], a[2 * i + 1]);
    return fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
#else
    float m = p[0];
    #pragma unroll
    for (int i = 1; i < 16; ++i) m = p[i] > m ? p[i] : m;
    return m;
#endif
}
__device__ __forceinline__ float a16_sum16(const float (&p)[16]) {
#if A16_TREE
    float a[8];
    #pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = p[2 * i] + p[2 * i + 1];
    #pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = a[2 * i] + a[2 * i + 1];
    return (a[0] + a[1]) + (a[2] + a[3]);
#else
    float s = 0.0f;
    #pragma unroll
    for (int i = 0; i < 16; ++i) s += p[i];
    return s;
#endif
}
// a value of this lane and of its partner in the other 32-lane half (which of the two is which differs between the halves: use them
// symmetrically), by v_permlane32_swap -- a vector instruction -- instead of a round trip through the LDS crossbar (__shfl_xor)
__device__ __forceinline__ void a16_halves(float x, float &u, float &v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    u = __uint_as_float(r[0]); v = __uint_as_float(r[1]);
}
// resident waves per SIMD the register allocation aims at (measured on MI355X, tools/exp/attn_probe.py, one launch at the train step's
// size: forward 0.866 ms at 2, 0.824 at 3; dQ + dK/dV 2.447 at 2 / 2, 2.631 with dQ at 3 -- it spills there)
#ifdef EMLOCO_EMU
#define A16_WAVES(n)
#else
#define A16_WAVES(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif
#ifndef A16_WPE_FWD
#define A16_WPE_FWD 3
#endif
#ifndef A16_WPE_DQ
#define A16_WPE_DQ 2
#endif
#ifndef A16_WPE_DQ2
#define A16_WPE_DQ2 2
#endif
#ifndef A16_WPE_DKV2
#define A16_WPE_DKV2 2
#endif
// keep factors (1 / 0) of a lane's 16 entries of a tile: entry r = (tile row kappa(r, hi), own row) -- `rows_are_keys`: the tile's rows
// are keys and the lane's own row is the query (forward, dQ), else the tile's rows are queries and the own row is a key (dK / dV).
// One hash serves four adjacent KEYS, a byte each (at_keep_bit): with keys along the registers that is registers 4 q .. 4 q + 3 (keys
// 8 q + 4 hi + 0 .. 3 of the tile); with queries along the registers every register has its own hash -- which the four lanes of a
// quad (keys own & ~3 .. + 3) share: each computes four of them and all read them through DPP.
template <bool ROWS_ARE_KEYS>
__device__ __forceinline__ void a16_keep16(const AttnArgs &a, unsigned hkey, int own, int t0, int hi, bool (&keep)[16]) {
    const unsigned thr = a.drop_thr;
    if (ROWS_ARE_KEYS) {
        const unsigned qk = hkey + (unsigned)own * 0x85EBCA6Bu;
        #pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned x = drop_hash(qk, (unsigned)((t0 >> 2) + 2 * q + hi));
            #pragma unroll
            for (int e = 0; e < 4; ++e) keep[4 * q + e] = ((x >> (8 * e)) & 0xffu) >= thr;
        }
    } else {
        const unsigned kk = drop_mul24((unsigned)own >> 2, 0xC2B2AFu);            // (drop_hash's column term: the lane's own key quad)
        const int e = own & 3;                                                     // = lane & 3: which of the quad's lanes this is, and its byte
        const unsigned q0k = hkey + (unsigned)(t0 + 4 * hi + 8 * e) * 0x85EBCA6Bu; // one slow multiply per tile; the registers' queries add constants
        unsigned xc[4];
        #pragma unroll
        for (int j = 0; j < 4; ++j) xc[j] = drop_mix24((q0k + (unsigned)j * 0x85EBCA6Bu) ^ kk);     // registers 4 e + j: queries t0 + 4 hi + 8 e + j
        #pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned x0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xc[j], 0x00, 0xF, 0xF, false);   // quad lane 0's: register j
            const unsigned x1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xc[j], 0x55, 0xF, 0xF, false);   // lane 1's: register 4 + j
            const unsigned x2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xc[j], 0xAA, 0xF, 0xF, false);   // lane 2's: register 8 + j
            const unsigned x3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)xc[j], 0xFF, 0xF, 0xF, false);   // lane 3's: register 12 + j
            keep[j] = ((x0 >> (8 * e)) & 0xffu) >= thr;
            keep[4 + j] = ((x1 >> (8 * e)) & 0xffu) >= thr;
            keep[8 + j] = ((x2 >> (8 * e)) & 0xffu) >= thr;
            keep[12 + j] = ((x3 >> (8 * e)) & 0xffu) >= thr;
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
// NP: pieces per operand (1 = bf16 operands, 3 = split mode); G: blocks of 32 queries per wave; MEM16: q|k|v in memory as bf16
template <int NP, int G, int DROP, int MEM16>
__global__ void __launch_bounds__(256) A16_WAVES(NP == 1 ? A16_WPE_FWD : 2)
attn16_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char Kt[2][NP * A16_TILE_BYTES], Vt[2][NP * A16_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) float Bs[2][AT_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.nhead, hd = bh - b * a.nhead;
    const unsigned hkey = at_head_key(a.drop_seed, (unsigned)bh);
    const long ld = 3L * a.d_model;
    const void *Q = MEM16 ? (const void *)((const unsigned short *)a.qkv + (long)b * a.S * ld + hd * AT_DH) : (const void *)(a.qkv + (long)b * a.S * ld + hd * AT_DH);
    const void *K = MEM16 ? (const void *)((const unsigned short *)Q + a.d_model) : (const void *)((const float *)Q + a.d_model);
    const void *V = MEM16 ? (const void *)((const unsigned short *)Q + 2 * a.d_model) : (const void *)((const float *)Q + 2 * a.d_model);
    const float *kb = a.key_bias ? a.key_bias + (long)b * a.S : nullptr;
    const int q0 = blockIdx.x * (128 * G) + wave * (32 * G);
    int query[G];
    A16Frag<NP> qf[G][2];
    bool live[G];
    at_f32x16 acc_o[G];
    float m[G], lsum[G];
    #pragma unroll
    for (int g = 0; g < G; ++g) {
        query[g] = q0 + 32 * g + l31;
        a16_own_row<NP, MEM16>(Q, ld, query[g], a.Sq, hi, qf[g], nullptr);
        live[g] = q0 + 32 * g < a.Sq;                        // wave-uniform: a block wholly past the live queries only helps staging
        #pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[g][r] = 0.0f;
        m[g] = AT_NEG; lsum[g] = 0.0f;
    }
    const float scale2 = a.scale * A16_LOG2E;
    const int ntiles = (a.S + AT_T - 1) / AT_T;
    // staging: thread = (tile row tid >> 3, 4 columns 4 (tid & 7)) of K and of V; thread tid < 32 the key bias of tile row tid
    const int sr = tid >> 3, sp = tid & 7, t31 = tid & 31;
    const float *kbp = kb ? kb : a.qkv;
    a16_u32x2 rk[NP], rv[NP];
    a16_fetch4<NP, MEM16>(K, (long)(sr < a.S ? sr : a.S - 1) * ld + 4 * sp, sr < a.S, rk);
    a16_fetch4<NP, MEM16>(V, (long)(sr < a.S ? sr : a.S - 1) * ld + 4 * sp, sr < a.S, rv);
    float rb = kbp[t31 < a.S ? t31 : a.S - 1];
    a16_put_rows<NP>(Kt[0], sr, sp, rk); a16_put_cols<NP>(Vt[0], sr, sp, rv);
    if (tid < AT_T) Bs[0][tid] = tid < a.S ? (kb ? rb * A16_LOG2E : 0.0f) : -INFINITY;
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * AT_T, k0n = k0 + AT_T;
        {
            const int kr = k0n + sr;
            const long kc = kr < a.S ? kr : a.S - 1;
            a16_fetch4<NP, MEM16>(K, kc * ld + 4 * sp, kr < a.S, rk); a16_fetch4<NP, MEM16>(V, kc * ld + 4 * sp, kr < a.S, rv);
            rb = kbp[k0n + t31 < a.S ? k0n + t31 : a.S - 1];
        }
        if (live[0]) {                                        // (a later block is never live without block 0)
            const A16Frag<NP> kf0 = a16_frag_n<NP>(Kt[buf], l31, hi), kf1 = a16_frag_n<NP>(Kt[buf], l31, 2 + hi);
            const A16Frag<NP> vf0 = a16_frag_n<NP>(Vt[buf], l31, hi), vf1 = a16_frag_n<NP>(Vt[buf], l31, 2 + hi);
            float bias[16];
            at_vec16(Bs[buf], hi, bias);
            #pragma unroll
            for (int g = 0; g < G; ++g) {
                at_f32x16 st;
                #pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = 0.0f;
                st = a16_mma<NP>(kf0, qf[g][0], st);              // scores^T: keys (rows) x own queries (lanes)
                st = a16_mma<NP>(kf1, qf[g][1], st);
                float p[16];
                #pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = fmaf(st[r], scale2, bias[r]);
                float tmax = fmaxf(a16_max16(p), AT_NEG);
                { float u, v; a16_halves(tmax, u, v); tmax = fmaxf(u, v); }
                const float m_new = tmax > m[g] ? tmax : m[g];
                const float alpha = A16_EXP2(m[g] - m_new);          // m = AT_NEG before the first live key: 0
                #pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = A16_EXP2(p[r] - m_new);                        // a masked key carries -inf: 0
                float tsum = a16_sum16(p);
                { float u, v; a16_halves(tsum, u, v); tsum = u + v; }
                lsum[g] = lsum[g] * alpha + tsum;
                m[g] = m_new;
                #pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[g][r] *= alpha;
                if (DROP) {                                       // the output sums the kept probabilities (scaled at the end), the normaliser all of them
                    bool keep[16];
                    a16_keep16<true>(a, hkey, query[g], k0, hi, keep);
                    #pragma unroll
                    for (int r = 0; r < 16; ++r) p[r] = keep[r] ? p[r] : 0.0f;
                }
                A16Frag<NP> pb[2];
                a16_pack16<NP>(p, pb);
                acc_o[g] = a16_mma<NP>(vf0, pb[0], acc_o[g]);     // O^T += V^T P^T
                acc_o[g] = a16_mma<NP>(vf1, pb[1], acc_o[g]);
            }
        }
        a16_put_rows<NP>(Kt[buf ^ 1], sr, sp, rk); a16_put_cols<NP>(Vt[buf ^ 1], sr, sp, rv);
        if (tid < AT_T) Bs[buf ^ 1][tid] = k0n + tid < a.S ? (kb ? rb * A16_LOG2E : 0.0f) : -INFINITY;
        __syncthreads();
    }
    #pragma unroll
    for (int g = 0; g < G; ++g)
        if (query[g] < a.Sq) {
            const float inv = lsum[g] > 0.0f ? (DROP ? a.drop_scale : 1.0f) / lsum[g] : 0.0f;       // fully masked row -> zeros ("safe softmax")
            at_store_rowT(a.out + ((long)b * a.Sq + query[g]) * a.d_model + hd * AT_DH, hi, acc_o[g], inv);
            if (hi == 0 && a.lse) a.lse[(long)bh * a.Sq + query[g]] = lsum[g] > 0.0f ? m[g] * A16_LN2 + logf(lsum[g]) : 3.0e38f;
        }
}

// ------------------------------------------------------------------------------------------------ backward 1: dQ (and D)
template <int NP, int G, int DROP, int MEM16>
__global__ void __launch_bounds__(256) A16_WAVES(NP == 2 ? A16_WPE_DQ2 : A16_WPE_DQ)
attn16_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char Kt[2][NP * A16_TILE_BYTES], Vr[2][NP * A16_TILE_BYTES], Kc[2][NP * A16_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) float Bs[2][AT_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.nhead, hd = bh - b * a.nhead;
    const unsigned hkey = at_head_key(a.drop_seed, (unsigned)bh);
    const long ld = 3L * a.d_model;
    const void *Q = MEM16 ? (const void *)((const unsigned short *)a.qkv + (long)b * a.S * ld + hd * AT_DH) : (const void *)(a.qkv + (long)b * a.S * ld + hd * AT_DH);
    const void *K = MEM16 ? (const void *)((const unsigned short *)Q + a.d_model) : (const void *)((const float *)Q + a.d_model);
    const void *V = MEM16 ? (const void *)((const unsigned short *)Q + 2 * a.d_model) : (const void *)((const float *)Q + 2 * a.d_model);
    const float *dO = a.dout + (long)b * a.Sq * a.d_model + hd * AT_DH, *O = a.out + (long)b * a.Sq * a.d_model + hd * AT_DH;
    const float *kb = a.key_bias ? a.key_bias + (long)b * a.S : nullptr;
    const int q0 = blockIdx.x * (128 * G) + wave * (32 * G);
    int query[G];
    A16Frag<NP> qf[G][2], dof[G][2];
    bool live[G];
    at_f32x16 acc[G];
    float lse2[G], dsum[G];
    #pragma unroll
    for (int g = 0; g < G; ++g) {
        query[g] = q0 + 32 * g + l31;
        live[g] = q0 + 32 * g < a.Sq;
        a16_own_row<NP, MEM16>(Q, ld, query[g], a.Sq, hi, qf[g], nullptr);
        float dor[16], orow[16];
        A16Frag<1> unused[2];
        a16_own_row<NP, 0>(dO, a.d_model, query[g], a.Sq, hi, dof[g], dor);
        a16_own_row<1, 0>(O, a.d_model, query[g], a.Sq, hi, unused, orow);
        float ds = 0.0f;                                      // D = sum_d dO O (the lane pair holds the two halves of d)
        #pragma unroll
        for (int e = 0; e < 16; ++e) ds = fmaf(dor[e], orow[e], ds);
        { float u, v; a16_halves(ds, u, v); ds = u + v; }
        dsum[g] = ds;
        const bool qok = query[g] < a.Sq;
        lse2[g] = qok ? a.lse[(long)bh * a.Sq + query[g]] * A16_LOG2E : 3.0e38f;
        if (qok && hi == 0) a.dsum[(long)bh * a.Sq + query[g]] = ds;
        #pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.0f;
    }
    const float scale2 = a.scale * A16_LOG2E;
    const float dscale = DROP ? a.drop_scale : 1.0f;
    const float sds = a.scale * dscale;
    float nsd[G];
    #pragma unroll
    for (int g = 0; g < G; ++g) nsd[g] = 0.0f - a.scale * dsum[g];
    const int ntiles = (a.S + AT_T - 1) / AT_T;
    const int sr = tid >> 3, sp = tid & 7, t31 = tid & 31;
    const float *kbp = kb ? kb : a.qkv;
    a16_u32x2 rk[NP], rv[NP];
    a16_fetch4<NP, MEM16>(K, (long)(sr < a.S ? sr : a.S - 1) * ld + 4 * sp, sr < a.S, rk);
    a16_fetch4<NP, MEM16>(V, (long)(sr < a.S ? sr : a.S - 1) * ld + 4 * sp, sr < a.S, rv);
    float rb = kbp[t31 < a.S ? t31 : a.S - 1];
    a16_put_rows<NP>(Kt[0], sr, sp, rk); a16_put_cols<NP>(Kc[0], sr, sp, rk); a16_put_rows<NP>(Vr[0], sr, sp, rv);
    if (tid < AT_T) Bs[0][tid] = tid < a.S ? (kb ? rb * A16_LOG2E : 0.0f) : -INFINITY;
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * AT_T, k0n = k0 + AT_T;
        {
            const int kr = k0n + sr;
            const long kc = kr < a.S ? kr : a.S - 1;
            a16_fetch4<NP, MEM16>(K, kc * ld + 4 * sp, kr < a.S, rk); a16_fetch4<NP, MEM16>(V, kc * ld + 4 * sp, kr < a.S, rv);
            rb = kbp[k0n + t31 < a.S ? k0n + t31 : a.S - 1];
        }
        if (live[0]) {
            float bias[16];
            at_vec16(Bs[buf], hi, bias);
            #pragma unroll
            for (int g = 0; g < G; ++g) {
                at_f32x16 st, dpt;
                #pragma unroll
                for (int r = 0; r < 16; ++r) { st[r] = 0.0f; dpt[r] = 0.0f; }
                st = a16_mma<NP>(a16_frag_n<NP>(Kt[buf], l31, hi), qf[g][0], st);       // S^T
                st = a16_mma<NP>(a16_frag_n<NP>(Kt[buf], l31, 2 + hi), qf[g][1], st);
                dpt = a16_mma<NP>(a16_frag_n<NP>(Vr[buf], l31, hi), dof[g][0], dpt);    // dP^T = V dO^T
                dpt = a16_mma<NP>(a16_frag_n<NP>(Vr[buf], l31, 2 + hi), dof[g][1], dpt);
                float ds[16];
                bool keep[16];
                if (DROP) a16_keep16<true>(a, hkey, query[g], k0, hi, keep);
                // dS = scale p (keep dP / (1 - p_drop) - D) as p * fma(keep dP, scale / (1 - p_drop), -scale D): the two constants are
                // formed once per launch / block (round 6: three multiplies and a subtraction per entry were five of its eight instructions)
                #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = A16_EXP2(fmaf(st[r], scale2, bias[r]) - lse2[g]);       // masked key: -inf -> 0; row past the end: lse = 3e38 -> 0
                    float dpr = dpt[r];                                               // d loss / d (dropped probability), before its scale
                    if (DROP) dpr = keep[r] ? dpr : 0.0f;
                    ds[r] = p * fmaf(dpr, sds, nsd[g]);
                }
                A16Frag<NP> db[2];
                a16_pack16<NP>(ds, db);
                acc[g] = a16_mma<NP>(a16_frag_n<NP>(Kc[buf], l31, hi), db[0], acc[g]);  // dQ^T += K^T dS^T
                acc[g] = a16_mma<NP>(a16_frag_n<NP>(Kc[buf], l31, 2 + hi), db[1], acc[g]);
            }
        }
        a16_put_rows<NP>(Kt[buf ^ 1], sr, sp, rk); a16_put_cols<NP>(Kc[buf ^ 1], sr, sp, rk); a16_put_rows<NP>(Vr[buf ^ 1], sr, sp, rv);
        if (tid < AT_T) Bs[buf ^ 1][tid] = k0n + tid < a.S ? (kb ? rb * A16_LOG2E : 0.0f) : -INFINITY;
        __syncthreads();
    }
    #pragma unroll
    for (int g = 0; g < G; ++g)
        if (query[g] < a.Sq) {
            if (MEM16) at_store_rowT<1>((float *)((unsigned short *)a.dqkv + ((long)b * a.S + query[g]) * ld + hd * AT_DH), hi, acc[g], 1.0f);
            else at_store_rowT<0>(a.dqkv + ((long)b * a.S + query[g]) * ld + hd * AT_DH, hi, acc[g], 1.0f);
        }
}

// ------------------------------------------------------------------------------------------------ backward 2: dK, dV
template <int NP, int G, int DROP, int MEM16>
__global__ void __launch_bounds__(256) A16_WAVES(NP == 2 ? A16_WPE_DKV2 : 2)       // (left alone: 247 + 64 accumulator registers at NP = 1, G = 2: one wave per SIMD)
attn16_bwd_dkv_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char Qr[2][NP * A16_TILE_BYTES], Or[2][NP * A16_TILE_BYTES], Qc[2][NP * A16_TILE_BYTES], Oc[2][NP * A16_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) float Ls[2][AT_T], Ds[2][AT_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.nhead, hd = bh - b * a.nhead;
    const unsigned hkey = at_head_key(a.drop_seed, (unsigned)bh);
    const long ld = 3L * a.d_model;
    const void *Q = MEM16 ? (const void *)((const unsigned short *)a.qkv + (long)b * a.S * ld + hd * AT_DH) : (const void *)(a.qkv + (long)b * a.S * ld + hd * AT_DH);
    const void *K = MEM16 ? (const void *)((const unsigned short *)Q + a.d_model) : (const void *)((const float *)Q + a.d_model);
    const void *V = MEM16 ? (const void *)((const unsigned short *)Q + 2 * a.d_model) : (const void *)((const float *)Q + 2 * a.d_model);
    const float *dO = a.dout + (long)b * a.Sq * a.d_model + hd * AT_DH;
    const float *lse = a.lse + (long)bh * a.Sq, *dsm = a.dsum + (long)bh * a.Sq;
    const int k0w = blockIdx.x * (128 * G) + wave * (32 * G);
    const float dscale = DROP ? a.drop_scale : 1.0f;
    const float ldsc = DROP ? log2f(a.drop_scale) : 0.0f;      // log2 of the dropout scale: rides in the key's bias
    const float dfac = 0.0f - a.scale / dscale;                 // staged D -> -D scale (1 - p_drop)
    int key[G];
    A16Frag<NP> kf[G][2], vf[G][2];
    bool live[G];
    float bias2[G];
    at_f32x16 acc_k[G], acc_v[G];
    #pragma unroll
    for (int g = 0; g < G; ++g) {
        key[g] = k0w + 32 * g + l31;
        live[g] = k0w + 32 * g < a.S;                         // wave-uniform
        a16_own_row<NP, MEM16>(K, ld, key[g], a.S, hi, kf[g], nullptr);
        a16_own_row<NP, MEM16>(V, ld, key[g], a.S, hi, vf[g], nullptr);
        bias2[g] = -INFINITY;
        if (key[g] < a.S) bias2[g] = (a.key_bias ? a.key_bias[(long)b * a.S + key[g]] * A16_LOG2E : 0.0f) + ldsc;
        #pragma unroll
        for (int r = 0; r < 16; ++r) { acc_k[g][r] = 0.0f; acc_v[g][r] = 0.0f; }
    }
    const float scale2 = a.scale * A16_LOG2E;
    const int ntiles = (a.Sq + AT_T - 1) / AT_T;               // walks the live queries
    const int sr = tid >> 3, sp = tid & 7, t31 = tid & 31;
    a16_u32x2 rq[NP], ro[NP];
    a16_fetch4<NP, MEM16>(Q, (long)(sr < a.Sq ? sr : a.Sq - 1) * ld + 4 * sp, sr < a.Sq, rq);
    a16_fetch4<NP, 0>(dO, (long)(sr < a.Sq ? sr : a.Sq - 1) * a.d_model + 4 * sp, sr < a.Sq, ro);
    float rl = lse[t31 < a.Sq ? t31 : a.Sq - 1], rd = dsm[t31 < a.Sq ? t31 : a.Sq - 1];
    a16_put_rows<NP>(Qr[0], sr, sp, rq); a16_put_cols<NP>(Qc[0], sr, sp, rq); a16_put_rows<NP>(Or[0], sr, sp, ro); a16_put_cols<NP>(Oc[0], sr, sp, ro);
    if (tid < AT_T) { Ls[0][tid] = tid < a.Sq ? rl * A16_LOG2E : 3.0e38f; Ds[0][tid] = tid < a.Sq ? rd * dfac : 0.0f; }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, q0 = t * AT_T, q0n = q0 + AT_T;
        {
            const int qr = q0n + sr;
            const long qc = qr < a.Sq ? qr : a.Sq - 1;
            a16_fetch4<NP, MEM16>(Q, qc * ld + 4 * sp, qr < a.Sq, rq); a16_fetch4<NP, 0>(dO, qc * a.d_model + 4 * sp, qr < a.Sq, ro);
            const int qs = q0n + t31 < a.Sq ? q0n + t31 : a.Sq - 1;
            rl = lse[qs]; rd = dsm[qs];
        }
        if (live[0]) {
            float lrow[16], drow[16];
            at_vec16(Ls[buf], hi, lrow);
            at_vec16(Ds[buf], hi, drow);
            #pragma unroll
            for (int g = 0; g < G; ++g) {
                at_f32x16 s, dp;
                #pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
                s = a16_mma<NP>(a16_frag_n<NP>(Qr[buf], l31, hi), kf[g][0], s);          // S: queries (rows) x own keys (lanes)
                s = a16_mma<NP>(a16_frag_n<NP>(Qr[buf], l31, 2 + hi), kf[g][1], s);
                dp = a16_mma<NP>(a16_frag_n<NP>(Or[buf], l31, hi), vf[g][0], dp);        // dP = dO V^T
                dp = a16_mma<NP>(a16_frag_n<NP>(Or[buf], l31, 2 + hi), vf[g][1], dp);
                float p[16], ds[16];
                bool keep[16];
                if (DROP) a16_keep16<false>(a, hkey, key[g], q0, hi, keep);
                // (round 6) the probability arrives scaled by 1 / (1 - p_drop) -- its logarithm rides in the key's bias -- as dV wants
                // it; dS = scale p (keep dP / (1 - p_drop) - D) is then p' * fma(keep dP, scale, -D scale (1 - p_drop)), the second
                // constant folded into the staged D: four multiplies and a subtraction per entry become one fma and one multiply
                #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = A16_EXP2(fmaf(s[r], scale2, bias2[g]) - lrow[r]);     // p / (1 - p_drop); rows past the sequence carry lse = 3e38 -> 0
                    float dpr = dp[r];
                    if (DROP) dpr = keep[r] ? dpr : 0.0f;
                    ds[r] = pr * fmaf(dpr, a.scale, drow[r]);
                    p[r] = DROP ? (keep[r] ? pr : 0.0f) : pr;                             // dV sums the dropped probabilities
                }
                {
                    A16Frag<NP> pb[2];
                    a16_pack16<NP>(p, pb);
                    acc_v[g] = a16_mma<NP>(a16_frag_n<NP>(Oc[buf], l31, hi), pb[0], acc_v[g]);   // dV^T += dO^T P
                    acc_v[g] = a16_mma<NP>(a16_frag_n<NP>(Oc[buf], l31, 2 + hi), pb[1], acc_v[g]);
                }
                {
                    A16Frag<NP> db[2];
                    a16_pack16<NP>(ds, db);
                    acc_k[g] = a16_mma<NP>(a16_frag_n<NP>(Qc[buf], l31, hi), db[0], acc_k[g]);   // dK^T += Q^T dS
                    acc_k[g] = a16_mma<NP>(a16_frag_n<NP>(Qc[buf], l31, 2 + hi), db[1], acc_k[g]);
                }
            }
        }
        a16_put_rows<NP>(Qr[buf ^ 1], sr, sp, rq); a16_put_cols<NP>(Qc[buf ^ 1], sr, sp, rq); a16_put_rows<NP>(Or[buf ^ 1], sr, sp, ro); a16_put_cols<NP>(Oc[buf ^ 1], sr, sp, ro);
        if (tid < AT_T) { Ls[buf ^ 1][tid] = q0n + tid < a.Sq ? rl * A16_LOG2E : 3.0e38f; Ds[buf ^ 1][tid] = q0n + tid < a.Sq ? rd * dfac : 0.0f; }
        __syncthreads();
    }
    // rows that do not attend (the live-query form, Sq < S) get dQ = 0 here -- every row of the sequence is a key of this kernel; the
    // launcher's 2-D memset of the Q third of ALL rows (475 MB per layer at the train step's size) is gone with it
    at_f32x16 zero;
    #pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
    #pragma unroll
    for (int g = 0; g < G; ++g)
        if (key[g] < a.S) {
            if (MEM16) {
                unsigned short *dst = (unsigned short *)a.dqkv + ((long)b * a.S + key[g]) * ld + hd * AT_DH;
                at_store_rowT<1>((float *)(dst + a.d_model), hi, acc_k[g], 1.0f);
                at_store_rowT<1>((float *)(dst + 2 * a.d_model), hi, acc_v[g], 1.0f);
                if (key[g] >= a.Sq) at_store_rowT<1>((float *)dst, hi, zero, 1.0f);
            } else {
                float *dst = a.dqkv + ((long)b * a.S + key[g]) * ld + hd * AT_DH;
                at_store_rowT<0>(dst + a.d_model, hi, acc_k[g], 1.0f);
                at_store_rowT<0>(dst + 2 * a.d_model, hi, acc_v[g], 1.0f);
                if (key[g] >= a.Sq) at_store_rowT<0>(dst, hi, zero, 1.0f);
            }
        }
}

}  // namespace emloco
