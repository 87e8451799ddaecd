// drop_hash.h -- the 32-bit mixer behind the counter-based dropout masks of the fused attention (at_keep_bit) and of the chained
// feed-forward block (ffn_keep16).  Round 5.
//
// A mask bit is a function of (seed, row, column) only, so forward and backward recompute it instead of storing it.  Rounds 3-4 used
// murmur3's 32-bit finaliser (two v_mul_lo_u32).  Measured on MI355X (tools/exp/ffn_probe.py: forward with / without the hidden-layer
// dropout, 4.75e8 hashes per launch): 43 SIMD-cycles per wave-hash -- v_mul_lo_u32 issues at a QUARTER of the rate of a plain
// vector instruction on gfx950 (8 cycles per wave64), three of them were 24 of a hash's 34 cycles, and in the attention kernels the mask was
// more than half of all vector work per probability.  The mixer below keeps the structure (fold, multiply, fold, multiply, fold) on the
// FULL-RATE 24-bit multiply (v_mul_u32_u24: low 32 bits of a 24 x 24-bit product): every fold brings the bits the next multiply
// cannot see (24..31) or has mixed least (the low ones) into play.  The row key it is applied to is itself a murmur3-mixed value
// (one slow multiply per lane and launch).  Statistical check (tools/exp/hash_quality.py: 16 head-sequences x 453 queries x 454 keys at
// p = 0.1): keep rate 0.9001, correlation between the two halves of a hash, adjacent keys, adjacent queries, adjacent heads all
// within +-1e-3 (the noise of 3.3 M samples), chi-square of both 16-bit halves ~1 per degree of freedom, variance of the kept count
// per row / column = binomial -- indistinguishable from the murmur finaliser it replaces.
#ifndef EMLOCO_DROP_HASH_H
#define EMLOCO_DROP_HASH_H
namespace emloco {
__host__ __device__ __forceinline__ unsigned drop_mul24(unsigned a, unsigned b) {          // low 32 bits of (a & 0xffffff) * (b & 0xffffff)
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (unsigned)(((unsigned long long)(a & 0xffffffu) * (unsigned long long)(b & 0xffffffu)) & 0xffffffffull);
#endif
}
__host__ __device__ __forceinline__ unsigned drop_mix24(unsigned x) {
    // (round 6: the first 24-bit multiply sees x only through the fold x ^ (x >> 16), in which the top byte meets bits 8..15: keys that
    // differ by (d << 24) | (d << 8) gave the SAME hash for every column -- two rows with one mask, ~25 000 row pairs per layer at
    // 927 744 token rows.  The key's upper 24 bits now also enter behind the first multiply, unmultiplied: a collision needs the
    // product to differ by exactly that pattern.  tools/exp/hash_quality.py: row-key collision test.)
    const unsigned hi = x >> 8;
    x ^= x >> 16; x = drop_mul24(x, 0x9E3779u); x ^= hi; x ^= x >> 13; x = drop_mul24(x, 0x85EBCBu); x ^= x >> 16;
    return x;
}
// hash of (row key, column index < 2^24): its low / high 16 bits are two independent uniform samples
__host__ __device__ __forceinline__ unsigned drop_hash(unsigned row_key, unsigned column) { return drop_mix24(row_key ^ drop_mul24(column, 0xC2B2AFu)); }
// Keep decision of element `idx` (flat index into the output of a launch with seed `seed`) at drop probability p -- the mask of the
// GEMM epilogues' inverted dropout, of emloco_act_bwd* (which recompute it) and of emloco_dropout_keep_mask (the host's copy).  Rounds
// 2-4 ran murmur3's 64-bit finaliser per element: three 64-bit multiplies = a dozen quarter-rate 32-bit ones, ~150 cycles per element,
// which made the mask 40 % of the time of the feed-forward's first GEMM (64 output elements per lane and tile, K = 128).  Now: one
// drop_hash per PAIR of adjacent elements (the pair index split 24 | rest: both 24-bit multiplies are full rate), 16 bits each
// against p 2^16 -- the resolution of p is 1.5e-5, as in the attention mask.
// the hash of a PAIR of adjacent elements (pair = idx >> 1): low 16 bits decide the even element, high 16 bits the odd one
__host__ __device__ __forceinline__ unsigned drop_pair_hash(unsigned seed, unsigned long long pair) {
    unsigned lo = (unsigned)pair & 0xffffffu;
    lo ^= (lo >> 9) ^ (lo << 11);                // (rows of a matrix are a power of two apart in the index: bring the bits that differ into the multiply's low end
                                                 //  -- without it the kept count per COLUMN of a 4096 x 1024 output had 1.15x the binomial variance)
    return drop_hash(seed + drop_mul24((unsigned)(pair >> 24), 0x7FEB35u), lo);
}
__host__ __device__ __forceinline__ bool drop_keep(unsigned seed, unsigned long long idx, float p) {
    const unsigned x = drop_pair_hash(seed, idx >> 1);
    return ((idx & 1ull) ? (x >> 16) : (x & 0xffffu)) >= (unsigned)(p * 65536.0f);
}
}  // namespace emloco
#endif
