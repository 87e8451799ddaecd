// mfma_bf16.h -- bf16 operands on the gfx950 matrix cores for the opt-in reduced-precision mode (EMLOCO_GEMM_BF16 /
// EMLOCO_ATTN_BF16): fp32 values are rounded to bf16 (v_cvt_pk_bf16_f32, round to nearest even) on their way into
// v_mfma_f32_32x32x16_bf16; accumulation stays fp32.  Lane (l & 31, h = l >> 5) supplies 8 consecutive reduction entries
// of its half; the reduction order of an MFMA chain is free, so callers only have to feed A and B with the same mapping.
// The CPU emulation header (tests/emu/hip) defines EMLOCO_EMU and supplies the same three names itself.
#ifndef EMLOCO_MFMA_BF16_H
#define EMLOCO_MFMA_BF16_H
#ifndef EMLOCO_EMU
namespace emloco {
typedef float mfma_f32x16 __attribute__((vector_size(64)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 gemm_bf16x8 __attribute__((ext_vector_type(8)));
template <class F4>
__device__ __forceinline__ gemm_bf16x8 gemm_pack_bf16(const F4 &lo, const F4 &hi) {
    typedef float vf4 __attribute__((ext_vector_type(4)));
    const vf4 l = {lo.x, lo.y, lo.z, lo.w}, h = {hi.x, hi.y, hi.z, hi.w};
    const bf16x4 a = __builtin_convertvector(l, bf16x4), b = __builtin_convertvector(h, bf16x4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ mfma_f32x16 gemm_mfma_bf16(gemm_bf16x8 a, gemm_bf16x8 b, mfma_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// The split mode (fp32 products rebuilt from bf16 pieces, predictor_kernels.hip) keeps operands as packed words: two bf16 per
// 32-bit word (low half = the first element), four words = the 8 reduction entries a lane feeds one matrix instruction.
struct __attribute__((aligned(16))) bf16w4 { unsigned x, y, z, w; };
__device__ __forceinline__ unsigned gemm_pack2_bf16(float a, float b) {                    // v_cvt_pk_bf16_f32: round to nearest even
    typedef float vf2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const vf2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ mfma_f32x16 gemm_mfma_bf16_w(const bf16w4 &a, const bf16w4 &b, mfma_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gemm_bf16x8, a), __builtin_bit_cast(gemm_bf16x8, b), c, 0, 0, 0);
}
}  // namespace emloco
#endif
#endif
