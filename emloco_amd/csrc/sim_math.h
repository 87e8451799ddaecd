// sim_math.h -- fused-multiply-add forms of the small vector / quaternion helpers, used ONLY by the rigid-body kernels
// (sim_kernels.hip maps the generic names onto these).  The library is built with -ffp-contract=off, so a fused
// operation exists only where it is spelled; the CPU checker of the rigid-body step spells the same chains, which keeps
// the HIP step on its bytes while halving the instruction count of the multiply-add heavy helpers.  The task / reset
// kernels keep the unfused helpers of dev_math.h (their references are torch formulas, compared with tolerances).
#pragma once
#include "dev_math.h"

namespace emloco {

__device__ __forceinline__ void fcross3(const float *a, const float *b, float *o) {
    const float x = fmaf(a[1], b[2], -(a[2] * b[1])), y = fmaf(a[2], b[0], -(a[0] * b[2])), z = fmaf(a[0], b[1], -(a[1] * b[0]));
    o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ float fdot3(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
__device__ __forceinline__ void fqmul(const float *a, const float *b, float *o) {
    const float x = fmaf(-a[2], b[1], fmaf(a[1], b[2], fmaf(a[0], b[3], a[3] * b[0])));
    const float y = fmaf(a[2], b[0], fmaf(a[1], b[3], fmaf(-a[0], b[2], a[3] * b[1])));
    const float z = fmaf(a[2], b[3], fmaf(-a[1], b[0], fmaf(a[0], b[1], a[3] * b[2])));
    const float w = fmaf(-a[2], b[2], fmaf(-a[1], b[1], fmaf(-a[0], b[0], a[3] * b[3])));
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
__device__ __forceinline__ void fqnormalize(float *q) {
    const float n = sqrtf(fmaf(q[3], q[3], fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0]))));
    const float s = 1.0f / n;
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
__device__ __forceinline__ void fq2mat(const float *q, float *R) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = fmaf(-2.0f, fmaf(y, y, z * z), 1.0f); R[1] = 2.0f * fmaf(x, y, -(z * w)); R[2] = 2.0f * fmaf(x, z, y * w);
    R[3] = 2.0f * fmaf(x, y, z * w); R[4] = fmaf(-2.0f, fmaf(x, x, z * z), 1.0f); R[5] = 2.0f * fmaf(y, z, -(x * w));
    R[6] = 2.0f * fmaf(x, z, -(y * w)); R[7] = 2.0f * fmaf(y, z, x * w); R[8] = fmaf(-2.0f, fmaf(x, x, y * y), 1.0f);
}
__device__ __forceinline__ void fmatvec3(const float *R, const float *v, float *o) {
    const float x = SOP3(R[0], v[0], R[1], v[1], R[2], v[2]);
    const float y = SOP3(R[3], v[0], R[4], v[1], R[5], v[2]);
    const float z = SOP3(R[6], v[0], R[7], v[1], R[8], v[2]);
    o[0] = x; o[1] = y; o[2] = z;
}
// sin / cos / atan from + - * / sqrt and fused multiply-adds (Horner steps), range reduction as in dev_math.h
__device__ __forceinline__ void fdet_sincos(float x, float *sn, float *cs) {
    const float inv_pi = 0.318309886f, pi_hi = 3.140625f, pi_lo = 9.67653589793e-4f;
    const float kf = floorf(fmaf(x, inv_pi, 0.5f));
    const float y = fmaf(-kf, pi_lo, fmaf(-kf, pi_hi, x));
    const float y2 = y * y;
    float ps = 1.0f / 6227020800.0f;
    ps = fmaf(y2, ps, -1.0f / 39916800.0f); ps = fmaf(y2, ps, 1.0f / 362880.0f); ps = fmaf(y2, ps, -1.0f / 5040.0f);
    ps = fmaf(y2, ps, 1.0f / 120.0f); ps = fmaf(y2, ps, -1.0f / 6.0f); ps = fmaf(y2, ps, 1.0f);
    float pc = -1.0f / 87178291200.0f;
    pc = fmaf(y2, pc, 1.0f / 479001600.0f); pc = fmaf(y2, pc, -1.0f / 3628800.0f); pc = fmaf(y2, pc, 1.0f / 40320.0f);
    pc = fmaf(y2, pc, -1.0f / 720.0f); pc = fmaf(y2, pc, 1.0f / 24.0f); pc = fmaf(y2, pc, -0.5f); pc = fmaf(y2, pc, 1.0f);
    const float sgn = (((long)kf) & 1) ? -1.0f : 1.0f;
    *sn = sgn * (y * ps);
    *cs = sgn * pc;
}
__device__ __forceinline__ float fdet_atan01(float t) {
    const float u = t / (1.0f + sqrtf(fmaf(t, t, 1.0f)));
    const float u2 = u * u;
    float p = 1.0f / 17.0f;
    p = fmaf(u2, p, -1.0f / 15.0f); p = fmaf(u2, p, 1.0f / 13.0f); p = fmaf(u2, p, -1.0f / 11.0f); p = fmaf(u2, p, 1.0f / 9.0f);
    p = fmaf(u2, p, -1.0f / 7.0f); p = fmaf(u2, p, 1.0f / 5.0f); p = fmaf(u2, p, -1.0f / 3.0f); p = fmaf(u2, p, 1.0f);
    return 2.0f * (u * p);
}
__device__ __forceinline__ float fdet_atan2_pos(float s, float w) {
    if (s <= w) return w > 0.0f ? fdet_atan01(s / w) : 0.0f;
    return 1.57079637f - fdet_atan01(w / s);
}
__device__ __forceinline__ void frotvec2quat(const float *e, float *q) {
    const float th2 = fdot3(e, e);
    const float th = sqrtf(th2);
    float k, c;
    if (th < 1e-4f) { k = fmaf(-th2, 1.0f / 48.0f, 0.5f); c = fmaf(-th2, 0.125f, 1.0f); }
    else { float sn; fdet_sincos(0.5f * th, &sn, &c); k = sn / th; }
    q[0] = e[0] * k; q[1] = e[1] * k; q[2] = e[2] * k; q[3] = c;
}
__device__ __forceinline__ void fquat2rotvec(const float *qin, float *e) {
    float q[4] = {qin[0], qin[1], qin[2], qin[3]};
    if (q[3] < 0.0f) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const float s = sqrtf(fdot3(q, q));
    float k;
    if (s < 1e-6f) k = 2.0f;
    else k = 2.0f * fdet_atan2_pos(s, q[3]) / s;
    e[0] = q[0] * k; e[1] = q[1] * k; e[2] = q[2] * k;
}

}  // namespace emloco
