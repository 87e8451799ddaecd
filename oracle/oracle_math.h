/*
 * oracle_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Plain-C restatement of the quaternion helpers the reference's task code is built on.
 * Quaternions are xyzw.  Every function cites the reference lines it follows; the arithmetic is
 * restated term by term (same association order) so fp32 results track the torch CPU path.
 *
 *   R1 = /root/reference/pacer/pacer/utils/torch_utils.py
 *   R2 = /root/reference/isaacgym/python/isaacgym/torch_utils.py
 */
#ifndef EMLOCO_ORACLE_MATH_H
#define EMLOCO_ORACLE_MATH_H
#include <math.h>

/* atan2 / sin / cos / acos of the torch CPU path: double-precision evaluation rounded once = correctly rounded fp32 (the
 * kernels do the same, emloco_amd/csrc/dev_math.h).  torch's own last bit is library-defined (MKL VML for sin / cos, Sleef
 * for atan2; ~5 % of values sit 1 ulp off the correctly rounded one), so this is the closest restatement there is. */
static inline float cr_sinf(float x) { return (float)sin((double)x); }
static inline float cr_cosf(float x) { return (float)cos((double)x); }
static inline float cr_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
static inline float cr_acosf(float x) { return (float)acos((double)x); }

/* R1:14-24 my_quat_rotate: a = v(2w^2-1); b = 2w (q x v); c = 2 q (q.v) */
static inline void orc_my_quat_rotate(const float *q, const float *v, float *o) {
    float w = q[3];
    float s = 2.0f * (w * w) - 1.0f;
    float cx = q[1] * v[2] - q[2] * v[1];
    float cy = q[2] * v[0] - q[0] * v[2];
    float cz = q[0] * v[1] - q[1] * v[0];
    float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    o[0] = (v[0] * s + cx * w * 2.0f) + q[0] * d * 2.0f;
    o[1] = (v[1] * s + cy * w * 2.0f) + q[1] * d * 2.0f;
    o[2] = (v[2] * s + cz * w * 2.0f) + q[2] * d * 2.0f;
}

/* R2:19-41 quat_mul, the 8-multiply form */
static inline void orc_quat_mul(const float *a, const float *b, float *o) {
    float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    float x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    float ww = (z1 + x1) * (x2 + y2);
    float yy = (w1 - y1) * (w2 + z2);
    float zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    float w = qq - ww + (z1 - y1) * (y2 - z2);
    float x = qq - xx + (x1 + w1) * (x2 + w2);
    float y = qq - yy + (w1 - x1) * (y2 + z2);
    float z = qq - zz + (z1 + y1) * (w2 - x2);
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

/* R2:49-56 quat_apply: t = 2 (xyz x b); b + w t + xyz x t */
static inline void orc_quat_apply(const float *a, const float *b, float *o) {
    float tx = (a[1] * b[2] - a[2] * b[1]) * 2.0f;
    float ty = (a[2] * b[0] - a[0] * b[2]) * 2.0f;
    float tz = (a[0] * b[1] - a[1] * b[0]) * 2.0f;
    float ux = a[1] * tz - a[2] * ty;
    float uy = a[2] * tx - a[0] * tz;
    float uz = a[0] * ty - a[1] * tx;
    o[0] = (b[0] + a[3] * tx) + ux;
    o[1] = (b[1] + a[3] * ty) + uy;
    o[2] = (b[2] + a[3] * tz) + uz;
}

/* R2:44-46 normalize (eps clamp 1e-9) */
static inline void orc_normalize(const float *x, int n, float *o) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += x[i] * x[i];
    float nrm = sqrtf(s);
    if (nrm < 1e-9f) nrm = 1e-9f;
    for (int i = 0; i < n; ++i) o[i] = x[i] / nrm;
}

/* R2:96-101 quat_from_angle_axis (normalises the axis, then re-normalises the quaternion) */
static inline void orc_quat_from_angle_axis(float angle, const float *axis, float *o) {
    float th = angle / 2.0f;
    float ax[3];
    orc_normalize(axis, 3, ax);
    float s = cr_sinf(th);
    float q[4] = {ax[0] * s, ax[1] * s, ax[2] * s, cr_cosf(th)};
    orc_normalize(q, 4, o);
}

/* R2:104-106 normalize_angle */
static inline float orc_normalize_angle(float x) { return cr_atan2f(cr_sinf(x), cr_cosf(x)); }

/* R1:137-149 calc_heading = atan2 of the rotated +x axis */
static inline float orc_calc_heading(const float *q) {
    const float ex[3] = {1.0f, 0.0f, 0.0f};
    float r[3];
    orc_my_quat_rotate(q, ex, r);
    return cr_atan2f(r[1], r[0]);
}

/* R1:151-162 / R1:164-175 */
static inline void orc_calc_heading_quat(const float *q, float *o) {
    const float ez[3] = {0.0f, 0.0f, 1.0f};
    orc_quat_from_angle_axis(orc_calc_heading(q), ez, o);
}
static inline void orc_calc_heading_quat_inv(const float *q, float *o) {
    const float ez[3] = {0.0f, 0.0f, 1.0f};
    orc_quat_from_angle_axis(-orc_calc_heading(q), ez, o);
}

/* R1:66-79 quat_to_tan_norm: [q (x) x^, q (x) z^] */
static inline void orc_quat_to_tan_norm(const float *q, float *o6) {
    const float ex[3] = {1.0f, 0.0f, 0.0f}, ez[3] = {0.0f, 0.0f, 1.0f};
    orc_my_quat_rotate(q, ex, o6);
    orc_my_quat_rotate(q, ez, o6 + 3);
}

/* R1:88-111 exp_map_to_angle_axis + exp_map_to_quat */
static inline void orc_exp_map_to_quat(const float *e, float *o) {
    float angle = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    float axis[3] = {e[0] / angle, e[1] / angle, e[2] / angle}; /* may be NaN, masked below */
    angle = orc_normalize_angle(angle);
    if (!(fabsf(angle) > 1e-5f)) {
        angle = 0.0f;
        axis[0] = 0.0f; axis[1] = 0.0f; axis[2] = 1.0f;
    }
    orc_quat_from_angle_axis(angle, axis, o);
}

/* R1:26-56 quat_to_angle_axis + quat_to_exp_map */
static inline void orc_quat_to_exp_map(const float *q, float *o) {
    float sin_theta = sqrtf(1.0f - q[3] * q[3]);
    float angle = 2.0f * cr_acosf(q[3]);
    angle = orc_normalize_angle(angle);
    if (fabsf(sin_theta) > 1e-5f) {
        o[0] = angle * (q[0] / sin_theta);
        o[1] = angle * (q[1] / sin_theta);
        o[2] = angle * (q[2] / sin_theta);
    } else {
        o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f * 1.0f;
    }
}

/* R1:113-135 slerp */
static inline void orc_slerp(const float *q0, const float *q1in, float t, float *o) {
    float q1[4] = {q1in[0], q1in[1], q1in[2], q1in[3]};
    float c = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
    if (c < 0.0f) { q1[0] = -q1[0]; q1[1] = -q1[1]; q1[2] = -q1[2]; q1[3] = -q1[3]; }
    c = fabsf(c);
    float half = cr_acosf(c);
    float s = sqrtf(1.0f - c * c);
    float ra = cr_sinf((1.0f - t) * half) / s;
    float rb = cr_sinf(t * half) / s;
    for (int i = 0; i < 4; ++i) {
        float v = ra * q0[i] + rb * q1[i];
        if (fabsf(s) < 0.001f) v = 0.5f * q0[i] + 0.5f * q1[i];
        if (fabsf(c) >= 1.0f) v = q0[i];
        o[i] = v;
    }
}

/* humanoid_pedestrain_terrain.py:1533-1538 quat_apply_yaw */
static inline void orc_quat_apply_yaw(const float *q, const float *v, float *o) {
    float qy[4] = {0.0f, 0.0f, q[2], q[3]}, qn[4];
    orc_normalize(qy, 4, qn);
    orc_quat_apply(qn, v, o);
}

#endif
