/*
 * oracle_sim.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Sequential fp32 restatement of this repo's rigid-body step, one env at a time.
 * It stands where the reference calls gym.simulate (pacer/env/tasks/base_task.py:792-797, engine
 * parameters pacer/utils/config.py:143-163 and pacer/data/cfg/pacer.yaml:93-104).
 *
 * PARITY UNPINNED against the reference: gym.simulate is Isaac Gym 1.0.preview4 / PhysX 5, closed
 * source and absent from /root/reference.  What is restated here is our own documented scheme:
 *
 *   per substep h:
 *     1. forward kinematics in world axes about O = root origin (all spatial quantities share O, so
 *        tree recursions need no frame transforms)
 *     2. bias forces (gravity + velocity products), Newton-Euler in spatial form
 *     3. articulated-body factorisation of  M^ = M + diag(armature + h kd + h^2 kp)  (implicit PD)
 *     4. unconstrained velocity  v_free
 *     5. ground contacts (spheres, capsule ends, box corners vs the plane), deepest ORC_MAXC kept
 *     6. contact matrix A = J M^^-1 J^T in Gram form from per-row chain propagation, projected
 *        Gauss-Seidel with friction-cone projection, warm started per contact candidate
 *     7. velocity update by a second articulated-body solve, semi-implicit Euler integration
 *
 * Spatial vectors are [angular(3); linear(3)]; quaternions xyzw.
 */
#include <math.h>
#include <string.h>
#ifndef ORC_WH_MAX
#define ORC_WH_MAX 1.0f   /* largest link rotation per substep [rad], see bias_and_drive: 120 rad/s at h = 1/120 -- above the asset's
                           * max_angular_velocity = 100 (humanoid.py:685-688), which is therefore the cap that binds */
#endif
#include "oracle_sim.h"

#define NB ORC_NB
#define YLEN 30 /* 6 root + 3 per chain level (depth <= 8) */

typedef struct {
    /* state */
    float p0[3], q0[4], V0[6];        /* root pose and spatial velocity [w; v] */
    float qj[NB][4], wj[NB][3];       /* joint rotation / joint-frame angular velocity (index = child body) */
    /* kinematics */
    float pw[NB][3], qw[NB][4], R[NB][9], r[NB][3];
    float S[NB][3][6];
    float V[NB][6], Aacc[NB][6];
    float I6[NB][36], f[NB][6];
    /* drive */
    float tau[NB][3], dd[NB][3];
    unsigned char sat[NB][3];
    /* factorisation */
    float W[NB][18];  /* 6x3 row-major */
    float K[NB][6];   /* k00 k10 k11 k20 k21 k22 */
    float L0[36];     /* Cholesky factor of the root articulated inertia, lower, row-major */
    float L0i[6];     /* reciprocals of its diagonal */
    int depth[NB];
    /* linear-momentum bookkeeping across the substeps of one call (see bias_and_drive) */
    float Pexp[3], Pcur[3], Mtot;
    int havP;
    /* angular-momentum bookkeeping about the centre of mass (see project_angular_momentum) */
    float Lexp[3], Lcur[3], com[3];
    int havL;
} Env;

/* ---------------------------------------------------------------- small helpers */
static void cross3(const float *a, const float *b, float *o) {
    float x = fmaf(a[1], b[2], -(a[2] * b[1])), y = fmaf(a[2], b[0], -(a[0] * b[2])), z = fmaf(a[0], b[1], -(a[1] * b[0]));   /* fused forms: emloco_amd/csrc/sim_math.h */
    o[0] = x; o[1] = y; o[2] = z;
}
static float dot3(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
/* fused forms of the factorisation / solves, as in emloco_amd/csrc/dev_math.h */
static float fdot6(const float *a, const float *b) {
    return fmaf(a[5], b[5], fmaf(a[4], b[4], fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])))));
}
#define SOP2(a0, b0, a1, b1) fmaf((a1), (b1), (a0) * (b0))
#define SOP3(a0, b0, a1, b1, a2, b2) fmaf((a2), (b2), fmaf((a1), (b1), (a0) * (b0)))
#define ADD_SOP3(c, a0, b0, a1, b1, a2, b2) fmaf((a2), (b2), fmaf((a1), (b1), fmaf((a0), (b0), (c))))
#define SUB_SOP3(c, a0, b0, a1, b1, a2, b2) fmaf(-(a2), (b2), fmaf(-(a1), (b1), fmaf(-(a0), (b0), (c))))
static float dot6(const float *a, const float *b) { return fdot6(a, b); }
static void qmul(const float *a, const float *b, float *o) { /* Hamilton product, xyzw */
    float x = fmaf(-a[2], b[1], fmaf(a[1], b[2], fmaf(a[0], b[3], a[3] * b[0])));
    float y = fmaf(a[2], b[0], fmaf(a[1], b[3], fmaf(-a[0], b[2], a[3] * b[1])));
    float z = fmaf(a[2], b[3], fmaf(-a[1], b[0], fmaf(a[0], b[1], a[3] * b[2])));
    float w = fmaf(-a[2], b[2], fmaf(-a[1], b[1], fmaf(-a[0], b[0], a[3] * b[3])));
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static void qnormalize(float *q) {
    float n = sqrtf(fmaf(q[3], q[3], fmaf(q[2], q[2], fmaf(q[1], q[1], q[0] * q[0]))));
    float s = 1.0f / n;
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}
static void q2mat(const float *q, float *R) { /* row-major */
    float x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = fmaf(-2.0f, fmaf(y, y, z * z), 1.0f); R[1] = 2.0f * fmaf(x, y, -(z * w)); R[2] = 2.0f * fmaf(x, z, y * w);
    R[3] = 2.0f * fmaf(x, y, z * w); R[4] = fmaf(-2.0f, fmaf(x, x, z * z), 1.0f); R[5] = 2.0f * fmaf(y, z, -(x * w));
    R[6] = 2.0f * fmaf(x, z, -(y * w)); R[7] = 2.0f * fmaf(y, z, x * w); R[8] = fmaf(-2.0f, fmaf(x, x, y * y), 1.0f);
}
static void matvec3(const float *R, const float *v, float *o) {
    float x = SOP3(R[0], v[0], R[1], v[1], R[2], v[2]);
    float y = SOP3(R[3], v[0], R[4], v[1], R[5], v[2]);
    float z = SOP3(R[6], v[0], R[7], v[1], R[8], v[2]);
    o[0] = x; o[1] = y; o[2] = z;
}
/* Deterministic sin/cos/atan built from +,-,*,/ and sqrt only (all correctly rounded in IEEE fp32), so the
 * GPU kernel, which uses the same operation sequence, reproduces the oracle bit for bit.  libm / ocml
 * transcendentals differ in the last ulp, and a falling humanoid amplifies that chaotically. */
static void det_sincos(float x, float *sn, float *cs) {
    /* reduce to [-pi/2, pi/2]: x = k*pi + y; Horner steps as fused multiply-adds */
    const float inv_pi = 0.318309886f, pi_hi = 3.140625f, pi_lo = 9.67653589793e-4f;
    float kf = floorf(fmaf(x, inv_pi, 0.5f));
    float y = fmaf(-kf, pi_lo, fmaf(-kf, pi_hi, x));
    float y2 = y * y;
    float ps = 1.0f / 6227020800.0f;
    ps = fmaf(y2, ps, -1.0f / 39916800.0f); ps = fmaf(y2, ps, 1.0f / 362880.0f); ps = fmaf(y2, ps, -1.0f / 5040.0f);
    ps = fmaf(y2, ps, 1.0f / 120.0f); ps = fmaf(y2, ps, -1.0f / 6.0f); ps = fmaf(y2, ps, 1.0f);
    float pc = -1.0f / 87178291200.0f;
    pc = fmaf(y2, pc, 1.0f / 479001600.0f); pc = fmaf(y2, pc, -1.0f / 3628800.0f); pc = fmaf(y2, pc, 1.0f / 40320.0f);
    pc = fmaf(y2, pc, -1.0f / 720.0f); pc = fmaf(y2, pc, 1.0f / 24.0f); pc = fmaf(y2, pc, -0.5f); pc = fmaf(y2, pc, 1.0f);
    float sgn = (((long)kf) & 1) ? -1.0f : 1.0f;
    *sn = sgn * (y * ps);
    *cs = sgn * pc;
}
/* atan(t) for t in [0, 1] */
static float det_atan01(float t) {
    float u = t / (1.0f + sqrtf(fmaf(t, t, 1.0f))); /* half-angle: atan(t) = 2 atan(u), u <= 0.4143 */
    float u2 = u * u;
    float p = 1.0f / 17.0f;
    p = fmaf(u2, p, -1.0f / 15.0f); p = fmaf(u2, p, 1.0f / 13.0f); p = fmaf(u2, p, -1.0f / 11.0f); p = fmaf(u2, p, 1.0f / 9.0f);
    p = fmaf(u2, p, -1.0f / 7.0f); p = fmaf(u2, p, 1.0f / 5.0f); p = fmaf(u2, p, -1.0f / 3.0f); p = fmaf(u2, p, 1.0f);
    return 2.0f * (u * p);
}
/* atan2(s, w) for s >= 0, w >= 0 */
static float det_atan2_pos(float s, float w) {
    if (s <= w) return w > 0.0f ? det_atan01(s / w) : 0.0f;
    return 1.57079637f - det_atan01(w / s);
}
/* rotation vector -> quaternion */
static void rotvec2quat(const float *e, float *q) {
    float th2 = dot3(e, e);
    float th = sqrtf(th2);
    float k, c;
    if (th < 1e-4f) { k = fmaf(-th2, 1.0f / 48.0f, 0.5f); c = fmaf(-th2, 0.125f, 1.0f); }
    else { float sn; det_sincos(0.5f * th, &sn, &c); k = sn / th; }
    q[0] = e[0] * k; q[1] = e[1] * k; q[2] = e[2] * k; q[3] = c;
}
/* quaternion -> rotation vector with angle in [0, pi] */
static void quat2rotvec(const float *qin, float *e) {
    float q[4] = {qin[0], qin[1], qin[2], qin[3]};
    if (q[3] < 0.0f) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    float s = sqrtf(dot3(q, q));
    float k;
    if (s < 1e-6f) k = 2.0f;
    else k = 2.0f * det_atan2_pos(s, q[3]) / s;
    e[0] = q[0] * k; e[1] = q[1] * k; e[2] = q[2] * k;
}

/* ---------------------------------------------------------------- model access */
typedef struct {
    const int32_t *parent, *gtype;
    const float *off, *mass, *com, *inertia, *ga, *gb, *gr, *kp, *kd, *arm, *eff;
    int sc_n;
    const uint8_t *sc_pairs;
    const float *sc_a, *sc_b, *sc_r;
    float sc_k, sc_c, sc_max_pen, sc_mu;
    int sc_nseg;
    const uint8_t *sc_segbody;
    const int16_t *hf;
    int hf_nx, hf_ny;
    float hf_hs, hf_inv_hs, hf_vs, hf_ox, hf_oy;
    const uint8_t *hf_mv;
} EnvModel;

static EnvModel env_model(const OrcModel *m, int e) {
    EnvModel x;
    x.parent = m->parent; x.gtype = m->geom_type;
    x.off = m->joint_off + (long)e * NB * 3; x.mass = m->mass + (long)e * NB;
    x.com = m->com + (long)e * NB * 3; x.inertia = m->inertia + (long)e * NB * 6;
    x.ga = m->geom_a + (long)e * NB * 3; x.gb = m->geom_b + (long)e * NB * 3; x.gr = m->geom_r + (long)e * NB;
    x.kp = m->kp + (long)e * ORC_NDOF; x.kd = m->kd + (long)e * ORC_NDOF;
    x.arm = m->armature + (long)e * ORC_NDOF; x.eff = m->effort + (long)e * ORC_NDOF;
    x.hf = m->hf; x.hf_nx = m->hf_nx; x.hf_ny = m->hf_ny; x.hf_hs = m->hf_hs; x.hf_vs = m->hf_vs; x.hf_ox = m->hf_ox; x.hf_oy = m->hf_oy;
    x.hf_inv_hs = m->hf ? 1.0f / m->hf_hs : 0.0f;
    x.hf_mv = m->hf ? m->hf_mv : 0;
    x.sc_n = m->sc_n; x.sc_pairs = m->sc_pairs; x.sc_k = m->sc_k; x.sc_c = m->sc_c; x.sc_max_pen = m->sc_max_pen; x.sc_mu = m->sc_mu;
    x.sc_nseg = (m->sc_nseg > 0 && m->sc_segbody) ? m->sc_nseg : NB;
    x.sc_segbody = (m->sc_nseg > 0) ? m->sc_segbody : 0;
    x.sc_a = m->sc_n > 0 ? m->sc_cap_a + (long)e * x.sc_nseg * 3 : 0;
    x.sc_b = m->sc_n > 0 ? m->sc_cap_b + (long)e * x.sc_nseg * 3 : 0;
    x.sc_r = m->sc_n > 0 ? m->sc_cap_r + (long)e * x.sc_nseg : 0;
    return x;
}

/* ---------------------------------------------------------------- 1. kinematics */
static void kinematics(Env *s, const EnvModel *m) {
    memcpy(s->pw[0], s->p0, 12); memcpy(s->qw[0], s->q0, 16);
    q2mat(s->qw[0], s->R[0]);
    s->r[0][0] = s->r[0][1] = s->r[0][2] = 0.0f;
    s->depth[0] = 0;
    for (int i = 1; i < NB; ++i) {
        int p = m->parent[i];
        float o[3];
        s->depth[i] = s->depth[p] + 1;
        matvec3(s->R[p], m->off + i * 3, o);
        for (int k = 0; k < 3; ++k) { s->pw[i][k] = s->pw[p][k] + o[k]; s->r[i][k] = s->pw[i][k] - s->p0[k]; }
        qmul(s->qw[p], s->qj[i], s->qw[i]);
        qnormalize(s->qw[i]);
        q2mat(s->qw[i], s->R[i]);
        for (int c = 0; c < 3; ++c) { /* motion subspace: joint axes = child-frame axes in world */
            float ax[3] = {s->R[i][c], s->R[i][3 + c], s->R[i][6 + c]};
            memcpy(s->S[i][c], ax, 12);
            cross3(s->r[i], ax, s->S[i][c] + 3);
        }
    }
}

/* spatial velocities from generalized velocities */
static void velocities(Env *s, const EnvModel *m, float (*V)[6], const float *V0, float (*wj)[3]) {
    memcpy(V[0], V0, 24);
    for (int i = 1; i < NB; ++i) {
        int p = m->parent[i];
        for (int k = 0; k < 6; ++k)
            V[i][k] = V[p][k] + SOP3(s->S[i][0][k], wj[i][0], s->S[i][1][k], wj[i][1], s->S[i][2][k], wj[i][2]);
    }
}

/* rigid-body spatial inertia about O in world axes */
static void body_inertia(const Env *s, const EnvModel *m, int i, float *I6, float *cw_out) {
    float Rc[9], Ic[9], cw[3], c[3];
    const float *in = m->inertia + i * 6;
    const float Ib[9] = {in[0], in[3], in[4], in[3], in[1], in[5], in[4], in[5], in[2]};
    const float *R = s->R[i];
    /* Ic = R Ib R^T */
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Rc[a * 3 + b] = SOP3(R[a * 3], Ib[b], R[a * 3 + 1], Ib[3 + b], R[a * 3 + 2], Ib[6 + b]);
    for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b) { /* upper triangle, mirrored: exactly symmetric */
            Ic[a * 3 + b] = SOP3(Rc[a * 3], R[b * 3], Rc[a * 3 + 1], R[b * 3 + 1], Rc[a * 3 + 2], R[b * 3 + 2]);
            Ic[b * 3 + a] = Ic[a * 3 + b];
        }
    matvec3(R, m->com + i * 3, cw);
    for (int k = 0; k < 3; ++k) c[k] = s->r[i][k] + cw[k];
    if (cw_out) memcpy(cw_out, c, 12);
    float ms = m->mass[i], cc = dot3(c, c);
    memset(I6, 0, 36 * sizeof(float));
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            I6[a * 6 + b] = Ic[a * 3 + b] + ms * ((a == b ? cc : 0.0f) - c[a] * c[b]);
    /* upper-right = m [c]x ; lower-left = transpose */
    const float cx[9] = {0.0f, -c[2], c[1], c[2], 0.0f, -c[0], -c[1], c[0], 0.0f};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) { I6[a * 6 + 3 + b] = ms * cx[a * 3 + b]; I6[(3 + b) * 6 + a] = ms * cx[a * 3 + b]; }
    for (int a = 0; a < 3; ++a) I6[(3 + a) * 6 + 3 + a] = ms;
}

static void mat6vec(const float *M, const float *v, float *o) {
    float t[6];
    for (int a = 0; a < 6; ++a) t[a] = dot6(M + a * 6, v);
    memcpy(o, t, 24);
}

/* ---------------------------------------------------------------- 1b. limb-limb penalty contacts (self-collision)
 * Our own scheme (the reference's engine resolves self-contacts as PhysX constraints; parity unpinned): sphere-swept
 * segments per body, closest points (Ericson RTCD 5.1.9), force k pen - c v_n >= 0 along the normal at the middle of the
 * overlap, equal and opposite on the two bodies; the first ORC_SC_MAXHITS hits in pair order are kept.  Same operation
 * order as emloco_amd/csrc/sim_kernels.hip phase 1b / dev_math.h seg_seg_closest. */
static float sc_clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
static void seg_seg_closest(const float *p0, const float *p1, const float *q0, const float *q1, float *c1, float *c2) {
    const float EPS = 1e-12f;
    float d1[3], d2[3], r[3];
    for (int k = 0; k < 3; ++k) { d1[k] = p1[k] - p0[k]; d2[k] = q1[k] - q0[k]; r[k] = p0[k] - q0[k]; }
    const float a = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
    const float e = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
    const float f = d2[0] * r[0] + d2[1] * r[1] + d2[2] * r[2];
    float s = 0.0f, t = 0.0f;
    if (a <= EPS && e <= EPS) { s = 0.0f; t = 0.0f; }
    else if (a <= EPS) { s = 0.0f; t = sc_clamp01(f / e); }
    else {
        const float c = d1[0] * r[0] + d1[1] * r[1] + d1[2] * r[2];
        if (e <= EPS) { t = 0.0f; s = sc_clamp01(-c / a); }
        else {
            const float b = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
            const float den = a * e - b * b;
            s = den > EPS ? sc_clamp01((b * f - c * e) / den) : 0.0f;
            t = (b * s + f) / e;
            if (t < 0.0f) { t = 0.0f; s = sc_clamp01(-c / a); }
            else if (t > 1.0f) { t = 1.0f; s = sc_clamp01((b - c) / a); }
        }
    }
    for (int k = 0; k < 3; ++k) { c1[k] = p0[k] + d1[k] * s; c2[k] = q0[k] + d2[k] * t; }
}

/* test hook: when set, self_contacts records per hit [bi, bj, pt (about the root origin) 3, n 3, F_normal, F total 3] */
static float (*g_sc_info)[12] = 0;
static int g_sc_ninfo = 0;

static void self_contacts(const Env *s, const EnvModel *m, float (*fext)[6]) {
    /* collision segments: one per body, and a second one for a box much wider than thick (the ankle boxes) */
    float seg[ORC_SC_MAXSEG][7];
    memset(fext, 0, sizeof(float) * NB * 6);
    for (int i = 0; i < m->sc_nseg; ++i) {
        float pa[3], pb[3];
        const int sb = m->sc_segbody ? m->sc_segbody[i] : i;
        matvec3(s->R[sb], m->sc_a + i * 3, pa); matvec3(s->R[sb], m->sc_b + i * 3, pb);
        for (int k = 0; k < 3; ++k) { seg[i][k] = s->r[sb][k] + pa[k]; seg[i][3 + k] = s->r[sb][k] + pb[k]; }
        seg[i][6] = m->sc_r[i];
    }
    float hitw[ORC_SC_MAXHITS][6];
    int hitb[ORC_SC_MAXHITS][2], nh = 0;
    for (int q = 0; q < m->sc_n && nh < ORC_SC_MAXHITS; ++q) {
        const int si = m->sc_pairs[2 * q], sj = m->sc_pairs[2 * q + 1];          /* segments ... */
        const int bi = m->sc_segbody ? m->sc_segbody[si] : si, bj = m->sc_segbody ? m->sc_segbody[sj] : sj;   /* ... and their bodies */
        float c1[3], c2[3];
        const float rsum = seg[si][6] + seg[sj][6];
        seg_seg_closest(seg[si], seg[si] + 3, seg[sj], seg[sj] + 3, c1, c2);
        const float dv[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
        const float dist2 = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
        if (!(dist2 < rsum * rsum && dist2 > 1e-12f)) continue;
        const float dist = sqrtf(dist2);
        float pen = rsum - dist;
        const float n[3] = {dv[0] / dist, dv[1] / dist, dv[2] / dist};
        const float off = seg[sj][6] - 0.5f * pen;
        const float pt[3] = {c2[0] + n[0] * off, c2[1] + n[1] * off, c2[2] + n[2] * off};
        float wi[3], wj[3];
        cross3(s->V[bi], pt, wi); cross3(s->V[bj], pt, wj);
        const float vn = ((s->V[bi][3] + wi[0]) - (s->V[bj][3] + wj[0])) * n[0] + ((s->V[bi][4] + wi[1]) - (s->V[bj][4] + wj[1])) * n[1]
                       + ((s->V[bi][5] + wi[2]) - (s->V[bj][5] + wj[2])) * n[2];
        if (pen > m->sc_max_pen) pen = m->sc_max_pen;
        const float F = m->sc_k * pen - m->sc_c * vn;
        if (!(F > 0.0f)) continue;
        float Fv[3] = {n[0] * F, n[1] * F, n[2] * F};
        if (m->sc_mu > 0.0f) {       /* Coulomb friction capped by the contact's damper: - min(mu F / |v_t|, c) v_t */
            const float vr[3] = {(s->V[bi][3] + wi[0]) - (s->V[bj][3] + wj[0]), (s->V[bi][4] + wi[1]) - (s->V[bj][4] + wj[1]),
                                 (s->V[bi][5] + wi[2]) - (s->V[bj][5] + wj[2])};
            const float vt[3] = {vr[0] - vn * n[0], vr[1] - vn * n[1], vr[2] - vn * n[2]};
            const float vt2 = vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2];
            if (vt2 > 1e-12f) {
                float g = m->sc_mu * F / sqrtf(vt2);
                if (g > m->sc_c) g = m->sc_c;
                for (int k = 0; k < 3; ++k) Fv[k] -= g * vt[k];
            }
        }
        if (g_sc_info) {
            float *o = g_sc_info[nh];
            o[0] = (float)bi; o[1] = (float)bj;
            for (int k = 0; k < 3; ++k) { o[2 + k] = pt[k]; o[5 + k] = n[k]; o[9 + k] = Fv[k]; }
            o[8] = F;
            g_sc_ninfo = nh + 1;
        }
        cross3(pt, Fv, hitw[nh]);
        hitw[nh][3] = Fv[0]; hitw[nh][4] = Fv[1]; hitw[nh][5] = Fv[2];
        hitb[nh][0] = bi; hitb[nh][1] = bj;
        ++nh;
    }
    for (int i = 0; i < NB; ++i)
        for (int h = 0; h < nh; ++h) {
            if (hitb[h][0] == i) for (int k = 0; k < 6; ++k) fext[i][k] += hitw[h][k];
            else if (hitb[h][1] == i) for (int k = 0; k < 6; ++k) fext[i][k] -= hitw[h][k];
        }
}

static float wave_sum_order(const float *v);

/* Linear-momentum balance.  The integrator is first order in the velocity products, so the velocity of the centre of mass
 * of a tumbling body would drift (measured: 5 % of g t in a tumbling free fall).  The total linear momentum is therefore
 * carried across the substeps of a call -- P_exp = P + h (M g + sum of the contact forces), all known exactly -- and the
 * momentum the new generalized velocities actually have (known once the kinematics of the new configuration are: next
 * substep, or the final pass that writes the body states) is shifted onto it by a uniform change dv of the linear
 * velocities (root and, through it, every body).  The velocity-product accelerations see dv through v_i x S_i qd_i: their
 * linear part gains dv x (w_i - w_0). */
static void project_momentum(Env *s, const EnvModel *m) {
    float lane_m[64], lane_p[3][64];
    for (int i = 0; i < 64; ++i) {
        lane_m[i] = 0.0f; lane_p[0][i] = lane_p[1][i] = lane_p[2][i] = 0.0f;
        if (i < NB) {
            float cw[3], rc[3], wx[3];
            matvec3(s->R[i], m->com + i * 3, cw);
            for (int k = 0; k < 3; ++k) rc[k] = s->r[i][k] + cw[k];
            cross3(s->V[i], rc, wx);
            lane_m[i] = m->mass[i];
            for (int k = 0; k < 3; ++k) lane_p[k][i] = m->mass[i] * (s->V[i][3 + k] + wx[k]);
        }
    }
    s->Mtot = wave_sum_order(lane_m);
    float Pact[3];
    for (int k = 0; k < 3; ++k) Pact[k] = wave_sum_order(lane_p[k]);
    if (s->havP) {
        float dv[3], w0[3] = {s->V[0][0], s->V[0][1], s->V[0][2]};
        for (int k = 0; k < 3; ++k) dv[k] = (s->Pexp[k] - Pact[k]) / s->Mtot;
        for (int i = 0; i < NB; ++i) {
            float wr[3] = {s->V[i][0] - w0[0], s->V[i][1] - w0[1], s->V[i][2] - w0[2]}, t[3];
            cross3(dv, wr, t);
            for (int k = 0; k < 3; ++k) { s->Aacc[i][3 + k] += t[k]; s->V[i][3 + k] += dv[k]; }
        }
        for (int k = 0; k < 3; ++k) { s->V0[3 + k] += dv[k]; s->Pcur[k] = s->Pexp[k]; }
    } else {
        for (int k = 0; k < 3; ++k) s->Pcur[k] = Pact[k];
    }
}

/* Angular-momentum balance.  The velocity products are integrated explicitly (first order): a free body tumbling at 13 rad/s gained
 * 50 % kinetic energy and lost 30 % of its angular momentum within a second, which is why the link speed used to be capped at
 * 48 rad/s.  The remedy is the one the linear momentum got: the angular momentum about the centre of mass is carried across the
 * substeps of a call -- L_exp = damp (L + sum over the contact impulses of (x - c) x impulse): gravity has no moment about c, drive
 * and limb-limb forces are internal, the angular damping scales every angular rate and with it L -- and the momentum the new
 * generalized velocities actually have in the new configuration is moved onto it by a RIGID rotation rate dw of the whole body
 * about c: J dw = L_exp - L with J the composite inertia about c (joints locked), every body twist gains [dw ; c x dw] (about O),
 * which leaves the linear momentum alone.  With L exact the kinetic energy of a free rigid motion stays between L^2 / 2 I_max and
 * L^2 / 2 I_min.  The velocity-product accelerations of this substep are left as computed (dw is itself second order in h).
 * Called right after project_momentum (V and Pcur current). */
static void project_angular_momentum(Env *s, const EnvModel *m) {
    /* The twelve wave sums are taken about O (none waits for another: the centre of mass is one of them) and moved to the centre of
     * mass c afterwards: L_c = L_O - c x P,  J_c = J_O - M ((c.c) E - c c^T). */
    float lane_c[3][64], lane_l[3][64], lane_j[6][64];
    for (int i = 0; i < 64; ++i) {
        for (int k = 0; k < 3; ++k) { lane_c[k][i] = 0.0f; lane_l[k][i] = 0.0f; }
        for (int k = 0; k < 6; ++k) lane_j[k][i] = 0.0f;
        if (i < NB) {
            const float *in = m->inertia + i * 6, *R = s->R[i];
            const float Ib[9] = {in[0], in[3], in[4], in[3], in[1], in[5], in[4], in[5], in[2]};
            float cw[3], rc[3], wx[3], vcb[3], Rc[9], Ic[6], ru[3], Iw[3];
            matvec3(R, m->com + i * 3, cw);
            for (int k = 0; k < 3; ++k) rc[k] = s->r[i][k] + cw[k];
            cross3(s->V[i], rc, wx);
            for (int k = 0; k < 3; ++k) vcb[k] = s->V[i][3 + k] + wx[k];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) Rc[a * 3 + b] = SOP3(R[a * 3], Ib[b], R[a * 3 + 1], Ib[3 + b], R[a * 3 + 2], Ib[6 + b]);
            /* Ic = R Ib R^T, upper triangle: 00 11 22 01 02 12 */
            static const int ja[6] = {0, 1, 2, 0, 0, 1}, jb[6] = {0, 1, 2, 1, 2, 2};
            for (int e = 0; e < 6; ++e) {
                int a = ja[e], b = jb[e];
                Ic[e] = SOP3(Rc[a * 3], R[b * 3], Rc[a * 3 + 1], R[b * 3 + 1], Rc[a * 3 + 2], R[b * 3 + 2]);
            }
            cross3(rc, vcb, ru);
            const float *w = s->V[i];
            Iw[0] = SOP3(Ic[0], w[0], Ic[3], w[1], Ic[4], w[2]);
            Iw[1] = SOP3(Ic[3], w[0], Ic[1], w[1], Ic[5], w[2]);
            Iw[2] = SOP3(Ic[4], w[0], Ic[5], w[1], Ic[2], w[2]);
            const float ms = m->mass[i], rr = dot3(rc, rc);
            for (int k = 0; k < 3; ++k) { lane_c[k][i] = ms * rc[k]; lane_l[k][i] = Iw[k] + ms * ru[k]; }
            for (int e = 0; e < 6; ++e) {
                int a = ja[e], b = jb[e];
                lane_j[e][i] = Ic[e] + ms * ((a == b ? rr : 0.0f) - rc[a] * rc[b]);
            }
        }
    }
    float C[3], LO[3], JO[6];
    for (int k = 0; k < 3; ++k) { C[k] = wave_sum_order(lane_c[k]) / s->Mtot; s->com[k] = C[k]; }
    for (int k = 0; k < 3; ++k) LO[k] = wave_sum_order(lane_l[k]);
    for (int e = 0; e < 6; ++e) JO[e] = wave_sum_order(lane_j[e]);
    float L[3], J[6], cP[3];
    cross3(C, s->Pcur, cP);
    for (int k = 0; k < 3; ++k) L[k] = LO[k] - cP[k];
    {
        static const int ja[6] = {0, 1, 2, 0, 0, 1}, jb[6] = {0, 1, 2, 1, 2, 2};
        const float cc = dot3(C, C);
        for (int e = 0; e < 6; ++e) J[e] = JO[e] - s->Mtot * ((ja[e] == jb[e] ? cc : 0.0f) - C[ja[e]] * C[jb[e]]);
    }
    if (!s->havL) { memcpy(s->Lcur, L, 12); return; }
    /* J dw = L_exp - L by cofactors (J symmetric positive definite: 00 11 22 01 02 12) */
    const float b0 = s->Lexp[0] - L[0], b1 = s->Lexp[1] - L[1], b2 = s->Lexp[2] - L[2];
    const float c00 = J[1] * J[2] - J[5] * J[5], c01 = J[4] * J[5] - J[3] * J[2], c02 = J[3] * J[5] - J[4] * J[1];
    const float c11 = J[0] * J[2] - J[4] * J[4], c12 = J[3] * J[4] - J[0] * J[5], c22 = J[0] * J[1] - J[3] * J[3];
    const float det = SOP3(J[0], c00, J[3], c01, J[4], c02);
    if (!(det > 1e-12f)) { memcpy(s->Lcur, L, 12); return; }
    float dw[3], dvO[3];
    dw[0] = SOP3(c00, b0, c01, b1, c02, b2) / det;
    dw[1] = SOP3(c01, b0, c11, b1, c12, b2) / det;
    dw[2] = SOP3(c02, b0, c12, b1, c22, b2) / det;
    cross3(C, dw, dvO);
    for (int i = 0; i < NB; ++i)
        for (int k = 0; k < 3; ++k) { s->V[i][k] += dw[k]; s->V[i][3 + k] += dvO[k]; }
    for (int k = 0; k < 3; ++k) { s->V0[k] += dw[k]; s->V0[3 + k] += dvO[k]; s->Lcur[k] = s->Lexp[k]; }
}

/* ---------------------------------------------------------------- 2. bias forces + drive */
static void bias_and_drive(Env *s, const EnvModel *m, const OrcSimParams *prm, const float *edof,
                           const float *tgt) {
    velocities(s, m, s->V, s->V0, s->wj);
    memset(s->Aacc[0], 0, 24);
    for (int i = 1; i < NB; ++i) {
        int p = m->parent[i];
        float wv[3], vj[3], t[3], c[6];
        for (int k = 0; k < 3; ++k)
            wv[k] = SOP3(s->S[i][0][k], s->wj[i][0], s->S[i][1][k], s->wj[i][1], s->S[i][2][k], s->wj[i][2]);
        cross3(s->V[i], s->r[i], t); /* w_i x r_i */
        for (int k = 0; k < 3; ++k) vj[k] = s->V[i][3 + k] + t[k];
        cross3(s->V[p], wv, c);      /* w_p x wv */
        float t1[3], t2[3];
        cross3(vj, wv, t1);
        cross3(s->r[i], c, t2);
        for (int k = 0; k < 3; ++k) c[3 + k] = t1[k] + t2[k];
        for (int k = 0; k < 6; ++k) s->Aacc[i][k] = s->Aacc[p][k] + c[k];
    }
    /* Link angular-speed limit.  The reference caps link angular velocities (AssetOptions.max_angular_velocity = 100,
     * humanoid.py:685-688); here the cap is min(that, ORC_WH_MAX / h): the velocity-product terms (Coriolis / centrifugal
     * accelerations, V x* I V) are integrated explicitly and beyond ~0.4 rad per substep they pump energy into fast
     * spinning links.  When the fastest link of the env exceeds the cap, all angular generalized velocities (root angular
     * velocity, joint rates) are scaled down uniformly by sc so that link just meets it, and the root's linear velocity is
     * shifted so the linear momentum is unchanged.  Body twists are linear in the generalized velocities:
     * V_i = [w_i ; v_0 + (v_i - v_0)] -> [sc w_i ; v_0' + sc (v_i - v_0)], v_0' = v_0 + (1 - sc)(v_com - v_0); the
     * velocity-product accelerations are quadratic in the angular rates (they do not involve v_0). */
    float w2max = 0.0f;
    for (int i = 0; i < NB; ++i) {
        float w2 = fmaf(s->V[i][0], s->V[i][0], fmaf(s->V[i][1], s->V[i][1], s->V[i][2] * s->V[i][2]));
        if (w2 > w2max) w2max = w2;
    }
    float wlim = ORC_WH_MAX / prm->h;
    if (prm->max_ang_vel < wlim) wlim = prm->max_ang_vel;
    if (w2max > wlim * wlim) {
        const float sc = wlim / sqrtf(w2max), sc2 = sc * sc;
        float lane_m[64], lane_p[3][64];
        for (int i = 0; i < 64; ++i) {
            lane_m[i] = 0.0f; lane_p[0][i] = lane_p[1][i] = lane_p[2][i] = 0.0f;
            if (i < NB) {
                float cw[3], rc[3], wx[3];
                matvec3(s->R[i], m->com + i * 3, cw);
                for (int k = 0; k < 3; ++k) rc[k] = s->r[i][k] + cw[k];
                cross3(s->V[i], rc, wx);
                lane_m[i] = m->mass[i];
                for (int k = 0; k < 3; ++k) lane_p[k][i] = m->mass[i] * (s->V[i][3 + k] + wx[k]);
            }
        }
        const float M = wave_sum_order(lane_m);
        const float v0[3] = {s->V0[3], s->V0[4], s->V0[5]};
        float v0n[3];
        for (int k = 0; k < 3; ++k) v0n[k] = v0[k] + (1.0f - sc) * (wave_sum_order(lane_p[k]) / M - v0[k]);
        for (int i = 0; i < NB; ++i) {
            for (int k = 0; k < 3; ++k) { s->V[i][k] *= sc; s->V[i][3 + k] = v0n[k] + sc * (s->V[i][3 + k] - v0[k]); }
            for (int k = 0; k < 6; ++k) s->Aacc[i][k] *= sc2;
            if (i >= 1) for (int k = 0; k < 3; ++k) s->wj[i][k] *= sc;
        }
        for (int k = 0; k < 3; ++k) { s->V0[k] *= sc; s->V0[3 + k] = v0n[k]; }
        for (int k = 0; k < 3; ++k) s->Lexp[k] *= sc;      /* every angular rate was scaled by sc: so was the angular momentum about c */
    }
    project_momentum(s, m);
    project_angular_momentum(s, m);
    for (int i = 0; i < NB; ++i) {
        float c[3], hI[6], IA[6], x1[3], x2[3];
        body_inertia(s, m, i, s->I6[i], c);
        mat6vec(s->I6[i], s->V[i], hI);
        mat6vec(s->I6[i], s->Aacc[i], IA);
        /* V x* h = [w x h_ang + v x h_lin ; w x h_lin] */
        cross3(s->V[i], hI, x1); cross3(s->V[i] + 3, hI + 3, x2);
        for (int k = 0; k < 3; ++k) s->f[i][k] = IA[k] + x1[k] + x2[k];
        cross3(s->V[i], hI + 3, x1);
        for (int k = 0; k < 3; ++k) s->f[i][3 + k] = IA[3 + k] + x1[k];
        /* gravity as an external force at the com: f -= [c x m g ; m g] */
        float fg[3] = {0.0f, 0.0f, m->mass[i] * prm->gravity_z}, ng[3];
        cross3(c, fg, ng);
        for (int k = 0; k < 3; ++k) { s->f[i][k] -= ng[k]; s->f[i][3 + k] -= fg[k]; }
    }
    if (m->sc_n > 0) {               /* limb-limb contact wrenches as external forces: f -= [p x F ; F] */
        float fext[NB][6];
        self_contacts(s, m, fext);
        for (int i = 0; i < NB; ++i)
            for (int k = 0; k < 6; ++k) s->f[i][k] -= fext[i][k];
    }
    /* implicit PD: tau~ = kp (q* - q) - (kd + h kp) qd ; diagonal d = armature + h kd + h^2 kp (effort limits: substep()) */
    float h = prm->h;
    for (int i = 1; i < NB; ++i)
        for (int k = 0; k < 3; ++k) {
            int d = (i - 1) * 3 + k;
            float e = tgt[d] - edof[d];
            if (prm->drive_mode == 1) {      /* effort drive: the given torque within the limit, nothing implicit */
                float t = tgt[d];
                s->sat[i][k] = 1;
                s->tau[i][k] = t > m->eff[d] ? m->eff[d] : (t < -m->eff[d] ? -m->eff[d] : t);
                s->dd[i][k] = m->arm[d];
                continue;
            }
            s->sat[i][k] = 0;
            s->tau[i][k] = m->kp[d] * e - (m->kd[d] + h * m->kp[d]) * s->wj[i][k];
            s->dd[i][k] = m->arm[d] + h * m->kd[d] + h * h * m->kp[d];
        }
}

/* Effort limits.  After a solve with every drive implicit, the torque a drive delivers over the substep is
 * tau~ - (h kd + h^2 kp) qdd; where that exceeds the limit the drive becomes a constant torque at the limit (no implicit
 * terms).  Returns whether any drive changed (the caller then factorises and solves once more). */
static int saturate_drives(Env *s, const EnvModel *m, const OrcSimParams *prm, float (*qdd)[3]) {
    float h = prm->h;
    int any = 0;
    for (int i = 1; i < NB; ++i)
        for (int k = 0; k < 3; ++k) {
            int d = (i - 1) * 3 + k;
            float ti = s->tau[i][k] - (h * m->kd[d] + h * h * m->kp[d]) * qdd[i][k];
            if (!s->sat[i][k] && fabsf(ti) > m->eff[d]) {
                s->sat[i][k] = 1;
                s->tau[i][k] = ti > 0.0f ? m->eff[d] : -m->eff[d];
                s->dd[i][k] = m->arm[d];
                any = 1;
            }
        }
    return any;
}

/* ---------------------------------------------------------------- 3. articulated-body factorisation */
static void factorize(Env *s, const EnvModel *m) {
    static __thread float IA[NB][36]; /* per-thread scratch (orc_sim_step runs envs on OpenMP threads) */
    float (*IAp)[36] = IA;
    memcpy(IAp, s->I6, sizeof(float) * NB * 36);
    for (int i = NB - 1; i >= 1; --i) {
        int p = m->parent[i];
        float U[18], D[9];
        for (int c = 0; c < 3; ++c) {
            for (int a = 0; a < 6; ++a) U[a * 3 + c] = fdot6(IAp[i] + a * 6, s->S[i][c]);
        }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                float acc = 0.0f;
                for (int k = 0; k < 6; ++k) acc = fmaf(s->S[i][a][k], U[k * 3 + b], acc);
                D[a * 3 + b] = acc + (a == b ? s->dd[i][a] : 0.0f);
            }
        float l00 = sqrtf(D[0]), k00 = 1.0f / l00, l10 = D[3] * k00, l20 = D[6] * k00;
        float l11 = sqrtf(fmaf(-l10, l10, D[4])), k11 = 1.0f / l11, l21 = fmaf(-l20, l10, D[7]) * k11;
        float l22 = sqrtf(fmaf(-l21, l21, fmaf(-l20, l20, D[8]))), k22 = 1.0f / l22;
        float k10 = -l10 * k00 * k11, k21 = -l21 * k11 * k22, k20 = -SOP2(l20, k00, l21, k10) * k22;
        float *K = s->K[i];
        K[0] = k00; K[1] = k10; K[2] = k11; K[3] = k20; K[4] = k21; K[5] = k22;
        float *W = s->W[i];
        for (int a = 0; a < 6; ++a) {
            W[a * 3 + 0] = U[a * 3] * k00;
            W[a * 3 + 1] = SOP2(U[a * 3], k10, U[a * 3 + 1], k11);
            W[a * 3 + 2] = SOP3(U[a * 3], k20, U[a * 3 + 1], k21, U[a * 3 + 2], k22);
        }
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b)
                IAp[p][a * 6 + b] += SUB_SOP3(IAp[i][a * 6 + b], W[a * 3], W[b * 3], W[a * 3 + 1], W[b * 3 + 1], W[a * 3 + 2], W[b * 3 + 2]);
    }
    /* Cholesky of the 6x6 root articulated inertia */
    float *L = s->L0;
    memset(L, 0, 36 * sizeof(float));
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b <= a; ++b) {
            float acc = IAp[0][a * 6 + b];
            for (int k = 0; k < b; ++k) acc = fmaf(-L[a * 6 + k], L[b * 6 + k], acc);
            if (a == b) { L[a * 6 + b] = sqrtf(acc); s->L0i[a] = 1.0f / L[a * 6 + b]; }
            else L[a * 6 + b] = acc * s->L0i[b];
        }
}

static void root_fwd(const float *L, const float *Li, const float *b, float *y) { /* L y = b */
    for (int a = 0; a < 6; ++a) {
        float acc = b[a];
        for (int k = 0; k < a; ++k) acc = fmaf(-L[a * 6 + k], y[k], acc);
        y[a] = acc * Li[a];
    }
}
static void root_bwd(const float *L, const float *Li, const float *y, float *x) { /* L^T x = y */
    for (int a = 5; a >= 0; --a) {
        float acc = y[a];
        for (int k = a + 1; k < 6; ++k) acc = fmaf(-L[k * 6 + a], x[k], acc);
        x[a] = acc * Li[a];
    }
}

/* solve  M^ x = tau - J^T(body forces pin)  with the factorisation: returns root accel a0 and joint qdd */
static void aba_solve(const Env *s, const EnvModel *m, float (*pin)[6], float (*tau)[3], float *a0,
                      float (*qdd)[3], float (*a)[6]) {
    float pA[NB][6], uh[NB][3];
    memcpy(pA, pin, sizeof(pA));
    for (int i = NB - 1; i >= 1; --i) {
        int p = m->parent[i];
        float u[3];
        for (int c = 0; c < 3; ++c) u[c] = (tau ? tau[i][c] : 0.0f) - fdot6(s->S[i][c], pA[i]);
        const float *K = s->K[i];
        uh[i][0] = K[0] * u[0];
        uh[i][1] = SOP2(K[1], u[0], K[2], u[1]);
        uh[i][2] = SOP3(K[3], u[0], K[4], u[1], K[5], u[2]);
        const float *W = s->W[i];
        for (int k = 0; k < 6; ++k)
            pA[p][k] += ADD_SOP3(pA[i][k], W[k * 3], uh[i][0], W[k * 3 + 1], uh[i][1], W[k * 3 + 2], uh[i][2]);
    }
    float y[6], nb[6];
    for (int k = 0; k < 6; ++k) nb[k] = -pA[0][k];
    root_fwd(s->L0, s->L0i, nb, y);
    root_bwd(s->L0, s->L0i, y, a0);
    memcpy(a[0], a0, 24);
    for (int i = 1; i < NB; ++i) {
        int p = m->parent[i];
        const float *W = s->W[i], *K = s->K[i];
        float t[3];
        for (int c = 0; c < 3; ++c) {
            float acc = 0.0f;
            for (int k = 0; k < 6; ++k) acc = fmaf(W[k * 3 + c], a[p][k], acc);
            t[c] = uh[i][c] - acc;
        }
        qdd[i][0] = SOP3(K[0], t[0], K[1], t[1], K[3], t[2]);
        qdd[i][1] = SOP2(K[2], t[1], K[4], t[2]);
        qdd[i][2] = K[5] * t[2];
        for (int k = 0; k < 6; ++k)
            a[i][k] = ADD_SOP3(a[p][k], s->S[i][0][k], qdd[i][0], s->S[i][1][k], qdd[i][1], s->S[i][2][k], qdd[i][2]);
    }
}

/* ---------------------------------------------------------------- 5. contact candidates */
int orc_sim_num_candidates(const int32_t *gtype) {
    int n = 0;
    for (int b = 0; b < NB; ++b) n += gtype[b] == ORC_GEOM_SPHERE ? 1 : (gtype[b] == ORC_GEOM_CAPSULE ? 2 : 8);
    return n;
}

/* candidate k of body b in the body frame */
static void cand_local(const EnvModel *m, int b, int k, float *pt) {
    const float *a = m->ga + b * 3, *bb = m->gb + b * 3;
    if (m->gtype[b] == ORC_GEOM_SPHERE) { memcpy(pt, a, 12); }
    else if (m->gtype[b] == ORC_GEOM_CAPSULE) { memcpy(pt, k == 0 ? a : bb, 12); }
    else {
        pt[0] = a[0] + ((k & 1) ? bb[0] : -bb[0]);
        pt[1] = a[1] + ((k & 2) ? bb[1] : -bb[1]);
        pt[2] = a[2] + ((k & 4) ? bb[2] : -bb[2]);
    }
}

/* contact frame D = [normal | tangent 1 | tangent 2]; the plane ground uses z, x, y */
typedef struct { int body, cand; float x[3], dist, D[9]; } Contact;

/* height-field ground under the world point (cx, cy): height of the cell triangle's plane there and its unit normal.  Cell
 * (i, j) holds the mesh triangles (v00, v10, v11) [u >= v] and (v00, v11, v01) [u < v] (terrain_utils.py:286-350 layout);
 * beyond the map the border cell's plane extends. */
static void hf_plane(const EnvModel *m, float cx, float cy, float *zt, float *n, int *id) {
    float gx = (cx - m->hf_ox) * m->hf_inv_hs, gy = (cy - m->hf_oy) * m->hf_inv_hs;
    int i = (int)floorf(gx), j = (int)floorf(gy);
    i = i < 0 ? 0 : (i > m->hf_nx - 2 ? m->hf_nx - 2 : i);
    j = j < 0 ? 0 : (j > m->hf_ny - 2 ? m->hf_ny - 2 : j);
    float u = gx - (float)i, v = gy - (float)j;
    *id = ((i << 15) + j) * 2 + (u >= v ? 1 : 0);               /* the triangle that was used: (cell i, cell j, which half) */
    const int16_t *c = m->hf + (long)i * m->hf_ny + j;
    float h00 = m->hf_vs * (float)c[0], h01 = m->hf_vs * (float)c[1];
    float h10 = m->hf_vs * (float)c[m->hf_ny], h11 = m->hf_vs * (float)c[m->hf_ny + 1];
    float zx, zy;
    if (u >= v) { zx = h10 - h00; zy = h11 - h10; } else { zy = h01 - h00; zx = h11 - h01; }
    *zt = fmaf(v, zy, fmaf(u, zx, h00));
    float sx = zx * m->hf_inv_hs, sy = zy * m->hf_inv_hs;
    float inv = 1.0f / sqrtf(fmaf(sx, sx, fmaf(sy, sy, 1.0f)));
    n[0] = 0.0f - sx * inv; n[1] = 0.0f - sy * inv; n[2] = inv;
}

/* the triangle under a point, without its plane */
static int hf_triangle(const EnvModel *m, float cx, float cy) {
    float gx = (cx - m->hf_ox) * m->hf_inv_hs, gy = (cy - m->hf_oy) * m->hf_inv_hs;
    int i = (int)floorf(gx), j = (int)floorf(gy);
    i = i < 0 ? 0 : (i > m->hf_nx - 2 ? m->hf_nx - 2 : i);
    j = j < 0 ? 0 : (j > m->hf_ny - 2 ? m->hf_ny - 2 : j);
    float u = gx - (float)i, v = gy - (float)j;
    return ((i << 15) + j) * 2 + (u >= v ? 1 : 0);
}


/* ---- the slope-corrected mesh (terrain_utils.py:313-325; built by the task at humanoid_pedestrain_terrain.py:859-881) ----
 * Where the height step between two neighbouring samples exceeds the slope threshold the LOWER vertex is moved one cell
 * sideways, under the upper one: the cell between them collapses to a vertical face (a stair riser), the cell on the low
 * side stretches.  hf_mv carries the moves; xy below are in grid units (integers as floats), z in metres. */
typedef struct { float x, y, z; } MeshV;
static MeshV mesh_vert(const EnvModel *m, int ci, int cj) {
    long k = (long)ci * m->hf_ny + cj;
    int b = m->hf_mv[k];
    MeshV v;
    v.x = (float)(ci + (b & 3) - 1); v.y = (float)(cj + ((b >> 2) & 3) - 1); v.z = m->hf_vs * (float)m->hf[k];
    return v;
}
/* cell (ci, cj), triangle t: t = 0 (v00, v10, v11) [id half 1], t = 1 (v00, v11, v01) [id half 0] -- the mesh's winding, normals out
 * of the solid */
static void mesh_tri(const EnvModel *m, int ci, int cj, int t, MeshV *A, MeshV *B, MeshV *Cc) {
    *A = mesh_vert(m, ci, cj);
    if (t == 0) { *B = mesh_vert(m, ci + 1, cj); *Cc = mesh_vert(m, ci + 1, cj + 1); }
    else { *B = mesh_vert(m, ci + 1, cj + 1); *Cc = mesh_vert(m, ci, cj + 1); }
}

/* the mesh surface under the world point (cx, cy): the highest of the (at most 18) triangles of the 3 x 3 cells around the point's
 * regular cell that cover it; cells whose 4 x 4 vertex block carries no move take the regular-grid formula (bit-equal to the
 * uncorrected height field), and so do points no triangle covers (beyond the map) */
static void mesh_plane(const EnvModel *m, float cx, float cy, float *zt, float *n, int *id) {
    if (!m->hf_mv) { hf_plane(m, cx, cy, zt, n, id); return; }
    float gx = (cx - m->hf_ox) * m->hf_inv_hs, gy = (cy - m->hf_oy) * m->hf_inv_hs;
    int i = (int)floorf(gx), j = (int)floorf(gy);
    i = i < 0 ? 0 : (i > m->hf_nx - 2 ? m->hf_nx - 2 : i);
    j = j < 0 ? 0 : (j > m->hf_ny - 2 ? m->hf_ny - 2 : j);
    if (!(m->hf_mv[(long)i * m->hf_ny + j] & 16)) { hf_plane(m, cx, cy, zt, n, id); return; }
    int found = 0, bid = 0;
    float bz = 0.0f, bsx = 0.0f, bsy = 0.0f;
    for (int a = -1; a <= 1; ++a) {
        int ci = i + a;
        if (ci < 0 || ci > m->hf_nx - 2) continue;
        for (int b = -1; b <= 1; ++b) {
            int cj = j + b;
            if (cj < 0 || cj > m->hf_ny - 2) continue;
            for (int t = 0; t < 2; ++t) {
                MeshV A, B, Cc;
                mesh_tri(m, ci, cj, t, &A, &B, &Cc);
                float bx = B.x - A.x, by = B.y - A.y, qx = Cc.x - A.x, qy = Cc.y - A.y;
                float ar = bx * qy - by * qx;
                if (ar == 0.0f) continue;                                   /* collapsed: a vertical face, see mesh_walls */
                float px = gx - A.x, py = gy - A.y;
                float eb = px * qy - py * qx, ec = bx * py - by * px;
                int in = ar > 0.0f ? (eb >= 0.0f && ec >= 0.0f && eb + ec <= ar) : (eb <= 0.0f && ec <= 0.0f && eb + ec >= ar);
                if (!in) continue;
                float zb = B.z - A.z, zc = Cc.z - A.z;
                float sx = (zb * qy - zc * by) / ar, sy = (zc * bx - zb * qx) / ar;
                float z = fmaf(py, sy, fmaf(px, sx, A.z));
                if (!found || z > bz) { found = 1; bz = z; bsx = sx; bsy = sy; bid = ((ci << 15) + cj) * 2 + (t == 0 ? 1 : 0); }
            }
        }
    }
    if (!found) { hf_plane(m, cx, cy, zt, n, id); return; }
    *zt = bz; *id = bid;
    float sx = bsx * m->hf_inv_hs, sy = bsy * m->hf_inv_hs;
    float inv = 1.0f / sqrtf(fmaf(sx, sx, fmaf(sy, sy, 1.0f)));
    n[0] = 0.0f - sx * inv; n[1] = 0.0f - sy * inv; n[2] = inv;
}

static float dot3f(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }

/* the vertical faces of the mesh near the world point P (centre of a contact sphere): every collapsed triangle of the 3 x 3 cells
 * around P's regular cell, closest point by regions (vertex, edge, face).  *dsel / nsel hold the signed distance of P to the
 * nearest surface found so far and its normal (>= 0: P is outside the terrain solid).  Outside: a face P is in front of wins when
 * it is nearer, normal from its closest point to P.  Inside: a face P is behind wins when the foot of P's perpendicular lies
 * in it and it is nearer than the surface above, normal = the face's. */
static void mesh_walls(const EnvModel *m, const float *P, float *dsel, float *nsel) {
    float gx = (P[0] - m->hf_ox) * m->hf_inv_hs, gy = (P[1] - m->hf_oy) * m->hf_inv_hs;
    int i = (int)floorf(gx), j = (int)floorf(gy);
    i = i < 0 ? 0 : (i > m->hf_nx - 2 ? m->hf_nx - 2 : i);
    j = j < 0 ? 0 : (j > m->hf_ny - 2 ? m->hf_ny - 2 : j);
    if (!(m->hf_mv[(long)i * m->hf_ny + j] & 16)) return;
    for (int a = -1; a <= 1; ++a) {
        int ci = i + a;
        if (ci < 0 || ci > m->hf_nx - 2) continue;
        for (int b = -1; b <= 1; ++b) {
            int cj = j + b;
            if (cj < 0 || cj > m->hf_ny - 2) continue;
            for (int t = 0; t < 2; ++t) {
                MeshV A, B, Cc;
                mesh_tri(m, ci, cj, t, &A, &B, &Cc);
                if ((B.x - A.x) * (Cc.y - A.y) - (B.y - A.y) * (Cc.x - A.x) != 0.0f) continue;
                /* corners relative to P, metres */
                float va[3] = {fmaf(A.x, m->hf_hs, m->hf_ox) - P[0], fmaf(A.y, m->hf_hs, m->hf_oy) - P[1], A.z - P[2]};
                float vb[3] = {fmaf(B.x, m->hf_hs, m->hf_ox) - P[0], fmaf(B.y, m->hf_hs, m->hf_oy) - P[1], B.z - P[2]};
                float vc[3] = {fmaf(Cc.x, m->hf_hs, m->hf_ox) - P[0], fmaf(Cc.y, m->hf_hs, m->hf_oy) - P[1], Cc.z - P[2]};
                float ab[3] = {vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]}, ac[3] = {vc[0] - va[0], vc[1] - va[1], vc[2] - va[2]};
                float fn[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
                float fn2 = dot3f(fn, fn);
                if (fn2 == 0.0f) continue;                                  /* no area in space either */
                /* closest point q of the triangle to the origin (= P), by regions */
                float ap[3] = {0.0f - va[0], 0.0f - va[1], 0.0f - va[2]}, bp[3] = {0.0f - vb[0], 0.0f - vb[1], 0.0f - vb[2]};
                float cp[3] = {0.0f - vc[0], 0.0f - vc[1], 0.0f - vc[2]};
                float d1 = dot3f(ab, ap), d2 = dot3f(ac, ap), d3 = dot3f(ab, bp), d4 = dot3f(ac, bp), d5 = dot3f(ab, cp), d6 = dot3f(ac, cp);
                float q[3];
                int face = 0;
                float vcc = d1 * d4 - d3 * d2, vbb = d5 * d2 - d1 * d6, vaa = d3 * d6 - d5 * d4;
                if (d1 <= 0.0f && d2 <= 0.0f) { q[0] = va[0]; q[1] = va[1]; q[2] = va[2]; }
                else if (d3 >= 0.0f && d4 <= d3) { q[0] = vb[0]; q[1] = vb[1]; q[2] = vb[2]; }
                else if (vcc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) { float w = d1 / (d1 - d3); for (int k = 0; k < 3; ++k) q[k] = fmaf(w, ab[k], va[k]); }
                else if (d6 >= 0.0f && d5 <= d6) { q[0] = vc[0]; q[1] = vc[1]; q[2] = vc[2]; }
                else if (vbb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) { float w = d2 / (d2 - d6); for (int k = 0; k < 3; ++k) q[k] = fmaf(w, ac[k], va[k]); }
                else if (vaa <= 0.0f && d4 - d3 >= 0.0f && d5 - d6 >= 0.0f) {
                    float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
                    for (int k = 0; k < 3; ++k) q[k] = fmaf(w, vc[k] - vb[k], vb[k]);
                } else {
                    float den = 1.0f / (vaa + vbb + vcc), v = vbb * den, w = vcc * den;
                    for (int k = 0; k < 3; ++k) q[k] = fmaf(w, ac[k], fmaf(v, ab[k], va[k]));
                    face = 1;
                }
                float dv[3] = {0.0f - q[0], 0.0f - q[1], 0.0f - q[2]};
                float side = dot3f(dv, fn);
                float ifn = 1.0f / sqrtf(fn2);
                if (*dsel >= 0.0f) {
                    float dist = sqrtf(dot3f(dv, dv));
                    if (side >= 0.0f && dist < *dsel) {
                        *dsel = dist;
                        if (dist > 1.0e-6f) { float id_ = 1.0f / dist; for (int k = 0; k < 3; ++k) nsel[k] = dv[k] * id_; }
                        else for (int k = 0; k < 3; ++k) nsel[k] = fn[k] * ifn;
                    }
                } else if (side < 0.0f && face) {
                    float dw = side * ifn;
                    if (dw > *dsel) { *dsel = dw; for (int k = 0; k < 3; ++k) nsel[k] = fn[k] * ifn; }
                }
            }
        }
    }
}

static int find_contacts(const Env *s, const EnvModel *m, const OrcSimParams *prm, Contact *out) {
    Contact all[ORC_MAXCAND];
    int n = 0, cid = 0;
    for (int b = 0; b < NB; ++b) {
        int nk = m->gtype[b] == ORC_GEOM_SPHERE ? 1 : (m->gtype[b] == ORC_GEOM_CAPSULE ? 2 : 8);
        for (int k = 0; k < nk; ++k, ++cid) {
            float lp[3], wp[3];
            cand_local(m, b, k, lp);
            matvec3(s->R[b], lp, wp);
            float rad = m->gr[b];
            float z = s->pw[b][2] + wp[2];
            static const float flat[9] = {0, 0, 1, 1, 0, 0, 0, 1, 0};
            Contact c;
            float dist;
            if (!m->hf) {
                dist = (z - prm->ground_z) - rad;
                c.x[0] = s->r[b][0] + wp[0]; c.x[1] = s->r[b][1] + wp[1]; c.x[2] = (s->r[b][2] + wp[2]) - rad;
                memcpy(c.D, flat, sizeof(flat));
            } else {
                /* sphere of the candidate against the plane of the terrain triangle under its centre -- and, for a sphere with a
                 * radius, against the triangles under four probes one radius out along +-x / +-y: a probed
                 * triangle counts when the foot of the centre's perpendicular lies in it and its plane is nearer than what has been
                 * found; one contact per candidate, the nearest */
                float zt, nn[3];
                int tid0;
                float pcx = s->pw[b][0] + wp[0], pcy = s->pw[b][1] + wp[1];
                mesh_plane(m, pcx, pcy, &zt, nn, &tid0);
                float dperp = (z - zt) * nn[2];
                if (rad > 0.0f)
                    for (int q = 0; q < 4; ++q) {
                        float ex = q == 0 ? rad : (q == 1 ? 0.0f - rad : 0.0f), ey = q == 2 ? rad : (q == 3 ? 0.0f - rad : 0.0f);
                        float ztq, nq[3];
                        int tidq;
                        mesh_plane(m, pcx + ex, pcy + ey, &ztq, nq, &tidq);
                        float dq = fmaf(z - ztq, nq[2], 0.0f - fmaf(ex, nq[0], ey * nq[1]));
                        int tidf;
                        if (m->hf_mv) { float zf, nf[3]; mesh_plane(m, pcx - dq * nq[0], pcy - dq * nq[1], &zf, nf, &tidf); }
                        else tidf = hf_triangle(m, pcx - dq * nq[0], pcy - dq * nq[1]);
                        if (tidq != tid0 && tidf == tidq && dq < dperp) { dperp = dq; nn[0] = nq[0]; nn[1] = nq[1]; nn[2] = nq[2]; }
                    }
                if (m->hf_mv) {                                  /* the corrected mesh's vertical faces */
                    float P[3] = {pcx, pcy, z};
                    mesh_walls(m, P, &dperp, nn);
                }
                dist = dperp - rad;
                for (int k = 0; k < 3; ++k) c.x[k] = (s->r[b][k] + wp[k]) - rad * nn[k];
                /* frame: normal, t1 = (y x n) / |y x n|, t2 = n x t1 */
                float l2 = fmaf(nn[2], nn[2], nn[0] * nn[0]);
                c.D[0] = nn[0]; c.D[1] = nn[1]; c.D[2] = nn[2];
                if (l2 >= 0.1f) {
                    float il = 1.0f / sqrtf(l2);
                    float t1x = nn[2] * il, t1z = 0.0f - nn[0] * il;
                    c.D[3] = t1x; c.D[4] = 0.0f; c.D[5] = t1z;
                    c.D[6] = nn[1] * t1z; c.D[7] = fmaf(nn[2], t1x, -(nn[0] * t1z)); c.D[8] = 0.0f - nn[1] * t1x;
                } else {              /* a face looking along y (a riser across the y axis): t1 = (n x x) / |n x x|, t2 = n x t1 */
                    float il = 1.0f / sqrtf(fmaf(nn[2], nn[2], nn[1] * nn[1]));
                    float t1y = nn[2] * il, t1z = 0.0f - nn[1] * il;
                    c.D[3] = 0.0f; c.D[4] = t1y; c.D[5] = t1z;
                    c.D[6] = fmaf(nn[1], t1z, -(nn[2] * t1y)); c.D[7] = 0.0f - nn[0] * t1z; c.D[8] = nn[0] * t1y;
                }
            }
            if (dist < prm->contact_offset) {
                c.body = b; c.cand = cid; c.dist = dist;
                all[n++] = c;
            }
        }
    }
    /* keep the ORC_MAXC deepest (drop the largest dist, ties: highest candidate id), keep candidate order */
    while (n > ORC_MAXC) {
        int w = 0;
        for (int i = 1; i < n; ++i)
            if (all[i].dist >= all[w].dist) w = i;
        for (int i = w; i < n - 1; ++i) all[i] = all[i + 1];
        --n;
    }
    memcpy(out, all, sizeof(Contact) * n);
    return n;
}

/* Sum of 64 lane values in the order of dev_math.h: wave_sum (DPP row_mirror, row_half_mirror, quad xor 1, quad xor 2
 * inside each 16-lane row, then ((r0 + r1) + r2) + r3 over the rows). */
static float wave_sum_order(const float *v) {
    float z[4];
    for (int r = 0; r < 4; ++r) {
        const float *x = v + 16 * r;
        float w[8], q[4];
        for (int i = 0; i < 8; ++i) w[i] = x[i] + x[15 - i];
        for (int i = 0; i < 4; ++i) q[i] = w[i] + w[7 - i];
        z[r] = (q[0] + q[1]) + (q[2] + q[3]);
    }
    return ((z[0] + z[1]) + z[2]) + z[3];
}

/* ---------------------------------------------------------------- substep */
static void substep(Env *s, const EnvModel *m, const OrcSimParams *prm, const float *tgt, float *edof,
                    float *lam_ws, float *cforce, float *dforce, int last) {
    const float h = prm->h;
    float a0[6], qdd[NB][3], acc[NB][6];
    kinematics(s, m);
    bias_and_drive(s, m, prm, edof, tgt);
    factorize(s, m);
    aba_solve(s, m, s->f, s->tau, a0, qdd, acc);
    if (saturate_drives(s, m, prm, qdd)) {
        factorize(s, m);
        aba_solve(s, m, s->f, s->tau, a0, qdd, acc);
    }

    /* 4. unconstrained velocities: generalized, and per body V + h a */
    float V0f[6], wjf[NB][3], Vf[NB][6];
    for (int k = 0; k < 6; ++k) V0f[k] = fmaf(h, a0[k], s->V0[k]);
    for (int i = 1; i < NB; ++i)
        for (int k = 0; k < 3; ++k) wjf[i][k] = fmaf(h, qdd[i][k], s->wj[i][k]);
    for (int i = 0; i < NB; ++i)
        for (int k = 0; k < 6; ++k) Vf[i][k] = fmaf(h, acc[i][k], s->V[i][k]);

    /* 5./6. contacts */
    Contact con[ORC_MAXC];
    int nc = find_contacts(s, m, prm, con);
    int nr = 3 * nc;
    float J[3 * ORC_MAXC][6], Y[3 * ORC_MAXC][YLEN], rhs[3 * ORC_MAXC], lam[3 * ORC_MAXC];
    for (int c = 0; c < nc; ++c) {
        for (int d = 0; d < 3; ++d) {
            int r = 3 * c + d;
            cross3(con[c].x, con[c].D + 3 * d, J[r]);
            memcpy(J[r] + 3, con[c].D + 3 * d, 12);
            float vel = dot6(J[r], Vf[con[c].body]);
            float bias = 0.0f;
            if (d == 0) {
                float dist = con[c].dist;
                if (dist > 0.0f) bias = dist / h;
                else { bias = prm->erp * dist / h; if (bias < -prm->max_depen_vel) bias = -prm->max_depen_vel; }
            }
            rhs[r] = vel + bias;
            /* chain propagation of a unit impulse along this row */
            float p[6];
            for (int k = 0; k < 6; ++k) p[k] = -J[r][k];
            memset(Y[r], 0, sizeof(float) * YLEN);
            for (int i = con[c].body; i >= 1; i = m->parent[i]) {
                float u[3], uh[3];
                for (int a = 0; a < 3; ++a) u[a] = -dot6(s->S[i][a], p);
                const float *K = s->K[i], *W = s->W[i];
                uh[0] = K[0] * u[0]; uh[1] = SOP2(K[1], u[0], K[2], u[1]); uh[2] = SOP3(K[3], u[0], K[4], u[1], K[5], u[2]);
                int slot = 6 + 3 * (s->depth[i] - 1);
                Y[r][slot] = uh[0]; Y[r][slot + 1] = uh[1]; Y[r][slot + 2] = uh[2];
                for (int k = 0; k < 6; ++k) p[k] = ADD_SOP3(p[k], W[k * 3], uh[0], W[k * 3 + 1], uh[1], W[k * 3 + 2], uh[2]);
            }
            root_fwd(s->L0, s->L0i, p, Y[r]);
            /* warm start */
            lam[r] = prm->warm * lam_ws[con[c].cand * 3 + d];
        }
    }
    /* Gram-form contact matrix: common chain = prefix up to the depth of the lowest common ancestor */
    static __thread float A[3 * ORC_MAXC][3 * ORC_MAXC];
    for (int c1 = 0; c1 < nc; ++c1)
        for (int c2 = 0; c2 < nc; ++c2) {
            int a = con[c1].body, b = con[c2].body;
            while (s->depth[a] > s->depth[b]) a = m->parent[a];
            while (s->depth[b] > s->depth[a]) b = m->parent[b];
            while (a != b) { a = m->parent[a]; b = m->parent[b]; }
            int len = 6 + 3 * s->depth[a];
            for (int d1 = 0; d1 < 3; ++d1)
                for (int d2 = 0; d2 < 3; ++d2) {
                    float acc = 0.0f;
                    for (int k = 0; k < len; ++k) {
                        acc = fmaf(Y[3 * c1 + d1][k], Y[3 * c2 + d2][k], acc);
                        /* the GPU pads every 3-wide tree-level block to two k = 2 matrix-core steps: one 0 * 0 term */
                        if (k >= 6 && (k - 6) % 3 == 2) acc = fmaf(0.0f, 0.0f, acc);
                    }
                    A[3 * c1 + d1][3 * c2 + d2] = acc;
                }
        }
    /* projected Gauss-Seidel with an incrementally maintained residual w = rhs + A lam: a row update touches w_r only,
     * then every w_s absorbs the change through column r (one fma per row -- no reduction, so no summation order to
     * mirror).  Per contact: normal first (>= 0), the two tangents, then the friction-cone projection. */
    float ainv[3 * ORC_MAXC], w[3 * ORC_MAXC];
    for (int r = 0; r < nr; ++r) { ainv[r] = 1.0f / (A[r][r] * (1.0f + prm->cfm)); w[r] = rhs[r]; }
    for (int r = 0; r < nr; ++r) {
        float lr = lam[r];                                   /* warm start */
        if (lr != 0.0f)
            for (int q = 0; q < nr; ++q) w[q] = fmaf(A[q][r], lr, w[q]);
    }
    for (int it = 0; it < prm->n_iter; ++it)
        for (int c = 0; c < nc; ++c) {
            for (int d = 0; d < 3; ++d) {
                int r = 3 * c + d;
                float nl = fmaf(-w[r], ainv[r], lam[r]);
                if (d == 0 && nl < 0.0f) nl = 0.0f;
                float delta = nl - lam[r];
                lam[r] = nl;
                for (int q = 0; q < nr; ++q) w[q] = fmaf(A[q][r], delta, w[q]);
            }
            float l1 = lam[3 * c + 1], l2 = lam[3 * c + 2];
            float lim = prm->mu * lam[3 * c];
            float m2 = fmaf(l1, l1, l2 * l2);
            if (m2 > lim * lim) {
                float sc = lim / sqrtf(m2);
                float n1 = l1 * sc, n2 = l2 * sc, d1 = n1 - l1, d2 = n2 - l2;
                lam[3 * c + 1] = n1; lam[3 * c + 2] = n2;
                for (int q = 0; q < nr; ++q) { w[q] = fmaf(A[q][3 * c + 1], d1, w[q]); w[q] = fmaf(A[q][3 * c + 2], d2, w[q]); }
            }
        }

    /* 7. velocity update from the contact impulses (second solve), integration */
    float pin[NB][6], da0[6], dq[NB][3];
    memset(pin, 0, sizeof(pin));
    memset(lam_ws, 0, sizeof(float) * 3 * ORC_MAXCAND);
    if (last) memset(cforce, 0, sizeof(float) * NB * 3);
    for (int c = 0; c < nc; ++c)
        for (int d = 0; d < 3; ++d) {
            int r = 3 * c + d;
            for (int k = 0; k < 6; ++k) pin[con[c].body][k] = fmaf(-J[r][k], lam[r], pin[con[c].body][k]);
            lam_ws[con[c].cand * 3 + d] = lam[r];
            if (last)
                for (int k = 0; k < 3; ++k) cforce[con[c].body * 3 + k] += con[c].D[3 * d + k] * lam[r] / h;
        }
    if (nc > 0) aba_solve(s, m, pin, 0, da0, dq, acc);
    else { memset(da0, 0, sizeof(da0)); memset(dq, 0, sizeof(dq)); }
    {   /* momentum the system must have after this substep: gravity and the contact impulses are the only external ones */
        float lane_i[3][64];
        memset(lane_i, 0, sizeof(lane_i));
        for (int r = 0; r < nr; ++r)
            for (int k = 0; k < 3; ++k) lane_i[k][r] = con[r / 3].D[3 * (r % 3) + k] * lam[r];
        for (int k = 0; k < 3; ++k) s->Pexp[k] = s->Pcur[k] + (nc > 0 ? wave_sum_order(lane_i[k]) : 0.0f);
        s->Pexp[2] = fmaf(s->Mtot * prm->gravity_z, h, s->Pexp[2]);
        s->havP = 1;
    }
    float damp = 1.0f / (1.0f + h * prm->ang_damping);
    {   /* angular momentum about the centre of mass after this substep: the moments of the contact impulses, then the damping */
        float lane_t[3][64];
        memset(lane_t, 0, sizeof(lane_t));
        for (int r = 0; r < nr; ++r) {
            const Contact *cc = &con[r / 3];
            float arm[3], imp[3], t[3];
            for (int k = 0; k < 3; ++k) { arm[k] = cc->x[k] - s->com[k]; imp[k] = cc->D[3 * (r % 3) + k] * lam[r]; }
            cross3(arm, imp, t);
            for (int k = 0; k < 3; ++k) lane_t[k][r] = t[k];
        }
        for (int k = 0; k < 3; ++k) s->Lexp[k] = (s->Lcur[k] + (nc > 0 ? wave_sum_order(lane_t[k]) : 0.0f)) * damp;
        s->havL = 1;
    }

    for (int k = 0; k < 6; ++k) s->V0[k] = V0f[k] + da0[k];
    for (int i = 1; i < NB; ++i)
        for (int k = 0; k < 3; ++k) {
            int d = (i - 1) * 3 + k;
            float wn = wjf[i][k] + dq[i][k];
            if (last) { /* drive torque applied over this substep (the contact impulses moved the implicit drive along;
                         * reported within the effort limit) */
                float tq = s->sat[i][k] ? s->tau[i][k] : m->kp[d] * (tgt[d] - edof[d] - h * wn) - m->kd[d] * wn;
                dforce[d] = tq > m->eff[d] ? m->eff[d] : (tq < -m->eff[d] ? -m->eff[d] : tq);
            }
            s->wj[i][k] = wn * damp;
        }
    for (int k = 0; k < 3; ++k) s->V0[k] *= damp;
    /* clamp angular speeds */
    {
        /* (a clamped rate changes the angular momentum in a way the balance does not predict: it is skipped once) */
        float n = sqrtf(dot3(s->V0, s->V0));
        if (n > prm->max_ang_vel) { float sc = prm->max_ang_vel / n; s->V0[0] *= sc; s->V0[1] *= sc; s->V0[2] *= sc; s->havL = 0; }
        for (int i = 1; i < NB; ++i) {
            float nj = sqrtf(dot3(s->wj[i], s->wj[i]));
            if (nj > prm->max_ang_vel) { float sc = prm->max_ang_vel / nj; s->wj[i][0] *= sc; s->wj[i][1] *= sc; s->wj[i][2] *= sc; s->havL = 0; }
        }
    }
    /* semi-implicit Euler: positions with the new velocities */
    float e[3], dqt[4], qn[4];
    for (int k = 0; k < 3; ++k) { s->p0[k] += h * s->V0[3 + k]; e[k] = h * s->V0[k]; }
    /* the reference point O of the spatial quantities moves with the root origin: re-base the root twist from O to
     * O + h v  (velocity of the body-fixed point there: v + w x (h v)); without it the root's linear velocity would not
     * turn with the body and linear momentum would not be conserved */
    {
        float wxv[3];
        cross3(s->V0, s->V0 + 3, wxv);
        for (int k = 0; k < 3; ++k) s->V0[3 + k] = fmaf(h, wxv[k], s->V0[3 + k]);
    }
    rotvec2quat(e, dqt);
    qmul(dqt, s->q0, qn); qnormalize(qn); memcpy(s->q0, qn, 16);   /* world-frame w: left multiply */
    for (int i = 1; i < NB; ++i) {
        for (int k = 0; k < 3; ++k) e[k] = h * s->wj[i][k];
        rotvec2quat(e, dqt);
        qmul(s->qj[i], dqt, qn); qnormalize(qn); memcpy(s->qj[i], qn, 16); /* joint-frame w: right multiply */
        quat2rotvec(s->qj[i], edof + (i - 1) * 3);
    }
}

static void load_state(Env *s, const float *root, const float *dof) {
    s->havP = 0; s->havL = 0;
    memcpy(s->p0, root, 12); memcpy(s->q0, root + 3, 16);
    qnormalize(s->q0);
    memcpy(s->V0, root + 10, 12); memcpy(s->V0 + 3, root + 7, 12);
    for (int i = 1; i < NB; ++i) {
        float e[3] = {dof[((i - 1) * 3) * 2], dof[((i - 1) * 3 + 1) * 2], dof[((i - 1) * 3 + 2) * 2]};
        rotvec2quat(e, s->qj[i]);
        for (int k = 0; k < 3; ++k) s->wj[i][k] = dof[((i - 1) * 3 + k) * 2 + 1];
    }
}

static void write_bodies(Env *s, const EnvModel *m, float *rb) {
    kinematics(s, m);
    velocities(s, m, s->V, s->V0, s->wj);
    if (s->havP) { project_momentum(s, m); project_angular_momentum(s, m); }   /* the last substep's balances, in the configuration it ended in */
    for (int i = 0; i < NB; ++i) {
        float *o = rb + i * 13, t[3];
        memcpy(o, s->pw[i], 12); memcpy(o + 3, s->qw[i], 16);
        cross3(s->V[i], s->r[i], t); /* classical velocity of the body origin */
        for (int k = 0; k < 3; ++k) { o[7 + k] = s->V[i][3 + k] + t[k]; o[10 + k] = s->V[i][k]; }
    }
}

void orc_sim_fk(int n_env, const OrcModel *mdl, const float *root_state, const float *dof_state, float *rb_state) {
    static Env s;
    for (int e = 0; e < n_env; ++e) {
        EnvModel m = env_model(mdl, e);
        load_state(&s, root_state + (long)e * 13, dof_state + (long)e * ORC_NDOF * 2);
        write_bodies(&s, &m, rb_state + (long)e * NB * 13);
    }
}

void orc_sim_step(int n_env, const OrcSimParams *prm, const OrcModel *mdl, float *root_state, float *dof_state,
                  const float *pd_target, float *rb_state, float *contact_force, float *dof_force,
                  float *lambda_ws) {
    /* envs are independent (no inter-env contacts, humanoid.py:838-841): one OpenMP thread per slice of them; every
     * env is computed by exactly the sequential code, so the bytes do not depend on the thread count (orc_set_threads) */
    static __thread Env s;
#pragma omp parallel for schedule(dynamic, 4)
    for (int e = 0; e < n_env; ++e) {
        EnvModel m = env_model(mdl, e);
        float *root = root_state + (long)e * 13, *dof = dof_state + (long)e * ORC_NDOF * 2;
        float edof[ORC_NDOF];
        load_state(&s, root, dof);
        for (int d = 0; d < ORC_NDOF; ++d) edof[d] = dof[d * 2];
        for (int i = 1; i < NB; ++i) quat2rotvec(s.qj[i], edof + (i - 1) * 3);
        for (int k = 0; k < prm->n_sub; ++k)
            substep(&s, &m, prm, pd_target + (long)e * ORC_NDOF, edof, lambda_ws + (long)e * ORC_MAXCAND * 3,
                    contact_force + (long)e * NB * 3, dof_force + (long)e * ORC_NDOF, k == prm->n_sub - 1);
        write_bodies(&s, &m, rb_state + (long)e * NB * 13);      /* also settles the root's linear velocity (momentum balance) */
        memcpy(root, s.p0, 12); memcpy(root + 3, s.q0, 16);
        memcpy(root + 7, s.V0 + 3, 12); memcpy(root + 10, s.V0, 12);
        for (int i = 1; i < NB; ++i)
            for (int k = 0; k < 3; ++k) {
                dof[((i - 1) * 3 + k) * 2] = edof[(i - 1) * 3 + k];
                dof[((i - 1) * 3 + k) * 2 + 1] = s.wj[i][k];
            }
    }
}

/* number of OpenMP threads of orc_sim_step (0: the runtime's default = all cores); returns the count in effect */
#ifdef _OPENMP
#include <omp.h>
int orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int orc_set_threads(int n) { (void)n; return 1; }
#endif

/* ---------------------------------------------------------------- test hooks */
void orc_sim_free_accel(const OrcSimParams *prm, const OrcModel *mdl, int env, const float *root_state,
                        const float *dof_state, const float *pd_target, float *qdd75) {
    static Env s;
    EnvModel m = env_model(mdl, env);
    float edof[ORC_NDOF], a0[6], qdd[NB][3];
    load_state(&s, root_state + (long)env * 13, dof_state + (long)env * ORC_NDOF * 2);
    for (int i = 1; i < NB; ++i) quat2rotvec(s.qj[i], edof + (i - 1) * 3);
    kinematics(&s, &m);
    bias_and_drive(&s, &m, prm, edof, pd_target + (long)env * ORC_NDOF);
    factorize(&s, &m);
    float acc[NB][6];
    aba_solve(&s, &m, s.f, s.tau, a0, qdd, acc);
    memcpy(qdd75, a0, 24);
    for (int i = 1; i < NB; ++i)
        for (int k = 0; k < 3; ++k) qdd75[6 + (i - 1) * 3 + k] = qdd[i][k];
}

/* the limb-limb contacts of one env at the given state: info [ORC_SC_MAXHITS][12] as described at g_sc_info; returns the count */
int orc_sim_self_contacts(const OrcSimParams *prm, const OrcModel *mdl, int env, const float *root_state, const float *dof_state,
                          float *info) {
    static Env s;
    static float fext[NB][6];
    EnvModel m = env_model(mdl, env);
    (void)prm;
    load_state(&s, root_state + (long)env * 13, dof_state + (long)env * ORC_NDOF * 2);
    kinematics(&s, &m);
    velocities(&s, &m, s.V, s.V0, s.wj);
    if (m.sc_n <= 0) return 0;
    g_sc_info = (float (*)[12])info; g_sc_ninfo = 0;
    self_contacts(&s, &m, fext);
    g_sc_info = 0;
    return g_sc_ninfo;
}

void orc_sim_dense_dynamics(const OrcSimParams *prm, const OrcModel *mdl, int env, const float *root_state,
                            const float *dof_state, const float *pd_target, double *M, double *rhs) {
    static Env s;
    EnvModel m = env_model(mdl, env);
    float edof[ORC_NDOF];
    load_state(&s, root_state + (long)env * 13, dof_state + (long)env * ORC_NDOF * 2);
    for (int i = 1; i < NB; ++i) quat2rotvec(s.qj[i], edof + (i - 1) * 3);
    kinematics(&s, &m);
    bias_and_drive(&s, &m, prm, edof, pd_target + (long)env * ORC_NDOF);
    const int N = 75;
    memset(M, 0, sizeof(double) * N * N);
    memset(rhs, 0, sizeof(double) * N);
    for (int i = 0; i < NB; ++i) {
        /* body Jacobian (6 x 75): identity for the root dofs, S_j for every joint j on the chain of i */
        static double Jb[6][75];
        memset(Jb, 0, sizeof(Jb));
        for (int k = 0; k < 6; ++k) Jb[k][k] = 1.0;
        for (int j = i; j >= 1; j = m.parent[j])
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < 6; ++k) Jb[k][6 + (j - 1) * 3 + c] = s.S[j][c][k];
        for (int a = 0; a < N; ++a) {
            double IJ[6];
            for (int k = 0; k < 6; ++k) {
                double acc = 0.0;
                for (int l = 0; l < 6; ++l) acc += (double)s.I6[i][k * 6 + l] * Jb[l][a];
                IJ[k] = acc;
            }
            for (int b = 0; b < N; ++b) {
                double acc = 0.0;
                for (int k = 0; k < 6; ++k) acc += Jb[k][b] * IJ[k];
                M[b * N + a] += acc;
            }
            double acc = 0.0;
            for (int k = 0; k < 6; ++k) acc += Jb[k][a] * (double)s.f[i][k];
            rhs[a] -= acc;
        }
    }
    for (int i = 1; i < NB; ++i)
        for (int k = 0; k < 3; ++k) {
            int a = 6 + (i - 1) * 3 + k;
            M[a * N + a] += s.dd[i][k];
            rhs[a] += s.tau[i][k];
        }
}
