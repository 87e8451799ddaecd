// reset_kernels.hip -- fused reset of finished envs for gfx950 (see include/emloco_task.h for the reference lines).
//
// Three launches per reset call, one wave per reset env:
//   reset_sample_kernel   motion sample (frame blend, slerp), random heading / forward speed, placement, state write
//   sim_fk_kernel         forward kinematics of the reset envs on their own skeletons (sim_kernels.hip)
//   reset_finish_kernel   ground-height fix from the lowest collision point, buffer zeroing, trajectory
//                         generation (random polyline | real path, heading alignment / inversion), LocoVal input
//                         capture, AMP history back-fill
// then emloco_task_post_physics(ids, OBS | AMP_ROW) writes the reset observations.
#include <hip/hip_runtime.h>
#include "dev_math.h"
#include "emloco_types.h"
#include "../../include/emloco_task.h"

namespace emloco {

#define RNB 24
#define RNDOF 69
#define RNV EMLOCO_TRAJ_VERTS

struct FrameBlend { long f0, f1; float blend; };

// motion_lib_smpl.py:596-606 _calc_frame_blend
__device__ __forceinline__ FrameBlend frame_blend(const EmlocoResetBufs &t, int mid, float time) {
    const float len = t.motion_len[mid], dt = t.motion_dt[mid];
    const long nf = t.motion_nframes[mid];
    float phase = time / len;
    phase = phase < 0.0f ? 0.0f : (phase > 1.0f ? 1.0f : phase);
    if (time < 0.0f) time = 0.0f;
    const long i0 = (long)(phase * (float)(nf - 1));
    const long i1 = (i0 + 1 < nf - 1) ? i0 + 1 : nf - 1;
    FrameBlend fb;
    fb.blend = (time - (float)i0 * dt) / dt;
    fb.f0 = i0 + t.motion_start[mid];
    fb.f1 = i1 + t.motion_start[mid];
    return fb;
}

__device__ __forceinline__ float lerp1(float a, float b, float w) { return (1.0f - w) * a + w * b; }

// Random rows for the entries of a (compacted, -1 padded) env list without a host-side generator: row bi of `rnd` gets
// EMLOCO_RESET_RND uniforms in [0, 1) from a stateless hash of (seed, bi, k) (two rounds of the murmur3 finaliser), only
// for the entries that are present -- the caller passes a fresh seed per call.  Replaces a torch.rand of the whole
// [n_env][512] block per step (8 MB at 4096 envs) when ~25 envs finish.
__device__ __forceinline__ unsigned fmix32(unsigned x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float reset_rnd_value(unsigned seed_lo, unsigned seed_hi, int bi, int k) {     // entry k of list entry bi's row
    const unsigned row = fmix32(seed_lo ^ ((unsigned)bi * 0x9E3779B1u)) + seed_hi;
    const unsigned x = fmix32(fmix32(row ^ ((unsigned)k * 0x27D4EB2Fu)) + 0x165667B1u);
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ void reset_fill_row(unsigned seed_lo, unsigned seed_hi, int bi, float *rnd, int lane, int nlanes) {
    for (int k = lane; k < EMLOCO_RESET_RND; k += nlanes) rnd[(long)bi * EMLOCO_RESET_RND + k] = reset_rnd_value(seed_lo, seed_hi, bi, k);
}
__global__ void reset_fill_rnd_kernel(const int32_t *ids, int n, unsigned seed_lo, unsigned seed_hi, float *rnd) {
    for (int bi = blockIdx.x; bi < n; bi += gridDim.x) {
        if (ids[bi] < 0) break;
        reset_fill_row(seed_lo, seed_hi, bi, rnd, threadIdx.x, blockDim.x);
    }
}

// the motion clip and start time a reset draws from its random row (one definition: the sample and the AMP history back-fill of
// a fused launch each evaluate it)
__device__ __forceinline__ void reset_pick_motion(const EmlocoResetBufs &t, float u_motion, float u_time, int *mid_out, float *time_out) {
    int mid = (int)(u_motion * (float)t.n_motions);
    if (mid > t.n_motions - 1) mid = t.n_motions - 1;
    *mid_out = mid;
    *time_out = u_time * t.motion_len[mid];
}

// root state of a reset before the terrain height is added -- lane 0's serial part: blended motion frame, random heading and
// forward speed, placement (humanoid_pedestrain_terrain.py:526-573)
__device__ __forceinline__ void reset_sample_root(const EmlocoResetBufs &t, const FrameBlend &fb, const float *u, float *pos, float *rot, float *vel, float *ang) {
    for (int k = 0; k < 3; ++k) {
        pos[k] = lerp1(t.gts[(fb.f0 * RNB) * 3 + k], t.gts[(fb.f1 * RNB) * 3 + k], fb.blend);
        vel[k] = lerp1(t.gvs[(fb.f0 * RNB) * 3 + k], t.gvs[(fb.f1 * RNB) * 3 + k], fb.blend);
        ang[k] = lerp1(t.gavs[(fb.f0 * RNB) * 3 + k], t.gavs[(fb.f1 * RNB) * 3 + k], fb.blend);
    }
    ref_slerp(t.grs + (fb.f0 * RNB) * 4, t.grs + (fb.f1 * RNB) * 4, fb.blend, rot);
    if (t.flags & EMLOCO_RESET_RANDOM_HEADING) {        // humanoid_pedestrain_terrain.py:556-569
        const float yaw = 3.14159265358979f * (2.0f * u[EMLOCO_RND_YAW] - 1.0f);
        const float hq[4] = {0.0f, 0.0f, sinf(0.5f * yaw), cosf(0.5f * yaw)};
        float r2[4], a2[3], hh[4], v2[3];
        ref_quat_mul(hq, rot, r2);
        ref_quat_apply(hq, ang, a2);
        for (int k = 0; k < 4; ++k) rot[k] = r2[k];
        for (int k = 0; k < 3; ++k) ang[k] = a2[k];
        ref_quat_about_z(ref_calc_heading(rot), hh);
        vel[0] = u[EMLOCO_RND_SPEED] * 0.5f + 1.0f;
        ref_quat_apply(hh, vel, v2);
        for (int k = 0; k < 3; ++k) vel[k] = v2[k];
    }
    if (t.flags & EMLOCO_RESET_FIXED_LOCATION) { pos[0] = t.fixed_x; pos[1] = t.fixed_y; }
    else {
        int li = (int)(u[EMLOCO_RND_LOC] * (float)t.n_valid);
        if (li > t.n_valid - 1) li = t.n_valid - 1;
        pos[0] = t.valid_x[li]; pos[1] = t.valid_y[li];
    }
}

// reset_sample of ONE list entry by one wave (u = its random row).  Nothing here depends on WHICH env is reset: the destinations are
// that env's simulator rows, or an entry of the pool of pre-drawn episodes (chain_kernels.hip).  dof_out [69][2] (position, velocity),
// root_out [13], *gh_out ground height under the pose, *mid_out / *time_out the clip and its start time.
__device__ __forceinline__ void reset_sample_to(const EmlocoResetBufs &t, const float *u, int lane, float *dof_out, float *root_out,
                                                float *gh_out, int64_t *mid_out, float *time_out) {
    int mid; float time;
    reset_pick_motion(t, u[EMLOCO_RND_MOTION], u[EMLOCO_RND_TIME], &mid, &time);
    const FrameBlend fb = frame_blend(t, mid, time);
    if (lane >= 1 && lane < RNB) {          // joint lane-1: local rotation -> rotation vector; dof velocity
        float q[4], e[3];
        ref_slerp(t.lrs + (fb.f0 * RNB + lane) * 4, t.lrs + (fb.f1 * RNB + lane) * 4, fb.blend, q);
        ref_quat_to_exp_map(q, e);
        float *ds = dof_out + (lane - 1) * 3 * 2;
        for (int k = 0; k < 3; ++k) {
            ds[2 * k] = e[k];
            ds[2 * k + 1] = lerp1(t.dvs[fb.f0 * RNDOF + (lane - 1) * 3 + k], t.dvs[fb.f1 * RNDOF + (lane - 1) * 3 + k], fb.blend);
        }
    }
    // root state: lane 0 blends / turns it, then the nine centre-height probes run on nine lanes (as a serial loop in lane 0 they were
    // nine dependent chains of map loads, most of this phase's latency), lane 0 averages them in the order the step uses (mean9)
    float pos[3] = {0.0f, 0.0f, 0.0f}, rot[4] = {0.0f, 0.0f, 0.0f, 1.0f}, vel[3] = {0.0f, 0.0f, 0.0f}, ang[3] = {0.0f, 0.0f, 0.0f};
    if (lane == 0) reset_sample_root(t, fb, u, pos, rot, vel, ang);
    // centre height: 3x3 yaw-only probes (humanoid_pedestrain_terrain.py:607,732-759), same device functions as the step
    float ppos[3], prot[4];
    for (int k = 0; k < 3; ++k) ppos[k] = __shfl(pos[k], 0);
    for (int k = 0; k < 4; ++k) prot[k] = __shfl(rot[k], 0);
    float hk = 0.0f;
    if (lane < 9) {
        float wx, wy;
        center_probe(ppos, prot, lane, &wx, &wy);
        hk = sample_height(t.heightfield, t.hf_rows, t.hf_cols, wx, wy, t.hscale, t.vscale);
    }
    float ch[9];
    for (int k = 0; k < 9; ++k) ch[k] = __shfl(hk, k);
    if (lane == 0) {
        const float gh = mean9(ch);
        pos[2] += gh;
        *gh_out = gh;
        for (int k = 0; k < 3; ++k) { root_out[k] = pos[k]; root_out[7 + k] = vel[k]; root_out[10 + k] = ang[k]; }
        for (int k = 0; k < 4; ++k) root_out[3 + k] = rot[k];
        *mid_out = mid;
        *time_out = time;
    }
}
__device__ __forceinline__ void reset_sample_env(const EmlocoResetBufs &t, const EmlocoSimDev &s, int env, const float *u, int lane) {
    reset_sample_to(t, u, lane, s.dof_state + (long)env * RNDOF * 2, s.root_state + (long)env * 13, t.ground_h + env, t.motion_ids + env,
                    t.motion_times + env);
}

__global__ void __launch_bounds__(64)
reset_sample_kernel(EmlocoResetBufs t, EmlocoSimDev s, const int32_t *ids, int n, const float *rnd) {
    // grid-stride over the id list: a device-compacted list (emloco_task_compact_done) holds its valid entries first and
    // -1 after them, so a small grid stops at the first padding entry instead of launching one workgroup per env
    const int lane = threadIdx.x;
    for (int bi = blockIdx.x; bi < n; bi += gridDim.x) {
        const int env = ids[bi];
        if (env < 0) break;
        reset_sample_env(t, s, env, rnd + (long)bi * EMLOCO_RESET_RND, lane);
        __syncthreads();                              // LDS is reused by the next list entry
    }
}

// traj_generator.py:278-296 calc_pos at one time
__device__ __forceinline__ void r_calc_pos(const float *verts, float time, float traj_dur, float *out) {
    float phase = time / traj_dur;
    phase = phase < 0.0f ? 0.0f : (phase > 1.0f ? 1.0f : phase);
    const float seg = phase * (float)(RNV - 1);
    const long i0 = (long)floorf(seg), i1 = (long)ceilf(seg);
    const float w = seg - (float)i0;
    for (int k = 0; k < 3; ++k) out[k] = (1.0f - w) * verts[i0 * 3 + k] + w * verts[i1 * 3 + k];
}

// traj_generator.py:60-237 TrajGenerator.reset for one env (one wave): random-walk polyline from the env's random row, then
// (flags) a real-world path instead, heading alignment / inversion.  Result in sh_v (LDS), the inversion flag in t.inverted.
//
// Real paths (:121-160): the reference draws `random.sample(range(data_num), real_data_num)` -- distinct rows for the envs
// of one reset call.  Here entry bi of the call's id list takes row P_key(bi mod n_real), P_key a keyed bijection of
// [0, n_real) (4-round Feistel network over the next even power of two, cycle-walked), so the rows of one call are distinct
// as well (for n <= n_real; the reference raises when more real rows are asked for than exist, here the list wraps).
// t.real_pick (optional) overrides it with explicit rows: tests replay the reference's own sample through it.
__device__ __forceinline__ unsigned real_pick_perm(unsigned x, unsigned n, unsigned key) {
    unsigned bits = 2;
    while ((1u << bits) < n) bits += 2;
    const unsigned half = bits >> 1, mask = (1u << half) - 1u;
    do {
        unsigned l = x >> half, r = x & mask;
        for (unsigned round = 0; round < 4; ++round) {
            const unsigned f = fmix32(r * 0x9E3779B1u + key + round * 0x85EBCA6Bu) & mask;
            const unsigned nl = r;
            r = l ^ f;
            l = nl;
        }
        x = (l << half) | r;
    } while (x >= n);
    return x;
}

__device__ __forceinline__ void reset_trajectory(const EmlocoResetBufs &t, const float *u, int bi, uint8_t *inv_out, int lane,
                                                 float ipx, float ipy, float rvx, float rvy, float rvz, float (*sh_v)[3], float *sm6) {
    // The heading / speed recurrences (clamped random walks) are sequential but cheap; the 100 cos / sin evaluations are
    // the cost, so lane 0 only produces theta_i and the segment length, all lanes evaluate the steps, and lane 0 adds
    // them up in the original order (same floating-point result as the one-lane loop, ~8x shorter critical path).
    float *sh_th = sm6, *sh_seg = sm6 + RNV;          // sm6: 6 x RNV floats of the workgroup's LDS
    // the four random streams of the walk are staged in LDS by all lanes first: lane 0's loop then runs on LDS latency
    // instead of one dependent global load per stream and step (that was most of this kernel's 30 us)
    float *sh_ud = sm6 + 2 * RNV, *sh_ub = sm6 + 3 * RNV, *sh_us = sm6 + 4 * RNV, *sh_ua = sm6 + 5 * RNV;
    for (int i = lane; i < RNV - 1; i += 64) {
        sh_ud[i] = u[EMLOCO_RND_DTHETA + i]; sh_ub[i] = u[EMLOCO_RND_BERN + i];
        sh_us[i] = u[EMLOCO_RND_SHARP + i]; sh_ua[i] = u[EMLOCO_RND_DSPEED + i];
    }
    __syncthreads();
    if (lane == 0) {
        const float vdt = t.vert_dt;
        float speed = (t.speed_max - t.speed_min) * u[EMLOCO_RND_SPEED0] + t.speed_min;
        float ratio = 1.0f;
        if (t.flags & EMLOCO_RESET_ADJUST_ROOT_VEL) {
            ratio = sqrtf(rvx * rvx + rvy * rvy) / speed;
        }
        float theta = 0.0f;
        for (int i = 0; i < RNV - 1; ++i) {
            float dth = (2.0f * sh_ud[i] - 1.0f) * (t.dtheta_max * vdt);
            if (sh_ub[i] < t.sharp_prob) dth = 3.14159265358979f * (2.0f * sh_us[i] - 1.0f);
            if (i == 0) dth = 3.14159265358979f * (2.0f * u[EMLOCO_RND_HEADING] - 1.0f);
            else {
                const float ds = (2.0f * sh_ua[i] - 1.0f) * (t.accel_max * vdt);
                speed = speed + ds;
                speed = speed < t.speed_min ? t.speed_min : (speed > t.speed_max ? t.speed_max : speed);
            }
            float sp = speed;
            if (t.flags & EMLOCO_RESET_ADJUST_ROOT_VEL) {
                sp = ratio * speed;
                sp = sp < t.speed_min ? t.speed_min : (sp > t.speed_max ? t.speed_max : sp);
            }
            theta += dth;
            sh_th[i] = theta;
            sh_seg[i] = sp * vdt;
        }
    }
    __syncthreads();
    for (int i = lane; i < RNV - 1; i += 64) {
        const float th = sh_th[i], seg = sh_seg[i];
        sh_th[i] = cosf(th) * seg;
        sh_seg[i] = -sinf(th) * seg;
    }
    __syncthreads();
    if (lane == 0) {
        float px = 0.0f, py = 0.0f;
        sh_v[0][0] = ipx; sh_v[0][1] = ipy; sh_v[0][2] = 0.0f;
        for (int i = 0; i < RNV - 1; ++i) {
            px += sh_th[i];
            py += sh_seg[i];
            sh_v[i + 1][0] = px + ipx; sh_v[i + 1][1] = py + ipy; sh_v[i + 1][2] = 0.0f;
        }
    }
    __syncthreads();
    const bool real = (t.flags & EMLOCO_RESET_REAL_PATH) && t.n_real > 0 && (u[EMLOCO_RND_REAL] > t.hybrid_prob);
    if (real) {                                     // :121-160
        int ri;
        if (t.real_pick) { ri = t.real_pick[bi]; ri = ri < 0 ? 0 : (ri > t.n_real - 1 ? t.n_real - 1 : ri); }
        else ri = (int)real_pick_perm((unsigned)bi % (unsigned)t.n_real, (unsigned)t.n_real, t.real_pick_key);
        const float *src = t.real_traj + (long)ri * RNV * 3;
        const float ox = src[0], oy = src[1];
        float sc = 1.0f;
        if (t.flags & EMLOCO_RESET_ADJUST_ROOT_VEL) {
            const float dx = src[3] - src[0], dy = src[4] - src[1], dzz = src[5] - src[2];
            float is = sqrtf(dx * dx + dy * dy + dzz * dzz);
            const float mn = t.speed_min * t.vert_dt;
            is = is < mn ? mn : is;
            sc = sqrtf(rvx * rvx + rvy * rvy) / is * t.vert_dt;
        }
        for (int i = lane; i < RNV; i += 64) {
            sh_v[i][0] = sc * (src[i * 3] - ox) + ipx;
            sh_v[i][1] = sc * (src[i * 3 + 1] - oy) + ipy;
            sh_v[i][2] = src[i * 3 + 2];
        }
        __syncthreads();
    }
    bool inv = false;
    if (t.flags & EMLOCO_RESET_INIT_HEADING) {      // :176-235
        const float ox = sh_v[0][0], oy = sh_v[0][1];
        const float dx = sh_v[1][0] - ox, dy = sh_v[1][1] - oy;
        const float rmag = sqrtf(rvx * rvx + rvy * rvy + rvz * rvz), dmag = sqrtf(dx * dx + dy * dy);
        const float root_rot = rmag > 0.0f ? atan2f(rvy, rvx) : 0.0f;
        const float ih = dmag > 0.0f ? atan2f(dy, dx) : 0.0f;
        float rd = ih - root_rot;
        if ((t.flags & EMLOCO_RESET_HEADING_INVERSION) && u[EMLOCO_RND_INVERSION] > 0.5f) { inv = true; rd = ih - root_rot + 3.14159265358979f; }
        const float c = cosf(rd), sn = sinf(rd);
        __syncthreads();
        for (int i = lane; i < RNV; i += 64) {
            const float x = sh_v[i][0] - ox, y = sh_v[i][1] - oy;
            sh_v[i][0] = (x * c + y * sn) + ox;
            sh_v[i][1] = (-x * sn + y * c) + oy;
        }
        if (lane == 0 && (t.flags & EMLOCO_RESET_HEADING_INVERSION)) *inv_out = inv ? 1 : 0;
        __syncthreads();
    }
}

// reset_finish of ONE list entry by one wave, in three pieces (the pooled reset of chain_kernels.hip runs the first and the last
// and copies what the middle one produced ahead of time):
//   reset_fix_height     a. lowest collision point -> vertical shift, b. buffers + warm-start impulses        (needs the env)
//   reset_traj_to        c. trajectory, d1. the LocoVal waypoints sampled from it                            (env-independent)
//   reset_capture_pose   d2. LocoVal pose / velocity inputs                                                    (needs the env)
__device__ __forceinline__ void reset_fix_height(const EmlocoResetBufs &t, const EmlocoSimDev &s, int env, int lane) {
    // ---- a. lowest collision point -> vertical shift (replaces the SMPL-mesh height fix, humanoid_amp.py:321-379)
    float low = 3.0e38f;
    for (int sl = 0; sl < 2; ++sl) {
        const int c = lane + 64 * sl;
        if (c < s.n_cand) {
            const int cp = s.topo[EMLOCO_TOPO_CAND + c], body = cp & 0xff, k = (cp >> 8) & 0xff, gt = cp >> 16;
            const long mb = (long)env * RNB + body;
            const float *ga = s.model + (size_t)env * EMLOCO_MODEL_WORDS + EMLOCO_MB_GEO + body * 8, *gb = ga + 4;   // a xyz, radius | b xyz
            float lp[3];
            if (gt == EMLOCO_GEOM_SPHERE) { lp[0] = ga[0]; lp[1] = ga[1]; lp[2] = ga[2]; }
            else if (gt == EMLOCO_GEOM_CAPSULE) { const float *src = k == 0 ? ga : gb; lp[0] = src[0]; lp[1] = src[1]; lp[2] = src[2]; }
            else {
                lp[0] = ga[0] + ((k & 1) ? gb[0] : -gb[0]);
                lp[1] = ga[1] + ((k & 2) ? gb[1] : -gb[1]);
                lp[2] = ga[2] + ((k & 4) ? gb[2] : -gb[2]);
            }
            const float *rb = s.rb_state + mb * 13;
            float R[9], wp[3];
            q2mat(rb + 3, R);
            matvec3(R, lp, wp);
            const float z = rb[2] + wp[2] - ga[3];
            low = z < low ? z : low;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(low, off); low = o < low ? o : low; }
    const float dz = low - t.ground_h[env] - t.height_tolerance;
    float *rs = s.root_state + (long)env * 13;
    if (lane < RNB) s.rb_state[((long)env * RNB + lane) * 13 + 2] -= dz;
    if (lane == 0) rs[2] -= dz;
    __syncthreads();

    // ---- b. buffers (humanoid.py:477-480) + warm-start impulses
    if (lane == 0) { t.progress_buf[env] = 0; t.reset_buf[env] = 0; t.terminate_buf[env] = 0; }
    for (int i = lane; i < RNB * 3; i += 64) s.contact_force[(long)env * RNB * 3 + i] = 0.0f;
    for (int i = lane; i < EMLOCO_MAXCAND * 3; i += 64) s.lambda_ws[(long)env * EMLOCO_MAXCAND * 3 + i] = 0.0f;
}

#define TRAJ_SM_FLOATS (RNV * 9)            /* LDS workspace: the vertices [RNV][3] + six RNV-float scratch rows */
__device__ __forceinline__ void reset_traj_to(const EmlocoResetBufs &t, const float *u, int bi, int lane, float ipx, float ipy, float rvx,
                                              float rvy, float rvz, float *verts_out, uint8_t *inv_out, float *way_out, float *sm) {
    float (*sh_v)[3] = (float (*)[3])sm;
    // ---- c. trajectory (traj_generator.py:60-237)
    reset_trajectory(t, u, bi, inv_out, lane, ipx, ipy, rvx, rvy, rvz, sh_v, sm + RNV * 3);
    for (int i = lane; i < RNV * 3; i += 64) verts_out[i] = (&sh_v[0][0])[i];
    // ---- d1. LocoVal waypoints captured at reset (humanoid_pedestrain_terrain.py:511-516)
    if (lane < EMLOCO_TRAJ_SAMPLES) {
        float p[3];
        r_calc_pos(&sh_v[0][0], (float)lane * t.sample_dt, t.traj_dur, p);
        for (int k = 0; k < 3; ++k) way_out[lane * 3 + k] = p[k];
    }
}

__device__ __forceinline__ void reset_capture_pose(const EmlocoResetBufs &t, const EmlocoSimDev &s, int env, int lane, float rvx, float rvy) {
    // ---- d2. LocoVal pose / velocity inputs captured at reset (humanoid_pedestrain_terrain.py:511-516)
    if (lane < RNB)
        for (int k = 0; k < 3; ++k) t.init_pose[((long)env * RNB + lane) * 3 + k] = s.rb_state[((long)env * RNB + lane) * 13 + k];
    if (lane == 0) { t.init_vel[(long)env * 2] = rvx; t.init_vel[(long)env * 2 + 1] = rvy; }
}

__device__ __forceinline__ void reset_finish_env(const EmlocoResetBufs &t, const EmlocoSimDev &s, int bi, int env, const float *u, int lane, float *sm) {
    reset_fix_height(t, s, env, lane);
    const float *rs = s.root_state + (long)env * 13;
    const float ipx = rs[0], ipy = rs[1];
    const float rvx = rs[7], rvy = rs[8];
    reset_traj_to(t, u, bi, lane, ipx, ipy, rvx, rvy, rs[9], t.traj_verts + (long)env * RNV * 3, t.inverted + env,
                  t.waypoint_traj + (long)env * EMLOCO_TRAJ_SAMPLES * 3, sm);
    reset_capture_pose(t, s, env, lane, rvx, rvy);
}

__global__ void __launch_bounds__(64)
reset_finish_kernel(EmlocoResetBufs t, EmlocoSimDev s, const int32_t *ids, int n, const float *rnd) {
    const int lane = threadIdx.x;
    for (int bi = blockIdx.x; bi < n; bi += gridDim.x) {
        const int env = ids[bi];
        if (env < 0) break;
        __shared__ float sm[TRAJ_SM_FLOATS];
        reset_finish_env(t, s, bi, env, rnd + (long)bi * EMLOCO_RESET_RND, lane, sm);
        __syncthreads();                              // LDS is reused by the next list entry
    }
}

// TrajGenerator.reset(env_ids, init_pos, root_vel) on its own (traj_generator.py:60-237): the trajectory part of the reset
// for callers that place the humanoids themselves; init_pos / root_vel are [n][3], one row per list entry.
__global__ void __launch_bounds__(64)
traj_reset_kernel(EmlocoResetBufs t, const int32_t *ids, int n, const float *rnd, const float *init_pos, const float *root_vel) {
    const int lane = threadIdx.x;
    for (int bi = blockIdx.x; bi < n; bi += gridDim.x) {
        const int env = ids[bi];
        if (env < 0) break;
        __shared__ float sm[TRAJ_SM_FLOATS];
        float (*sh_v)[3] = (float (*)[3])sm;
        const float *ip = init_pos + (long)bi * 3, *rv = root_vel + (long)bi * 3;
        reset_trajectory(t, rnd + (long)bi * EMLOCO_RESET_RND, bi, t.inverted + env, lane, ip[0], ip[1], rv[0], rv[1], rv[2], sh_v, sm + RNV * 3);
        float *vout = t.traj_verts + (long)env * RNV * 3;
        for (int i = lane; i < RNV * 3; i += 64) vout[i] = (&sh_v[0][0])[i];
        __syncthreads();                              // LDS is reused by the next list entry
    }
}

// ---- e. AMP history rows 1..14 from the motion library at t - k dt (humanoid_amp.py:486-535): one workgroup per
// (finished env, history row) -- the rows are independent, a single wave walking all 14 was the longest serial path of a reset.
#define HIST_SM_FLOATS (16 + 2 * RNDOF + 12)
__device__ __forceinline__ void reset_amp_history_row_to(const EmlocoResetBufs &t, const float *betas, float *out, int k, int mid, float mt, int lane, float *sm) {
    float *sh_root = sm, *sh_dp = sm + 16, *sh_dv = sh_dp + RNDOF, *sh_key = sh_dv + RNDOF;
    const FrameBlend fb = frame_blend(t, mid, mt - t.dt * (float)k);
    if (lane >= 1 && lane < RNB) {
        float q[4], e[3];
        ref_slerp(t.lrs + (fb.f0 * RNB + lane) * 4, t.lrs + (fb.f1 * RNB + lane) * 4, fb.blend, q);
        ref_quat_to_exp_map(q, e);
        for (int c = 0; c < 3; ++c) {
            sh_dp[(lane - 1) * 3 + c] = e[c];
            sh_dv[(lane - 1) * 3 + c] = lerp1(t.dvs[fb.f0 * RNDOF + (lane - 1) * 3 + c], t.dvs[fb.f1 * RNDOF + (lane - 1) * 3 + c], fb.blend);
        }
    }
    if (lane == 0) {
        for (int c = 0; c < 3; ++c) {
            sh_root[c] = lerp1(t.gts[(fb.f0 * RNB) * 3 + c], t.gts[(fb.f1 * RNB) * 3 + c], fb.blend);
            sh_root[7 + c] = lerp1(t.gvs[(fb.f0 * RNB) * 3 + c], t.gvs[(fb.f1 * RNB) * 3 + c], fb.blend);
            sh_root[10 + c] = lerp1(t.gavs[(fb.f0 * RNB) * 3 + c], t.gavs[(fb.f1 * RNB) * 3 + c], fb.blend);
        }
        ref_slerp(t.grs + (fb.f0 * RNB) * 4, t.grs + (fb.f1 * RNB) * 4, fb.blend, sh_root + 3);
    }
    if (lane < 4) {
        const int kb = t.key_bodies[lane];
        for (int c = 0; c < 3; ++c)
            sh_key[lane * 3 + c] = lerp1(t.gts[(fb.f0 * RNB + kb) * 3 + c], t.gts[(fb.f1 * RNB + kb) * 3 + c], fb.blend);
    }
    __syncthreads();
    amp_row(lane, sh_root, sh_root + 3, sh_root + 7, sh_root + 10, sh_dp, sh_dv, 1, sh_key, betas, t.dof_subset, t.n_dof_subset, out);
}
__device__ __forceinline__ void reset_amp_history_row(const EmlocoResetBufs &t, int env, int k, int mid, float mt, int lane, float *sm) {
    reset_amp_history_row_to(t, t.betas + (long)env * 17, t.amp_obs_buf + ((long)env * EMLOCO_AMP_STEPS + EMLOCO_AMP_PHYS_ROW(t.amp_ring, k)) * EMLOCO_AMP_ROW,
                             k, mid, mt, lane, sm);
}

__global__ void __launch_bounds__(64)
reset_amp_history_kernel(EmlocoResetBufs t, const int32_t *ids, int n) {
    const int lane = threadIdx.x;
    for (int bi = blockIdx.x; bi < n; bi += gridDim.x) {
        const int env = ids[bi];
        if (env < 0) break;
        __shared__ float sm[HIST_SM_FLOATS];
        reset_amp_history_row(t, env, 1 + (int)blockIdx.y, (int)t.motion_ids[env], t.motion_times[env], lane, sm);
        __syncthreads();                              // LDS is reused by the next list entry
    }
}

}  // namespace emloco
