// task_device.h -- the device functions of the fused post-physics pass (one wave per env), shared by task_kernels.hip (the
// post_physics_kernel launch), chain_kernels.hip (the observation roles of reset_obs_kernel) and sim_kernels.hip (progress / reward /
// flags in the last workgroup of an env's rigid-body step, emloco_sim_attach_post).  Reference lines: include/emloco_task.h.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_math.h"
#include "../../include/emloco_task.h"
#include "locoval_returns_device.h"

namespace emloco {

#define TNB 24
#define TNDOF 69

// np.linspace(lo, hi, n)[i] in fp64, then cast to fp32 (humanoid_pedestrain_terrain.py:650-668)
__device__ __forceinline__ float linspace_f(double lo, double hi, int n, int i) {
    if (i == n - 1) return (float)hi;
    const double step = (hi - lo) / (double)(n - 1);
    return (float)(lo + (double)i * step);
}

// traj_generator.py:278-296 calc_pos
__device__ __forceinline__ void calc_pos(const float *verts, float time, float traj_dur, float *out) {
    float phase = time / traj_dur;
    if (phase < 0.0f) phase = 0.0f;
    if (phase > 1.0f) phase = 1.0f;
    const float seg_idx = phase * (float)(EMLOCO_TRAJ_VERTS - 1);
    const long i0 = (long)floorf(seg_idx);
    const long i1 = (long)ceilf(seg_idx);
    const float lerp = seg_idx - (float)i0;
    for (int k = 0; k < 3; ++k) out[k] = (1.0f - lerp) * verts[i0 * 3 + k] + lerp * verts[i1 * 3 + k];
}

// humanoid_pedestrain_terrain.py:1212-1218 world_points_to_map: (p / horizontal_scale).long(), clipped to [0, shape - 2]
// (32-bit: a float -> int64 conversion is ~15 instructions with double-precision steps on this target, 41 of them per env.  The
// quotient is clamped to [-1, shape] as a float first, so the truncation never leaves the int range and the result equals the
// 64-bit form's for every input: below -1 -> 0, above shape -> shape - 2, NaN -> 0.)
__device__ __forceinline__ int map_index1(float x, float hscale, int n) {
    float v = x / hscale;
    v = v > -1.0f ? v : -1.0f;                     // NaN -> -1
    v = v < (float)n ? v : (float)n;
    int p = (int)v;
    if (p < 0) p = 0;
    if (p > n - 2) p = n - 2;
    return p;
}
__device__ __forceinline__ void map_index(int rows, int cols, float x, float y, float hscale, int *opx, int *opy) {
    *opx = map_index1(x, hscale, rows);
    *opy = map_index1(y, hscale, cols);
}
// :1282-1288 sample_height_points: min of the cell's two diagonal corners
__device__ __forceinline__ float sample_height_at(const int16_t *hf, int cols, int px, int py, float vscale) {
    const int16_t h1 = hf[px * cols + py], h2 = hf[(px + 1) * cols + (py + 1)];
    const int16_t hm = h1 < h2 ? h1 : h2;
    return (float)hm * vscale;
}
__device__ __forceinline__ float sample_height(const int16_t *hf, int rows, int cols, float x, float y, float hscale, float vscale) {
    int px, py;
    map_index(rows, cols, x, y, hscale, &px, &py);
    return sample_height_at(hf, cols, px, py, vscale);
}
// probe idx = 32 i + j of the 32x32 grid (+-2 m, meshgrid 'ij', :650-668) rotated by the heading quaternion hq about `origin`
// (:778-787 get_heights): world position
__device__ __forceinline__ void grid_probe(const float *hq, const float *origin, int idx, float *wx, float *wy) {
    const int i = idx >> 5, j = idx & 31;
    const float pt[3] = {linspace_f(-2.0, 2.0, 32, i), linspace_f(-2.0, 2.0, 32, j), 0.0f};
    float rr[3];
    ref_quat_apply(hq, pt, rr);
    *wx = rr[0] + origin[0];
    *wy = rr[1] + origin[1];
}
// probe k = 3 i + j of the 3x3 centre grid (x +-0.1, y +-0.2, :633-648), yaw-only rotation of the root (:747-750, quat_apply_yaw :1533-1538)
__device__ __forceinline__ void center_probe(const float *root_pos, const float *root_rot, int k, float *wx, float *wy) {
    const int i = k / 3, j = k - 3 * i;
    const float pt[3] = {linspace_f(-0.1, 0.1, 3, i), linspace_f(-0.2, 0.2, 3, j), 0.0f};
    float qy[4] = {0.0f, 0.0f, root_rot[2], root_rot[3]}, rr[3];
    float nn = sqrtf(qy[2] * qy[2] + qy[3] * qy[3]);
    if (nn < 1e-9f) nn = 1e-9f;
    qy[0] = 0.0f / nn; qy[1] = 0.0f / nn; qy[2] = qy[2] / nn; qy[3] = qy[3] / nn;
    ref_quat_apply(qy, pt, rr);
    *wx = rr[0] + root_pos[0];
    *wy = rr[1] + root_pos[1];
}

// torch's .mean(dim=-1) over the 9 centre probes (:427-429): sum / 9, the sum in the order of torch's scalar row reduction
// (8 partial sums + remainder): ((c0 + c8) + c1) + c2 + ... + c7
__device__ __forceinline__ float mean9(const float *c) {
    float s = c[0] + c[8];
    for (int k = 1; k < 8; ++k) s += c[k];
    return s / 9.0f;
}

// one body's share of compute_humanoid_observations_smpl_max (humanoid.py:1625-1687)
__device__ __forceinline__ void self_obs_body(int b, const float *root_pos, const float *hinv, const float *pos,
                                              const float *rot, const float *vel, const float *ang, float *obs) {
    if (b >= 1) {
        const float dlt[3] = {pos[0] - root_pos[0], pos[1] - root_pos[1], pos[2] - root_pos[2]};
        ref_quat_rotate(hinv, dlt, obs + (b - 1) * 3);
    }
    float lq[4];
    ref_quat_mul(hinv, rot, lq);
    ref_quat_to_tan_norm(lq, obs + 69 + b * 6);
    ref_quat_rotate(hinv, vel, obs + 69 + 144 + b * 3);
    ref_quat_rotate(hinv, ang, obs + 69 + 144 + 72 + b * 3);
}

// one AMP row (humanoid_amp.py:917-971); lanes cooperate, `out` points at the row
__device__ __forceinline__ void amp_row(int lane, const float *root_pos, const float *root_rot, const float *root_vel,
                                        const float *root_ang, const float *dof_pos, const float *dof_vel,
                                        int dof_stride /* 2 when both point into dof_state */, const float *key_pos /* [4][3] */,
                                        const float *betas, const int32_t *subset, int n_sub, float *out) {
    float hinv[4];
    ref_quat_about_z(-ref_calc_heading(root_rot), hinv);
    if (lane == 0) {
        float lq[4], t6[6], t3[3];
        ref_quat_mul(hinv, root_rot, lq);
        ref_quat_to_tan_norm(lq, t6);
        for (int k = 0; k < 6; ++k) out[k] = t6[k];
        ref_quat_rotate(hinv, root_vel, t3);
        for (int k = 0; k < 3; ++k) out[6 + k] = t3[k];
        ref_quat_rotate(hinv, root_ang, t3);
        for (int k = 0; k < 3; ++k) out[9 + k] = t3[k];
    }
    const int nj = n_sub / 3;
    if (lane < nj) {
        float em[3], q[4], t6[6];
        for (int k = 0; k < 3; ++k) {
            const int dd = subset[lane * 3 + k];
            em[k] = dof_pos[dd * dof_stride];
        }
        ref_exp_map_to_quat(em, q);
        ref_quat_to_tan_norm(q, t6);
        for (int k = 0; k < 6; ++k) out[12 + lane * 6 + k] = t6[k];
    }
    if (lane < n_sub) {
        const int dd = subset[lane];
        out[12 + nj * 6 + lane] = dof_vel[dd * dof_stride];
    }
    if (lane < 4) {
        float dlt[3], t3[3];
        for (int k = 0; k < 3; ++k) dlt[k] = key_pos[lane * 3 + k] - root_pos[k];
        ref_quat_rotate(hinv, dlt, t3);
        for (int k = 0; k < 3; ++k) out[12 + nj * 6 + n_sub + lane * 3 + k] = t3[k];
    }
    if (lane < 11) out[12 + nj * 6 + n_sub + 12 + lane] = betas[lane];
}

// The post-physics work of ONE env by one wave (every early return is wave-uniform): post_physics_kernel below and the
// observation roles of reset_obs_kernel (chain_kernels.hip) run this body.
#ifdef EMLOCO_POST_PROFILE
__device__ long long *g_post_prof = nullptr;
#define PPSTAMP(i) do { if (g_post_prof && lane == 0 && env == t.n_env - 1) g_post_prof[i] = wall_clock64(); } while (0)
#else
#define PPSTAMP(i) do { } while (0)
#endif
// `sm`: POST_SM_FLOATS floats of LDS owned by the calling workgroup (the caller decides what else lives there before and after:
// the fused launches overlay it with their other roles' arrays)
#define POST_SM_FLOATS (TNB * 13 + EMLOCO_TRAJ_SAMPLES * 3 + 12 + TNB * 3 + 2 * EMLOCO_SELF_OBS + 12)
// `lv` (optional): the LocoVal return bookkeeping of this env (locoval_returns_device.h) runs right behind its reward and reset flag
// -- needs EMLOCO_POST_REWARD and EMLOCO_POST_RESET in `mode`; `lv_inverted` [n_env] bytes (or NULL) are the heading-inversion flags.
__device__ __forceinline__ void post_physics_env(const EmlocoTaskBufs &t, int mode, int env, int lane, float *sm,
                                                 const EmlocoLocoValStep *lv = nullptr, const uint8_t *lv_inverted = nullptr) {
    PPSTAMP(0);
    float (*sh_body)[13] = (float (*)[13])sm;
    float (*sh_samp)[3] = (float (*)[3])(sm + TNB * 13);
    float *sh_center = sm + TNB * 13 + EMLOCO_TRAJ_SAMPLES * 3;
    float (*sh_cf)[3] = (float (*)[3])(sh_center + 12);
    float *sh_obs = sh_center + 12 + TNB * 3, *sh_fobs = sh_obs + EMLOCO_SELF_OBS;
    float (*sh_key)[3] = (float (*)[3])(sh_fobs + EMLOCO_SELF_OBS);

    int64_t prog = t.progress_buf[env];
    if (mode & EMLOCO_POST_ADVANCE) prog += 1;      // stored after the barrier below, once every lane has read it
    if (lane < TNB) {
        const float *src = t.rb_state + ((long)env * TNB + lane) * 13;
        for (int k = 0; k < 13; ++k) sh_body[lane][k] = src[k];
        if (mode & EMLOCO_POST_RESET)
            for (int k = 0; k < 3; ++k) sh_cf[lane][k] = t.contact_force[((long)env * TNB + lane) * 3 + k];
    }
    // trajectory samples (lane k < 15): t + k * sample_dt; sample 0 is the reward / reset target
    if (lane < EMLOCO_TRAJ_SAMPLES) {
        const float beg = (float)prog * t.dt;
        float s[3];
        calc_pos(t.traj_verts + (long)env * EMLOCO_TRAJ_VERTS * 3, beg + (float)lane * t.sample_dt, t.traj_dur, s);
        for (int k = 0; k < 3; ++k) sh_samp[lane][k] = s[k];
    }
    __syncthreads();
    PPSTAMP(1);
    if ((mode & EMLOCO_POST_ADVANCE) && lane == 0) t.progress_buf[env] = prog;
    const float *root = sh_body[0];

    if (mode & EMLOCO_POST_OBS) {
        float *obs = t.obs_buf + (long)env * EMLOCO_OBS;
        float *fobs = t.flip_obs_buf + (long)env * EMLOCO_OBS;
        float hinv[4], hinv_f[4];
        ref_quat_about_z(-ref_calc_heading(root + 3), hinv);
        const float froot_rot[4] = {-root[3], root[4], -root[5], root[6]};
        ref_quat_about_z(-ref_calc_heading(froot_rot), hinv_f);
        PPSTAMP(2);
        // ---- self obs + mirrored self obs (lane = body); staged in LDS so the row is written coalesced
        if (lane < TNB) {
            const float *bd = sh_body[lane];
            self_obs_body(lane, root, hinv, bd, bd + 3, bd + 7, bd + 10, sh_obs);
            const float *sb = sh_body[t.left_to_right[lane]];
            const float fp[3] = {sb[0], -sb[1], sb[2]};
            const float fr[4] = {-sb[3], sb[4], -sb[5], sb[6]};
            const float fv[3] = {sb[7], -sb[8], sb[9]};
            const float fa[3] = {-sb[10], sb[11], -sb[12]};
            const float froot_pos[3] = {root[0], -root[1], root[2]};
            self_obs_body(lane, froot_pos, hinv_f, fp, fr, fv, fa, sh_fobs);
        } else if (lane < TNB + 11) {
            const float bv = t.betas[(long)env * 17 + (lane - TNB)];
            sh_obs[357 + lane - TNB] = bv;
            sh_fobs[357 + lane - TNB] = bv;
        }
        // ---- location obs (lane = sample)
        if (lane < EMLOCO_TRAJ_SAMPLES) {
            const float dlt[3] = {sh_samp[lane][0] - root[0], sh_samp[lane][1] - root[1], sh_samp[lane][2] - root[2]};
            float rr[3];
            ref_quat_rotate(hinv, dlt, rr);
            obs[EMLOCO_SELF_OBS + 2 * lane] = rr[0];
            obs[EMLOCO_SELF_OBS + 2 * lane + 1] = rr[1];
            fobs[EMLOCO_SELF_OBS + 2 * lane] = rr[0];
            fobs[EMLOCO_SELF_OBS + 2 * lane + 1] = -rr[1];
        }
        // ---- centre-height probes (3x3, yaw only) around the root
        if (lane < 9) {
            float wx, wy;
            center_probe(root, root + 3, lane, &wx, &wy);
            sh_center[lane] = sample_height(t.heightfield, t.hf_rows, t.hf_cols, wx, wy, t.hscale, t.vscale);
        }
        __syncthreads();
        PPSTAMP(3);
        for (int i = lane; i < EMLOCO_SELF_OBS; i += 64) { obs[i] = sh_obs[i]; fobs[i] = sh_fobs[i]; }
        const float cmean = mean9(sh_center);
        // ---- 32x32 height grid around the head, rotated by the head's heading (16 points per lane)
        const float *head = sh_body[t.head_body];
        float hq[4];
        ref_quat_about_z(ref_calc_heading(head + 3), hq);
        PPSTAMP(4);
        float *hobs = obs + EMLOCO_SELF_OBS + 2 * EMLOCO_TRAJ_SAMPLES;
        float *fhobs = fobs + EMLOCO_SELF_OBS + 2 * EMLOCO_TRAJ_SAMPLES;
        // two passes: all sixteen probes of a lane first fetch their two map cells (the loads of one probe do not wait for the stores of
        // the one before it: sixteen dependent round trips to the map were most of this kernel's latency), then the rows are written
        const int16_t *__restrict__ hf = t.heightfield;
        int16_t c1[EMLOCO_HEIGHT_POINTS / 64], c2[EMLOCO_HEIGHT_POINTS / 64];
#pragma unroll
        for (int it = 0; it < EMLOCO_HEIGHT_POINTS / 64; ++it) {
            float wx, wy;
            int px, py;
            grid_probe(hq, head, lane + 64 * it, &wx, &wy);
            map_index(t.hf_rows, t.hf_cols, wx, wy, t.hscale, &px, &py);
            c1[it] = hf[px * t.hf_cols + py];
            c2[it] = hf[(px + 1) * t.hf_cols + (py + 1)];
        }
#pragma unroll
        for (int it = 0; it < EMLOCO_HEIGHT_POINTS / 64; ++it) {
            const int idx = lane + 64 * it;
            const int i = idx >> 5, j = idx & 31;
            const int16_t hm = c1[it] < c2[it] ? c1[it] : c2[it];           // sample_height_at
            const float hh = (float)hm * t.vscale;
            float v = cmean - hh;
            if (v < -3.0f) v = -3.0f;
            if (v > 3.0f) v = 3.0f;
            v *= 5.0f;
            hobs[idx] = v;
            fhobs[i * 32 + (31 - j)] = v;
        }
    }

    PPSTAMP(5);
    const float *tar = sh_samp[0];
    float rew_now = 0.0f;                        // lane 0: this step's reward (the returns hook below reads it)
    if (mode & EMLOCO_POST_REWARD) {
        float part = 0.0f;
        for (int dd = lane; dd < TNDOF; dd += 64)
            part += fabsf(t.dof_force[(long)env * TNDOF + dd] * t.dof_state[((long)env * TNDOF + dd) * 2 + 1]);
        const float power = wave_sum(part);
        if (lane == 0) {
            const float dx = tar[0] - root[0], dy = tar[1] - root[1];
            const float err = dx * dx + dy * dy;
            const float loc = expf(-2.0f * err);
            const float pw = -t.power_coef * power;
            rew_now = loc + pw;
            t.rew_buf[env] = rew_now;
            t.reward_raw[(long)env * 2] = loc;
            t.reward_raw[(long)env * 2 + 1] = pw;
        }
    }
    int done_now = 0;                            // this env's reset flag as this launch leaves it (lane 0; broadcast below)
    if ((mode & EMLOCO_POST_RESET) && lane == 0) {
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (int bb = 0; bb < TNB; ++bb) {
            const bool masked = t.contact_body_mask[bb] != 0;
            sx += masked ? 0.0f : sh_cf[bb][0];
            sy += masked ? 0.0f : sh_cf[bb][1];
            sz += masked ? 0.0f : sh_cf[bb][2];
        }
        const float ax = fabsf(sx), ay = fabsf(sy), az = fabsf(sz);
        const float mag = sqrtf(ax * ax + ay * ay + az * az);
        const bool fallen = (mag > 50.0f) && (prog > 1);
        const float dx = tar[0] - root[0], dy = tar[1] - root[1];
        const float d2 = dx * dx + dy * dy;
        const bool far = d2 > t.fail_dist * t.fail_dist;
        const int64_t term = (fallen || far) ? 1 : 0;
        t.terminate_buf[env] = term;
        done_now = ((float)prog >= t.max_episode_length - 1.0f) ? 1 : (int)term;
        t.reset_buf[env] = done_now;
    }
    if (lv && (mode & EMLOCO_POST_REWARD) && (mode & EMLOCO_POST_RESET))
        locoval_returns_env(*lv, env, lane, rew_now, 0.0f, done_now != 0, lv_inverted && lv_inverted[env]);
    if (mode & EMLOCO_POST_AMP_DONE_ONLY) {      // the AMP rows of the finished envs only (block-uniform)
        done_now = (mode & EMLOCO_POST_RESET) ? __shfl(done_now, 0) : (int)(t.reset_buf[env] != 0);
        if (!done_now) return;
    }

    if (mode & (EMLOCO_POST_AMP_SHIFT | EMLOCO_POST_AMP_ROW)) {
        float *amp = t.amp_obs_buf + (long)env * EMLOCO_AMP_STEPS * EMLOCO_AMP_ROW;
        if ((mode & EMLOCO_POST_AMP_SHIFT) && !t.amp_ring) {         // (ring: the caller moved the head, nothing is copied)
            // rows 0..13 -> rows 1..14; every lane first loads all of its elements, then stores
            constexpr int NEL = (EMLOCO_AMP_STEPS - 1) * EMLOCO_AMP_ROW;
            constexpr int PER = (NEL + 63) / 64;
            float keep[PER];
            for (int j = 0; j < PER; ++j) { const int e = lane + 64 * j; keep[j] = e < NEL ? amp[e] : 0.0f; }
            __syncthreads();
            for (int j = 0; j < PER; ++j) { const int e = lane + 64 * j; if (e < NEL) amp[EMLOCO_AMP_ROW + e] = keep[j]; }
        }
        PPSTAMP(6);
        if (mode & EMLOCO_POST_AMP_ROW) {
            if (lane < 4) for (int k = 0; k < 3; ++k) sh_key[lane][k] = sh_body[t.key_bodies[lane]][k];
            __syncthreads();
            const float *ds = t.dof_state + (long)env * TNDOF * 2;
            amp_row(lane, root, root + 3, root + 7, root + 10, ds, ds + 1, 2, &sh_key[0][0], t.betas + (long)env * 17, t.dof_subset, t.n_dof_subset,
                    amp + EMLOCO_AMP_PHYS_ROW(t.amp_ring, 0) * EMLOCO_AMP_ROW);
        }
    }
    PPSTAMP(7);
}

}  // namespace emloco
