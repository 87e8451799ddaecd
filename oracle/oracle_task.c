/*
 * oracle_task.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement (plain C, fp32, one env at a time) of the reference's non-physics share of
 * env.step: observations, reward, reset masks, trajectory sampling, height sampling, AMP rows,
 * PD-target map.  Pinned against the tests/golden fixtures, which were produced by running the
 * reference's own torch functions (tests/golden/gen_golden.py).
 *
 * Reference files (all under /root/reference/pacer/pacer/):
 *   H   = env/tasks/humanoid.py
 *   HA  = env/tasks/humanoid_amp.py
 *   HT  = env/tasks/humanoid_traj.py
 *   HPT = env/tasks/humanoid_pedestrain_terrain.py
 *   TG  = env/util/traj_generator.py
 */
#include <stdint.h>
#include <string.h>
#include "oracle_math.h"

#define NB 24
#define NDOF 69
#define SELF_OBS 368
#define NSAMP 15
#define NVERT 101
#define NHP 1024
#define TASK_OBS (2 * NSAMP + NHP)
#define AMP_ROW 206

/* H:334 left_to_right_index */
static const int L2R[NB] = {0, 5, 6, 7, 8, 1, 2, 3, 4, 9, 10, 11, 12, 13, 19, 20, 21, 22, 23, 14, 15, 16, 17, 18};

/* H:1625-1687 compute_humanoid_observations_smpl_max with
 * local_root_obs=True, root_height_obs=False, upright=True, has_smpl_params=True, has_limb_weight=False.
 * Layout (H:1678-1686): [local pos 23x3 | tan-norm 24x6 | local vel 24x3 | local ang vel 24x3 | betas[:11]] */
static void self_obs_one(const float *pos, const float *rot, const float *vel, const float *ang,
                         const float *betas17, float *obs) {
    float hinv[4];
    orc_calc_heading_quat_inv(rot, hinv); /* root_rot = body_rot[:,0] */
    float *o = obs;
    for (int b = 1; b < NB; ++b) { /* root pos dropped (H:1653) */
        float d[3] = {pos[b * 3] - pos[0], pos[b * 3 + 1] - pos[1], pos[b * 3 + 2] - pos[2]};
        orc_my_quat_rotate(hinv, d, o);
        o += 3;
    }
    for (int b = 0; b < NB; ++b) {
        float lq[4];
        orc_quat_mul(hinv, rot + b * 4, lq);
        orc_quat_to_tan_norm(lq, o);
        o += 6;
    }
    for (int b = 0; b < NB; ++b) { orc_my_quat_rotate(hinv, vel + b * 3, o); o += 3; }
    for (int b = 0; b < NB; ++b) { orc_my_quat_rotate(hinv, ang + b * 3, o); o += 3; }
    for (int k = 0; k < 11; ++k) *o++ = betas17[k]; /* smpl_params[:, :-6] */
}

void orc_self_obs(int E, const float *pos, const float *rot, const float *vel, const float *ang,
                  const float *betas, float *obs) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e)
        self_obs_one(pos + e * NB * 3, rot + e * NB * 4, vel + e * NB * 3, ang + e * NB * 3,
                     betas + e * 17, obs + e * SELF_OBS);
}

/* H:1066-1108 _compute_flip_humanoid_obs: negate y of pos/vel, x,z of quat xyz-part / ang vel, then L/R permute */
void orc_flip_self_obs(int E, const float *pos, const float *rot, const float *vel, const float *ang,
                       const float *betas, float *obs) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        float fp[NB * 3], fr[NB * 4], fv[NB * 3], fa[NB * 3];
        const float *p = pos + e * NB * 3, *r = rot + e * NB * 4, *v = vel + e * NB * 3, *a = ang + e * NB * 3;
        for (int b = 0; b < NB; ++b) {
            int s = L2R[b];
            fp[b * 3] = p[s * 3]; fp[b * 3 + 1] = -p[s * 3 + 1]; fp[b * 3 + 2] = p[s * 3 + 2];
            fr[b * 4] = -r[s * 4]; fr[b * 4 + 1] = r[s * 4 + 1]; fr[b * 4 + 2] = -r[s * 4 + 2]; fr[b * 4 + 3] = r[s * 4 + 3];
            fv[b * 3] = v[s * 3]; fv[b * 3 + 1] = -v[s * 3 + 1]; fv[b * 3 + 2] = v[s * 3 + 2];
            fa[b * 3] = -a[s * 3]; fa[b * 3 + 1] = a[s * 3 + 1]; fa[b * 3 + 2] = -a[s * 3 + 2];
        }
        self_obs_one(fp, fr, fv, fa, betas + e * 17, obs + e * SELF_OBS);
    }
}

/* TG:278-296 calc_pos for one (trajectory, time). traj_dur = num_verts * dt_vert (TG:270-273). */
static void calc_pos_one(const float *verts, float time, float traj_dur, float *out) {
    float phase = time / traj_dur;
    if (phase < 0.0f) phase = 0.0f;
    if (phase > 1.0f) phase = 1.0f;
    float seg_idx = phase * (float)(NVERT - 1);
    long i0 = (long)floorf(seg_idx);
    long i1 = (long)ceilf(seg_idx);
    float lerp = seg_idx - (float)i0;
    for (int k = 0; k < 3; ++k)
        out[k] = (1.0f - lerp) * verts[i0 * 3 + k] + lerp * verts[i1 * 3 + k];
}

void orc_traj_calc_pos(int E, const float *verts, const int64_t *progress, float dt, float traj_dur, float *out) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e)
        calc_pos_one(verts + (long)e * NVERT * 3, (float)progress[e] * dt, traj_dur, out + e * 3);
}

/* HT:208-224 _fetch_traj_samples: 15 samples at t + k*0.4 s */
void orc_traj_samples(int E, const float *verts, const int64_t *progress, float dt, float traj_dur,
                      float sample_dt, float *out) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        float beg = (float)progress[e] * dt;
        for (int k = 0; k < NSAMP; ++k)
            calc_pos_one(verts + (long)e * NVERT * 3, beg + (float)k * sample_dt, traj_dur,
                         out + ((long)e * NSAMP + k) * 3);
    }
}

/* HPT:1549-1577 compute_location_observations (upright=True) */
void orc_location_obs(int E, const float *root_states, const float *samples, float *obs) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        const float *rs = root_states + e * 13;
        float hinv[4];
        orc_calc_heading_quat_inv(rs + 3, hinv);
        for (int k = 0; k < NSAMP; ++k) {
            const float *s = samples + ((long)e * NSAMP + k) * 3;
            float d[3] = {s[0] - rs[0], s[1] - rs[1], s[2] - rs[2]}, r[3];
            orc_my_quat_rotate(hinv, d, r);
            obs[e * 2 * NSAMP + 2 * k] = r[0];
            obs[e * 2 * NSAMP + 2 * k + 1] = r[1];
        }
    }
}

/* HPT:1212-1218 world_points_to_map: (points / horizontal_scale).long() (true fp32 division by fp32(0.1), truncation
 * toward zero), clipped to [0, shape - 2] */
static void world_point_to_map(int rows, int cols, float x, float y, float hscale, long *px, long *py) {
    long ix = (long)(x / hscale);
    long iy = (long)(y / hscale);
    if (ix < 0) ix = 0;
    if (ix > rows - 2) ix = rows - 2;
    if (iy < 0) iy = 0;
    if (iy > cols - 2) iy = cols - 2;
    *px = ix; *py = iy;
}

/* HPT:1282-1288 sample_height_points (no root_points, no velocity map): min of the cell's two diagonal corners */
static float sample_height_at(const int16_t *hf, int cols, long px, long py, float vscale) {
    int16_t h1 = hf[px * cols + py], h2 = hf[(px + 1) * cols + (py + 1)];
    int16_t h = h1 < h2 ? h1 : h2;
    return (float)h * vscale;
}

/* np.linspace(-ext, ext, n)[i] cast to fp32 (HPT:650-668 init_square_height_points; meshgrid 'ij') */
static float linspace_f(double lo, double hi, int n, int i) {
    if (i == n - 1) return (float)hi;
    double step = (hi - lo) / (double)(n - 1);
    return (float)(lo + (double)i * step);
}

/* HPT:761-815 get_heights with terrain_obs_root == "head" (HPT:410-412): rotate the 32x32 grid by the
 * heading of `pose` (pos3, quat4), sample the map.  out (E,1024) row-major over meshgrid(x,y) 'ij'.
 * heading_q (optional, [E][4]): use this heading quaternion instead of computing it (the sine / cosine / arctangent behind it
 * are the one step of the chain whose last bit is library-defined: torch's CPU path calls MKL's closed VML there);
 * px / py (optional, [E][1024] int64): the map indices. */
void orc_get_heights_ex(int E, const float *pose7, const float *heading_q, const int16_t *hf, int rows, int cols, float hscale,
                        float vscale, float *out, int64_t *out_px, int64_t *out_py) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        const float *p = pose7 + e * 7;
        float hq[4];
        if (heading_q) { for (int k = 0; k < 4; ++k) hq[k] = heading_q[e * 4 + k]; }
        else orc_calc_heading_quat(p + 3, hq);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float pt[3] = {linspace_f(-2.0, 2.0, 32, i), linspace_f(-2.0, 2.0, 32, j), 0.0f}, r[3];
                long px, py;
                orc_quat_apply(hq, pt, r);
                world_point_to_map(rows, cols, r[0] + p[0], r[1] + p[1], hscale, &px, &py);
                const long o = (long)e * NHP + i * 32 + j;
                if (out) out[o] = sample_height_at(hf, cols, px, py, vscale);
                if (out_px) { out_px[o] = px; out_py[o] = py; }
            }
    }
}

void orc_get_heights(int E, const float *pose7, const int16_t *hf, int rows, int cols, float hscale,
                     float vscale, float *out) {
    orc_get_heights_ex(E, pose7, 0, hf, rows, cols, hscale, vscale, out, 0, 0);
}

/* HPT:732-759 get_center_heights: 3x3 probe (x in linspace(-.1,.1,3), y in linspace(-.2,.2,3)), yaw-only */
void orc_get_center_heights_ex(int E, const float *root_states, const int16_t *hf, int rows, int cols,
                               float hscale, float vscale, float *out9, int64_t *out_px, int64_t *out_py) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        const float *rs = root_states + e * 13;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float pt[3] = {linspace_f(-0.1, 0.1, 3, i), linspace_f(-0.2, 0.2, 3, j), 0.0f}, r[3];
                long px, py;
                orc_quat_apply_yaw(rs + 3, pt, r);
                world_point_to_map(rows, cols, r[0] + rs[0], r[1] + rs[1], hscale, &px, &py);
                if (out9) out9[e * 9 + i * 3 + j] = sample_height_at(hf, cols, px, py, vscale);
                if (out_px) { out_px[e * 9 + i * 3 + j] = px; out_py[e * 9 + i * 3 + j] = py; }
            }
    }
}

void orc_get_center_heights(int E, const float *root_states, const int16_t *hf, int rows, int cols,
                            float hscale, float vscale, float *out9) {
    orc_get_center_heights_ex(E, root_states, hf, rows, cols, hscale, vscale, out9, 0, 0);
}

/* HPT:427-437 height obs = clip(mean(center) - h, -3, 3) * 5 (use_center_height: true) */
void orc_height_obs(int E, const float *center9, const float *heights, float *obs) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        /* torch's .mean(dim=-1) over the 9 probes = sum / 9 with the sum in the order of its scalar row reduction
         * (8 partial sums + remainder, aten SumKernel row_sum): ((c0 + c8) + c1) + c2 + ... + c7 */
        const float *c = center9 + e * 9;
        float s = c[0] + c[8];
        for (int k = 1; k < 8; ++k) s += c[k];
        float m = s / 9.0f;
        for (int k = 0; k < NHP; ++k) {
            float v = m - heights[(long)e * NHP + k];
            if (v < -3.0f) v = -3.0f;
            if (v > 3.0f) v = 3.0f;
            obs[(long)e * NHP + k] = v * 5.0f;
        }
    }
}

/* HPT:455-491 _compute_flip_task_obs: negate traj y, mirror the height grid along its 2nd axis */
void orc_flip_task_obs(int E, const float *task_obs, float *out) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        const float *t = task_obs + (long)e * TASK_OBS;
        float *o = out + (long)e * TASK_OBS;
        for (int k = 0; k < NSAMP; ++k) { o[2 * k] = t[2 * k]; o[2 * k + 1] = -t[2 * k + 1]; }
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) o[2 * NSAMP + i * 32 + j] = t[2 * NSAMP + i * 32 + (31 - j)];
    }
}

/* HPT:907-930 _compute_reward + HPT:1581-1592 compute_location_reward.  rew = loc + power (power_reward True) */
void orc_reward(int E, const float *root_pos, const float *tar_pos, const float *dof_force,
                const float *dof_vel, float power_coef, float *rew, float *reward_raw2) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        float dx = tar_pos[e * 3] - root_pos[e * 3], dy = tar_pos[e * 3 + 1] - root_pos[e * 3 + 1];
        float err = dx * dx + dy * dy;
        float loc = expf(-2.0f * err);
        float power = 0.0f;
        for (int d = 0; d < NDOF; ++d) power += fabsf(dof_force[e * NDOF + d] * dof_vel[e * NDOF + d]);
        float pw = -power_coef * power;
        rew[e] = loc + pw;
        reward_raw2[e * 2] = loc;
        reward_raw2[e * 2 + 1] = pw;
    }
}

/* HPT:1468-1530 compute_humanoid_reset (enable_early_termination=True, disableCollision=False):
 * terminated = (|sum of non-foot contact forces| > 50 and progress > 1) or (|tar - root|^2 > fail_dist^2)
 * reset = progress >= max_episode_length - 1 ? 1 : terminated.   All int64, bit-exact. */
void orc_reset(int E, const int64_t *progress, const float *contact, const int *contact_body_ids, int n_cb,
               const float *body_pos, const float *tar_pos, float max_episode_length, float fail_dist,
               int64_t *reset, int64_t *terminate) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
        for (int b = 0; b < NB; ++b) {
            int masked = 0;
            for (int k = 0; k < n_cb; ++k) masked |= (contact_body_ids[k] == b);
            const float *f = contact + ((long)e * NB + b) * 3;
            sx += masked ? 0.0f : f[0];
            sy += masked ? 0.0f : f[1];
            sz += masked ? 0.0f : f[2];
        }
        float ax = fabsf(sx), ay = fabsf(sy), az = fabsf(sz);
        float mag = sqrtf(ax * ax + ay * ay + az * az);
        int fallen = (mag > 50.0f) && (progress[e] > 1);
        const float *rp = body_pos + (long)e * NB * 3;
        float dx = tar_pos[e * 3] - rp[0], dy = tar_pos[e * 3 + 1] - rp[1];
        float d2 = dx * dx + dy * dy;
        int far = d2 > fail_dist * fail_dist;
        int64_t term = (fallen || far) ? 1 : 0;
        terminate[e] = term;
        reset[e] = ((float)progress[e] >= max_episode_length - 1.0f) ? 1 : term;
    }
}

/* HA:917-971 build_amp_observations_smpl with local_root_obs=True, root_height_obs=False,
 * has_dof_subset=True, has_shape_obs_disc=True, has_limb_weight_obs=False, upright=True.
 * Row (206) = [root tan-norm 6 | local root vel 3 | local root ang vel 3 | dof tan-norm 19x6 | dof vel 57 |
 *              key pos 4x3 | betas[:11]]; H:1327-1338 dof_to_obs_smpl. */
void orc_amp_obs(int E, const float *root_pos, const float *root_rot, const float *root_vel,
                 const float *root_ang, const float *dof_pos, const float *dof_vel, const float *key_pos,
                 const float *betas, const int *dof_subset, int n_sub, float *out) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        float hinv[4], lq[4];
        float *o = out + (long)e * AMP_ROW;
        orc_calc_heading_quat_inv(root_rot + e * 4, hinv);
        orc_quat_mul(hinv, root_rot + e * 4, lq);
        orc_quat_to_tan_norm(lq, o); o += 6;
        orc_my_quat_rotate(hinv, root_vel + e * 3, o); o += 3;
        orc_my_quat_rotate(hinv, root_ang + e * 3, o); o += 3;
        for (int j = 0; j < n_sub / 3; ++j) {
            float em[3], q[4];
            for (int k = 0; k < 3; ++k) em[k] = dof_pos[e * NDOF + dof_subset[j * 3 + k]];
            orc_exp_map_to_quat(em, q);
            orc_quat_to_tan_norm(q, o); o += 6;
        }
        for (int j = 0; j < n_sub; ++j) *o++ = dof_vel[e * NDOF + dof_subset[j]];
        for (int k = 0; k < 4; ++k) {
            float d[3];
            for (int c = 0; c < 3; ++c) d[c] = key_pos[(e * 4 + k) * 3 + c] - root_pos[e * 3 + c];
            orc_my_quat_rotate(hinv, d, o); o += 3;
        }
        for (int k = 0; k < 11; ++k) *o++ = betas[e * 17 + k];
    }
}

/* H:1281-1283 _action_to_pd_targets + H:1190-1196 zeroing of L/R_Hand and (freeze_toe) L/R_Toe targets */
void orc_pd_targets(int E, const float *actions, const float *offset, const float *scale,
                    const unsigned char *zero_mask, float *out) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e)
        for (int d = 0; d < NDOF; ++d)
            out[e * NDOF + d] = zero_mask[d] ? 0.0f : offset[d] + scale[d] * actions[e * NDOF + d];
}

/* vectorised wrappers over the helpers in oracle_math.h, for the golden-vector pin test */
void orc_vec_my_quat_rotate(int n, const float *q, const float *v, float *o) { for (int i = 0; i < n; ++i) orc_my_quat_rotate(q + 4 * i, v + 3 * i, o + 3 * i); }
void orc_vec_quat_mul(int n, const float *a, const float *b, float *o) { for (int i = 0; i < n; ++i) orc_quat_mul(a + 4 * i, b + 4 * i, o + 4 * i); }
void orc_vec_quat_apply(int n, const float *a, const float *b, float *o) { for (int i = 0; i < n; ++i) orc_quat_apply(a + 4 * i, b + 3 * i, o + 3 * i); }
void orc_vec_calc_heading(int n, const float *q, float *o) { for (int i = 0; i < n; ++i) o[i] = orc_calc_heading(q + 4 * i); }
void orc_vec_calc_heading_quat(int n, const float *q, float *o) { for (int i = 0; i < n; ++i) orc_calc_heading_quat(q + 4 * i, o + 4 * i); }
void orc_vec_calc_heading_quat_inv(int n, const float *q, float *o) { for (int i = 0; i < n; ++i) orc_calc_heading_quat_inv(q + 4 * i, o + 4 * i); }
void orc_vec_quat_to_tan_norm(int n, const float *q, float *o) { for (int i = 0; i < n; ++i) orc_quat_to_tan_norm(q + 4 * i, o + 6 * i); }
void orc_vec_exp_map_to_quat(int n, const float *e, float *o) { for (int i = 0; i < n; ++i) orc_exp_map_to_quat(e + 3 * i, o + 4 * i); }
void orc_vec_quat_to_exp_map(int n, const float *q, float *o) { for (int i = 0; i < n; ++i) orc_quat_to_exp_map(q + 4 * i, o + 3 * i); }
void orc_vec_slerp(int n, const float *a, const float *b, const float *t, float *o) { for (int i = 0; i < n; ++i) orc_slerp(a + 4 * i, b + 4 * i, t[i], o + 4 * i); }
void orc_vec_quat_from_angle_axis(int n, const float *ang, const float *ax, float *o) { for (int i = 0; i < n; ++i) orc_quat_from_angle_axis(ang[i], ax + 3 * i, o + 4 * i); }
void orc_vec_quat_apply_yaw(int n, const float *q, const float *v, float *o) { for (int i = 0; i < n; ++i) orc_quat_apply_yaw(q + 4 * i, v + 3 * i, o + 3 * i); }
