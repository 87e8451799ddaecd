// task_capi.hip -- C ABI (include/emloco_task.h) over the fused post-physics kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include "task_kernels.hip"
#include "reset_kernels.hip"
#include "chain_kernels.hip"
#include "sim_state.h"
#include "../../include/emloco_predictor.h"


namespace {
hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
bool g_timing = false, g_pending = false;
float g_last_ms = -1.0f;
thread_local std::string g_terr;
long long *g_chain_prof = nullptr;
int tfail(int code, const char *what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof(buf), "%s", what);
    g_terr = buf;
    fprintf(stderr, "[emloco] %s\n", buf);
    return code;
}
}  // namespace

#define THIPCHK(expr)                                                  \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return tfail(-2, #expr, e_);             \
    } while (0)

extern "C" {

int emloco_task_enable_timing(int on) {
    if (on && !g_ev0) {
        THIPCHK(hipEventCreate(&g_ev0));
        THIPCHK(hipEventCreate(&g_ev1));
    }
    g_timing = on != 0;
    return 0;
}

float emloco_task_last_ms(void) {
    if (g_pending) {
        if (hipEventSynchronize(g_ev1) == hipSuccess) {
            float ms = -1.0f;
            if (hipEventElapsedTime(&ms, g_ev0, g_ev1) == hipSuccess) g_last_ms = ms;
        }
        g_pending = false;
    }
    return g_last_ms;
}

int emloco_task_post_physics(const EmlocoTaskBufs *b, int mode, const int32_t *dev_env_ids, int n, void *stream) {
    if (!b) return tfail(-1, "emloco_task_post_physics: null buffers");
    if (b->n_env < 1 || b->hf_rows < 2 || b->hf_cols < 2) return tfail(-1, "emloco_task_post_physics: bad sizes");
    if (!b->rb_state || !b->progress_buf || !b->traj_verts) return tfail(-1, "emloco_task_post_physics: missing tensors");
    if ((mode & EMLOCO_POST_OBS) && (!b->obs_buf || !b->flip_obs_buf || !b->heightfield || !b->betas || !b->left_to_right))
        return tfail(-1, "emloco_task_post_physics: observation buffers missing");
    if ((mode & EMLOCO_POST_REWARD) && (!b->rew_buf || !b->reward_raw || !b->dof_force || !b->dof_state))
        return tfail(-1, "emloco_task_post_physics: reward buffers missing");
    if ((mode & EMLOCO_POST_SKIP_DONE) && !b->reset_buf) return tfail(-1, "emloco_task_post_physics: SKIP_DONE needs reset_buf");
    if ((mode & EMLOCO_POST_RESET) && (!b->reset_buf || !b->terminate_buf || !b->contact_force || !b->contact_body_mask))
        return tfail(-1, "emloco_task_post_physics: reset buffers missing");
    if ((mode & (EMLOCO_POST_AMP_ROW | EMLOCO_POST_AMP_SHIFT)) && (!b->amp_obs_buf || !b->dof_subset || !b->key_bodies || b->n_dof_subset > 64 || b->n_dof_subset % 3))
        return tfail(-1, "emloco_task_post_physics: AMP buffers missing");
    const int count = dev_env_ids ? n : b->n_env;
    if (count < 0 || count > b->n_env) return tfail(-1, "emloco_task_post_physics: bad env count");
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (g_timing) THIPCHK(hipEventRecord(g_ev0, st));
    hipLaunchKernelGGL(emloco::post_physics_kernel, dim3((unsigned)count), dim3(64), 0, st, *b, mode, dev_env_ids, count);
    THIPCHK(hipGetLastError());
    if (g_timing) { THIPCHK(hipEventRecord(g_ev1, st)); g_pending = true; }
    return 0;
}

int emloco_task_post_physics_returns(const EmlocoTaskBufs *b, int mode, const void *step_, const uint8_t *dev_inverted, void *stream) {
    const EmlocoLocoValStep *step = (const EmlocoLocoValStep *)step_;
    if (!b || !step) return tfail(-1, "emloco_task_post_physics_returns: null argument");
    if (!(mode & EMLOCO_POST_REWARD) || !(mode & EMLOCO_POST_RESET) || (mode & EMLOCO_POST_SKIP_DONE))
        return tfail(-1, "emloco_task_post_physics_returns: the return bookkeeping follows the reward and the reset flag of the same launch, for every env");
    if (step->n_env != b->n_env || !step->current_rewards || !step->current_lengths || !step->current_combined_rewards || !step->discount_coefs ||
        !step->waypoint_traj || !step->init_pose || !step->init_vel || !step->traj13 || !step->pose || !step->vel || !step->target || !step->weight)
        return tfail(-1, "emloco_task_post_physics_returns: LocoVal step buffers missing or for another env count");
    if ((step->staged_reward == nullptr) != (step->staged_done == nullptr))
        return tfail(-1, "emloco_task_post_physics_returns: staged_reward and staged_done go together");
    if (b->n_env < 1 || b->hf_rows < 2 || b->hf_cols < 2 || !b->rb_state || !b->progress_buf || !b->traj_verts || !b->rew_buf || !b->reward_raw ||
        !b->dof_force || !b->dof_state || !b->reset_buf || !b->terminate_buf || !b->contact_force || !b->contact_body_mask)
        return tfail(-1, "emloco_task_post_physics_returns: task buffers missing");
    if ((mode & EMLOCO_POST_OBS) && (!b->obs_buf || !b->flip_obs_buf || !b->heightfield || !b->betas || !b->left_to_right))
        return tfail(-1, "emloco_task_post_physics_returns: observation buffers missing");
    if ((mode & (EMLOCO_POST_AMP_ROW | EMLOCO_POST_AMP_SHIFT)) && (!b->amp_obs_buf || !b->dof_subset || !b->key_bodies || !b->betas || b->n_dof_subset > 64 || b->n_dof_subset % 3))
        return tfail(-1, "emloco_task_post_physics_returns: AMP buffers missing");
    hipStream_t st = (hipStream_t)stream;
    if (g_timing) THIPCHK(hipEventRecord(g_ev0, st));
    hipLaunchKernelGGL(emloco::post_physics_returns_kernel, dim3((unsigned)b->n_env), dim3(64), 0, st, *b, mode, *step, dev_inverted);
    THIPCHK(hipGetLastError());
    if (g_timing) { THIPCHK(hipEventRecord(g_ev1, st)); g_pending = true; }
    return 0;
}

int emloco_task_amp_rows(int n, const float *root_pos, const float *root_rot, const float *root_vel,
                         const float *root_ang_vel, const float *dof_pos, const float *dof_vel,
                         const float *key_pos, const float *betas, const int32_t *dof_subset,
                         int n_dof_subset, float *out, void *stream) {
    if (n < 0 || !root_pos || !root_rot || !root_vel || !root_ang_vel || !dof_pos || !dof_vel || !key_pos || !betas || !dof_subset || !out)
        return tfail(-1, "emloco_task_amp_rows: bad argument");
    if (n_dof_subset > 64 || n_dof_subset % 3) return tfail(-1, "emloco_task_amp_rows: dof subset must be <= 64 and a multiple of 3");
    if (n == 0) return 0;
    hipLaunchKernelGGL(emloco::amp_rows_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, n, root_pos, root_rot,
                       root_vel, root_ang_vel, dof_pos, dof_vel, key_pos, betas, dof_subset, n_dof_subset, out);
    THIPCHK(hipGetLastError());
    return 0;
}

int emloco_task_reset(EmlocoSim *sim, const EmlocoResetBufs *b, const int32_t *dev_env_ids, int n, const float *dev_rnd, void *stream) {
    if (!sim || !b || !dev_env_ids || !dev_rnd) return tfail(-1, "emloco_task_reset: null argument");
    if (!sim->prepared) return tfail(-3, "emloco_task_reset: sim not prepared");
    if (n < 0 || n > sim->n_env) return tfail(-1, "emloco_task_reset: bad env count");
    if (n == 0) return 0;
    if (!b->gts || !b->grs || !b->lrs || !b->gvs || !b->gavs || !b->dvs || !b->motion_len || !b->motion_dt || !b->motion_nframes ||
        !b->motion_start || b->n_motions < 1 || !b->heightfield || !b->betas || !b->key_bodies || !b->dof_subset || !b->traj_verts ||
        !b->inverted || !b->progress_buf || !b->reset_buf || !b->terminate_buf || !b->waypoint_traj || !b->init_pose || !b->init_vel ||
        !b->amp_obs_buf || !b->motion_ids || !b->motion_times || !b->ground_h)
        return tfail(-1, "emloco_task_reset: missing buffers");
    if (!(b->flags & EMLOCO_RESET_FIXED_LOCATION) && (!b->valid_x || !b->valid_y || b->n_valid < 1))
        return tfail(-1, "emloco_task_reset: no valid locations");
    if ((b->flags & EMLOCO_RESET_REAL_PATH) && b->n_real > 0 && !b->real_traj) return tfail(-1, "emloco_task_reset: real_path without data");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(n < 256 ? n : 256);        // grid-stride kernels: see reset_kernels.hip
    hipLaunchKernelGGL(emloco::reset_sample_kernel, dim3(grid), dim3(64), 0, st, *b, sim->dev, dev_env_ids, n, dev_rnd);
    THIPCHK(hipGetLastError());
    const int rc = emloco_sim_fk_indexed(sim, dev_env_ids, n, stream);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(emloco::reset_finish_kernel, dim3(grid), dim3(64), 0, st, *b, sim->dev, dev_env_ids, n, dev_rnd);
    THIPCHK(hipGetLastError());
    if (b->flags & EMLOCO_RESET_NO_AMP_HISTORY) return 0;
    hipLaunchKernelGGL(emloco::reset_amp_history_kernel, dim3(grid, EMLOCO_AMP_STEPS - 1), dim3(64), 0, st, *b, dev_env_ids, n);
    THIPCHK(hipGetLastError());
    return 0;
}

int emloco_task_reset_amp_history(const EmlocoResetBufs *b, const int32_t *dev_env_ids, int n, void *stream) {
    if (!b || !dev_env_ids) return tfail(-1, "emloco_task_reset_amp_history: null argument");
    if (n < 0) return tfail(-1, "emloco_task_reset_amp_history: bad env count");
    if (n == 0) return 0;
    if (!b->gts || !b->grs || !b->lrs || !b->gvs || !b->gavs || !b->dvs || !b->motion_len || !b->motion_dt || !b->motion_nframes ||
        !b->motion_start || !b->betas || !b->key_bodies || !b->dof_subset || !b->amp_obs_buf || !b->motion_ids || !b->motion_times)
        return tfail(-1, "emloco_task_reset_amp_history: missing buffers");
    const unsigned grid = (unsigned)(n < 256 ? n : 256);
    hipLaunchKernelGGL(emloco::reset_amp_history_kernel, dim3(grid, EMLOCO_AMP_STEPS - 1), dim3(64), 0, (hipStream_t)stream, *b, dev_env_ids, n);
    THIPCHK(hipGetLastError());
    return 0;
}

int emloco_task_reset_seeded(EmlocoSim *sim, const EmlocoResetBufs *b, const int32_t *dev_env_ids, int n, uint64_t seed,
                             float *dev_rnd_ws, void *stream) {
    if (!dev_env_ids || !dev_rnd_ws) return tfail(-1, "emloco_task_reset_seeded: null argument");
    if (n < 0) return tfail(-1, "emloco_task_reset_seeded: bad env count");
    if (n == 0) return 0;
    const unsigned grid = (unsigned)(n < 256 ? n : 256);
    hipLaunchKernelGGL(emloco::reset_fill_rnd_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, dev_env_ids, n,
                       (unsigned)(seed & 0xffffffffu), (unsigned)(seed >> 32), dev_rnd_ws);
    THIPCHK(hipGetLastError());
    if (!b) return tfail(-1, "emloco_task_reset_seeded: null argument");
    EmlocoResetBufs keyed = *b;                 // a fresh real-path permutation per call (reset_kernels.hip: real_pick_perm)
    keyed.real_pick = nullptr;
    keyed.real_pick_key = (uint32_t)((seed * 0xD6E8FEB86659FD93ull) >> 32);
    return emloco_task_reset(sim, &keyed, dev_env_ids, n, dev_rnd_ws, stream);
}

int emloco_task_traj_reset(const EmlocoResetBufs *b, const int32_t *dev_env_ids, int n, const float *dev_rnd,
                           const float *dev_init_pos, const float *dev_root_vel, void *stream) {
    if (!b || !dev_env_ids || !dev_rnd || !dev_init_pos || !dev_root_vel) return tfail(-1, "emloco_task_traj_reset: null argument");
    if (n < 0) return tfail(-1, "emloco_task_traj_reset: bad env count");
    if (n == 0) return 0;
    if (!b->traj_verts || !b->inverted) return tfail(-1, "emloco_task_traj_reset: missing buffers");
    if ((b->flags & EMLOCO_RESET_REAL_PATH) && b->n_real > 0 && !b->real_traj) return tfail(-1, "emloco_task_traj_reset: real_path without data");
    const unsigned grid = (unsigned)(n < 256 ? n : 256);
    hipLaunchKernelGGL(emloco::traj_reset_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, *b, dev_env_ids, n, dev_rnd,
                       dev_init_pos, dev_root_vel);
    THIPCHK(hipGetLastError());
    return 0;
}

int emloco_task_get_heights(const int16_t *dev_heightfield, int rows, int cols, float hscale, float vscale, const float *dev_pose7,
                            int n, int grid, float *dev_heights, int64_t *dev_px, int64_t *dev_py, void *stream) {
    if (!dev_heightfield || !dev_pose7 || rows < 2 || cols < 2 || n < 0) return tfail(-1, "emloco_task_get_heights: bad argument");
    if ((dev_px == nullptr) != (dev_py == nullptr)) return tfail(-1, "emloco_task_get_heights: px and py go together");
    if (n == 0) return 0;
    hipLaunchKernelGGL(emloco::get_heights_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, dev_heightfield, rows, cols,
                       hscale, vscale, dev_pose7, n, grid, dev_heights, dev_px, dev_py);
    THIPCHK(hipGetLastError());
    return 0;
}

int emloco_task_compact_done_snapshot(const int64_t *dev_flags, int n, int32_t *dev_ids, int64_t *dev_snapshot, void *stream) {
    if (!dev_flags || !dev_ids || n < 1) return tfail(-1, "emloco_task_compact_done: bad argument (ids holds n + 1 entries)");
    hipLaunchKernelGGL(emloco::compact_flags_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dev_flags, n, dev_ids, dev_snapshot);
    THIPCHK(hipGetLastError());
    return 0;
}

int emloco_task_compact_done(const int64_t *dev_flags, int n, int32_t *dev_ids, void *stream) {
    return emloco_task_compact_done_snapshot(dev_flags, n, dev_ids, nullptr, stream);
}

int emloco_task_pd_targets_copy(int n_env, const float *actions, const float *offset, const float *scale,
                                const uint8_t *zero_mask, float *pd_targets, float *actions_copy, void *stream) {
    if (n_env < 1 || !actions || !offset || !scale || !zero_mask || !pd_targets) return tfail(-1, "emloco_task_pd_targets: bad argument");
    if (actions_copy == actions) actions_copy = nullptr;
    const int total = n_env * 69;
    hipLaunchKernelGGL(emloco::pd_targets_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       total, actions, offset, scale, zero_mask, pd_targets, actions_copy);
    THIPCHK(hipGetLastError());
    return 0;
}

int emloco_task_pd_targets(int n_env, const float *actions, const float *offset, const float *scale,
                           const uint8_t *zero_mask, float *pd_targets, void *stream) {
    return emloco_task_pd_targets_copy(n_env, actions, offset, scale, zero_mask, pd_targets, nullptr, stream);
}

int emloco_task_compact_done_order(EmlocoSim *sim, const int64_t *dev_flags, int n, int32_t *dev_ids, int64_t *dev_snapshot, void *stream) {
    if (!dev_flags || !dev_ids || n < 1) return tfail(-1, "emloco_task_compact_done_order: bad argument (ids holds n + 1 entries)");
    const bool sort = sim && sim->prepared && sim->cost_order && sim->d_ticks.p && sim->d_order.p;
    hipLaunchKernelGGL(emloco::compact_order_kernel, dim3(sort ? 2 : 1), dim3(1024), 0, (hipStream_t)stream, dev_flags, n, dev_ids, dev_snapshot,
                       sort ? sim->d_ticks.p : nullptr, sort ? sim->n_env : 0, sort ? sim->d_order.p : nullptr, sort ? sim->d_order_ws.p : nullptr);
    THIPCHK(hipGetLastError());
    if (sort) sim->order_ready = true;
    return 0;
}

int emloco_task_reset_obs_pooled(EmlocoSim *sim, const EmlocoResetBufs *rb, const EmlocoTaskBufs *pb, int live_mode, const int64_t *dev_skip,
                                 const int32_t *dev_env_ids, int n, uint64_t seed, float *dev_rnd_ws, const float *dev_rnd,
                                 const EmlocoResetPool *pool, void *stream);

// internal diagnostic (not in the header): the first call allocates 16 wall-clock stamps (100 MHz) that every later
// emloco_task_reset_obs launch overwrites -- [0..5] reset slot 0: start, random row, sample, kinematics, finish, observations;
// [6, 7] last AMP history row of entry 0; [8, 9] / [10, 11] the observation workgroups of env 0 / the last env -- and copies them out
int emloco_task_chain_profile(long long *host16) {
    if (!g_chain_prof) { THIPCHK(hipMalloc((void **)&g_chain_prof, 16 * sizeof(long long))); THIPCHK(hipMemset(g_chain_prof, 0, 16 * sizeof(long long))); }
    THIPCHK(hipDeviceSynchronize());
    if (host16) THIPCHK(hipMemcpy(host16, g_chain_prof, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}

#ifdef EMLOCO_POST_PROFILE
// internal diagnostic (builds with -DEMLOCO_POST_PROFILE only): 8 wall-clock stamps of the last env's post-physics workgroup
int emloco_task_post_profile(long long *host8) {
    static long long *buf = nullptr;
    if (!buf) {
        THIPCHK(hipMalloc((void **)&buf, 8 * sizeof(long long)));
        THIPCHK(hipMemset(buf, 0, 8 * sizeof(long long)));
        THIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(emloco::g_post_prof), &buf, sizeof(buf)));
    }
    THIPCHK(hipDeviceSynchronize());
    if (host8) THIPCHK(hipMemcpy(host8, buf, 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif

int emloco_task_reset_obs(EmlocoSim *sim, const EmlocoResetBufs *rb, const EmlocoTaskBufs *pb, int live_mode, const int64_t *dev_skip,
                          const int32_t *dev_env_ids, int n, uint64_t seed, float *dev_rnd_ws, const float *dev_rnd, void *stream) {
    return emloco_task_reset_obs_pooled(sim, rb, pb, live_mode, dev_skip, dev_env_ids, n, seed, dev_rnd_ws, dev_rnd, nullptr, stream);
}

int emloco_task_reset_obs_pooled(EmlocoSim *sim, const EmlocoResetBufs *rb, const EmlocoTaskBufs *pb, int live_mode, const int64_t *dev_skip,
                                 const int32_t *dev_env_ids, int n, uint64_t seed, float *dev_rnd_ws, const float *dev_rnd,
                                 const EmlocoResetPool *pool, void *stream) {
    if (!sim || !rb || !pb || !dev_env_ids) return tfail(-1, "emloco_task_reset_obs: null argument");
    if (pool && (pool->k < 0 || pool->k > 4096 || (pool->k > 0 && ((pool->cur && !pool->cur_tag) || (pool->next && !pool->next_tag)))))
        return tfail(-1, "emloco_task_reset_obs: bad pool (k in [0, 4096], every pool buffer with its tag array)");
    if (!sim->prepared) return tfail(-3, "emloco_task_reset_obs: sim not prepared");
    if (n < 0 || n > sim->n_env) return tfail(-1, "emloco_task_reset_obs: bad env count");
    if (!dev_rnd && !dev_rnd_ws) return tfail(-1, "emloco_task_reset_obs: neither random rows nor a workspace for them");
    if (live_mode && !dev_skip) return tfail(-1, "emloco_task_reset_obs: the live envs' role needs the flag snapshot");
    if (live_mode & ~(EMLOCO_POST_OBS | EMLOCO_POST_AMP_SHIFT | EMLOCO_POST_AMP_ROW | EMLOCO_POST_SKIP_DONE))
        return tfail(-1, "emloco_task_reset_obs: the live envs' role builds observations / AMP rows only (progress, reward and flags belong to the flags launch)");
    if (pb->n_env != sim->n_env) return tfail(-1, "emloco_task_reset_obs: task buffers and simulator disagree on the env count");
    const EmlocoResetBufs *b = rb;
    if (!b->gts || !b->grs || !b->lrs || !b->gvs || !b->gavs || !b->dvs || !b->motion_len || !b->motion_dt || !b->motion_nframes ||
        !b->motion_start || b->n_motions < 1 || !b->heightfield || !b->betas || !b->key_bodies || !b->dof_subset || !b->traj_verts ||
        !b->inverted || !b->progress_buf || !b->reset_buf || !b->terminate_buf || !b->waypoint_traj || !b->init_pose || !b->init_vel ||
        !b->amp_obs_buf || !b->motion_ids || !b->motion_times || !b->ground_h)
        return tfail(-1, "emloco_task_reset_obs: missing reset buffers");
    if (!(b->flags & EMLOCO_RESET_FIXED_LOCATION) && (!b->valid_x || !b->valid_y || b->n_valid < 1))
        return tfail(-1, "emloco_task_reset_obs: no valid locations");
    if ((b->flags & EMLOCO_RESET_REAL_PATH) && b->n_real > 0 && !b->real_traj) return tfail(-1, "emloco_task_reset_obs: real_path without data");
    if (!pb->rb_state || !pb->progress_buf || !pb->traj_verts || !pb->obs_buf || !pb->flip_obs_buf || !pb->heightfield || !pb->betas ||
        !pb->left_to_right || !pb->amp_obs_buf || !pb->dof_subset || !pb->key_bodies || !pb->dof_state || pb->n_dof_subset > 64 || pb->n_dof_subset % 3)
        return tfail(-1, "emloco_task_reset_obs: missing observation buffers");
    emloco::ChainArgs a;
    a.n = n;
    a.n_slots = n < 256 ? (n > 0 ? n : 1) : 256;
    a.h_slots = n < 256 ? (n > 0 ? n : 1) : 256;
    a.n_hist = (b->flags & EMLOCO_RESET_NO_AMP_HISTORY) ? 0 : EMLOCO_AMP_STEPS - 1;
    a.live_mode = live_mode & ~EMLOCO_POST_SKIP_DONE;
    a.reset_mode = EMLOCO_POST_OBS | EMLOCO_POST_AMP_ROW;
    a.seeded = dev_rnd ? 0 : 1;
    a.seed_lo = (unsigned)(seed & 0xffffffffu); a.seed_hi = (unsigned)(seed >> 32);
    a.ids = dev_env_ids; a.skip = dev_skip; a.rnd_in = dev_rnd; a.rnd_ws = dev_rnd_ws;
    a.prof = g_chain_prof;
    const bool pooled = pool && pool->k > 0 && a.seeded;
    a.pool_k = pooled ? pool->k : 0;
    a.pool_cur = pooled ? pool->cur : nullptr; a.tag_cur = pooled ? (const unsigned long long *)pool->cur_tag : nullptr;
    a.pool_next = pooled ? pool->next : nullptr; a.tag_next = pooled ? (unsigned long long *)pool->next_tag : nullptr;
    const uint64_t nseed = pooled ? pool->next_seed : 0;
    a.nseed_lo = (unsigned)(nseed & 0xffffffffu); a.nseed_hi = (unsigned)(nseed >> 32);
    a.next_key = (uint32_t)((nseed * 0xD6E8FEB86659FD93ull) >> 32);
    EmlocoResetBufs keyed = *rb;
    if (a.seeded) {                              // a fresh real-path permutation per call, as emloco_task_reset_seeded
        keyed.real_pick = nullptr;
        keyed.real_pick_key = (uint32_t)((seed * 0xD6E8FEB86659FD93ull) >> 32);
    }
    const unsigned grid = (unsigned)(a.n_slots + a.h_slots * a.n_hist + (a.pool_next ? a.pool_k * 2 : 0) + (a.live_mode ? pb->n_env : 0));
    if (n == 0 && !a.live_mode && !a.pool_next) return 0;
    hipLaunchKernelGGL(emloco::reset_obs_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, *pb, keyed, sim->dev, a);
    THIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
