// TEST INFRASTRUCTURE ONLY: runs emloco_amd/csrc/task_kernels.hip on the CPU through tests/emu/hip/.
#include <cstring>
#include "hip/hip_runtime.h"
#include "../../emloco_amd/csrc/task_kernels.hip"
#include "../../emloco_amd/csrc/reset_kernels.hip"
#include "../../emloco_amd/csrc/chain_kernels.hip"

extern "C" int emu_task_post_physics(const EmlocoTaskBufs *b, int mode, const int32_t *env_ids, int n) {
    const int count = env_ids ? n : b->n_env;
    EmlocoTaskBufs t = *b;
    emu::launch((unsigned)count, 64, [&] { emloco::post_physics_kernel(t, mode, env_ids, count); });
    return 0;
}

extern "C" int emu_task_amp_rows(int n, const float *root_pos, const float *root_rot, const float *root_vel,
                                 const float *root_ang, const float *dof_pos, const float *dof_vel, const float *key_pos,
                                 const float *betas, const int32_t *subset, int n_sub, float *out) {
    emu::launch((unsigned)n, 64, [&] {
        emloco::amp_rows_kernel(n, root_pos, root_rot, root_vel, root_ang, dof_pos, dof_vel, key_pos, betas, subset, n_sub, out);
    });
    return 0;
}

extern "C" int emu_task_pd_targets(int n_env, const float *actions, const float *offset, const float *scale,
                                   const uint8_t *zero_mask, float *out) {
    const int total = n_env * 69;
    emu::launch((unsigned)((total + 255) / 256), 256, [&] { emloco::pd_targets_kernel(total, actions, offset, scale, zero_mask, out, (float *)nullptr); });
    return 0;
}

extern "C" int emu_compact_flags(const int64_t *flags, int n, int32_t *ids) {
    emu::launch(1, 1024, [&] { emloco::compact_flags_kernel(flags, n, ids, (int64_t *)nullptr); });
    return 0;
}

extern "C" int emu_task_traj_reset(const EmlocoResetBufs *b, const int32_t *env_ids, int n, const float *rnd,
                                   const float *init_pos, const float *root_vel) {
    EmlocoResetBufs t = *b;
    emu::launch((unsigned)n, 64, [&] { emloco::traj_reset_kernel(t, env_ids, n, rnd, init_pos, root_vel); });
    return 0;
}

extern "C" int emu_task_get_heights(const int16_t *hf, int rows, int cols, float hscale, float vscale, const float *pose7, int n,
                                    int grid, float *out_h, int64_t *out_px, int64_t *out_py) {
    emu::launch((unsigned)n, 64, [&] { emloco::get_heights_kernel(hf, rows, cols, hscale, vscale, pose7, n, grid, out_h, out_px, out_py); });
    return 0;
}

// the two-workgroup launch of the fused chain: block 0 compacts the flags (+ snapshot), block 1 sorts the dispatch order
extern "C" int emu_compact_order(const int64_t *flags, int n, int32_t *ids, int64_t *snapshot, const unsigned *ticks, int n_order, int *order,
                                 unsigned char *bucket_ws) {
    emu::launch(ticks ? 2u : 1u, 1024, [&] { emloco::compact_order_kernel(flags, n, ids, snapshot, ticks, n_order, order, bucket_ws); });
    return 0;
}

// reset_obs_kernel with an empty finished-env list: only the observation role does work (post-physics pass `live_mode` of every env
// whose snapshot entry is zero) -- the role arithmetic of the launch on the CPU
extern "C" int emu_reset_obs_live(const EmlocoTaskBufs *pb, int live_mode, const int64_t *skip, const int32_t *ids, int n, int n_slots, int h_slots, int n_hist) {
    EmlocoResetBufs rb; memset(&rb, 0, sizeof(rb));
    EmlocoSimDev sd; memset(&sd, 0, sizeof(sd));
    emloco::ChainArgs a; memset(&a, 0, sizeof(a));
    a.n = n; a.n_slots = n_slots; a.h_slots = h_slots; a.n_hist = n_hist; a.live_mode = live_mode; a.reset_mode = EMLOCO_POST_OBS | EMLOCO_POST_AMP_ROW;
    a.seeded = 1; a.ids = ids; a.skip = skip;
    const unsigned grid = (unsigned)(n_slots + h_slots * n_hist + pb->n_env);
    emu::launch(grid, 64, [&] { emloco::reset_obs_kernel(*pb, rb, sd, a); });
    return 0;
}

// the flags launch with the LocoVal return bookkeeping of every env behind its reward and reset flag
extern "C" int emu_task_post_physics_returns(const EmlocoTaskBufs *b, int mode, const EmlocoLocoValStep *step, const uint8_t *inverted) {
    EmlocoLocoValStep lv = *step;
    emu::launch((unsigned)b->n_env, 64, [&] { emloco::post_physics_returns_kernel(*b, mode, lv, inverted); });
    return 0;
}
