// locoval_returns_device.h -- the per-env bookkeeping of the LocoVal return (amp_continuous_value.py:63-64,93-118,126-129;
// vec_task_wrappers.py:50-66) as a device function of one wave: locoval_returns_kernel (predictor_kernels.hip) runs it as a launch of
// its own, the task's flags launch (task_kernels.hip: post_physics_kernel with a LocoVal step attached) runs it for the env whose
// reward and reset flag it has just computed -- one launch less on the chain between two rigid-body steps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/emloco_predictor.h"

namespace emloco {

// One wave per env: lanes copy the LocoVal inputs the task captured at reset into origin-relative form (first 13 waypoints), lane 0
// advances the per-env bookkeeping of the discounted return (inversion penalty, gamma^t, the step_to_pred cut-off) and emits the
// normalised target / weight of this step's fit.  r / a / done / inverted are read in lane 0 only.
//
// Two halves, because the two inputs of the bookkeeping become available at different times when an AMP discriminator scores the
// step: what the TASK knows (the inputs it captured at reset, the reward, the reset flag, the heading inversion) exists right
// behind the flags and is overwritten by the resets that follow; the discriminator's reward arrives three GEMMs later.
//   locoval_stage_env    the copies, and (staged mode) the penalised reward / the done flag into the step's staging arrays
//   locoval_advance_env  lane 0's bookkeeping from (r after the penalty, a, done)
// locoval_returns_env = both in one go (no discriminator, or one that is waited for); a step whose EmlocoLocoValStep carries
// staging arrays is staged by the task's flags launch and finished by locoval_returns_finish_kernel once `a` exists -- the same
// operations in the same order on the same values, so the results are the one-phase kernel's bit for bit.
__device__ __forceinline__ void locoval_stage_env(const EmlocoLocoValStep &t, int e, int lane) {
    const float *wp = t.waypoint_traj + (long)e * 45, *ip = t.init_pose + (long)e * 72;
    if (lane < 39) t.traj13[(long)e * 39 + lane] = wp[lane] - wp[lane % 3];
    for (int k = lane; k < 72; k += 64) t.pose[(long)e * 72 + k] = ip[k] - ip[k % 3];
    if (lane < 2) t.vel[(long)e * 2 + lane] = t.init_vel[(long)e * 2 + lane];
}

__device__ __forceinline__ float locoval_penalised(const EmlocoLocoValStep &t, float r, bool inverted) {
#ifndef EMLOCO_EMU
#pragma clang fp contract(off)
#endif
    if (inverted) r = r * (-t.inversion_penalty);
    return r;
}

__device__ __forceinline__ void locoval_advance_env(const EmlocoLocoValStep &t, int e, float r, float a, bool done) {
    // the bookkeeping follows the reference's torch expressions operation by operation (fixture locoval_returns.npz is matched
    // bit for bit): no multiply-add contraction here, whatever the translation unit's default is
#ifndef EMLOCO_EMU
#pragma clang fp contract(off)
#endif
    const float nd = done ? 0.0f : 1.0f;
    const float cr = t.current_rewards[e] + r;
    const float len = t.current_lengths[e] + 1.0f;
    const float coef = t.discount_coefs[e];
    const float comb = t.current_combined_rewards[e] + (r + a) * coef;
    const bool emit = done ? (len <= (float)t.step_to_pred) : (len == (float)t.step_to_pred);
    const float G = emit ? comb : 0.0f;
    t.target[e] = (G - t.min_cum_rewards) / (t.max_cum_rewards - t.min_cum_rewards);
    t.weight[e] = G != 0.0f ? 1.0f : 0.0f;
    t.current_combined_rewards[e] = comb * nd;
    t.discount_coefs[e] = done ? 1.0f : coef * t.gamma;
    t.current_rewards[e] = cr * nd;
    t.current_lengths[e] = len * nd;
}

__device__ __forceinline__ void locoval_returns_env(const EmlocoLocoValStep &t, int e, int lane, float r, float a, bool done, bool inverted) {
    locoval_stage_env(t, e, lane);
    if (lane == 0) {
        r = locoval_penalised(t, r, inverted);
        if (t.staged_reward) {                   // staged mode: the bookkeeping waits for the discriminator's reward
            t.staged_reward[e] = r;
            t.staged_done[e] = done ? 1 : 0;
        } else {
            locoval_advance_env(t, e, r, a, done);
        }
    }
}

}  // namespace emloco
