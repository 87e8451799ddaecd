// TEST INFRASTRUCTURE ONLY: runs emloco_amd/csrc/ppo_kernels.hip (the PPO learner's loss heads) on the CPU through the emulation header.
#include "hip/hip_runtime.h"
#include "../../emloco_amd/csrc/ppo_kernels.hip"
using namespace emloco;

extern "C" int emu_ppo_actor_head(int B, int A, const float *mu, const float *logstd, const float *actions, const float *old_neglogp,
                                  const float *adv, const float *old_mu, const float *old_sigma, float e_clip, float *rows, float *out5,
                                  const float *grad3, float *dmu, float *dlogstd) {
    emu::launch((unsigned)((B + 3) / 4), 256, [&] { ppo_actor_head_fwd_kernel(B, A, mu, logstd, actions, old_neglogp, adv, old_mu, old_sigma, e_clip, rows); });
    emu::launch(1, 256, [&] { ppo_rows_mean_kernel(B, PPO_ACTOR_COLS, rows, out5); });
    emu::launch((unsigned)((B + 3) / 4), 256, [&] { ppo_actor_head_bwd_kernel(B, A, mu, logstd, actions, old_neglogp, adv, e_clip, grad3, dmu, dlogstd); });
    return 0;
}
extern "C" int emu_ppo_critic_head(int B, const float *v, const float *v_old, const float *ret, float e_clip, int clip_value, float *rows,
                                   float *out1, const float *grad1, float *dv) {
    emu::launch((unsigned)((B + 255) / 256), 256, [&] { ppo_critic_head_fwd_kernel(B, v, v_old, ret, e_clip, clip_value, rows); });
    emu::launch(1, 256, [&] { ppo_rows_mean_kernel(B, 1, rows, out1); });
    emu::launch((unsigned)((B + 255) / 256), 256, [&] { ppo_critic_head_bwd_kernel(B, v, v_old, ret, e_clip, clip_value, grad1, dv); });
    return 0;
}
extern "C" int emu_ppo_disc_head(int na, int nd, const float *a, const float *d, float *rows, float *out4, const float *grad2, float *da, float *dd) {
    const int n = na + nd;
    emu::launch((unsigned)((n + 255) / 256), 256, [&] { ppo_disc_head_fwd_kernel(na, nd, a, d, rows); });
    emu::launch(1, 256, [&] { ppo_rows_mean_kernel(na, 2, rows, out4); });
    emu::launch(1, 256, [&] { ppo_rows_mean_kernel(nd, 2, rows + 2 * (long)na, out4 + 2); });
    emu::launch((unsigned)((n + 255) / 256), 256, [&] { ppo_disc_head_bwd_kernel(na, nd, a, d, grad2, da, dd); });
    return 0;
}
