// ppo_capi.hip -- C ABI of the PPO + AMP learner's loss heads (include/emloco_predictor.h, "PPO loss heads").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdlib.h>
#include "ppo_kernels.hip"
#include "../../include/emloco_predictor.h"

namespace {
int lfail(int code, const char *what, hipError_t e = hipSuccess) {
    if (e != hipSuccess) fprintf(stderr, "[emloco] %s: %s\n", what, hipGetErrorString(e));
    else fprintf(stderr, "[emloco] %s\n", what);
    return code;
}
int launched(const char *what) {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : lfail(-2, what, e);
}
}  // namespace

extern "C" {

int emloco_ppo_actor_head_fwd(int B, int A, const float *mu, const float *logstd, const float *actions, const float *old_neglogp,
                              const float *advantages, const float *old_mu, const float *old_sigma, float e_clip, float *rows,
                              float *out5, void *stream) {
    if (B < 1 || A < 1 || !mu || !logstd || !actions || !old_neglogp || !advantages || !rows || !out5 || (!old_mu) != (!old_sigma))
        return lfail(-1, "emloco_ppo_actor_head_fwd: bad argument (rows = B x 5 floats of workspace, out5 = 5 floats)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(emloco::ppo_actor_head_fwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, B, A, mu, logstd, actions, old_neglogp,
                       advantages, old_mu, old_sigma, e_clip, rows);
    hipLaunchKernelGGL(emloco::ppo_rows_mean_kernel, dim3(1), dim3(256), 0, st, B, PPO_ACTOR_COLS, (const float *)rows, out5);
    return launched("emloco_ppo_actor_head_fwd launch");
}

int emloco_ppo_actor_head_bwd(int B, int A, const float *mu, const float *logstd, const float *actions, const float *old_neglogp,
                              const float *advantages, float e_clip, const float *grad3, float *dmu, float *dlogstd, void *stream) {
    if (B < 1 || A < 1 || !mu || !logstd || !actions || !old_neglogp || !advantages || !grad3 || !dmu)
        return lfail(-1, "emloco_ppo_actor_head_bwd: bad argument");
    hipLaunchKernelGGL(emloco::ppo_actor_head_bwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, B, A, mu, logstd, actions,
                       old_neglogp, advantages, e_clip, grad3, dmu, dlogstd);
    return launched("emloco_ppo_actor_head_bwd launch");
}

int emloco_ppo_critic_head_fwd(int B, const float *values, const float *old_values, const float *returns, float e_clip, int clip_value,
                               float *rows, float *out1, void *stream) {
    if (B < 1 || !values || !returns || (clip_value && !old_values) || !rows || !out1) return lfail(-1, "emloco_ppo_critic_head_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(emloco::ppo_critic_head_fwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, B, values, old_values, returns, e_clip,
                       clip_value, rows);
    hipLaunchKernelGGL(emloco::ppo_rows_mean_kernel, dim3(1), dim3(256), 0, st, B, 1, (const float *)rows, out1);
    return launched("emloco_ppo_critic_head_fwd launch");
}

int emloco_ppo_critic_head_bwd(int B, const float *values, const float *old_values, const float *returns, float e_clip, int clip_value,
                               const float *grad1, float *dvalues, void *stream) {
    if (B < 1 || !values || !returns || (clip_value && !old_values) || !grad1 || !dvalues) return lfail(-1, "emloco_ppo_critic_head_bwd: bad argument");
    hipLaunchKernelGGL(emloco::ppo_critic_head_bwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, values, old_values,
                       returns, e_clip, clip_value, grad1, dvalues);
    return launched("emloco_ppo_critic_head_bwd launch");
}

int emloco_ppo_disc_head_fwd(int n_agent, int n_demo, const float *agent_logits, const float *demo_logits, float *rows, float *out4, void *stream) {
    if (n_agent < 1 || n_demo < 1 || !agent_logits || !demo_logits || !rows || !out4)
        return lfail(-1, "emloco_ppo_disc_head_fwd: bad argument (rows = (n_agent + n_demo) x 2 floats of workspace, out4 = 4 floats)");
    hipStream_t st = (hipStream_t)stream;
    const int n = n_agent + n_demo;
    hipLaunchKernelGGL(emloco::ppo_disc_head_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_agent, n_demo, agent_logits, demo_logits, rows);
    hipLaunchKernelGGL(emloco::ppo_rows_mean_kernel, dim3(1), dim3(256), 0, st, n_agent, 2, (const float *)rows, out4);
    hipLaunchKernelGGL(emloco::ppo_rows_mean_kernel, dim3(1), dim3(256), 0, st, n_demo, 2, (const float *)(rows + 2 * (long)n_agent), out4 + 2);
    return launched("emloco_ppo_disc_head_fwd launch");
}

int emloco_ppo_disc_head_bwd(int n_agent, int n_demo, const float *agent_logits, const float *demo_logits, const float *grad2,
                             float *d_agent, float *d_demo, void *stream) {
    if (n_agent < 1 || n_demo < 1 || !agent_logits || !demo_logits || !grad2 || !d_agent || !d_demo) return lfail(-1, "emloco_ppo_disc_head_bwd: bad argument");
    const int n = n_agent + n_demo;
    hipLaunchKernelGGL(emloco::ppo_disc_head_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_agent, n_demo, agent_logits,
                       demo_logits, grad2, d_agent, d_demo);
    return launched("emloco_ppo_disc_head_bwd launch");
}

int emloco_ppo_gather_rows(int n_tables, int n_rows, const int64_t *idx, const float *const *src, float *const *dst, const int *cols, void *stream) {
    if (n_tables < 1 || n_tables > PPO_GATHER_MAX || n_rows < 1 || !idx || !src || !dst || !cols)
        return lfail(-1, "emloco_ppo_gather_rows: bad argument (1 .. 16 tables; src / dst / cols are HOST arrays of device pointers / row widths)");
    emloco::PpoGatherArgs a{};
    a.n_tables = n_tables; a.n_rows = n_rows; a.idx = (const long long *)idx;
    for (int t = 0; t < n_tables; ++t) {
        if (!src[t] || !dst[t] || cols[t] < 1) return lfail(-1, "emloco_ppo_gather_rows: null table or empty rows");
        a.src[t] = src[t]; a.dst[t] = dst[t]; a.cols[t] = cols[t];
    }
    const unsigned gx = (unsigned)((n_rows + 3) / 4 < 1024 ? (n_rows + 3) / 4 : 1024);
    hipLaunchKernelGGL(emloco::ppo_gather_rows_kernel, dim3(gx, (unsigned)n_tables), dim3(256), 0, (hipStream_t)stream, a);
    return launched("emloco_ppo_gather_rows launch");
}

}  // extern "C"
