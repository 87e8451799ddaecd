// ppo_kernels.hip -- the loss heads of the PPO + AMP learner (configs[1]'s update, amp_continuous.py:335-425 through
// common_agent.py:426-468 and the rl_games 1.1.4 pieces restated in learning/amp_agent.py) as a few launches.
//
// In torch the tail between the networks' outputs and the scalar loss is ~140 elementwise / reduction launches of 4-5 us each per
// optimiser step (profiles/r05_ppo_step_trace.txt: 300 elementwise launches, 1.45 ms of a 5.4 ms step, most of them here), forward and
// backward: neglogp, the clipped surrogate, the bound loss, the entropy, the policy KL; the clipped value loss; the discriminator's two
// binary cross entropies and accuracies.  Three heads, each one forward launch over the rows + one fixed-order mean + one backward launch.
// Rows are independent: one wave per row for the actor head (69 actions: two per lane), one thread per row for the scalar heads.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_math.h"

namespace emloco {

#define PPO_ACTOR_COLS 5      /* per row: surrogate, entropy, bound loss, clipped (0 / 1), KL(new || old) */

__device__ __forceinline__ float ppo_sq(float x) { return x * x; }

// One wave per row.  neglogp = 0.5 sum(((x - mu) / sigma)^2) + 0.5 log(2 pi) A + sum(logstd)     (ModelA2CContinuousLogStd.neglogp)
// surrogate = max(-adv ratio, -adv clamp(ratio, 1 - e, 1 + e)), ratio = exp(old_neglogp - neglogp)   (common_agent.py actor_loss)
// entropy = sum(0.5 + 0.5 log(2 pi) + logstd);  bound = sum(min(mu + 1, 0)^2 + max(mu - 1, 0)^2)      (bound_loss, soft bound 1.0)
// kl = sum(log(sigma_old / sigma + 1e-5) + (sigma^2 + (mu_old - mu)^2) / (2 (sigma_old^2 + 1e-5)) - 0.5)   (torch_ext.policy_kl)
__global__ void __launch_bounds__(256)
ppo_actor_head_fwd_kernel(int B, int A, const float *mu, const float *logstd, const float *actions, const float *old_neglogp,
                          const float *adv, const float *old_mu, const float *old_sigma, float e_clip, float *rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const long o = (long)row * A;
    float sq = 0.0f, ls = 0.0f, bd = 0.0f, kl = 0.0f;
    for (int j = lane; j < A; j += 64) {
        const float m = mu[o + j], l = logstd[o + j], s = expf(l);
        const float t = (actions[o + j] - m) / s;
        sq += t * t;
        ls += l;
        bd += ppo_sq(fminf(m + 1.0f, 0.0f)) + ppo_sq(fmaxf(m - 1.0f, 0.0f));
        if (old_mu) {
            const float s1 = old_sigma[o + j], m1 = old_mu[o + j];
            kl += logf(s1 / s + 1e-5f) + (s * s + ppo_sq(m1 - m)) / (2.0f * (s1 * s1 + 1e-5f)) - 0.5f;
        }
    }
    sq = wave_sum(sq); ls = wave_sum(ls); bd = wave_sum(bd); kl = wave_sum(kl);
    if (lane == 0) {
        const float half_log_2pi = 0.91893853320467274178f;
        const float nlp = 0.5f * sq + half_log_2pi * (float)A + ls;
        const float ratio = expf(old_neglogp[row] - nlp), a = adv[row];
        const float s1 = -a * ratio, s2 = -a * fminf(fmaxf(ratio, 1.0f - e_clip), 1.0f + e_clip);
        float *r = rows + (long)row * PPO_ACTOR_COLS;
        r[0] = fmaxf(s1, s2);
        r[1] = (0.5f + half_log_2pi) * (float)A + ls;
        r[2] = bd;
        r[3] = fabsf(ratio - 1.0f) > e_clip ? 1.0f : 0.0f;
        r[4] = kl;
    }
}

// out[c] = mean over the rows of rows[.][c], fixed order (one workgroup: thread t sums rows t, t + 256, ..., then a tree)
__global__ void __launch_bounds__(256)
ppo_rows_mean_kernel(int B, int ncol, const float *rows, float *out) {
    __shared__ float sh[256];
    for (int c = 0; c < ncol; ++c) {
        float s = 0.0f;
        for (int r = threadIdx.x; r < B; r += 256) s += rows[(long)r * ncol + c];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[c] = sh[0] / (float)B;
        __syncthreads();
    }
}

// d(ga mean(surrogate) + ge mean(entropy) + gb mean(bound)) / d(mu, logstd); g = device scalars [ga, ge, gb].
// torch's maximum hands a tie half to each side and clamp passes the gradient on its closed interval: inside the clip range both
// sides of the max carry the same derivative, outside it the clamped side carries none.
__global__ void __launch_bounds__(256)
ppo_actor_head_bwd_kernel(int B, int A, const float *mu, const float *logstd, const float *actions, const float *old_neglogp,
                          const float *adv, float e_clip, const float *g, float *dmu, float *dlogstd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const long o = (long)row * A;
    float sq = 0.0f, ls = 0.0f;
    for (int j = lane; j < A; j += 64) {
        const float l = logstd[o + j];
        const float t = (actions[o + j] - mu[o + j]) / expf(l);
        sq += t * t;
        ls += l;
    }
    sq = wave_sum(sq); ls = wave_sum(ls);
    const float half_log_2pi = 0.91893853320467274178f;
    const float nlp = 0.5f * sq + half_log_2pi * (float)A + ls;
    const float ratio = expf(old_neglogp[row] - nlp), a = adv[row];
    const float lo = 1.0f - e_clip, hi = 1.0f + e_clip;
    const float s1 = -a * ratio, s2 = -a * fminf(fmaxf(ratio, lo), hi);
    const float w1 = s1 > s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f), w2 = 1.0f - w1;
    const float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    const float invB = 1.0f / (float)B;
    const float c_nlp = g[0] * invB * a * ratio * (w1 + w2 * inr);        // d(ga mean surrogate) / d neglogp of this row
    const float c_ent = g[1] * invB, c_bd = g[2] * invB;
    for (int j = lane; j < A; j += 64) {
        const float m = mu[o + j], s = expf(logstd[o + j]);
        const float t = (actions[o + j] - m) / s;
        dmu[o + j] = c_nlp * (0.0f - t / s) + c_bd * 2.0f * (fminf(m + 1.0f, 0.0f) + fmaxf(m - 1.0f, 0.0f));
        if (dlogstd) dlogstd[o + j] = c_nlp * (1.0f - t * t) + c_ent;
    }
}

// Clipped value loss (common_agent.py critic_loss): max((v - R)^2, (v_old + clamp(v - v_old, -e, e) - R)^2), or (R - v)^2 unclipped;
// rows[r] = the row's loss (ppo_rows_mean_kernel reduces it)
__global__ void ppo_critic_head_fwd_kernel(int B, const float *v, const float *v_old, const float *ret, float e_clip, int clip_value, float *rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B) return;
    const float d1 = v[r] - ret[r];
    float l = d1 * d1;
    if (clip_value) {
        const float vc = v_old[r] + fminf(fmaxf(v[r] - v_old[r], 0.0f - e_clip), e_clip);
        const float d2 = vc - ret[r];
        l = fmaxf(l, d2 * d2);
    }
    rows[r] = l;
}
__global__ void ppo_critic_head_bwd_kernel(int B, const float *v, const float *v_old, const float *ret, float e_clip, int clip_value, const float *g, float *dv) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B) return;
    const float d1 = v[r] - ret[r];
    float grad = 2.0f * d1;
    if (clip_value) {
        const float dlt = v[r] - v_old[r];
        const float vc = v_old[r] + fminf(fmaxf(dlt, 0.0f - e_clip), e_clip);
        const float d2 = vc - ret[r];
        const float l1 = d1 * d1, l2 = d2 * d2;
        const float w1 = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f);
        const float inr = (dlt >= 0.0f - e_clip && dlt <= e_clip) ? 1.0f : 0.0f;
        grad = w1 * 2.0f * d1 + (1.0f - w1) * 2.0f * d2 * inr;
    }
    dv[r] = g[0] * grad / (float)B;
}

// The discriminator's cross entropies (amp_continuous.py:515-558: BCEWithLogitsLoss against 0 for the agent / replay rows, against
// 1 for the demo rows) and accuracies.  rows [n_agent + n_demo][2]: (bce, correct); the two means are taken per group by the host
// wrapper (two ppo_rows_mean_kernel launches).  bce(x, t) = max(x, 0) - x t + log(1 + exp(-|x|)).
__global__ void ppo_disc_head_fwd_kernel(int n_agent, int n_demo, const float *agent_logit, const float *demo_logit, float *rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_agent + n_demo) return;
    const bool demo = r >= n_agent;
    const float x = demo ? demo_logit[r - n_agent] : agent_logit[r];
    const float sp = fmaxf(x, 0.0f) + log1pf(expf(0.0f - fabsf(x)));
    rows[2 * r] = demo ? sp - x : sp;
    rows[2 * r + 1] = demo ? (x > 0.0f ? 1.0f : 0.0f) : (x < 0.0f ? 1.0f : 0.0f);
}
// d(g[0] mean_agent(bce) + g[1] mean_demo(bce)) / d logits: (sigmoid(x) - t) / n
__global__ void ppo_disc_head_bwd_kernel(int n_agent, int n_demo, const float *agent_logit, const float *demo_logit, const float *g,
                                         float *d_agent, float *d_demo) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_agent + n_demo) return;
    const bool demo = r >= n_agent;
    const float x = demo ? demo_logit[r - n_agent] : agent_logit[r];
    const float sg = 1.0f / (1.0f + expf(0.0f - x));
    if (demo) d_demo[r - n_agent] = g[1] * (sg - 1.0f) / (float)n_demo;
    else d_agent[r] = g[0] * sg / (float)n_agent;
}

// Minibatch gather (AMPDataset._get_item, amp_datasets.py:16-33: every tensor of the dataset indexed with the same shuffled row ids):
// all tables in ONE launch -- dst_t[r][:] = src_t[idx[r]][:] for up to PPO_GATHER_MAX fp32 tables; blockIdx.y = table, a wave per row,
// 16-byte copies where a table's rows are 16-byte aligned.  torch issues one index_select per tensor (13 launches of 5-25 us).
#define PPO_GATHER_MAX 16
struct PpoGatherArgs {
    int n_tables, n_rows;
    const long long *idx;
    const float *src[PPO_GATHER_MAX];
    float *dst[PPO_GATHER_MAX];
    int cols[PPO_GATHER_MAX];
};
__global__ void __launch_bounds__(256)
ppo_gather_rows_kernel(PpoGatherArgs a) {
    const int t = blockIdx.y, lane = threadIdx.x & 63;
    const int cols = a.cols[t];
    const float *src = a.src[t];
    float *dst = a.dst[t];
    const bool vec = (cols & 3) == 0 && (((unsigned long long)src | (unsigned long long)dst) & 15ull) == 0;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < a.n_rows; r += gridDim.x * 4) {
        const float *s = src + a.idx[r] * (long)cols;
        float *d = dst + (long)r * cols;
        if (vec) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            for (int c = lane; c < cols / 4; c += 64) ((f4 *)d)[c] = ((const f4 *)s)[c];
        } else {
            for (int c = lane; c < cols; c += 64) d[c] = s[c];
        }
    }
}

}  // namespace emloco
