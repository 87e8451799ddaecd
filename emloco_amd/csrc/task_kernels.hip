// task_kernels.hip -- fused post-physics kernels of the humanoid rollout for gfx950 (MI355X).
//
// One wave (64 lanes) per env computes, in one launch, what the reference's post_physics_step does
// with ~100 small torch kernels (see include/emloco_task.h for the reference lines of each part).
// The arithmetic of every observation restates the reference's formulas through the ref_* helpers in
// dev_math.h; integer masks are computed with the same operation order as the CPU oracle so they are
// bit-exact.  HBM-bound: per env.step it reads ~15 KB (body/dof state, AMP history, trajectory and
// height-map gathers) and writes ~24 KB (obs, mirrored obs, AMP buffer); all obs stores are coalesced
// (lane i writes float i of a row segment).
#include <hip/hip_runtime.h>
#include "dev_math.h"
#include "../../include/emloco_task.h"

#include "task_device.h"

namespace emloco {

__global__ void __launch_bounds__(64)
post_physics_kernel(EmlocoTaskBufs t, int mode, const int32_t *env_ids, int n_ids) {
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= n_ids) return;
    const int env = env_ids ? env_ids[blockIdx.x] : (int)blockIdx.x;
    if (env < 0) return;                     // padding entry of a device-compacted id list
    if ((mode & EMLOCO_POST_SKIP_DONE) && t.reset_buf[env] != 0) return;      // block-uniform
    __shared__ float sm[POST_SM_FLOATS];
    post_physics_env(t, mode, env, lane, sm);
}

// the same launch with the LocoVal return bookkeeping of every env behind its flags (emloco_task_post_physics_returns)
__global__ void __launch_bounds__(64)
post_physics_returns_kernel(EmlocoTaskBufs t, int mode, EmlocoLocoValStep lv, const uint8_t *lv_inverted) {
    const int lane = threadIdx.x, env = (int)blockIdx.x;
    if (env >= t.n_env) return;
    __shared__ float sm[POST_SM_FLOATS];
    post_physics_env(t, mode, env, lane, sm, &lv, lv_inverted);
}

__global__ void __launch_bounds__(64)
amp_rows_kernel(int n, const float *root_pos, const float *root_rot, const float *root_vel, const float *root_ang,
                const float *dof_pos, const float *dof_vel, const float *key_pos, const float *betas,
                const int32_t *subset, int n_sub, float *out) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    amp_row(lane, root_pos + (long)i * 3, root_rot + (long)i * 4, root_vel + (long)i * 3, root_ang + (long)i * 3,
            dof_pos + (long)i * TNDOF, dof_vel + (long)i * TNDOF, 1, key_pos + (long)i * 12, betas + (long)i * 17, subset, n_sub,
            out + (long)i * EMLOCO_AMP_ROW);
}

// pre_physics_step: pd_tar = offset + scale * a, zero where masked (humanoid.py:1184-1202,1281-1283)
// actions_copy (optional): the task's own copy of the actions (the reference clones them, humanoid.py:1185) written by the same launch
__global__ void pd_targets_kernel(int total, const float *actions, const float *offset, const float *scale,
                                  const uint8_t *zero_mask, float *out, float *actions_copy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int dd = i % TNDOF;
    const float a = actions[i];
    if (actions_copy) actions_copy[i] = a;
    out[i] = zero_mask[dd] ? 0.0f : offset[dd] + scale[dd] * a;
}

// Device-side replacement of `reset_buf.nonzero()` (amp_continuous_value.py:46,74 does it on the host, which stalls the
// launch queue every step): ids[0..count) = ascending indices of the non-zero flags, ids[count..n) = -1, ids[n] = count.
// One 1024-thread workgroup; ballot + popcount inside a wave, LDS prefix over the 16 waves, running base over chunks.
__device__ __forceinline__ void compact_flags(const int64_t *flags, int n, int32_t *ids, int64_t *snapshot) {
    // ONE pass (round 5; was a loop of 1024-flag chunks with three barriers each): thread t owns the run of `per` consecutive flags
    // [t per, (t + 1) per), counts its non-zero ones, the counts are scanned inside each wave (six shuffle steps) and across the 16
    // waves (LDS), and the thread writes its ids behind its offset -- ascending, as the chunked loop left them.
    __shared__ int sh_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + 1023) / 1024;
    const int i0 = tid * per;
    int cnt = 0;
    for (int k = 0; k < per; ++k) {
        const int i = i0 + k;
        if (i < n) {
            const int64_t fl = flags[i];
            if (snapshot) snapshot[i] = fl;
            cnt += fl != 0 ? 1 : 0;
        }
    }
    int inc = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl(inc, lane - d < 0 ? 0 : lane - d);
        if (lane >= d) inc += up;
    }
    if (lane == 63) sh_wave[wave] = inc;
    __syncthreads();
    int off = inc - cnt, count = 0;
    for (int w = 0; w < 16; ++w) { const int c = sh_wave[w]; off += w < wave ? c : 0; count += c; }
    for (int k = 0; k < per; ++k) {
        const int i = i0 + k;
        if (i < n && flags[i] != 0) ids[off++] = i;
    }
    for (int i = count + tid; i < n; i += 1024) ids[i] = -1;
    if (tid == 0) ids[n] = count;
}
__global__ void __launch_bounds__(1024)
compact_flags_kernel(const int64_t *flags, int n, int32_t *ids, int64_t *snapshot) { compact_flags(flags, n, ids, snapshot); }

// get_heights / get_center_heights (humanoid_pedestrain_terrain.py:761-815, 732-759) for arbitrary poses, with the integer map
// indices: pose7 [n][7] (pos3, quat4 xyzw); grid = 1: 32x32 grid rotated by the pose's heading -> out [n][1024];
// grid = 0: 3x3 yaw-only centre probes -> out [n][9].  One wave per pose.
__global__ void __launch_bounds__(64)
get_heights_kernel(const int16_t *hf, int rows, int cols, float hscale, float vscale, const float *pose7, int n, int grid,
                   float *out_h, int64_t *out_px, int64_t *out_py) {
    const int e = blockIdx.x, lane = threadIdx.x;
    if (e >= n) return;
    const float *p = pose7 + (long)e * 7;
    const int np = grid ? EMLOCO_HEIGHT_POINTS : 9;
    float hq[4] = {0.0f, 0.0f, 0.0f, 1.0f};
    if (grid) ref_quat_about_z(ref_calc_heading(p + 3), hq);
    for (int idx = lane; idx < np; idx += 64) {
        float wx, wy;
        int px, py;
        if (grid) grid_probe(hq, p, idx, &wx, &wy);
        else center_probe(p, p + 3, idx, &wx, &wy);
        map_index(rows, cols, wx, wy, hscale, &px, &py);
        const long o = (long)e * np + idx;
        if (out_h) out_h[o] = sample_height_at(hf, cols, px, py, vscale);
        if (out_px) { out_px[o] = px; out_py[o] = py; }
    }
}

}  // namespace emloco
