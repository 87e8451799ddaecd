// chain_kernels.hip -- the launches between two rigid-body steps, folded together (gfx950).
//
// In the reference's order (amp_continuous_value.py:45-75: reset the finished envs, read the observations, act, step) everything
// between two `gym.simulate` calls is ONE dependent chain on the caller's stream: reset / terminate flags -> which envs finished
// -> their reset (motion sample, kinematics, height fix + trajectory + LocoVal inputs, AMP history, observations) -> policy ->
// PD targets -> dispatch order -> next step.  As separate launches that chain was 11 kernels and ~170 us beside a 390 us
// rigid-body launch (profiles/r03_env_step_trace.txt): a few dozen one-wave workgroups per launch, each launch paying its ramp
// and drain, two event packets (~7 us each) to run the live envs' observations on a side stream and a 13 us hand-back.
// Here:
//   compact_order_kernel   2 workgroups: the finished-env compaction (+ flag snapshot) | the next step's cost-ordered dispatch
//   reset_obs_kernel       ONE launch, three roles by workgroup index:
//                            [0, S)             reset chain of list entry bi (bi = slot, slot + S, ...): random row, sample,
//                                               kinematics, finish, observations + newest AMP row -- one wave walks the
//                                               phases with workgroup barriers between them (the separate kernels' bodies)
//                            then 2 K           the pool of pre-drawn episodes for the NEXT call (below)
//                            then E             post-physics observations (+ AMP shift / row) of the envs that did NOT finish
//                                               (flag snapshot), i.e. the launch that ran on the observation stream
//                            then 14 S          AMP history row k of list entry bi (independent of the chain: the clip and its
//                                               start time are functions of the random row alone)
// The reset workgroups come first in dispatch order: the longest serial path of the launch starts first, the 4096 short
// observation workgroups fill the device beside it.  Same device functions as the separate kernels -> same bytes
// (tests/test_emu_kernels.py, tests/test_gpu_env.py).
#include <hip/hip_runtime.h>
#include "dev_math.h"
#include "emloco_types.h"
#include "fk_device.h"
#include "order_device.h"
#include "../../include/emloco_task.h"

namespace emloco {

__global__ void __launch_bounds__(1024)
compact_order_kernel(const int64_t *flags, int n, int32_t *ids, int64_t *snapshot,
                     const unsigned *ticks, int n_order, int *order, unsigned char *bucket_ws) {
    if (blockIdx.x == 0) compact_flags(flags, n, ids, snapshot);
    else order_sort(ticks, n_order, order, bucket_ws);
}

// Pool of pre-drawn episodes.  What a reset draws -- clip and start time, joint state, root state on the terrain, trajectory,
// LocoVal waypoints -- is a function of (call seed, position in the finished-env list) alone; only the kinematics on the env's
// own skeleton, the height fix and the observations need to know WHICH env finished.  So every launch also draws the leading
// entries of the NEXT call (whose seed the host knows: base + call counter) into a pool -- two workgroups per entry, sample and
// trajectory side by side -- and the reset chain of an entry whose pool slot carries this call's seed copies it instead of walking
// sample -> trajectory itself: the launch's longest serial path loses its two longest phases (82 -> 38 us for one entry).  How many
// entries are drawn follows the number of envs that finished THIS step (+ 25 % + 32, at most pool_k).  Entries beyond the pool and
// slots drawn for another seed (first call, caller-supplied random rows) take the direct path; the same device functions fill
// both, so the bytes are the same (tests/test_gpu_env.py: pooled launches against the separate kernels).
// Entry layout in floats (16-byte aligned blocks; EMLOCO_POOL_FLOATS per entry):
#define POOL_ROOT 0        /* [13] root state (z includes the ground height) */
#define POOL_GH 13         /* ground height under the pose */
#define POOL_TIME 14       /* clip start time */
#define POOL_INV 15        /* heading-inversion flag: first byte */
#define POOL_MID 16        /* int64 clip id (2 floats) */
#define POOL_DOF 20        /* [69][2] */
#define POOL_VERTS 160     /* [101][3] */
#define POOL_WAY 464       /* [15][3] */
static_assert(POOL_WAY + EMLOCO_TRAJ_SAMPLES * 3 <= EMLOCO_POOL_FLOATS, "pool entry size (include/emloco_task.h)");
static_assert(POOL_DOF + EMLOCO_NDOF * 2 <= POOL_VERTS && POOL_VERTS + EMLOCO_TRAJ_VERTS * 3 <= POOL_WAY, "pool layout");

struct ChainArgs {
    int n;                      // capacity of the id list (entries beyond the finished envs are -1)
    int n_slots;                // S: workgroups of the reset role (grid-stride over the list)
    int h_slots;                // workgroups per AMP history row (grid-stride over the list)
    int n_hist;                 // AMP history rows back-filled per reset env (14, or 0: EMLOCO_RESET_NO_AMP_HISTORY)
    int live_mode;              // post-physics mode of the envs that did not finish (0: that role is absent)
    int reset_mode;             // post-physics mode of the reset envs (OBS | AMP_ROW)
    int seeded;                 // 1: random rows from (seed, entry) into rnd_ws; 0: rows supplied in rnd_in
    unsigned seed_lo, seed_hi;
    const int32_t *ids;
    const int64_t *skip;        // flag snapshot [n_env]: the live role leaves envs with a non-zero entry alone
    const float *rnd_in;
    float *rnd_ws;
    // pool of pre-drawn episodes (all NULL / 0: every entry takes the direct path, nothing is drawn ahead)
    int pool_k;
    const float *pool_cur;                   // [pool_k][EMLOCO_POOL_FLOATS] drawn by the previous launch ...
    const unsigned long long *tag_cur;       // ... for the seed in its tag [pool_k]
    float *pool_next;                        // filled by this launch for the next call's seed
    unsigned long long *tag_next;
    unsigned nseed_lo, nseed_hi, next_key;   // next call: seed halves, real-path permutation key
    long long *prof;            // diagnostic (emloco_task_chain_profile): wall-clock stamps of reset slot 0's phases and of the last
                                // observation workgroup, else NULL
};

// Between the phases of the reset role the lane index and the env are redefined through an empty asm: what a phase derived from
// them (addresses, masks) dies with the phase instead of being held in registers for the whole chain (the kernel must fit the
// register budget of the 4096 observation workgroups that share the launch).
#ifdef EMLOCO_EMU
#define PHASE_FENCE(ln, ev) do { } while (0)
#else
#define PHASE_FENCE(ln, ev) do { int e_ = (ev); asm volatile("" : "+v"(ln), "+v"(e_)); (ev) = __builtin_amdgcn_readfirstlane(e_); } while (0)
#endif

__device__ __forceinline__ bool pool_hit(const ChainArgs &a, int bi) {
    return a.seeded && a.pool_cur && bi < a.pool_k && a.tag_cur[bi] == (((unsigned long long)a.seed_hi << 32) | a.seed_lo);
}

// three waves per SIMD (168 registers): with the roles' LDS overlaid (5.7 KB) four (128 registers) or five (96) would fit, and were
// measured slower -- 9.07 / 8.85 / 8.46 M env-steps/s: what the observation workgroups gain in residency the reset roles lose to spills
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8)))
reset_obs_kernel(EmlocoTaskBufs pt, EmlocoResetBufs rt, EmlocoSimDev s, ChainArgs a) {
    const int lane = threadIdx.x;
    const int b = (int)blockIdx.x, S = a.n_slots;
    // ONE block of LDS for whatever role this workgroup plays (a role's phases follow each other behind barriers and reuse it):
    // post-physics pass 4.8 KB | kinematics 2.1 KB | trajectory 3.6 KB | history row 0.7 KB | pool draw: random row 2 KB + trajectory
    constexpr int SM_FLOATS = EMLOCO_RESET_RND + TRAJ_SM_FLOATS > POST_SM_FLOATS ? EMLOCO_RESET_RND + TRAJ_SM_FLOATS : POST_SM_FLOATS;
    static_assert(SM_FLOATS >= FK_SM_FLOATS && SM_FLOATS >= HIST_SM_FLOATS && SM_FLOATS >= TRAJ_SM_FLOATS, "role workspace");
    __shared__ float sm[SM_FLOATS];
    if (b < S) {                                                    // ---- reset chain of the list entries b, b + S, ...
#ifndef EMLOCO_EMU
        __builtin_amdgcn_s_setprio(3);                              // the launch's longest serial path: ahead of the observation waves
#endif
        for (int bi = b; bi < a.n; bi += S) {
            const int env = a.ids[bi];
            if (env < 0) break;
#ifdef EMLOCO_EMU
            int ln = lane, ev = env;
#else
            int ln = lane, ev = __builtin_amdgcn_readfirstlane(env);
#endif
            const bool stamp = a.prof && b == 0 && bi == 0 && lane == 0;
            if (stamp) a.prof[0] = wall_clock64();
            const bool pooled = pool_hit(a, bi);                    // wave-uniform
            const float *u = a.rnd_in + (long)bi * EMLOCO_RESET_RND;
            const float *pe = a.pool_cur + (long)bi * EMLOCO_POOL_FLOATS;
            if (pooled) {                                           // the episode drawn ahead: joint state, root state, clip
                float *dst = s.dof_state + (long)ev * EMLOCO_NDOF * 2;
                for (int i = ln; i < EMLOCO_NDOF * 2; i += 64) dst[i] = pe[POOL_DOF + i];
                if (ln < 13) s.root_state[(long)ev * 13 + ln] = pe[POOL_ROOT + ln];
                if (ln == 0) {
                    rt.ground_h[ev] = pe[POOL_GH];
                    rt.motion_times[ev] = pe[POOL_TIME];
                    rt.motion_ids[ev] = *(const int64_t *)(pe + POOL_MID);
                }
                if (stamp) a.prof[1] = wall_clock64();
            } else {
                if (a.seeded) {
                    reset_fill_row(a.seed_lo, a.seed_hi, bi, a.rnd_ws, lane, 64);
                    u = a.rnd_ws + (long)bi * EMLOCO_RESET_RND;
                    __syncthreads();                                // workgroup barrier + fence: the row is read by other lanes
                }
                if (stamp) a.prof[1] = wall_clock64();
                PHASE_FENCE(ln, ev);
                reset_sample_env(rt, s, ev, u, ln);
            }
            __syncthreads();
            if (stamp) a.prof[2] = wall_clock64();
            PHASE_FENCE(ln, ev);
            fk_env(s, ev, ln, sm);
            __syncthreads();
            if (stamp) a.prof[3] = wall_clock64();
            PHASE_FENCE(ln, ev);
            if (pooled) {
                reset_fix_height(rt, s, ev, ln);
                float *vo = rt.traj_verts + (long)ev * EMLOCO_TRAJ_VERTS * 3;
                for (int i = ln; i < EMLOCO_TRAJ_VERTS * 3; i += 64) vo[i] = pe[POOL_VERTS + i];
                if (ln < EMLOCO_TRAJ_SAMPLES * 3) rt.waypoint_traj[(long)ev * EMLOCO_TRAJ_SAMPLES * 3 + ln] = pe[POOL_WAY + ln];
                if (ln == 0 && (rt.flags & EMLOCO_RESET_INIT_HEADING) && (rt.flags & EMLOCO_RESET_HEADING_INVERSION)) rt.inverted[ev] = *(const uint8_t *)(pe + POOL_INV);
                reset_capture_pose(rt, s, ev, ln, pe[POOL_ROOT + 7], pe[POOL_ROOT + 8]);
            } else {
                reset_finish_env(rt, s, bi, ev, u, ln, sm);
            }
            __syncthreads();
            if (stamp) a.prof[4] = wall_clock64();
            PHASE_FENCE(ln, ev);
            post_physics_env(pt, a.reset_mode, ev, ln, sm);
            __syncthreads();                                        // LDS is reused by the next list entry
            if (stamp) a.prof[5] = wall_clock64();
        }
        return;
    }
    // dispatch order behind the chain: pool draw (long, few), the observation workgroups (4096, ~20 us each, most of the launch's
    // work), the AMP history rows last -- ~2000 short workgroups that are not on anybody's path and fill the second round; ahead of
    // the observation workgroups they took the first round's slots and the last observation workgroup started 34 us into the launch
    const int Sh = a.h_slots;
    const int n_draw = a.pool_next ? a.pool_k * 2 : 0;
    const int n_live = a.live_mode ? pt.n_env : 0;
    const int rb = b - S;
    int hb = rb - n_draw - n_live;                                  // index among the history workgroups (< 0: another role)
    if (hb >= 0) {
      if (hb < Sh * a.n_hist) {                                       // ---- AMP history row k of the list entries slot, slot + Sh, ...
        const int slot = hb % Sh, k = 1 + hb / Sh;
        for (int bi = slot; bi < a.n; bi += Sh) {
            const int env = a.ids[bi];
            if (env < 0) break;
            const bool stamp = a.prof && hb == Sh * a.n_hist - Sh && bi == 0 && lane == 0;    // last history row of entry 0
            if (stamp) a.prof[6] = wall_clock64();
            const float um = a.seeded ? reset_rnd_value(a.seed_lo, a.seed_hi, bi, EMLOCO_RND_MOTION) : a.rnd_in[(long)bi * EMLOCO_RESET_RND + EMLOCO_RND_MOTION];
            const float ut = a.seeded ? reset_rnd_value(a.seed_lo, a.seed_hi, bi, EMLOCO_RND_TIME) : a.rnd_in[(long)bi * EMLOCO_RESET_RND + EMLOCO_RND_TIME];
            int mid; float mt;
            reset_pick_motion(rt, um, ut, &mid, &mt);
            reset_amp_history_row(rt, env, k, mid, mt, lane, sm);
            __syncthreads();
            if (stamp) a.prof[7] = wall_clock64();
        }
      }
      return;
    }
    hb = rb;
    if (hb < n_draw) {                                              // ---- draw entry e of the NEXT call into the pool
        const int e = hb % a.pool_k, kind = hb / a.pool_k;         // kind 0: sample, 1: trajectory
        const int now = a.ids[a.n];                                 // envs that finished this step (emloco_task_compact_done*: ids[n] = count)
        if (e >= now + (now >> 2) + 32) return;                     // the next step will not need more (if it does: direct path)
        float *pe = a.pool_next + (long)e * EMLOCO_POOL_FLOATS;
        float *sh_u = sm;                                           // the random row, then the trajectory workspace behind it
        for (int k = lane; k < EMLOCO_RESET_RND; k += 64) sh_u[k] = reset_rnd_value(a.nseed_lo, a.nseed_hi, e, k);
        __syncthreads();
        const bool stamp = a.prof && e == 0 && lane == 0;                               // the first entry's sample / trajectory workgroups
        if (stamp) a.prof[12 + 2 * kind] = wall_clock64();
        if (kind == 0) {
            reset_sample_to(rt, sh_u, lane, pe + POOL_DOF, pe + POOL_ROOT, pe + POOL_GH, (int64_t *)(pe + POOL_MID), pe + POOL_TIME);
            if (lane == 0) a.tag_next[e] = ((unsigned long long)a.nseed_hi << 32) | a.nseed_lo;
            __syncthreads();
            if (stamp) a.prof[13] = wall_clock64();
        } else if (kind == 1) {
            // the trajectory starts at the root's place and follows its velocity: lane 0 repeats the root part of the sample
            int mid; float time;
            reset_pick_motion(rt, sh_u[EMLOCO_RND_MOTION], sh_u[EMLOCO_RND_TIME], &mid, &time);
            const FrameBlend fb = frame_blend(rt, mid, time);
            float pos[3] = {0.0f, 0.0f, 0.0f}, rot[4] = {0.0f, 0.0f, 0.0f, 1.0f}, vel[3] = {0.0f, 0.0f, 0.0f}, ang[3] = {0.0f, 0.0f, 0.0f};
            if (lane == 0) reset_sample_root(rt, fb, sh_u, pos, rot, vel, ang);
            const float ipx = __shfl(pos[0], 0), ipy = __shfl(pos[1], 0);
            const float rvx = __shfl(vel[0], 0), rvy = __shfl(vel[1], 0), rvz = __shfl(vel[2], 0);
            EmlocoResetBufs nx = rt;
            nx.real_pick = nullptr;
            nx.real_pick_key = a.next_key;
            if (lane == 0) *(uint8_t *)(pe + POOL_INV) = 0;
            reset_traj_to(nx, sh_u, e, lane, ipx, ipy, rvx, rvy, rvz, pe + POOL_VERTS, (uint8_t *)(pe + POOL_INV), pe + POOL_WAY, sm + EMLOCO_RESET_RND);
            __syncthreads();
            if (stamp) a.prof[15] = wall_clock64();
        }
        return;
    }
    hb -= n_draw;
    if (!a.live_mode || hb >= n_live) return;
    const int env = hb;                                             // ---- observations of an env that did not finish
    if (env >= pt.n_env) return;
    if (a.skip[env] != 0) return;
    const bool stamp = a.prof && lane == 0 && (env == 0 || env == pt.n_env - 1);
    if (stamp) a.prof[env == 0 ? 8 : 10] = wall_clock64();
    post_physics_env(pt, a.live_mode, env, lane, sm);
    if (stamp) a.prof[env == 0 ? 9 : 11] = wall_clock64();
}

}  // namespace emloco
