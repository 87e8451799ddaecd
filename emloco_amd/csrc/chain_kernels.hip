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
//                            [S, S + 14 S)      AMP history row k of list entry bi (independent of the chain: the clip and its
//                                               start time are functions of the random row alone)
//                            [S + 14 S, .. + E) post-physics observations (+ AMP shift / row) of the envs that did NOT finish
//                                               (flag snapshot), i.e. the launch that ran on the observation stream
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

struct ChainArgs {
    int n;                      // capacity of the id list (entries beyond the finished envs are -1)
    int n_slots;                // S: workgroups of the reset role (grid-stride over the list)
    int n_hist;                 // AMP history rows back-filled per reset env (14, or 0: EMLOCO_RESET_NO_AMP_HISTORY)
    int live_mode;              // post-physics mode of the envs that did not finish (0: that role is absent)
    int reset_mode;             // post-physics mode of the reset envs (OBS | AMP_ROW)
    int seeded;                 // 1: random rows from (seed, entry) into rnd_ws; 0: rows supplied in rnd_in
    unsigned seed_lo, seed_hi;
    const int32_t *ids;
    const int64_t *skip;        // flag snapshot [n_env]: the live role leaves envs with a non-zero entry alone
    const float *rnd_in;
    float *rnd_ws;
    long long *prof;            // diagnostic (emloco_task_chain_profile): wall-clock stamps of reset slot 0's phases and of the last
                                // observation workgroup, else NULL
};

// Between the phases of the reset role the lane index, the env and the random-row pointer are redefined through an empty asm:
// what a phase derived from them (addresses, masks) dies with the phase instead of being held in registers for the whole chain
// (the kernel must fit the register budget of the 4096 observation workgroups that share the launch).
#ifdef EMLOCO_EMU
#define PHASE_FENCE(ln, ev, u) do { } while (0)
#else
#define PHASE_FENCE(ln, ev, u) asm volatile("" : "+v"(ln), "+s"(ev))
#endif

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))
reset_obs_kernel(EmlocoTaskBufs pt, EmlocoResetBufs rt, EmlocoSimDev s, ChainArgs a) {
    const int lane = threadIdx.x;
    const int b = (int)blockIdx.x, S = a.n_slots;
    if (b < S) {                                                    // ---- reset chain of the list entries b, b + S, ...
#ifndef EMLOCO_EMU
        __builtin_amdgcn_s_setprio(3);                              // the launch's longest serial path: ahead of the observation waves
#endif
        for (int bi = b; bi < a.n; bi += S) {
            const int env = a.ids[bi];
            if (env < 0) break;
            const float *u = a.rnd_in + (long)bi * EMLOCO_RESET_RND;
            if (a.seeded) {
                reset_fill_row(a.seed_lo, a.seed_hi, bi, a.rnd_ws, lane, 64);
                u = a.rnd_ws + (long)bi * EMLOCO_RESET_RND;
                __syncthreads();                                    // workgroup barrier + fence: the row is read by other lanes
            }
            #ifdef EMLOCO_EMU
            int ln = lane, ev = env;
#else
            int ln = lane, ev = __builtin_amdgcn_readfirstlane(env);
#endif
            const bool stamp = a.prof && b == 0 && bi == 0 && lane == 0;
            if (stamp) a.prof[0] = wall_clock64();
            PHASE_FENCE(ln, ev, u);
            if (stamp) a.prof[1] = wall_clock64();
            reset_sample_env(rt, s, ev, u, ln);
            __syncthreads();
            if (stamp) a.prof[2] = wall_clock64();
            PHASE_FENCE(ln, ev, u);
            fk_env(s, ev, ln);
            __syncthreads();
            if (stamp) a.prof[3] = wall_clock64();
            PHASE_FENCE(ln, ev, u);
            reset_finish_env(rt, s, bi, ev, u, ln);
            __syncthreads();
            if (stamp) a.prof[4] = wall_clock64();
            PHASE_FENCE(ln, ev, u);
            post_physics_env(pt, a.reset_mode, ev, ln);
            __syncthreads();                                        // LDS is reused by the next list entry
            if (stamp) a.prof[5] = wall_clock64();
        }
        return;
    }
    const int hb = b - S;
    if (hb < S * a.n_hist) {                                        // ---- AMP history row k of the list entries slot, slot + S, ...
        const int slot = hb % S, k = 1 + hb / S;
        for (int bi = slot; bi < a.n; bi += S) {
            const int env = a.ids[bi];
            if (env < 0) break;
            const float um = a.seeded ? reset_rnd_value(a.seed_lo, a.seed_hi, bi, EMLOCO_RND_MOTION) : a.rnd_in[(long)bi * EMLOCO_RESET_RND + EMLOCO_RND_MOTION];
            const float ut = a.seeded ? reset_rnd_value(a.seed_lo, a.seed_hi, bi, EMLOCO_RND_TIME) : a.rnd_in[(long)bi * EMLOCO_RESET_RND + EMLOCO_RND_TIME];
            int mid; float mt;
            reset_pick_motion(rt, um, ut, &mid, &mt);
            const bool stamp = a.prof && hb == S * a.n_hist - S && bi == 0 && lane == 0;      // last history row of entry 0
            if (stamp) a.prof[6] = wall_clock64();
            reset_amp_history_row(rt, env, k, mid, mt, lane);
            __syncthreads();
            if (stamp) a.prof[7] = wall_clock64();
        }
        return;
    }
    if (!a.live_mode) return;
    const int env = hb - S * a.n_hist;                              // ---- observations of an env that did not finish
    if (env >= pt.n_env) return;
    if (a.skip[env] != 0) return;
    const bool stamp = a.prof && lane == 0 && (env == 0 || env == pt.n_env - 1);
    if (stamp) a.prof[env == 0 ? 8 : 10] = wall_clock64();
    post_physics_env(pt, a.live_mode, env, lane);
    if (stamp) a.prof[env == 0 ? 9 : 11] = wall_clock64();
}

}  // namespace emloco
