/*
 * emloco_task.h -- C ABI of libemloco_hip.so, part 2: the fused post-physics task kernels.
 *
 * One launch replaces the ~100 small torch kernels of the reference's post_physics_step
 * (pacer/pacer/env/tasks/humanoid.py:1211-1232, humanoid_amp.py:139-157):
 *   progress += 1                        humanoid.py:1213
 *   self observations (368)              humanoid.py:1625-1687 (compute_humanoid_observations_smpl_max)
 *   mirrored observations                humanoid.py:1066-1108, humanoid_pedestrain_terrain.py:455-491
 *   trajectory samples + location obs    humanoid_traj.py:208-224, traj_generator.py:278-296,
 *                                        humanoid_pedestrain_terrain.py:1549-1577
 *   height-map observations (32x32)      humanoid_pedestrain_terrain.py:394-442,732-815,1212-1288
 *   reward                               humanoid_pedestrain_terrain.py:907-930,1581-1592
 *   reset / terminate masks (int64)      humanoid_pedestrain_terrain.py:883-905,1468-1530
 *   AMP history shift + new AMP row      humanoid_amp.py:585-657,917-971
 * Buffers are caller-owned device memory (the task object allocates them, base_task.py:96-111).
 */
#ifndef EMLOCO_TASK_H
#define EMLOCO_TASK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMLOCO_SELF_OBS 368
#define EMLOCO_TRAJ_SAMPLES 15
#define EMLOCO_TRAJ_VERTS 101
#define EMLOCO_HEIGHT_POINTS 1024
#define EMLOCO_TASK_OBS (2 * EMLOCO_TRAJ_SAMPLES + EMLOCO_HEIGHT_POINTS)
#define EMLOCO_OBS (EMLOCO_SELF_OBS + EMLOCO_TASK_OBS)
#define EMLOCO_AMP_ROW 206
#define EMLOCO_AMP_STEPS 15

/* what one launch computes (bit-or) */
enum {
    EMLOCO_POST_ADVANCE = 1,    /* progress_buf += 1 */
    EMLOCO_POST_OBS = 2,        /* obs_buf and flip_obs_buf */
    EMLOCO_POST_REWARD = 4,     /* rew_buf, reward_raw */
    EMLOCO_POST_RESET = 8,      /* reset_buf, terminate_buf */
    EMLOCO_POST_AMP_SHIFT = 16, /* history shift of amp_obs_buf */
    EMLOCO_POST_AMP_ROW = 32,   /* newest AMP row */
    EMLOCO_POST_STEP = 63,      /* everything post_physics_step does */
    EMLOCO_POST_SKIP_DONE = 64, /* leave the envs whose reset_buf is set alone (their observation rows are rebuilt by the reset
                                 * path): lets the observation launch of a step run beside that step's resets on another stream */
    EMLOCO_POST_AMP_DONE_ONLY = 128 /* AMP_SHIFT / AMP_ROW act on the envs whose reset flag is set after this launch only: the
                                 * terminal AMP observations of the finished envs (amp_continuous_value.py:90-96 scores them) are
                                 * written by the flags launch, BEFORE the reset path overwrites those envs' state; the side
                                 * launch (SKIP_DONE) writes the live envs' rows */
};

typedef struct {
    int32_t n_env;
    int32_t hf_rows, hf_cols;       /* height map shape (first index = x) */
    int32_t head_body;              /* sensor frame body (terrain_obs_root "head" -> 13) */
    int32_t n_dof_subset;           /* 57 */
    float dt;                       /* control dt = controlFrequencyInv * sim.dt */
    float traj_dur;                 /* num_verts * vertex dt (traj_generator.py:270-273) */
    float sample_dt;                /* trajSampleTimestep */
    float hscale, vscale;           /* 0.1, 0.005 */
    float power_coef;               /* power_coefficient */
    float fail_dist;                /* 4.0 (humanoid_traj.py:30) */
    float max_episode_length;       /* episodeLength */
    /* simulator tensors (device) */
    const float *rb_state;          /* [E][24][13] */
    const float *dof_state;         /* [E][69][2] */
    const float *dof_force;         /* [E][69] */
    const float *contact_force;     /* [E][24][3] */
    /* task data (device) */
    const float *betas;             /* [E][17] humanoid_betas */
    const float *traj_verts;        /* [E][101][3] */
    const int16_t *heightfield;     /* [hf_rows][hf_cols] */
    const int32_t *left_to_right;   /* [24] */
    const uint8_t *contact_body_mask; /* [24] 1 = foot body excluded from the fall test */
    const int32_t *key_bodies;      /* [4] */
    const int32_t *dof_subset;      /* [n_dof_subset] */
    /* task buffers (device, in/out) */
    int64_t *progress_buf;          /* [E] */
    int64_t *reset_buf;             /* [E] */
    int64_t *terminate_buf;         /* [E] */
    float *obs_buf;                 /* [E][1422] */
    float *flip_obs_buf;            /* [E][1422] */
    float *rew_buf;                 /* [E] */
    float *reward_raw;              /* [E][2] */
    float *amp_obs_buf;             /* [E][15][206], index 0 = newest */
    /* 0: amp_obs_buf is the reference's layout (humanoid_amp.py:585-594): EMLOCO_POST_AMP_SHIFT moves rows 0..13 to 1..14 every step
     * -- 23 KB of the ~38 KB an env-step moves.  1 + h: amp_obs_buf is a RING: the row the reference calls k lives in physical row
     * (h + k) % 15, EMLOCO_POST_AMP_SHIFT is a no-op and EMLOCO_POST_AMP_ROW writes physical row h.  The caller moves h back by one
     * (h <- (h + 14) % 15) once per step, ahead of that step's launches, and hands the same value to EmlocoResetBufs.amp_ring. */
    int32_t amp_ring;
} EmlocoTaskBufs;
#define EMLOCO_AMP_PHYS_ROW(amp_ring, k) ((amp_ring) ? ((amp_ring) - 1 + (k)) % EMLOCO_AMP_STEPS : (k))

/* post_physics_step for all envs (dev_env_ids == NULL) or for the listed envs
 * (_compute_observations(env_ids) on reset, humanoid.py:459-465). */
int emloco_task_post_physics(const EmlocoTaskBufs *bufs, int mode, const int32_t *dev_env_ids, int n, void *stream);

/* emloco_task_post_physics for all envs with the LocoVal return bookkeeping of amp_continuous_value.py:63-64,93-129 behind it in the
 * SAME launch: what emloco_locoval_returns (include/emloco_predictor.h) does with rewards = rew_buf, dones = reset_buf and no AMP
 * reward, for each env right after its reward and reset flag exist.  `mode` must hold EMLOCO_POST_REWARD | EMLOCO_POST_RESET.
 * `step` is an EmlocoLocoValStep (emloco_predictor.h), dev_inverted the heading-inversion flags [n_env] (bytes) or NULL.  A step
 * with staging arrays (staged_reward / staged_done) is only STAGED here; emloco_locoval_returns_finish completes it once the AMP
 * discriminator's reward of the step exists. */
int emloco_task_post_physics_returns(const EmlocoTaskBufs *bufs, int mode, const void *step /* const EmlocoLocoValStep * */,
                                     const uint8_t *dev_inverted, void *stream);

/* AMP rows from explicit states (history back-fill from the motion library, humanoid_amp.py:486-535):
 * n rows; inputs [n][3|4|3|3|69|69|4*3|17]; out [n][206]. */
int emloco_task_amp_rows(int n, const float *root_pos, const float *root_rot, const float *root_vel,
                         const float *root_ang_vel, const float *dof_pos, const float *dof_vel,
                         const float *key_pos, const float *betas, const int32_t *dof_subset,
                         int n_dof_subset, float *out, void *stream);

/* pre_physics_step: pd_tar = offset + scale * action, zeroed where mask != 0 (humanoid.py:1184-1202,1281-1283) */
int emloco_task_pd_targets(int n_env, const float *actions, const float *offset, const float *scale,
                           const uint8_t *zero_mask, float *pd_targets, void *stream);

/* Same, and a copy of the actions as they were read into `actions_copy` ([n_env][69], optional, ignored when it aliases
 * `actions`): the reference keeps `self.actions = actions.clone()` (humanoid.py:1185); one launch instead of a copy and a launch. */
int emloco_task_pd_targets_copy(int n_env, const float *actions, const float *offset, const float *scale,
                                const uint8_t *zero_mask, float *pd_targets, float *actions_copy, void *stream);

/* wall-clock of the last emloco_task_post_physics launch measured with HIP events [ms]; <0 if none */
float emloco_task_last_ms(void);
int emloco_task_enable_timing(int on);


/* ------------------------------------------------------------------------------------------------------------
 * Fused reset of finished envs (reference-state init + task reset; per-episode, but with 4096 envs some env
 * resets almost every step).  Replaces the ~1 100 small torch launches of the host-side path:
 *   _sample_ref_state / get_motion_state_smpl      humanoid_pedestrain_terrain.py:526-573, motion_lib_smpl.py:485-606
 *   _reset_ref_state_init, _set_env_state          humanoid_pedestrain_terrain.py:575-631, humanoid_amp.py:537-563
 *   _reset_env_tensors                             humanoid.py:467-481
 *   TrajGenerator.reset                            env/util/traj_generator.py:60-237
 *   _reset_task (LocoVal input capture)            humanoid_pedestrain_terrain.py:493-523
 *   _init_amp_obs_ref (history back-fill)          humanoid_amp.py:486-535
 * Random numbers are supplied by the caller (`rnd`, uniform [0,1), EMLOCO_RESET_RND floats per env) so a test can
 * drive the kernels and the host mirror with the same draws. */
#define EMLOCO_RESET_RND 512
enum {
    EMLOCO_RESET_RANDOM_HEADING = 1, EMLOCO_RESET_INIT_HEADING = 2, EMLOCO_RESET_HEADING_INVERSION = 4,
    EMLOCO_RESET_ADJUST_ROOT_VEL = 8, EMLOCO_RESET_REAL_PATH = 16, EMLOCO_RESET_FIXED_LOCATION = 32,
    EMLOCO_RESET_NO_AMP_HISTORY = 64      /* emloco_task_reset leaves the AMP history back-fill to emloco_task_reset_amp_history */
};
/* layout of one env's random row */
enum {
    EMLOCO_RND_MOTION = 0, EMLOCO_RND_TIME = 1, EMLOCO_RND_YAW = 2, EMLOCO_RND_SPEED = 3, EMLOCO_RND_LOC = 4,
    EMLOCO_RND_REAL = 5, EMLOCO_RND_REAL_PICK = 6, EMLOCO_RND_INVERSION = 7, EMLOCO_RND_HEADING = 8, EMLOCO_RND_SPEED0 = 9,
    EMLOCO_RND_DTHETA = 16, EMLOCO_RND_SHARP = 116, EMLOCO_RND_BERN = 216, EMLOCO_RND_DSPEED = 316
};

typedef struct {
    int32_t flags;
    int32_t n_motions, n_real, n_valid, n_dof_subset;
    int32_t hf_rows, hf_cols;
    float fixed_x, fixed_y;
    float dt;                       /* control dt */
    float height_tolerance;         /* lowest collision point above ground after reset (0.02) */
    float vert_dt, dtheta_max, speed_min, speed_max, accel_max, sharp_prob, hybrid_prob;   /* TrajGenerator */
    float traj_dur, sample_dt, hscale, vscale;
    /* motion cache (motion_lib_smpl.py:334-341) */
    const float *gts, *grs, *lrs, *gvs, *gavs, *dvs;
    const float *motion_len, *motion_dt;
    const int64_t *motion_nframes, *motion_start;
    const float *real_traj;         /* [n_real][101][3] or NULL */
    const int16_t *heightfield;
    const float *valid_x, *valid_y; /* walkable sample locations [n_valid] */
    const float *betas;             /* [E][17] */
    const int32_t *key_bodies, *dof_subset;
    /* task buffers written */
    float *traj_verts;              /* [E][101][3] */
    uint8_t *inverted;              /* [E] (bool) */
    int64_t *progress_buf, *reset_buf, *terminate_buf;
    float *waypoint_traj;           /* [E][15][3] */
    float *init_pose;               /* [E][24][3] */
    float *init_vel;                /* [E][2] */
    float *amp_obs_buf;             /* [E][15][206]: rows 1..14 back-filled here, row 0 by emloco_task_post_physics */
    int64_t *motion_ids;            /* [E] */
    float *motion_times;            /* [E] */
    float *ground_h;                /* [E] scratch: ground height under the reset pose */
    /* real paths (traj_generator.py:121-160): list entry i takes row P_key(i mod n_real) of real_traj, P_key a keyed
     * bijection of [0, n_real) -- distinct rows within one call, like the reference's random.sample(range(n), k);
     * real_pick (device, [n] rows, optional) replaces the permutation by explicit rows */
    const int32_t *real_pick;
    uint32_t real_pick_key;
    int32_t amp_ring;               /* as EmlocoTaskBufs.amp_ring: where the back-filled history rows 1..14 go */
} EmlocoResetBufs;

struct EmlocoSim;
/* resets the listed envs of `sim` (device env ids, n entries, rnd [n][EMLOCO_RESET_RND]) */
int emloco_task_reset(struct EmlocoSim *sim, const EmlocoResetBufs *bufs, const int32_t *dev_env_ids, int n,
                      const float *dev_rnd, void *stream);

/* Same, with the random rows produced on the device: row i of `dev_rnd_ws` ([n][EMLOCO_RESET_RND] floats, caller-owned
 * workspace) is filled with uniforms in [0, 1) from a stateless hash of (seed, i, k) for the list entries that are present,
 * then emloco_task_reset runs on it.  Pass a fresh seed per call (e.g. base seed + call counter; the reference draws from
 * torch's global generator: humanoid_amp.py:284-379, traj_generator.py:60-237).  With a compacted done-list this avoids
 * generating n_env rows per step when a few dozen envs finish. */
int emloco_task_reset_seeded(struct EmlocoSim *sim, const EmlocoResetBufs *bufs, const int32_t *dev_env_ids, int n,
                             uint64_t seed, float *dev_rnd_ws, void *stream);

/* The AMP history back-fill of a reset (_init_amp_obs_ref, humanoid_amp.py:486-535) alone: rows 1..14 of the listed envs'
 * AMP observations from the motion ids / start times the reset sampled.  It reads nothing the simulator writes, so a caller
 * that passed EMLOCO_RESET_NO_AMP_HISTORY may run it on another stream beside the rest of the reset chain. */
int emloco_task_reset_amp_history(const EmlocoResetBufs *bufs, const int32_t *dev_env_ids, int n, void *stream);

/* TrajGenerator.reset(env_ids, init_pos, root_vel) alone (traj_generator.py:60-237): writes traj_verts / inverted of the
 * listed envs from the random rows; dev_init_pos, dev_root_vel [n][3] (one row per list entry).  Only the trajectory
 * fields of `bufs` are read (flags, vert_dt ... hybrid_prob, n_real, real_traj, real_pick*, traj_verts, inverted). */
int emloco_task_traj_reset(const EmlocoResetBufs *bufs, const int32_t *dev_env_ids, int n, const float *dev_rnd,
                           const float *dev_init_pos, const float *dev_root_vel, void *stream);

/* HumanoidPedestrianTerrain.get_heights / get_center_heights for arbitrary poses (humanoid_pedestrain_terrain.py:761-815,
 * 732-759; Terrain.world_points_to_map / sample_height_points :1212-1218,1282-1288): dev_pose7 [n][7] = position + xyzw
 * rotation; grid = 1: the 32x32 sensor grid rotated by the pose's heading -> [n][1024]; grid = 0: the 3x3 yaw-only centre
 * probes -> [n][9].  dev_heights (metres) and the int64 map indices dev_px / dev_py are each optional.  The fused
 * post-physics kernel evaluates the same device functions. */
int emloco_task_get_heights(const int16_t *dev_heightfield, int rows, int cols, float hscale, float vscale, const float *dev_pose7,
                            int n, int grid, float *dev_heights, int64_t *dev_px, int64_t *dev_py, void *stream);

/* Device-side `reset_buf.nonzero()`: dev_ids[0..count) = ascending indices of the non-zero flags, the rest of the n
 * entries = -1, dev_ids[n] = count.  Every *_indexed / env-id-list entry point of this library skips negative ids, so
 *   emloco_task_compact_done(reset_buf, E, ids, s); emloco_task_reset(sim, bufs, ids, E, rnd, s);
 *   emloco_task_post_physics(bufs, EMLOCO_POST_OBS | EMLOCO_POST_AMP_ROW, ids, E, s);
 * resets exactly the finished envs without the host ever reading the count (the reference's loop,
 * amp_continuous_value.py:46,74, synchronises on `dones.nonzero()` every step). */
int emloco_task_compact_done(const int64_t *dev_flags, int n, int32_t *dev_ids, void *stream);
/* Same, and a copy of the flags as they were (`dev_snapshot`, n entries): the reset kernels clear the flags of the envs they
 * reset, a launch that runs beside them (emloco_sim_step_subset) follows the snapshot. */
int emloco_task_compact_done_snapshot(const int64_t *dev_flags, int n, int32_t *dev_ids, int64_t *dev_snapshot, void *stream);


/* ------------------------------------------------------------------------------------------------------------
 * The chain between two rigid-body steps in three launches ("around the kernel").  In the reference's
 * loop (amp_continuous_value.py:45-75: env_reset(done_indices) -> obs -> policy -> env.step) everything between two
 * gym.simulate calls is one dependent chain; as separate launches it was 11 kernels and two stream hand-overs.
 *
 * emloco_task_compact_done_order: emloco_task_compact_done_snapshot and -- when `sim` has the cost-ordered dispatch on
 * (emloco_sim_set_cost_order) -- the sort of the NEXT rigid-body launch's dispatch order, as two workgroups of one launch;
 * the next emloco_sim_step / emloco_sim_step_subset then uses that order instead of sorting again.  `sim` may be NULL. */
int emloco_task_compact_done_order(struct EmlocoSim *sim, const int64_t *dev_flags, int n, int32_t *dev_ids,
                                   int64_t *dev_snapshot, void *stream);

/* emloco_task_reset_obs: ONE launch for
 *   (a) the reset of the listed envs: emloco_task_reset[_seeded] (random rows, sample, kinematics, height fix, trajectory,
 *       LocoVal inputs, AMP history unless EMLOCO_RESET_NO_AMP_HISTORY) followed by
 *       emloco_task_post_physics(EMLOCO_POST_OBS | EMLOCO_POST_AMP_ROW) of the same envs (humanoid.py:459-465), and
 *   (b) with live_mode != 0: emloco_task_post_physics(live_mode) of every env whose `dev_skip` entry (the flag snapshot
 *       of emloco_task_compact_done_order) is zero -- the observation / AMP part of post_physics_step (humanoid.py:1214,
 *       humanoid_amp.py:150-155) for the envs that did not finish, which the caller left out of its flags launch.
 * live_mode may hold EMLOCO_POST_OBS, _AMP_SHIFT, _AMP_ROW.  Random rows: `dev_rnd` [n][EMLOCO_RESET_RND] as for
 * emloco_task_reset, or NULL: made on the device from `seed` into `dev_rnd_ws` as for emloco_task_reset_seeded.
 * Results are those of the separate calls, byte for byte (the same device functions run). */
int emloco_task_reset_obs(struct EmlocoSim *sim, const EmlocoResetBufs *reset_bufs, const EmlocoTaskBufs *task_bufs, int live_mode,
                          const int64_t *dev_skip, const int32_t *dev_env_ids, int n, uint64_t seed, float *dev_rnd_ws,
                          const float *dev_rnd, void *stream);

/* The same launch with a pool of pre-drawn episodes.  What a reset draws (clip, joint and root state on the terrain, trajectory,
 * LocoVal waypoints) depends on (seed, position in the list) only -- not on which env finished -- so the launch also draws the
 * leading entries (as many as finished this step + 25 % + 32, at most k) of the NEXT call (seed `next_seed`) into `next` in
 * workgroups of its own, and the reset chain of an entry whose slot in `cur` carries THIS call's seed copies it instead of
 * computing it: the longest serial path of the launch loses its two longest phases.  Entries beyond the pool and slots tagged
 * with another seed take the direct path; same bytes either way.
 * The caller alternates two buffers: this call's `next` is the following call's `cur`.  Ignored (direct path, nothing drawn)
 * when dev_rnd is given.  With a pool, dev_env_ids must be a list of emloco_task_compact_done* (n + 1 entries: the draw count
 * follows dev_env_ids[n], the number of finished envs). */
#define EMLOCO_POOL_FLOATS 512      /* per entry: root 13 | misc | dof 138 | trajectory 303 | waypoints 45 */
typedef struct {
    int32_t k;                      /* entries per pool buffer (0: no pool) */
    const float *cur;               /* [k][EMLOCO_POOL_FLOATS] or NULL */
    const uint64_t *cur_tag;        /* [k] seed each entry of `cur` was drawn for */
    float *next;                    /* filled by this launch, or NULL */
    uint64_t *next_tag;
    uint64_t next_seed;             /* the seed the caller will pass to its next call */
} EmlocoResetPool;
int emloco_task_reset_obs_pooled(struct EmlocoSim *sim, const EmlocoResetBufs *reset_bufs, const EmlocoTaskBufs *task_bufs, int live_mode,
                                 const int64_t *dev_skip, const int32_t *dev_env_ids, int n, uint64_t seed, float *dev_rnd_ws,
                                 const float *dev_rnd, const EmlocoResetPool *pool, void *stream);

/* Diagnostic (tools/exp/chain_prof.py; the product never calls it): the first call allocates 16 wall-clock stamps (100 MHz) that
 * every later emloco_task_reset_obs* launch overwrites -- [0..5] reset entry 0: start, random row, sample, kinematics, finish,
 * observations; [6, 7] last AMP history row of entry 0; [8, 9] / [10, 11] the observation workgroups of env 0 / the last env -- and
 * copies them to host16 (may be NULL).  Synchronises the device. */
int emloco_task_chain_profile(long long *host16);

#ifdef __cplusplus
}
#endif
#endif