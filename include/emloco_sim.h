/*
 * emloco_sim.h -- C ABI of libemloco_hip.so, part 1: the vectorised humanoid rollout (boundary 1).
 *
 * Plain C, plain pointers and sizes, no torch types.  Every entry point returns 0 on success and a
 * negative EMLOCO_E_* code on failure (the reference's gym API returns bool/None and never throws,
 * isaacgym/docs/api/python/gym_py.html; the Python shim maps codes back to that convention).
 * Device pointers belong to the HIP device the sim was created on.  `stream` is a hipStream_t
 * passed as void* (0 = the default stream); the library never synchronises unless asked to.
 *
 * Each entry point names the reference interface it stands behind (file:line under
 * /root/reference/); the reference-side binding is shown in INTEGRATION.md.
 */
#ifndef EMLOCO_SIM_H
#define EMLOCO_SIM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMLOCO_NB 24        /* rigid bodies per humanoid (humanoid.py:264) */
#define EMLOCO_NDOF 69      /* 23 joints x 3 (humanoid.py:516-521) */
#define EMLOCO_MAXC 20      /* contacts kept per env and substep */
#define EMLOCO_MAXCAND 96   /* contact-candidate slots per env (warm-start storage) */

enum { EMLOCO_GEOM_SPHERE = 0, EMLOCO_GEOM_CAPSULE = 1, EMLOCO_GEOM_BOX = 2 };

enum {
    EMLOCO_OK = 0,
    EMLOCO_E_ARG = -1,      /* bad argument */
    EMLOCO_E_HIP = -2,      /* a HIP runtime call failed (emloco_last_error() has the text) */
    EMLOCO_E_STATE = -3,    /* call out of order (e.g. step before prepare) */
    EMLOCO_E_NODEV = -4     /* no usable gfx950 device */
};

/* gymapi.SimParams subset that reaches the step (pacer/pacer/utils/config.py:141-174,
 * pacer/pacer/data/cfg/pacer.yaml:93-104). */
typedef struct {
    int32_t n_sub;          /* substeps fused per emloco_sim_step call-unit: sim.substeps */
    int32_t n_iter;         /* physx.num_position_iterations */
    float h;                /* sim.dt / sim.substeps */
    float gravity_z;        /* sim.gravity.z */
    float contact_offset;   /* physx.contact_offset */
    float erp;              /* penetration fraction corrected per substep */
    float max_depen_vel;    /* physx.max_depenetration_velocity */
    float mu;               /* friction: plane/terrain staticFriction (pacer.yaml:70-73,90-92) */
    float ang_damping;      /* AssetOptions.angular_damping (humanoid.py:685) */
    float max_ang_vel;      /* AssetOptions.max_angular_velocity (humanoid.py:687) */
    float ground_z;         /* plane height (gymapi.PlaneParams.distance) */
    float cfm;              /* relative diagonal regularisation of the contact matrix */
    float warm;             /* contact warm-start factor */
    int32_t drive_mode;     /* 0: position drives (gymapi.DOF_MODE_POS, implicit PD towards the targets of
                             * emloco_sim_set_pd_targets); 1: effort drives (gymapi.DOF_MODE_EFFORT, `pdControl: False`,
                             * humanoid.py:905-913,1203-1207): the torques of emloco_sim_set_dof_actuation_force, clipped to the
                             * effort limits, no drive stiffness or damping */
} EmlocoSimParams;

/* Host description of the humanoids, one model per env (gym.load_asset + create_actor,
 * humanoid.py:720,864; per-env shapes: humanoid.py:597-633).  All arrays are host memory, fp32/int32. */
typedef struct {
    int32_t n_env;
    const int32_t *parent;     /* [24] body tree, depth-first order */
    const int32_t *geom_type;  /* [24] */
    const float *joint_off;    /* [n_env][24][3] */
    const float *mass;         /* [n_env][24] */
    const float *com;          /* [n_env][24][3] */
    const float *inertia;      /* [n_env][24][6] xx yy zz xy xz yz about com, body frame */
    const float *geom_a;       /* [n_env][24][3] */
    const float *geom_b;       /* [n_env][24][3] */
    const float *geom_r;       /* [n_env][24] */
    const float *kp, *kd, *armature, *effort; /* [n_env][69] (gym.set_actor_dof_properties, humanoid.py:904-914) */
} EmlocoModelDesc;

/* Limb-limb contact (`has_self_collision: True`, pacer.yaml:17; per-shape filters humanoid.py:917-944).  Penalty
 * contacts between the bodies' collision capsules, evaluated every substep inside the step kernel: for each listed
 * pair, closest points of the two sphere-swept segments; on overlap a force k * pen - c * v_n (>= 0, pen capped at
 * max_pen) along the contact normal acts on both bodies at the contact point (equal and opposite), plus Coulomb friction
 * regularised by the same damper: - min(mu F / |v_t|, c) v_t with v_t the tangential relative velocity there.  Optional: call
 * between emloco_sim_set_models and emloco_sim_prepare; never calling it (or n_pairs = 0) leaves self-collision off. */
#define EMLOCO_SC_MAXPAIRS 320
#define EMLOCO_SC_MAXSEG 32     /* collision segments per humanoid: one per body + up to 8 second ones (wide boxes: the feet) */
#define EMLOCO_SC_MAXHITS 32    /* simultaneous limb-limb contacts kept per env and substep (lowest pair indices first) */
typedef struct {
    int32_t n_pairs;
    const uint8_t *pairs;      /* [n_pairs][2] SEGMENT indices, i < j, shared by all envs (segment = body while n_seg = 0) */
    const float *cap_a;        /* [n_env][n_seg or 24][3] capsule end 0, frame of the segment's body */
    const float *cap_b;        /* [n_env][n_seg or 24][3] capsule end 1 */
    const float *cap_r;        /* [n_env][n_seg or 24] radius */
    float k;                   /* stiffness [N/m] */
    float c;                   /* normal damping [N s/m] */
    float max_pen;             /* penetration used for the spring is capped here [m] */
    float mu;                  /* friction coefficient of the limb-limb contacts (0: frictionless) */
    /* Round 4: a body may carry more than one sphere-swept segment -- a box much wider than thick (the SMPL humanoid's ankle boxes,
     * 17 x 9.7 x 4.2 cm, smpl_humanoid.xml) is two parallel capsules along its long edges, not one down its middle.  n_seg = 0 (or
     * seg_body NULL): 24 segments, segment i on body i.  Else 24 <= n_seg <= EMLOCO_SC_MAXSEG, seg_body[i] = i for i < 24 and the
     * body of every further segment; pairs never join two segments of one body. */
    int32_t n_seg;
    const uint8_t *seg_body;   /* [n_seg] */
} EmlocoSelfCollisionDesc;

/* state tensors a caller may alias (gym.acquire_*_tensor, humanoid.py:137-148) */
enum {
    EMLOCO_T_ROOT_STATE = 0,    /* f32 [n_env][13]      acquire_actor_root_state_tensor */
    EMLOCO_T_DOF_STATE = 1,     /* f32 [n_env*69][2]    acquire_dof_state_tensor */
    EMLOCO_T_RIGID_BODY = 2,    /* f32 [n_env*24][13]   acquire_rigid_body_state_tensor */
    EMLOCO_T_CONTACT_FORCE = 3, /* f32 [n_env*24][3]    acquire_net_contact_force_tensor */
    EMLOCO_T_DOF_FORCE = 4,     /* f32 [n_env*69]       acquire_dof_force_tensor */
    EMLOCO_T_PD_TARGET = 5,     /* f32 [n_env][69]      internal copy of the last set_dof_position_target_tensor */
    EMLOCO_T_WARM_START = 6,    /* f32 [n_env][EMLOCO_MAXCAND*3]  contact impulses carried between steps (solver state: save /
                                   restore it with the five state tensors to resume a rollout bit for bit) */
    EMLOCO_T_COUNT = 7
};

typedef struct EmlocoSim EmlocoSim;

const char *emloco_last_error(void);
int emloco_device_count(void);

/* gym.create_sim(compute_device, graphics_device, type, params) -- base_task.py:238 */
int emloco_sim_create(const EmlocoSimParams *params, int device, EmlocoSim **out);
/* gym.destroy_sim */
int emloco_sim_destroy(EmlocoSim *sim);
/* gym.load_asset + create_env/create_actor + set_actor_dof_properties for all envs -- humanoid.py:720,809,864,914 */
int emloco_sim_set_models(EmlocoSim *sim, const EmlocoModelDesc *desc);
/* gym.prepare_sim -- base_task.py:128: allocates the device state, uploads the models */
int emloco_sim_set_self_collision(EmlocoSim *sim, const EmlocoSelfCollisionDesc *desc);
/* gym.add_triangle_mesh for a height-field terrain -- humanoid_pedestrain_terrain.py:859-881 (mesh built by
 * terrain_utils.convert_heightfield_to_trimesh from Terrain.height_field_raw, :1135-1194).  `samples` is the int16 field
 * [nx][ny] (first axis = x) in units of `vertical_scale` metres on a `horizontal_scale`-metre grid; sample (0, 0) sits at
 * world (origin_x, origin_y) (the mesh transform, zero in the reference).  Host pointer, copied.  Each cell collides as the
 * mesh's two triangles (v00, v10, v11) / (v00, v11, v01); a body's contact sphere is tested against the plane of the
 * triangle under its centre and, when it has a radius, of the triangles under four probes one radius out along +-x / +-y (the nearest
 * plane whose perpendicular foot lies in its own triangle wins: a neighbouring face is met when the sphere's surface reaches it).  NULL samples restore the plane z = ground_z.  Call before emloco_sim_prepare. */
int emloco_sim_set_ground_heightfield(EmlocoSim *sim, const int16_t *samples, int nx, int ny, float horizontal_scale,
                                      float vertical_scale, float origin_x, float origin_y);
/* The SLOPE-CORRECTED terrain mesh -- terrain_utils.py:313-325 (convert_heightfield_to_trimesh with slope_threshold, the mesh
 * humanoid_pedestrain_terrain.py:859-881 hands to gym.add_triangle_mesh): where the step between two neighbouring samples exceeds the
 * threshold the lower vertex sits one cell sideways, under the upper one, so a stair riser is a vertical face.  `move_x`, `move_y`
 * [nx][ny]: how many cells (-1, 0, +1) vertex (i, j) of the mesh sits away from its grid position along x / y (host pointers, copied;
 * both NULL: none).  Call after emloco_sim_set_ground_heightfield and before emloco_sim_prepare.  With moves set a contact sphere is
 * tested against the mesh triangle that covers its centre (highest of the 3 x 3 cells' triangles; cells without a moved vertex nearby
 * keep the regular-grid formula bit for bit), the four probed triangles, and the mesh's vertical faces (closest point by vertex / edge /
 * face regions); still ONE contact per candidate: the nearest surface on the centre's side -- outside the solid the nearest face in
 * front of the centre, inside it (a box corner that crossed a riser) the nearest face behind it. */
int emloco_sim_set_ground_mesh_moves(EmlocoSim *sim, const int8_t *move_x, const int8_t *move_y);
int emloco_sim_prepare(EmlocoSim *sim);
/* gym.get_sim_params / set_sim_params -- base_task.py:151 */
int emloco_sim_get_params(EmlocoSim *sim, EmlocoSimParams *out);
/* (set_params validates like emloco_sim_create; once the sim is prepared `drive_mode` is ignored: the live mode belongs to the
 * last emloco_sim_set_pd_targets / emloco_sim_set_dof_actuation_force upload) */
int emloco_sim_set_params(EmlocoSim *sim, const EmlocoSimParams *in);
/* gym.acquire_*_tensor -- humanoid.py:137-148: device pointer + shape of a state tensor the sim owns */
int emloco_sim_tensor(EmlocoSim *sim, int kind, void **dev_ptr, int64_t shape[2]);
/* gym.set_dof_position_target_tensor -- humanoid.py:1201-1202 (device pointer, [n_env][69] f32) */
int emloco_sim_set_pd_targets(EmlocoSim *sim, const float *dev_targets, void *stream);
/* gym.set_dof_actuation_force_tensor -- humanoid.py:1206-1207 (device pointer, [n_env][69] f32 joint torques).  Switches the
 * sim to effort drives (drive_mode = 1) for the steps that follow; emloco_sim_set_pd_targets switches back. */
int emloco_sim_set_dof_actuation_force(EmlocoSim *sim, const float *dev_forces, void *stream);
/* gym.simulate x n_calls -- base_task.py:792-797 (n_calls = controlFrequencyInv); one fused launch */
int emloco_sim_step(EmlocoSim *sim, int n_calls, void *stream);
/* The same step for a subset of the envs (extension; envs are independent, humanoid.py:838-841, so a step of all envs may be
 * issued as two launches on two streams: the envs that just finished an episode are reset and stepped beside the others,
 * amp_continuous_value.py:46,74 env_reset(done_indices) followed by env.step).  Exactly one of the two selectors:
 * `dev_skip` (int64 per env, the task's reset_buf layout): envs with a non-zero entry are left untouched;
 * `dev_env_ids` (int32, n_ids entries): a device-compacted list, valid ids first and -1 after them (emloco_task_compact_done). */
int emloco_sim_step_subset(EmlocoSim *sim, int n_calls, const int64_t *dev_skip, const int32_t *dev_env_ids, int n_ids, void *stream);
/* Split launch (extension, off = 1 part by default; results do not depend on it).  The substeps of an env.step run as
 * `n_parts` workgroups per env in ONE launch -- all first parts, then all second parts, ... -- a later part continuing from
 * the registers its predecessor published (plain copies: the parts together are the fused step bit for bit).  The launch is
 * a whole number of resident rounds of waves only on paper: envs differ in cost, and with one workgroup per env the wave
 * slots of the cheap ones idle until the last workgroup ends; with parts they are refilled at half / quarter granularity.
 * n_sub * n_calls must be a multiple of n_parts to split evenly (otherwise the parts are uneven, still correct). */
int emloco_sim_set_split(EmlocoSim *sim, int n_parts);
/* Cost-ordered dispatch of the step launch (extension, off by default; results do not depend on it: envs are independent).
 * Every workgroup records how long its env's step took; the next launch hands the envs to the CUs longest first (a counting
 * sort on the device, one small launch ahead of the step).  The launch is two resident rounds of waves, so its length is set
 * by what the last workgroups cost: airborne humanoids (no contact solve) take ~60 % of the time of fallen ones. */
int emloco_sim_set_cost_order(EmlocoSim *sim, int on);
/* gym.fetch_results(sim, True) -- base_task.py:258: host waits for the stream */
int emloco_sim_sync(EmlocoSim *sim, void *stream);
/* Fault injection for the split launch's hand-over (no counterpart in the gym API; the product never calls it).  The first
 * part of `env` withholds its hand-over flag in the split launches that follow (env = -1: back to normal), and a later part's
 * wait is bounded by `spin_max` sleeps (<= 0: the default, 1 << 22).  A wait that runs out raises the device error word that
 * emloco_sim_sync / the next emloco_sim_step return as EMLOCO_E_HIP: this call lets a test see that path in milliseconds
 * instead of seconds (tests/test_gpu_sim.py).  Costs the kernel one scalar compare per part. */
int emloco_sim_debug_poison_part(EmlocoSim *sim, int env, int spin_max);
/* Diagnostics and one cross-unit helper (no counterpart in the gym API; the product's Python never calls the three diagnostic
 * ones -- tools/sim_phase_profile.py, tools/exp/cost_hist.py do).  They synchronise the device and copy to HOST buffers.
 *   emloco_sim_fk_indexed   forward kinematics of the listed envs (device int32 ids); the reset chain of emloco_task.h calls it
 *                           (the two translation units link through this symbol)
 *   emloco_sim_cost_ticks   per-env durations (100 MHz ticks) of the latest step, the key of the cost-ordered dispatch
 *                           (emloco_sim_set_cost_order must be on) and, from the second call on, each workgroup's start stamps
 *                           (host_start: 2 n_envs entries, or NULL)
 *   emloco_sim_profile      wall-clock stamps of env 0, 16 per substep (kernels built with -DEMLOCO_SIM_PROFILE; without it the
 *                           buffer stays zero).  The first call allocates the stamp buffer and returns. */
int emloco_sim_fk_indexed(EmlocoSim *sim, const int32_t *dev_env_ids, int n, void *stream);
int emloco_sim_cost_ticks(EmlocoSim *sim, unsigned *host_ticks, unsigned long long *host_start, int n);
int emloco_sim_profile(EmlocoSim *sim, long long *host_out, int n);
/* gym.set_actor_root_state_tensor_indexed / set_dof_state_tensor_indexed -- humanoid.py:470-475.
 * `dev_full` is the full tensor (may be the sim's own alias); rows of the listed envs are applied and the
 * rigid-body state of those envs is recomputed.  `dev_env_ids` int32 device pointer, n entries. */
int emloco_sim_set_root_state_indexed(EmlocoSim *sim, const float *dev_full, const int32_t *dev_env_ids, int n, void *stream);
int emloco_sim_set_dof_state_indexed(EmlocoSim *sim, const float *dev_full, const int32_t *dev_env_ids, int n, void *stream);
/* gym.refresh_rigid_body_state_tensor after host-side state edits: recompute body states of all envs */
int emloco_sim_refresh_bodies(EmlocoSim *sim, void *stream);
/* number of ground-contact candidates of the loaded model (68 for the SMPL humanoid) */
int emloco_sim_num_candidates(EmlocoSim *sim);
/* wall-clock of the last emloco_sim_step launch measured with HIP events on its stream [ms]; <0 if none */
float emloco_sim_last_step_ms(EmlocoSim *sim);
/* enable/disable HIP-event timing of step launches (events are recorded on the launch stream; enabling resets the log);
 * on = 1: every launch, on = N > 1: every N-th launch (an event record is a packet of its own in the stream, ~5 us each) */
int emloco_sim_enable_timing(EmlocoSim *sim, int on);
/* number and summed duration [ms] of the step launches recorded since timing was enabled / last queried (<= 1024 kept) */
int emloco_sim_timing_stats(EmlocoSim *sim, int *n_launches, float *total_ms);

#ifdef __cplusplus
}
#endif
#endif
