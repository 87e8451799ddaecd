/*
 * oracle_sim.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Sequential CPU restatement of this repo's rigid-body step, the stage that stands
 * where the reference calls gym.simulate (base_task.py:792-797).  PARITY UNPINNED against the
 * reference: its engine (Isaac Gym 1.0.preview4 / PhysX 5) is absent from /root/reference.
 */
#ifndef EMLOCO_ORACLE_SIM_H
#define EMLOCO_ORACLE_SIM_H
#include <stdint.h>

#define ORC_NB 24
#define ORC_NJ 23
#define ORC_NDOF 69
#define ORC_SC_MAXHITS 32
#define ORC_SC_MAXSEG 32
#define ORC_MAXCAND 96
#define ORC_MAXC 20

enum { ORC_GEOM_SPHERE = 0, ORC_GEOM_CAPSULE = 1, ORC_GEOM_BOX = 2 };

typedef struct {
    int32_t n_sub;         /* substeps fused per call (controlFrequencyInv * substeps) */
    int32_t n_iter;        /* contact solver sweeps per substep (num_position_iterations) */
    float h;               /* substep length [s] */
    float gravity_z;       /* [m/s^2], negative */
    float contact_offset;  /* contacts are generated below this distance [m] */
    float erp;             /* fraction of penetration corrected per substep */
    float max_depen_vel;   /* cap of the penetration-correction speed [m/s] */
    float mu;              /* Coulomb friction coefficient */
    float ang_damping;     /* angular damping [1/s] */
    float max_ang_vel;     /* clamp of joint / root angular speed [rad/s] */
    float ground_z;        /* height of the ground plane */
    float cfm;             /* relative diagonal regularisation of the contact matrix */
    float warm;            /* warm-start factor for contact impulses */
    int32_t drive_mode;    /* 0: implicit PD drives towards pd_target; 1: effort drives -- pd_target holds joint torques, applied
                            * clipped to the effort limits with no drive stiffness / damping (gymapi.DOF_MODE_EFFORT) */
} OrcSimParams;

typedef struct {
    /* topology shared by all envs */
    const int32_t *parent;     /* [24] */
    const int32_t *geom_type;  /* [24] */
    /* per-env (leading dim n_env) */
    const float *joint_off;    /* [E][24][3] body origin in the parent frame */
    const float *mass;         /* [E][24] */
    const float *com;          /* [E][24][3] body frame */
    const float *inertia;      /* [E][24][6] about com, body frame: xx yy zz xy xz yz */
    const float *geom_a;       /* [E][24][3] sphere centre | capsule end 0 | box centre */
    const float *geom_b;       /* [E][24][3] unused        | capsule end 1 | box half extents */
    const float *geom_r;       /* [E][24] radius (0 for boxes) */
    const float *kp, *kd, *armature, *effort; /* [E][69] */
    /* optional limb-limb penalty contacts (sc_n = 0: off); mirrors EmlocoSelfCollisionDesc */
    int32_t sc_n;
    const uint8_t *sc_pairs;   /* [sc_n][2] */
    const float *sc_cap_a, *sc_cap_b, *sc_cap_r;   /* [E][sc_nseg or 24][3|3|1] */
    float sc_k, sc_c, sc_max_pen, sc_mu;
    int32_t sc_nseg;           /* collision segments per env (0: 24, segment i on body i); sc_pairs index segments */
    const uint8_t *sc_segbody; /* [sc_nseg] body of each segment */
    /* optional height-field ground (hf = NULL: the plane z = ground_z); mirrors emloco_sim_set_ground_heightfield */
    const int16_t *hf;         /* [hf_nx][hf_ny] in units of hf_vs metres on an hf_hs-metre grid, sample (0,0) at (hf_ox, hf_oy) */
    int32_t hf_nx, hf_ny;
    float hf_hs, hf_vs, hf_ox, hf_oy;
    /* optional vertex moves of the slope-corrected mesh (terrain_utils.py:313-325), NULL: none.  One byte per sample: bits 0-1 =
     * move along x + 1, bits 2-3 = move along y + 1 (each move -1, 0 or +1 cells), bit 4 = some vertex of the 4 x 4 block around
     * cell (i, j) moved (the cell's lookups take the mesh path); mirrors emloco_sim_set_ground_mesh_moves */
    const uint8_t *hf_mv;
} OrcModel;

/* one call = n_sub substeps for every env */
void orc_sim_step(int n_env, const OrcSimParams *prm, const OrcModel *mdl,
                  float *root_state,     /* [E][13] in/out: pos3 quat4(xyzw) linvel3 angvel3 */
                  float *dof_state,      /* [E][69][2] in/out: exp-map position, joint-frame velocity */
                  const float *pd_target,/* [E][69] */
                  float *rb_state,       /* [E][24][13] out */
                  float *contact_force,  /* [E][24][3] out (last substep) */
                  float *dof_force,      /* [E][69] out (last substep) */
                  float *lambda_ws);     /* [E][ORC_MAXCAND][3] in/out warm-start impulses per candidate */

/* forward kinematics only: fills rb_state from root_state / dof_state */
void orc_sim_fk(int n_env, const OrcModel *mdl, const float *root_state, const float *dof_state, float *rb_state);

/* test hooks */
int orc_sim_num_candidates(const int32_t *geom_type);
/* dense generalized mass matrix (75x75, row-major; root = [ang3, lin3] world axes about the root origin,
 * then 3 per joint) and bias force for one env, straight from body Jacobians: used to validate the ABA. */
void orc_sim_dense_dynamics(const OrcSimParams *prm, const OrcModel *mdl, int env,
                            const float *root_state, const float *dof_state, const float *pd_target,
                            double *M75x75, double *rhs75);
/* unconstrained (contact-free) acceleration of one env by the ABA factorisation: qdd[75] */
int orc_sim_self_contacts(const OrcSimParams *prm, const OrcModel *mdl, int env, const float *root_state, const float *dof_state,
                          float *info);
void orc_sim_free_accel(const OrcSimParams *prm, const OrcModel *mdl, int env,
                        const float *root_state, const float *dof_state, const float *pd_target, float *qdd75);
#endif
