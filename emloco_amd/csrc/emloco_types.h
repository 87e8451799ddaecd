// emloco_types.h -- device-side argument structs shared by the kernels and the C-ABI layer.
#pragma once
#define EMLOCO_PART_WORDS 416   /* 24 x 12 joint registers | 16 root | 8 momentum | 64 multipliers | 32 slot map | 4 misc | pad: 16-byte granules */
#include "../../include/emloco_sim.h"
/* hand-over tag of part `part` of the launch numbered `seq`: unique whatever n_parts the launches before it used (list launches
 * and unsplit launches bump the sequence number without publishing, so a stale flag can never equal an awaited tag) */
#define EMLOCO_PART_TAG(seq, part) ((seq) * 4u + (unsigned)(part))
#define EMLOCO_ERR_PART_TIMEOUT 1u

// Topology block (ints, shared by all envs; built by model_pack.h): one pointer instead of eight
#define EMLOCO_TOPO_PARENT 0      /* [24] parent body (-1: root) */
#define EMLOCO_TOPO_DEPTH 24      /* [24] tree depth */
#define EMLOCO_TOPO_CHILD 48      /* [24][3] children, descending body index, -1 padded */
#define EMLOCO_TOPO_PDPACK 120    /* [24] parent (the root: 31) | depth << 5 | index among the bodies of its depth << 9 | children << 12, 17, 22 */
#define EMLOCO_TOPO_CAND 144      /* [96] ground-contact candidate: body | k << 8 | geom type << 16 */
#define EMLOCO_TOPO_SCPAIR 240    /* [320] limb-limb pair: segment i | segment j << 8 | body of i << 16 | body of j << 24 */
#define EMLOCO_TOPO_WORDS 560

// Per-env model block (floats, [n_env][EMLOCO_MODEL_WORDS]; built by model_pack.h).  16-byte records so that a lane fetches what a
// phase needs with a few dwordx4 loads off ONE workgroup-uniform base (scalar registers) and a 32-bit lane offset:
#define EMLOCO_MB_DYN 0           /* [24][16]  joint offset xyz, mass | com xyz, 0 | Ixx Iyy Izz Ixy | Ixz Iyz 0 0 */
#define EMLOCO_MB_GEO 384         /* [24][8]   geom a xyz, geom radius | geom b xyz, 0 */
#define EMLOCO_MB_CAP 576         /* [32][8]   collision segment: capsule end a xyz, radius | end b xyz, its body  (self-collision; zeros when off) */
#define EMLOCO_MB_DRV 832         /* [69][4]   kp, kd, armature, effort limit */
#define EMLOCO_MODEL_WORDS 1112

// Device pointers handed to the rollout kernels by value (kernarg segment).
struct EmlocoSimDev {
    int n_env, n_cand, max_depth, sc_n;       /* sc_n: limb-limb pairs (0: self-collision off) */
    int sc_nseg;                              /* collision segments per env (24 .. EMLOCO_SC_MAXSEG) */
    const int *topo;                          /* EMLOCO_TOPO_* */
    const float *model;                       /* EMLOCO_MB_*, one block per env */
    // state
    float *root_state, *dof_state;
    const float *pd_target;
    float *rb_state, *contact_force, *dof_force, *lambda_ws;
    float sc_k, sc_c, sc_max_pen, sc_mu;      /* limb-limb penalty contacts */
    // height-field ground (hf = NULL: the plane z = ground_z).  hf[ix * hf_ny + iy] in units of hf_vs metres on an
    // hf_hs-metre grid whose sample (0, 0) sits at world (hf_ox, hf_oy)
    const short *hf;
    int hf_nx, hf_ny;
    float hf_hs, hf_inv_hs, hf_vs, hf_ox, hf_oy, hf_pad_;
    // vertex moves of the slope-corrected mesh (emloco_sim_set_ground_mesh_moves), NULL: none.  One byte per sample: bits 0-1 move
    // along x + 1, bits 2-3 move along y + 1, bit 4 = some vertex of the 4 x 4 block around cell (i, j) moved
    const unsigned char *hf_mv;
    long long *prof;   /* optional (built with -DEMLOCO_SIM_PROFILE): per-phase cycle stamps of env 0, else NULL */
    // subset launches (emloco_sim_step_subset), both NULL for the plain step: envs whose step_skip entry is non-zero are left
    // untouched; with step_ids workgroup i steps env step_ids[i] of a device-compacted list (valid ids first, -1 after them)
    const long long *step_skip;
    const int *step_ids;
    int n_slots;                      /* env slots of this launch (n_env, or the length of the id list): workgroup i steps slots 2 i and 2 i + 1 */
    // cost-ordered dispatch (emloco_sim_set_cost_order): workgroup i of the full launch steps env step_order[i]; every
    // workgroup leaves its contact work in step_ticks[env], the key of the next launch's order
    const int *step_order;
    unsigned *step_ticks;
    // split launch (emloco_sim_set_split): the substeps of an env.step as n_parts workgroups per env, part p + 1 continuing
    // from the registers / LDS part p left in part_state [n_env][EMLOCO_PART_WORDS] once part_flag[env] holds the predecessor's tag (EMLOCO_PART_TAG)
    int n_parts; unsigned part_seq;
    float *part_state;
    unsigned *part_flag;
    unsigned *err;                    /* device error word (sim_capi.hip: surfaced by emloco_sim_sync and by the next step): bit 0 = a part of a
                                         split launch gave up waiting for its predecessor's hand-over; that env's step was abandoned */
    int part_spin_max;                /* bound of a part's wait for its predecessor, in 16 x 64-clock sleeps (default 1 << 22: seconds) */
    int part_poison;                  /* test hook (emloco_sim_debug_poison_part): the first part of this env withholds its flag; -1: none */
    unsigned long long *step_start;   /* diagnostic (emloco_sim_cost_ticks): wall clock at the start of each env's workgroup, else NULL */
};
