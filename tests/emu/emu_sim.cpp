#include <vector>
#include <cstdlib>
// TEST INFRASTRUCTURE ONLY: runs emloco_amd/csrc/sim_kernels.hip on the CPU through the emulation
// header in tests/emu/hip/, so the kernel's logic can be compared with the oracle without a GPU.
#include "hip/hip_runtime.h"
#ifdef EMLOCO_SIM_PAIR      /* EMLOCO_EMU_EXTRA="-DEMLOCO_SIM_PAIR=1": the two-envs-per-wave kernel (round 6, experimental) */
#include "../../emloco_amd/csrc/sim_pair_kernels.hip"
#define EMU_ENVS_PER_WG 2
#else
#include "../../emloco_amd/csrc/sim_kernels.hip"
#define EMU_ENVS_PER_WG 1
#endif
#include "../../emloco_amd/csrc/topology.h"
#include "../../emloco_amd/csrc/model_pack.h"

static const EmlocoSelfCollisionDesc *g_sc = nullptr;      // set by emu_sim_set_self_collision for the next emu_sim_step calls
extern "C" void emu_sim_set_self_collision(const EmlocoSelfCollisionDesc *sc) { g_sc = sc; }

struct EmuHeightfield { const short *samples; int nx, ny; float hs, vs, ox, oy; const unsigned char *mv; };
static EmuHeightfield g_hf = {nullptr, 0, 0, 0, 0, 0, 0, nullptr};   // set by emu_sim_set_heightfield for the next emu_sim_step calls
extern "C" void emu_sim_set_heightfield(const short *samples, int nx, int ny, float hs, float vs, float ox, float oy, const unsigned char *mv) {
    g_hf = {samples, nx, ny, hs, vs, ox, oy, mv};       // mv: packed vertex moves of the slope-corrected mesh (emloco_types.h: hf_mv) or NULL
}

extern "C" int emu_sim_step(const EmlocoSimParams *prm, const EmlocoModelDesc *m, float *root_state,
                            float *dof_state, const float *pd_target, float *rb_state, float *contact_force,
                            float *dof_force, float *lambda_ws, int n_calls) {
    emloco::Topology t;
    if (!t.build(m->parent, m->geom_type)) return -1;
    EmlocoSimDev d{};
    d.n_env = m->n_env; d.n_cand = t.n_cand; d.max_depth = t.max_depth;
    const bool sc_on = g_sc && g_sc->n_pairs > 0;
    const int n_seg = (sc_on && g_sc->n_seg > 0 && g_sc->seg_body) ? g_sc->n_seg : EMLOCO_NB;
    const unsigned char *seg_body = (sc_on && g_sc->n_seg > 0) ? g_sc->seg_body : nullptr;
    const std::vector<int32_t> topo = emloco::pack_topology(t, sc_on ? g_sc->pairs : nullptr, sc_on ? g_sc->n_pairs : 0, seg_body);
    const std::vector<float> mdl = emloco::pack_models(m->n_env, m->joint_off, m->mass, m->com, m->inertia, m->geom_a, m->geom_b, m->geom_r,
                                                       m->kp, m->kd, m->armature, m->effort, sc_on ? g_sc->cap_a : nullptr,
                                                       sc_on ? g_sc->cap_b : nullptr, sc_on ? g_sc->cap_r : nullptr, n_seg, seg_body);
    d.topo = topo.data(); d.model = mdl.data();
    d.root_state = root_state; d.dof_state = dof_state; d.pd_target = pd_target;
    d.rb_state = rb_state; d.contact_force = contact_force; d.dof_force = dof_force; d.lambda_ws = lambda_ws;
    if (sc_on) { d.sc_nseg = n_seg; d.sc_n = g_sc->n_pairs; d.sc_k = g_sc->k; d.sc_c = g_sc->c; d.sc_max_pen = g_sc->max_pen; d.sc_mu = g_sc->mu; }
    if (g_hf.samples) {
        d.hf = g_hf.samples; d.hf_nx = g_hf.nx; d.hf_ny = g_hf.ny; d.hf_hs = g_hf.hs; d.hf_inv_hs = 1.0f / g_hf.hs; d.hf_vs = g_hf.vs;
        d.hf_ox = g_hf.ox; d.hf_oy = g_hf.oy; d.hf_mv = g_hf.mv;
    }
    EmlocoSimParams p = *prm;
    p.n_sub = prm->n_sub * n_calls;
    // EMLOCO_EMU_PARTS=n: the split launch (emloco_sim_set_split) -- n workgroups per env, all first parts first
    const char *pe = getenv("EMLOCO_EMU_PARTS");
    const int n_parts = pe ? atoi(pe) : 1;
    std::vector<float> part_state((size_t)m->n_env * EMLOCO_PART_WORDS, 0.0f);
    std::vector<unsigned> part_flag((size_t)m->n_env, 0u);
    d.n_parts = n_parts; d.part_seq = 1; d.part_state = part_state.data(); d.part_flag = part_flag.data();
    // EMLOCO_EMU_POISON=env: the first part of that env withholds its hand-over flag (emloco_sim_debug_poison_part); the
    // device error word is the return value (0: fine)
    const char *po = getenv("EMLOCO_EMU_POISON");
    unsigned err = 0u;
    d.part_spin_max = 4; d.part_poison = po ? atoi(po) : -1; d.err = &err;
    d.n_slots = m->n_env;
    emu::launch((unsigned)(((m->n_env + EMU_ENVS_PER_WG - 1) / EMU_ENVS_PER_WG) * n_parts), 64, [&] { if (d.hf) emloco::sim_step_kernel<1>(p, d); else emloco::sim_step_kernel<0>(p, d); });
    return (int)err;
}

extern "C" int emu_sim_fk(const EmlocoModelDesc *m, float *root_state, float *dof_state, float *rb_state) {
    emloco::Topology t;
    if (!t.build(m->parent, m->geom_type)) return -1;
    EmlocoSimDev d{};
    d.n_env = m->n_env; d.n_cand = t.n_cand; d.max_depth = t.max_depth;
    const std::vector<int32_t> topo = emloco::pack_topology(t, nullptr, 0);
    const std::vector<float> mdl = emloco::pack_models(m->n_env, m->joint_off, m->mass, m->com, m->inertia, m->geom_a, m->geom_b, m->geom_r,
                                                       m->kp, m->kd, m->armature, m->effort, nullptr, nullptr, nullptr);
    d.topo = topo.data(); d.model = mdl.data();
    d.root_state = root_state; d.dof_state = dof_state; d.rb_state = rb_state;
    emu::launch((unsigned)m->n_env, 64, [&] { emloco::sim_fk_kernel(d, nullptr, m->n_env); });
    return 0;
}
