// sim_capi.hip -- C ABI (include/emloco_sim.h) over the rollout kernels: owns device state, uploads
// the per-env models, launches the fused step.  Host code is C++; nothing here touches torch.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
// EMLOCO_SIM_PAIR=1 (round 6, experimental: python -m emloco_amd.build with EMLOCO_HIPCC_EXTRA_SIM="-DEMLOCO_SIM_PAIR=1"): the rigid-body
// step with TWO envs per 64-lane wave (sim_pair_kernels.hip) instead of one (sim_kernels.hip).  Same entry points, same bytes.
#ifdef EMLOCO_SIM_PAIR
#include "sim_pair_kernels.hip"
#define EMLOCO_SIM_ENVS_PER_WG 2
#else
#include "sim_kernels.hip"
#define EMLOCO_SIM_ENVS_PER_WG 1
#endif
#include "topology.h"
#include "model_pack.h"
#include "sim_state.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof(buf), "%s", what);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                            \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) return fail(EMLOCO_E_HIP, #expr, e_);             \
    } while (0)

// the device error word (EmlocoSimDev::err) lives in pinned host memory: reading it costs nothing and needs no copy
int check_device_error(EmlocoSim *s, const char *where) {
    if (!s->h_err) return EMLOCO_OK;
    const unsigned e = __atomic_exchange_n(s->h_err, 0u, __ATOMIC_RELAXED);     // reported once, then cleared
    if (e == 0u) return EMLOCO_OK;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: device error word 0x%x%s", where, e,
             (e & EMLOCO_ERR_PART_TIMEOUT) ? " (split launch: a part's wait for its predecessor's hand-over ran out; the step of that env was abandoned)" : "");
    return fail(EMLOCO_E_HIP, buf);
}

__global__ void copy_rows_kernel(const float *src, float *dst, const int *ids, int n_ids, int row_len) {
    const int i = blockIdx.x;
    if (i >= n_ids) return;
    const long base = (long)ids[i] * row_len;
    for (int k = threadIdx.x; k < row_len; k += blockDim.x) dst[base + k] = src[base + k];
}

__global__ void fill_quat_kernel(float *root, int n_env) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_env) root[(long)e * 13 + 6] = 1.0f;
}

}  // namespace

extern "C" {

const char *emloco_last_error(void) { return g_err.c_str(); }

int emloco_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int emloco_sim_create(const EmlocoSimParams *params, int device, EmlocoSim **out) {
    if (!params || !out) return fail(EMLOCO_E_ARG, "emloco_sim_create: null argument");
    if (params->n_sub < 1 || params->h <= 0.0f || params->n_iter < 0 || params->drive_mode < 0 || params->drive_mode > 1)
        return fail(EMLOCO_E_ARG, "emloco_sim_create: n_sub >= 1, h > 0, n_iter >= 0, drive_mode 0 | 1 required");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(EMLOCO_E_NODEV, "no HIP device visible");
    if (device < 0 || device >= n) return fail(EMLOCO_E_ARG, "emloco_sim_create: device index out of range");
    HIPCHK(hipSetDevice(device));
    EmlocoSim *s = new EmlocoSim();
    s->device = device;
    s->prm = *params;
    *out = s;
    return EMLOCO_OK;
}

int emloco_sim_destroy(EmlocoSim *s) {
    if (!s) return EMLOCO_OK;
    (void)hipSetDevice(s->device);
    s->d_topo.release(); s->d_model.release();
    s->d_root.release(); s->d_dof.release(); s->d_tgt.release(); s->d_rb.release();
    s->d_cf.release(); s->d_df.release(); s->d_lws.release();
    s->d_ticks.release(); s->d_order.release(); s->d_order_ws.release();
    s->d_part_state.release(); s->d_part_flag.release();
    if (s->h_err) (void)hipHostFree(s->h_err);
    for (auto e : s->ev0) (void)hipEventDestroy(e);
    for (auto e : s->ev1) (void)hipEventDestroy(e);
    delete s;
    return EMLOCO_OK;
}

int emloco_sim_set_models(EmlocoSim *s, const EmlocoModelDesc *m) {
    if (!s || !m) return fail(EMLOCO_E_ARG, "emloco_sim_set_models: null argument");
    if (s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_set_models: sim already prepared");
    if (m->n_env < 1) return fail(EMLOCO_E_ARG, "emloco_sim_set_models: n_env < 1");
    if (!s->topo.build(m->parent, m->geom_type))
        return fail(EMLOCO_E_ARG, "emloco_sim_set_models: body tree must be depth-first, depth <= 8, <= 3 children, <= 96 contact candidates");
    const size_t E = (size_t)m->n_env, NBs = EMLOCO_NB, ND = EMLOCO_NDOF;
    s->n_env = m->n_env;
    s->h_off.assign(m->joint_off, m->joint_off + E * NBs * 3);
    s->h_mass.assign(m->mass, m->mass + E * NBs);
    s->h_com.assign(m->com, m->com + E * NBs * 3);
    s->h_inertia.assign(m->inertia, m->inertia + E * NBs * 6);
    s->h_ga.assign(m->geom_a, m->geom_a + E * NBs * 3);
    s->h_gb.assign(m->geom_b, m->geom_b + E * NBs * 3);
    s->h_gr.assign(m->geom_r, m->geom_r + E * NBs);
    s->h_kp.assign(m->kp, m->kp + E * ND);
    s->h_kd.assign(m->kd, m->kd + E * ND);
    s->h_arm.assign(m->armature, m->armature + E * ND);
    s->h_eff.assign(m->effort, m->effort + E * ND);
    for (size_t i = 0; i < E * NBs; ++i)
        if (!(s->h_mass[i] > 0.0f)) return fail(EMLOCO_E_ARG, "emloco_sim_set_models: body masses must be positive");
    s->have_model = true;
    return EMLOCO_OK;
}

int emloco_sim_set_self_collision(EmlocoSim *s, const EmlocoSelfCollisionDesc *c) {
    if (!s || !c) return fail(EMLOCO_E_ARG, "emloco_sim_set_self_collision: null argument");
    if (!s->have_model) return fail(EMLOCO_E_STATE, "emloco_sim_set_self_collision: set the models first");
    if (s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_set_self_collision: sim already prepared");
    if (c->n_pairs < 0 || c->n_pairs > EMLOCO_SC_MAXPAIRS) return fail(EMLOCO_E_ARG, "emloco_sim_set_self_collision: at most EMLOCO_SC_MAXPAIRS pairs");
    s->h_sc_pairs.clear();
    if (c->n_pairs == 0) return EMLOCO_OK;
    if (!c->pairs || !c->cap_a || !c->cap_b || !c->cap_r || !(c->k > 0.0f) || c->c < 0.0f || !(c->max_pen > 0.0f) || !(c->mu >= 0.0f))
        return fail(EMLOCO_E_ARG, "emloco_sim_set_self_collision: missing arrays or non-positive stiffness / penetration cap");
    const int n_seg = (c->n_seg > 0 && c->seg_body) ? c->n_seg : EMLOCO_NB;
    if (n_seg < EMLOCO_NB || n_seg > EMLOCO_SC_MAXSEG) return fail(EMLOCO_E_ARG, "emloco_sim_set_self_collision: 24 <= n_seg <= EMLOCO_SC_MAXSEG");
    s->h_sc_segbody.resize((size_t)n_seg);
    for (int i = 0; i < n_seg; ++i) {
        const int b = (c->n_seg > 0 && c->seg_body) ? c->seg_body[i] : i;
        if (b >= EMLOCO_NB || (i < EMLOCO_NB && b != i)) return fail(EMLOCO_E_ARG, "emloco_sim_set_self_collision: seg_body[i] = i for i < 24, a body index beyond");
        s->h_sc_segbody[(size_t)i] = (unsigned char)b;
    }
    for (int i = 0; i < c->n_pairs; ++i) {
        const int si = c->pairs[2 * i], sj = c->pairs[2 * i + 1];
        if (si >= sj || sj >= n_seg || s->h_sc_segbody[(size_t)si] == s->h_sc_segbody[(size_t)sj])
            return fail(EMLOCO_E_ARG, "emloco_sim_set_self_collision: pairs must be segments (i < j < n_seg) of two different bodies");
    }
    s->sc_nseg = n_seg;
    const size_t E = (size_t)s->n_env, NBs = (size_t)n_seg;
    s->h_sc_pairs.assign(c->pairs, c->pairs + 2 * (size_t)c->n_pairs);
    s->h_sc_a.assign(c->cap_a, c->cap_a + E * NBs * 3);
    s->h_sc_b.assign(c->cap_b, c->cap_b + E * NBs * 3);
    s->h_sc_r.assign(c->cap_r, c->cap_r + E * NBs);
    s->sc_k = c->k; s->sc_c = c->c; s->sc_max_pen = c->max_pen; s->sc_mu = c->mu;
    return EMLOCO_OK;
}

int emloco_sim_set_ground_heightfield(EmlocoSim *s, const int16_t *samples, int nx, int ny, float horizontal_scale,
                                      float vertical_scale, float origin_x, float origin_y) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_set_ground_heightfield: null sim");
    if (s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_set_ground_heightfield: sim already prepared");
    s->h_hf.clear(); s->h_hf_mv.clear();
    if (!samples) return EMLOCO_OK;                                   // back to the plane
    if (nx < 2 || ny < 2 || !(horizontal_scale > 0.0f) || !(vertical_scale > 0.0f))
        return fail(EMLOCO_E_ARG, "emloco_sim_set_ground_heightfield: need a >= 2 x 2 grid and positive scales");
    s->h_hf.assign(samples, samples + (size_t)nx * (size_t)ny);
    s->hf_nx = nx; s->hf_ny = ny; s->hf_hs = horizontal_scale; s->hf_vs = vertical_scale; s->hf_ox = origin_x; s->hf_oy = origin_y;
    return EMLOCO_OK;
}

int emloco_sim_set_ground_mesh_moves(EmlocoSim *s, const int8_t *move_x, const int8_t *move_y) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_set_ground_mesh_moves: null sim");
    if (s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_set_ground_mesh_moves: sim already prepared");
    s->h_hf_mv.clear();
    if (!move_x && !move_y) return EMLOCO_OK;
    if (!move_x || !move_y) return fail(EMLOCO_E_ARG, "emloco_sim_set_ground_mesh_moves: need both move arrays (or neither)");
    if (s->h_hf.empty()) return fail(EMLOCO_E_STATE, "emloco_sim_set_ground_mesh_moves: set the height field first");
    const int nx = s->hf_nx, ny = s->hf_ny;
    std::vector<uint8_t> mv((size_t)nx * ny);
    bool any = false;
    for (size_t k = 0; k < mv.size(); ++k) {
        const int mx = move_x[k], my = move_y[k];
        if (mx < -1 || mx > 1 || my < -1 || my > 1) return fail(EMLOCO_E_ARG, "emloco_sim_set_ground_mesh_moves: a move must be -1, 0 or +1 cells");
        any = any || mx != 0 || my != 0;
        mv[k] = (uint8_t)((mx + 1) | ((my + 1) << 2));
    }
    if (!any) return EMLOCO_OK;                                      // an uncorrected mesh: the regular-grid path
    // bit 4 of sample (i, j): some vertex of the block (i - 1 .. i + 2) x (j - 1 .. j + 2) moved -- lookups from cell (i, j) take the mesh path
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j) {
            bool f = false;
            for (int a = i - 1; a <= i + 2 && !f; ++a)
                for (int b = j - 1; b <= j + 2 && !f; ++b)
                    if (a >= 0 && a < nx && b >= 0 && b < ny && (mv[(size_t)a * ny + b] & 15) != 5) f = true;
            if (f) mv[(size_t)i * ny + j] |= 16;
        }
    s->h_hf_mv.swap(mv);
    return EMLOCO_OK;
}

int emloco_sim_prepare(EmlocoSim *s) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_prepare: null sim");
    if (!s->have_model) return fail(EMLOCO_E_STATE, "emloco_sim_prepare: no models set");
    if (s->prepared) return EMLOCO_OK;
    HIPCHK(hipSetDevice(s->device));
    const emloco::Topology &t = s->topo;
    const bool sc_on = !s->h_sc_pairs.empty();
    {   // the two blocks the kernels read: topology tables, one model record block per env (model_pack.h)
        const std::vector<int32_t> topo = emloco::pack_topology(t, sc_on ? s->h_sc_pairs.data() : nullptr, sc_on ? (int)(s->h_sc_pairs.size() / 2) : 0,
                                                                sc_on ? s->h_sc_segbody.data() : nullptr);
        const std::vector<float> mdl = emloco::pack_models(s->n_env, s->h_off.data(), s->h_mass.data(), s->h_com.data(), s->h_inertia.data(),
                                                           s->h_ga.data(), s->h_gb.data(), s->h_gr.data(), s->h_kp.data(), s->h_kd.data(),
                                                           s->h_arm.data(), s->h_eff.data(), sc_on ? s->h_sc_a.data() : nullptr,
                                                           sc_on ? s->h_sc_b.data() : nullptr, sc_on ? s->h_sc_r.data() : nullptr,
                                                           sc_on ? s->sc_nseg : EMLOCO_NB, sc_on ? s->h_sc_segbody.data() : nullptr);
        HIPCHK(s->d_topo.upload(topo.data(), topo.size()));
        HIPCHK(s->d_model.upload(mdl.data(), mdl.size()));
    }
    const size_t E = (size_t)s->n_env;
    HIPCHK(s->d_root.alloc(E * 13)); HIPCHK(hipMemset(s->d_root.p, 0, E * 13 * 4));
    HIPCHK(s->d_dof.alloc(E * EMLOCO_NDOF * 2)); HIPCHK(hipMemset(s->d_dof.p, 0, E * EMLOCO_NDOF * 2 * 4));
    HIPCHK(s->d_tgt.alloc(E * EMLOCO_NDOF)); HIPCHK(hipMemset(s->d_tgt.p, 0, E * EMLOCO_NDOF * 4));
    HIPCHK(s->d_rb.alloc(E * EMLOCO_NB * 13)); HIPCHK(hipMemset(s->d_rb.p, 0, E * EMLOCO_NB * 13 * 4));
    HIPCHK(s->d_cf.alloc(E * EMLOCO_NB * 3)); HIPCHK(hipMemset(s->d_cf.p, 0, E * EMLOCO_NB * 3 * 4));
    HIPCHK(s->d_df.alloc(E * EMLOCO_NDOF)); HIPCHK(hipMemset(s->d_df.p, 0, E * EMLOCO_NDOF * 4));
    HIPCHK(s->d_lws.alloc(E * EMLOCO_MAXCAND * 3)); HIPCHK(hipMemset(s->d_lws.p, 0, E * EMLOCO_MAXCAND * 3 * 4));
    hipLaunchKernelGGL(fill_quat_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, 0, s->d_root.p, s->n_env);
    HIPCHK(hipGetLastError());
    EmlocoSimDev &d = s->dev;
    d.n_env = s->n_env; d.n_cand = t.n_cand; d.max_depth = t.max_depth;
    d.topo = s->d_topo.p; d.model = s->d_model.p;
    d.root_state = s->d_root.p; d.dof_state = s->d_dof.p; d.pd_target = s->d_tgt.p;
    d.rb_state = s->d_rb.p; d.contact_force = s->d_cf.p; d.dof_force = s->d_df.p; d.lambda_ws = s->d_lws.p;
    HIPCHK(hipHostMalloc((void **)&s->h_err, sizeof(unsigned), hipHostMallocMapped));
    *s->h_err = 0u;
    HIPCHK(hipHostGetDevicePointer((void **)&d.err, s->h_err, 0));
    d.part_spin_max = s->part_spin_max; d.part_poison = -1;
    if (const char *pad = getenv("EMLOCO_SIM_LDS_PAD")) s->lds_pad = atoi(pad);
    d.sc_n = 0;
    if (sc_on) {
        d.sc_n = (int)(s->h_sc_pairs.size() / 2);
        d.sc_nseg = s->sc_nseg;
        d.sc_k = s->sc_k; d.sc_c = s->sc_c; d.sc_max_pen = s->sc_max_pen; d.sc_mu = s->sc_mu;
    }
    d.hf = nullptr; d.hf_mv = nullptr;
    if (!s->h_hf.empty()) {
        HIPCHK(s->d_hf.upload(s->h_hf.data(), s->h_hf.size()));
        d.hf = s->d_hf.p; d.hf_nx = s->hf_nx; d.hf_ny = s->hf_ny;
        d.hf_hs = s->hf_hs; d.hf_inv_hs = 1.0f / s->hf_hs; d.hf_vs = s->hf_vs; d.hf_ox = s->hf_ox; d.hf_oy = s->hf_oy; d.hf_pad_ = 0.0f;
        d.hf_mv = nullptr;
        if (s->h_hf_mv.size() == s->h_hf.size()) {
            HIPCHK(s->d_hf_mv.upload(s->h_hf_mv.data(), s->h_hf_mv.size()));
            d.hf_mv = s->d_hf_mv.p;
        }
    }
    HIPCHK(hipDeviceSynchronize());
    s->prepared = true;
    return EMLOCO_OK;
}

int emloco_sim_get_params(EmlocoSim *s, EmlocoSimParams *out) {
    if (!s || !out) return fail(EMLOCO_E_ARG, "emloco_sim_get_params: null argument");
    *out = s->prm;
    return EMLOCO_OK;
}

int emloco_sim_set_params(EmlocoSim *s, const EmlocoSimParams *in) {
    if (!s || !in) return fail(EMLOCO_E_ARG, "emloco_sim_set_params: null argument");
    if (in->n_sub < 1 || in->h <= 0.0f || in->n_iter < 0) return fail(EMLOCO_E_ARG, "emloco_sim_set_params: invalid parameters");
    if (in->drive_mode < 0 || in->drive_mode > 1) return fail(EMLOCO_E_ARG, "emloco_sim_set_params: drive_mode 0 | 1 required");
    // the drive mode is a property of what was uploaded last (emloco_sim_set_pd_targets / emloco_sim_set_dof_actuation_force:
    // both travel in one buffer), not of the solver parameters: a parameter update keeps the live mode
    const int live_mode = s->prm.drive_mode;
    s->prm = *in;
    if (s->prepared) s->prm.drive_mode = live_mode;
    return EMLOCO_OK;
}

int emloco_sim_tensor(EmlocoSim *s, int kind, void **dev_ptr, int64_t shape[2]) {
    if (!s || !dev_ptr || !shape) return fail(EMLOCO_E_ARG, "emloco_sim_tensor: null argument");
    if (!s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_tensor: call emloco_sim_prepare first");
    const int64_t E = s->n_env;
    switch (kind) {
        case EMLOCO_T_ROOT_STATE: *dev_ptr = s->d_root.p; shape[0] = E; shape[1] = 13; break;
        case EMLOCO_T_DOF_STATE: *dev_ptr = s->d_dof.p; shape[0] = E * EMLOCO_NDOF; shape[1] = 2; break;
        case EMLOCO_T_RIGID_BODY: *dev_ptr = s->d_rb.p; shape[0] = E * EMLOCO_NB; shape[1] = 13; break;
        case EMLOCO_T_CONTACT_FORCE: *dev_ptr = s->d_cf.p; shape[0] = E * EMLOCO_NB; shape[1] = 3; break;
        case EMLOCO_T_DOF_FORCE: *dev_ptr = s->d_df.p; shape[0] = E * EMLOCO_NDOF; shape[1] = 1; break;
        case EMLOCO_T_PD_TARGET: *dev_ptr = s->d_tgt.p; shape[0] = E; shape[1] = EMLOCO_NDOF; break;
        case EMLOCO_T_WARM_START: *dev_ptr = s->d_lws.p; shape[0] = E; shape[1] = EMLOCO_MAXCAND * 3; break;
        default: return fail(EMLOCO_E_ARG, "emloco_sim_tensor: unknown tensor kind");
    }
    return EMLOCO_OK;
}

int emloco_sim_set_pd_targets(EmlocoSim *s, const float *dev_targets, void *stream) {
    if (!s || !dev_targets) return fail(EMLOCO_E_ARG, "emloco_sim_set_pd_targets: null argument");
    if (!s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_set_pd_targets: sim not prepared");
    if (dev_targets != s->d_tgt.p)
        HIPCHK(hipMemcpyAsync(s->d_tgt.p, dev_targets, (size_t)s->n_env * EMLOCO_NDOF * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    s->prm.drive_mode = 0;
    return EMLOCO_OK;
}

int emloco_sim_set_dof_actuation_force(EmlocoSim *s, const float *dev_forces, void *stream) {
    if (!s || !dev_forces) return fail(EMLOCO_E_ARG, "emloco_sim_set_dof_actuation_force: null argument");
    if (!s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_set_dof_actuation_force: sim not prepared");
    // the torques travel in the buffer the position targets travel in (one of the two is live, EmlocoSimParams::drive_mode says which)
    if (dev_forces != s->d_tgt.p)
        HIPCHK(hipMemcpyAsync(s->d_tgt.p, dev_forces, (size_t)s->n_env * EMLOCO_NDOF * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    s->prm.drive_mode = 1;
    return EMLOCO_OK;
}

// the dispatch order of the full launch: sorted on the caller's stream right ahead of it (a stream of the simulator's own
// for the sort was measured and lost: with torch's pool streams it ended up sharing a hardware queue with the caller's)
static int launch_order(EmlocoSim *s, hipStream_t st) {
    if (s->order_ready) { s->order_ready = false; return EMLOCO_OK; }       // sorted earlier in the chain, behind the previous step
    hipLaunchKernelGGL(emloco::sim_order_kernel, dim3(1), dim3(1024), 0, st, s->d_ticks.p, s->n_env, s->d_order.p, s->d_order_ws.p);
    HIPCHK(hipGetLastError());
    return EMLOCO_OK;
}

int emloco_sim_step(EmlocoSim *s, int n_calls, void *stream) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_step: null sim");
    if (!s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_step: sim not prepared");
    if (n_calls < 1) return fail(EMLOCO_E_ARG, "emloco_sim_step: n_calls < 1");
    if (const int rc = check_device_error(s, "emloco_sim_step")) return rc;
    EmlocoSimParams p = s->prm;
    p.n_sub = s->prm.n_sub * n_calls;
    hipStream_t st = (hipStream_t)stream;
    EmlocoSimDev d = s->dev;
    if (s->cost_order) {                     // longest-first dispatch order from the durations of the previous launch
        const int rc = launch_order(s, st);
        if (rc != EMLOCO_OK) return rc;
        d.step_order = s->d_order.p;
        d.step_ticks = s->d_ticks.p;
    }
    const int slot = s->ev_head;
    const bool timed = s->timing && (s->launch_no++ % s->timing_stride) == 0;
    if (timed) HIPCHK(hipEventRecord(s->ev0[slot], st));
    d.n_parts = s->n_parts < p.n_sub ? s->n_parts : p.n_sub;       // at least one substep per part
    d.part_seq = ++s->part_seq; d.part_state = s->d_part_state.p; d.part_flag = s->d_part_flag.p;
    d.part_spin_max = s->part_spin_max; d.part_poison = s->part_poison;
    d.n_slots = s->n_env;
    const unsigned grid = (unsigned)(((s->n_env + EMLOCO_SIM_ENVS_PER_WG - 1) / EMLOCO_SIM_ENVS_PER_WG) * d.n_parts);       // one 64-lane workgroup per env (pair) and part
    if (d.hf) hipLaunchKernelGGL(emloco::sim_step_kernel<1>, dim3(grid), dim3(64), (size_t)s->lds_pad, st, p, d);
    else hipLaunchKernelGGL(emloco::sim_step_kernel<0>, dim3(grid), dim3(64), (size_t)s->lds_pad, st, p, d);
    HIPCHK(hipGetLastError());
    if (timed) {
        HIPCHK(hipEventRecord(s->ev1[slot], st));
        s->ev_head = (slot + 1) % EmlocoSim::kRing;
        if (s->ev_count < EmlocoSim::kRing) ++s->ev_count;
    }
    return EMLOCO_OK;
}

int emloco_sim_step_subset(EmlocoSim *s, int n_calls, const int64_t *dev_skip, const int32_t *dev_ids, int n_ids, void *stream) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_step_subset: null sim");
    if (!s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_step_subset: sim not prepared");
    if (n_calls < 1) return fail(EMLOCO_E_ARG, "emloco_sim_step_subset: n_calls < 1");
    if ((dev_skip == nullptr) == (dev_ids == nullptr)) return fail(EMLOCO_E_ARG, "emloco_sim_step_subset: exactly one of skip flags / id list");
    if (dev_ids && (n_ids < 0 || n_ids > s->n_env)) return fail(EMLOCO_E_ARG, "emloco_sim_step_subset: bad id count");
    if (dev_ids && n_ids == 0) return EMLOCO_OK;
    if (const int rc = check_device_error(s, "emloco_sim_step_subset")) return rc;
    EmlocoSimParams p = s->prm;
    p.n_sub = s->prm.n_sub * n_calls;
    EmlocoSimDev d = s->dev;
    d.step_skip = (const long long *)dev_skip;
    d.step_ids = (const int *)dev_ids;
    hipStream_t st = (hipStream_t)stream;
    if (s->cost_order) {
        d.step_ticks = s->d_ticks.p;
        if (dev_skip) {
            const int rc = launch_order(s, st);
            if (rc != EMLOCO_OK) return rc;
            d.step_order = s->d_order.p;
        }
    }
    // the launch over everything but the flagged envs is the one the timing log follows (it stands where emloco_sim_step stood)
    const bool timed = s->timing && dev_skip && (s->launch_no++ % s->timing_stride) == 0;
    const int slot = s->ev_head;
    if (timed) HIPCHK(hipEventRecord(s->ev0[slot], st));
    d.n_parts = dev_ids ? 1 : (s->n_parts < p.n_sub ? s->n_parts : p.n_sub);      // the list launch (a few dozen envs) is not split
    d.part_seq = ++s->part_seq; d.part_state = s->d_part_state.p; d.part_flag = s->d_part_flag.p;
    d.part_spin_max = s->part_spin_max; d.part_poison = s->part_poison;
    d.n_slots = dev_ids ? n_ids : s->n_env;
    const unsigned grid = (unsigned)(((d.n_slots + EMLOCO_SIM_ENVS_PER_WG - 1) / EMLOCO_SIM_ENVS_PER_WG) * d.n_parts);      // one 64-lane workgroup per slot (pair) and part
    if (d.hf) hipLaunchKernelGGL(emloco::sim_step_kernel<1>, dim3(grid), dim3(64), (size_t)s->lds_pad, st, p, d);
    else hipLaunchKernelGGL(emloco::sim_step_kernel<0>, dim3(grid), dim3(64), (size_t)s->lds_pad, st, p, d);
    HIPCHK(hipGetLastError());
    if (timed) {
        HIPCHK(hipEventRecord(s->ev1[slot], st));
        s->ev_head = (slot + 1) % EmlocoSim::kRing;
        if (s->ev_count < EmlocoSim::kRing) ++s->ev_count;
    }
    return EMLOCO_OK;
}

int emloco_sim_set_split(EmlocoSim *s, int n_parts) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_set_split: null sim");
    if (!s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_set_split: sim not prepared");
    if (n_parts < 1 || n_parts > 4) return fail(EMLOCO_E_ARG, "emloco_sim_set_split: 1 to 4 parts");
    if (n_parts > 1 && !s->d_part_state.p) {
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(s->d_part_state.alloc((size_t)s->n_env * EMLOCO_PART_WORDS));
        HIPCHK(s->d_part_flag.alloc((size_t)s->n_env));
        HIPCHK(hipMemset(s->d_part_flag.p, 0, sizeof(unsigned) * (size_t)s->n_env));
    }
    s->n_parts = n_parts;
    return EMLOCO_OK;
}

int emloco_sim_set_cost_order(EmlocoSim *s, int on) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_set_cost_order: null sim");
    if (!s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_set_cost_order: sim not prepared");
    if (on && !s->d_ticks.p) {
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(s->d_ticks.alloc((size_t)s->n_env));
        HIPCHK(s->d_order.alloc((size_t)s->n_env));
        if (s->n_env > EMLOCO_ORDER_LDS_ENVS) HIPCHK(s->d_order_ws.alloc((size_t)s->n_env));
        HIPCHK(hipMemset(s->d_ticks.p, 0, sizeof(unsigned) * (size_t)s->n_env));
    }
    s->cost_order = on != 0;
    return EMLOCO_OK;
}

int emloco_sim_sync(EmlocoSim *s, void *stream) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_sync: null sim");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return check_device_error(s, "emloco_sim_sync");
}

// fault injection (include/emloco_sim.h): the first part of `env` withholds its hand-over flag in the split launches that
// follow (-1: back to normal) and a part's wait is bounded by `spin_max` sleeps (<= 0: the default) -- lets a test see the
// timeout of a lost hand-over surface as EMLOCO_E_HIP instead of waiting seconds for it
int emloco_sim_debug_poison_part(EmlocoSim *s, int env, int spin_max) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_debug_poison_part: null sim");
    s->part_poison = env;
    s->part_spin_max = spin_max > 0 ? spin_max : (1 << 22);
    return EMLOCO_OK;
}

static int set_indexed(EmlocoSim *s, const float *dev_full, float *own, int row_len, const int32_t *ids, int n, void *stream) {
    if (!s || !dev_full || (!ids && n > 0)) return fail(EMLOCO_E_ARG, "set_*_indexed: null argument");
    if (!s->prepared) return fail(EMLOCO_E_STATE, "set_*_indexed: sim not prepared");
    if (n < 0 || n > s->n_env) return fail(EMLOCO_E_ARG, "set_*_indexed: bad count");
    if (n == 0) return EMLOCO_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dev_full != own) {
        hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)n), dim3(64), 0, st, dev_full, own, (const int *)ids, n, row_len);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(emloco::sim_fk_kernel, dim3((unsigned)(n > 256 ? 256 : n)), dim3(64), 0, st, s->dev, (const int *)ids, n);
    HIPCHK(hipGetLastError());
    return EMLOCO_OK;
}

int emloco_sim_set_root_state_indexed(EmlocoSim *s, const float *dev_full, const int32_t *ids, int n, void *stream) {
    return set_indexed(s, dev_full, s ? s->d_root.p : nullptr, 13, ids, n, stream);
}

int emloco_sim_set_dof_state_indexed(EmlocoSim *s, const float *dev_full, const int32_t *ids, int n, void *stream) {
    return set_indexed(s, dev_full, s ? s->d_dof.p : nullptr, EMLOCO_NDOF * 2, ids, n, stream);
}

int emloco_sim_refresh_bodies(EmlocoSim *s, void *stream) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_refresh_bodies: null sim");
    if (!s->prepared) return fail(EMLOCO_E_STATE, "emloco_sim_refresh_bodies: sim not prepared");
    hipLaunchKernelGGL(emloco::sim_fk_kernel, dim3((unsigned)s->n_env), dim3(64), 0, (hipStream_t)stream, s->dev, (const int *)nullptr, s->n_env);
    HIPCHK(hipGetLastError());
    return EMLOCO_OK;
}

// internal (not in the public header): forward kinematics of the listed envs, used by emloco_task_reset
int emloco_sim_fk_indexed(EmlocoSim *s, const int32_t *ids, int n, void *stream) {
    if (!s || !s->prepared || !ids || n < 1) return fail(EMLOCO_E_ARG, "emloco_sim_fk_indexed: bad argument");
    hipLaunchKernelGGL(emloco::sim_fk_kernel, dim3((unsigned)(n > 256 ? 256 : n)), dim3(64), 0, (hipStream_t)stream, s->dev, (const int *)ids, n);
    HIPCHK(hipGetLastError());
    return EMLOCO_OK;
}

// internal diagnostic: the per-env durations (100 MHz ticks) the cost-ordered dispatch sorts by and, from the second call on,
// the wall clock at which each env's workgroup started in the latest launch
int emloco_sim_cost_ticks(EmlocoSim *s, unsigned *host_ticks, unsigned long long *host_start, int n) {
    if (!s || !host_ticks || n < 1 || !s->d_ticks.p) return fail(EMLOCO_E_ARG, "emloco_sim_cost_ticks: bad argument / cost order off");
    HIPCHK(hipDeviceSynchronize());
    const size_t m = (size_t)(n < s->n_env ? n : s->n_env);
    HIPCHK(hipMemcpy(host_ticks, s->d_ticks.p, sizeof(unsigned) * m, hipMemcpyDeviceToHost));
    if (host_start) {
        if (!s->dev.step_start) {
            unsigned long long *p = nullptr;
            HIPCHK(hipMalloc((void **)&p, sizeof(unsigned long long) * 2 * (size_t)s->n_env));
            HIPCHK(hipMemset(p, 0, sizeof(unsigned long long) * 2 * (size_t)s->n_env));
            s->dev.step_start = p;
        }
        HIPCHK(hipMemcpy(host_start, s->dev.step_start, sizeof(unsigned long long) * 2 * (size_t)s->n_env, hipMemcpyDeviceToHost));
    }
    return EMLOCO_OK;
}

// internal profiling hook (kernels built with -DEMLOCO_SIM_PROFILE): cycle stamps of env 0, 16 per substep
int emloco_sim_profile(EmlocoSim *s, long long *host_out, int n) {
    if (!s || !host_out || n < 1) return fail(EMLOCO_E_ARG, "emloco_sim_profile: bad argument");
    if (!s->dev.prof) {
        long long *p = nullptr;
        HIPCHK(hipMalloc((void **)&p, 16 * 16 * sizeof(long long)));
        HIPCHK(hipMemset(p, 0, 16 * 16 * sizeof(long long)));
        s->dev.prof = p;
        return EMLOCO_OK;
    }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host_out, s->dev.prof, sizeof(long long) * (n < 256 ? n : 256), hipMemcpyDeviceToHost));
    return EMLOCO_OK;
}

int emloco_sim_num_candidates(EmlocoSim *s) { return s ? s->topo.n_cand : EMLOCO_E_ARG; }

int emloco_sim_enable_timing(EmlocoSim *s, int on) {
    if (!s) return fail(EMLOCO_E_ARG, "emloco_sim_enable_timing: null sim");
    if (on && s->ev0.empty()) {
        HIPCHK(hipSetDevice(s->device));
        s->ev0.resize(EmlocoSim::kRing); s->ev1.resize(EmlocoSim::kRing);
        for (int i = 0; i < EmlocoSim::kRing; ++i) { HIPCHK(hipEventCreate(&s->ev0[i])); HIPCHK(hipEventCreate(&s->ev1[i])); }
    }
    s->timing = on != 0;
    s->timing_stride = on > 1 ? on : 1;
    s->launch_no = 0;
    s->ev_head = 0; s->ev_count = 0;
    return EMLOCO_OK;
}

int emloco_sim_timing_stats(EmlocoSim *s, int *n_launches, float *total_ms) {
    if (!s || !n_launches || !total_ms) return fail(EMLOCO_E_ARG, "emloco_sim_timing_stats: null argument");
    *n_launches = 0; *total_ms = 0.0f;
    for (int k = 0; k < s->ev_count; ++k) {
        const int slot = (s->ev_head - 1 - k + 2 * EmlocoSim::kRing) % EmlocoSim::kRing;
        HIPCHK(hipEventSynchronize(s->ev1[slot]));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, s->ev0[slot], s->ev1[slot]));
        if (k == 0) s->last_ms = ms;
        *total_ms += ms; ++*n_launches;
    }
    s->ev_count = 0;
    return EMLOCO_OK;
}

float emloco_sim_last_step_ms(EmlocoSim *s) {
    if (!s) return -1.0f;
    if (s->ev_count > 0) {
        const int slot = (s->ev_head - 1 + EmlocoSim::kRing) % EmlocoSim::kRing;
        if (hipEventSynchronize(s->ev1[slot]) == hipSuccess) {
            float ms = -1.0f;
            if (hipEventElapsedTime(&ms, s->ev0[slot], s->ev1[slot]) == hipSuccess) s->last_ms = ms;
        }
    }
    return s->last_ms;
}

}  // extern "C"
