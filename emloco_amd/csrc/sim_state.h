// sim_state.h -- internal: the simulator object behind the opaque EmlocoSim handle (shared by the C-ABI units).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "emloco_types.h"
#include "topology.h"

template <class T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    hipError_t alloc(size_t count) { n = count; return hipMalloc((void **)&p, count * sizeof(T)); }
    hipError_t upload(const T *src, size_t count) {
        hipError_t e = alloc(count);
        if (e != hipSuccess) return e;
        return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};


struct EmlocoSim {
    int device = 0;
    EmlocoSimParams prm{};
    bool have_model = false, prepared = false, timing = false;
    int n_env = 0;
    emloco::Topology topo;
    // host copies of the model until prepare()
    std::vector<float> h_off, h_mass, h_com, h_inertia, h_ga, h_gb, h_gr, h_kp, h_kd, h_arm, h_eff;
    // optional self-collision description (host copies until prepare())
    std::vector<unsigned char> h_sc_pairs, h_sc_segbody;
    int sc_nseg = EMLOCO_NB;
    std::vector<float> h_sc_a, h_sc_b, h_sc_r;
    float sc_k = 0.0f, sc_c = 0.0f, sc_max_pen = 0.0f, sc_mu = 0.0f;
    // optional height-field ground (host copy until prepare())
    std::vector<short> h_hf;
    std::vector<uint8_t> h_hf_mv;      // packed vertex moves of the slope-corrected mesh (emloco_types.h: hf_mv), empty: none
    int hf_nx = 0, hf_ny = 0;
    float hf_hs = 0.0f, hf_vs = 0.0f, hf_ox = 0.0f, hf_oy = 0.0f;
    DevBuf<short> d_hf;
    DevBuf<unsigned char> d_hf_mv;
    // device
    DevBuf<int> d_topo;                        // EMLOCO_TOPO_* tables
    DevBuf<float> d_model;                     // EMLOCO_MB_* records, one block per env
    DevBuf<float> d_root, d_dof, d_tgt, d_rb, d_cf, d_df, d_lws;
    EmlocoSimDev dev{};
    // cost-ordered dispatch of the full launch (emloco_sim_set_cost_order): per-env duration of the last step, env ids sorted by it
    int n_parts = 1;                           // split launch (emloco_sim_set_split)
    unsigned part_seq = 0;
    DevBuf<float> d_part_state;
    DevBuf<unsigned> d_part_flag;
    unsigned *h_err = nullptr;                 // device error word in pinned host memory (kernels raise bits with system-scope atomics)
    int part_spin_max = 1 << 22, part_poison = -1;
    int lds_pad = 0;                           // diagnostic (EMLOCO_SIM_LDS_PAD): extra dynamic LDS per workgroup of the step launch, caps the residency
    bool cost_order = false;
    bool order_ready = false;                   // d_order already holds the next full launch's order (emloco_task_compact_done_order sorted it)
    DevBuf<unsigned> d_ticks;
    DevBuf<int> d_order;
    DevBuf<unsigned char> d_order_ws;           // bucket of every env, beyond the 16384 envs the sort keeps in LDS
    // HIP-event timing of step launches: a ring of event pairs recorded on the launch stream
    static constexpr int kRing = 1024;
    std::vector<hipEvent_t> ev0, ev1;
    int ev_head = 0, ev_count = 0;
    int timing_stride = 1, launch_no = 0;      // every timing_stride-th step launch is timed (the event packets cost ~5 us each)
    float last_ms = -1.0f;
};

