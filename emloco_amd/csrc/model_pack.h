// model_pack.h -- host side: lays the topology tables and the per-env models out in the two blocks the rollout kernels read
// (emloco_types.h: EMLOCO_TOPO_*, EMLOCO_MB_*).  Shared by the C-ABI layer (sim_capi.hip) and the CPU emulation harness of the tests.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include "emloco_types.h"
#include "topology.h"

namespace emloco {

inline std::vector<int32_t> pack_topology(const Topology &t, const unsigned char *sc_pairs, int sc_n, const unsigned char *seg_body = nullptr) {
    std::vector<int32_t> o(EMLOCO_TOPO_WORDS, 0);
    for (int b = 0; b < EMLOCO_NB; ++b) {
        o[EMLOCO_TOPO_PARENT + b] = t.parent[b];
        o[EMLOCO_TOPO_DEPTH + b] = t.depth[b];
        for (int k = 0; k < 3; ++k) o[EMLOCO_TOPO_CHILD + b * 3 + k] = t.children[b * 3 + k];
        o[EMLOCO_TOPO_PDPACK + b] = t.pd_pack[b];
    }
    for (int c = 0; c < t.n_cand; ++c)
        o[EMLOCO_TOPO_CAND + c] = t.cand_body[c] | (t.cand_k[c] << 8) | (t.geom_type[t.cand_body[c]] << 16);
    for (int q = 0; q < sc_n; ++q) {
        const int si = sc_pairs[2 * q], sj = sc_pairs[2 * q + 1];
        const int bi = seg_body ? seg_body[si] : si, bj = seg_body ? seg_body[sj] : sj;
        o[EMLOCO_TOPO_SCPAIR + q] = si | (sj << 8) | (bi << 16) | (bj << 24);
    }
    return o;
}

// all arrays [n_env][24][k] / [n_env][69] as EmlocoModelDesc / EmlocoSelfCollisionDesc hold them; the capsule arrays may be NULL
inline std::vector<float> pack_models(int n_env, const float *joint_off, const float *mass, const float *com, const float *inertia,
                                      const float *geom_a, const float *geom_b, const float *geom_r, const float *kp, const float *kd,
                                      const float *armature, const float *effort, const float *cap_a, const float *cap_b, const float *cap_r,
                                      int n_seg = EMLOCO_NB, const unsigned char *seg_body = nullptr) {
    std::vector<float> o((size_t)n_env * EMLOCO_MODEL_WORDS, 0.0f);
    for (int e = 0; e < n_env; ++e) {
        float *m = o.data() + (size_t)e * EMLOCO_MODEL_WORDS;
        for (int b = 0; b < EMLOCO_NB; ++b) {
            const size_t eb = (size_t)e * EMLOCO_NB + b;
            float *r = m + EMLOCO_MB_DYN + b * 16;
            for (int k = 0; k < 3; ++k) { r[k] = joint_off[eb * 3 + k]; r[4 + k] = com[eb * 3 + k]; }
            r[3] = mass[eb];
            for (int k = 0; k < 4; ++k) r[8 + k] = inertia[eb * 6 + k];
            r[12] = inertia[eb * 6 + 4]; r[13] = inertia[eb * 6 + 5];
            float *g = m + EMLOCO_MB_GEO + b * 8;
            for (int k = 0; k < 3; ++k) { g[k] = geom_a[eb * 3 + k]; g[4 + k] = geom_b[eb * 3 + k]; }
            g[3] = geom_r[eb];
        }
        if (cap_a)
            for (int sg = 0; sg < n_seg; ++sg) {                 // collision segments: [n_env][n_seg]
                const size_t es = (size_t)e * n_seg + sg;
                float *c = m + EMLOCO_MB_CAP + sg * 8;
                for (int k = 0; k < 3; ++k) { c[k] = cap_a[es * 3 + k]; c[4 + k] = cap_b[es * 3 + k]; }
                c[3] = cap_r[es];
                c[7] = (float)(seg_body ? seg_body[sg] : sg);
            }
        for (int d = 0; d < EMLOCO_NDOF; ++d) {
            const size_t ed = (size_t)e * EMLOCO_NDOF + d;
            float *r = m + EMLOCO_MB_DRV + d * 4;
            r[0] = kp[ed]; r[1] = kd[ed]; r[2] = armature[ed]; r[3] = effort[ed];
        }
    }
    return o;
}

}  // namespace emloco
