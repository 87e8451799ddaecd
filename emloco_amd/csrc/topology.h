// topology.h -- host-side derivation of the per-model lookup tables the rollout kernels read:
// body depths, child lists (descending body index, the order the articulated-body passes add them),
// lowest-common-ancestor depths (prefix length of the Gram-form contact matrix) and the fixed
// enumeration of ground-contact candidates (sphere: centre; capsule: two end centres; box: 8 corners).
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/emloco_sim.h"

namespace emloco {

struct Topology {
    std::vector<int32_t> parent, depth, children, geom_type, cand_body, cand_k;
    std::vector<int32_t> pd_pack;   // per body: parent (the root: 31) | depth << 5 | index among the bodies of its depth << 9 | children << 12, 17, 22 (31: none)
    std::vector<uint8_t> lca_depth;
    int max_depth = 0;
    int n_cand = 0;

    // returns false when the tree is not in depth-first order, a body has more than 3 children,
    // or the candidates do not fit the warm-start storage
    bool build(const int32_t *par, const int32_t *gtype) {
        const int nb = EMLOCO_NB;
        parent.assign(par, par + nb);
        geom_type.assign(gtype, gtype + nb);
        depth.assign(nb, 0);
        children.assign(nb * 3, -1);
        if (parent[0] != -1) return false;
        for (int i = 1; i < nb; ++i) {
            if (parent[i] < 0 || parent[i] >= i) return false;
            depth[i] = depth[parent[i]] + 1;
            if (depth[i] > max_depth) max_depth = depth[i];
        }
        if (max_depth > 8) return false;
        for (int i = nb - 1; i >= 1; --i) {   // descending, so each list is in descending order
            int *c = &children[parent[i] * 3];
            int k = 0;
            while (k < 3 && c[k] >= 0) ++k;
            if (k == 3) return false;
            c[k] = i;
        }
        pd_pack.assign(nb, 0);
        for (int i = 0; i < nb; ++i) {
            int lslot = 0;
            for (int j = 0; j < i; ++j) lslot += depth[j] == depth[i];
            if (lslot > 6) return false;                          // the Gram build keeps slot + 1 in three bits
            pd_pack[i] = (parent[i] < 0 ? 31 : parent[i]) | (depth[i] << 5) | (lslot << 9);
            for (int k = 0; k < 3; ++k) pd_pack[i] |= (children[i * 3 + k] < 0 ? 31 : children[i * 3 + k]) << (12 + 5 * k);
        }
        lca_depth.assign(nb * nb, 0);
        for (int a = 0; a < nb; ++a)
            for (int b = 0; b < nb; ++b) {
                int x = a, y = b;
                while (depth[x] > depth[y]) x = parent[x];
                while (depth[y] > depth[x]) y = parent[y];
                while (x != y) { x = parent[x]; y = parent[y]; }
                lca_depth[a * nb + b] = (uint8_t)depth[x];
            }
        cand_body.clear(); cand_k.clear();
        for (int b = 0; b < nb; ++b) {
            const int nk = gtype[b] == EMLOCO_GEOM_SPHERE ? 1 : (gtype[b] == EMLOCO_GEOM_CAPSULE ? 2 : 8);
            for (int k = 0; k < nk; ++k) { cand_body.push_back(b); cand_k.push_back(k); }
        }
        n_cand = (int)cand_body.size();
        return n_cand <= EMLOCO_MAXCAND;
    }
};

}  // namespace emloco
