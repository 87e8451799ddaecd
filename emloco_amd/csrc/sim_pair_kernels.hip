// sim_pair_kernels.hip -- rigid-body step of the vectorised humanoid rollout for gfx950 (MI355X), TWO envs per wave (round 6,
// experimental: built instead of sim_kernels.hip with -DEMLOCO_SIM_PAIR=1; see DESIGN.md section 5 for what it measured).
//
// Stands where the reference calls gym.simulate (pacer/pacer/env/tasks/base_task.py:792-797; engine
// parameters pacer/pacer/utils/config.py:143-163, pacer/pacer/data/cfg/pacer.yaml:93-104).  The reference's
// engine (PhysX 5) is absent; the scheme is this repo's own:
// articulated-body dynamics with implicit PD drives + maximal-coordinate ground contacts solved by
// projected Gauss-Seidel on the Gram-form contact matrix.
//
// Mapping (round 6): ONE 64-lane wave per env PAIR (workgroup = 1 wave, so __syncthreads() is free of cross-wave
// waits); the substeps of an env.step are fused in one launch (or handed from workgroup to workgroup, emloco_sim_set_split)
// and the envs' state lives in LDS / registers between them.  Lane roles change per phase:
//   lane = 32 x env + body (2 x 24 active)  kinematics, inertia, bias forces, articulated-body passes (level-synchronous), BOTH envs
//   lane = candidate (<=128, 2/lane)        ground-contact detection, wave ballot + popcount compaction   } one env wide, the two
//   lane = contact row (<=60)               chain propagation, contact-matrix column, Gauss-Seidel multiplier } envs one after the other
// Gauss-Seidel row products are reduced on the DPP crossbar in a fixed association order (wave_sum), which is
// also the order the CPU oracle uses, so multipliers agree bit for bit.
//
// HBM traffic per env.step is ~9 KB (state in/out + per-env model), the kernel is latency/occupancy
// bound, not bandwidth bound.
#include <hip/hip_runtime.h>
#include "dev_math.h"
#include "sim_math.h"
// the rigid-body kernels use the fused helper set (sim_math.h); undone at the end of this file
#define cross3 fcross3
#define dot3 fdot3
#define dot6 fdot6
#define qmul fqmul
#define qnormalize fqnormalize
#define q2mat fq2mat
#define matvec3 fmatvec3
#define rotvec2quat frotvec2quat
#define quat2rotvec fquat2rotvec
#include "emloco_types.h"
#include "fk_device.h"
#include "order_device.h"

namespace emloco {

#define NB EMLOCO_NB
#define NDOF EMLOCO_NDOF
#define MAXC EMLOCO_MAXC
#define MAXR (3 * EMLOCO_MAXC)
#define MAXCAND EMLOCO_MAXCAND
#define EMLOCO_WH_MAX 1.0f /* largest link rotation per substep [rad]: 120 rad/s at h = 1/120 -- above the asset's max_angular_velocity = 100
                            * (humanoid.py:685-688), which is therefore the cap that binds (phase 1; 0.4 = 48 rad/s before the angular-momentum balance) */
#define YLEN 30 /* chain-propagation vector: 6 root + 3 per tree level (depth <= 8) */

// index into a packed symmetric 6x6 (upper triangle, row-major): (a<=b)
__device__ __forceinline__ constexpr int sidx(int a, int b) {
    return a <= b ? (a * (13 - a)) / 2 + (b - a) : (b * (13 - b)) / 2 + (a - b);
}

// (a, b) of a symmetric matrix stored as its packed lower triangle, branch-free
__device__ __forceinline__ int tri_index(int a, int b) {
    const int hi = a > b ? a : b, lo = a > b ? b : a;
    return ((hi * (hi + 1)) >> 1) + lo;
}

// Hand-over of a split launch (emloco_sim_set_split): 16-byte granules written with `sc1` (write-through) stores and read
// with `sc1` loads -- coherent across the XCDs' L2s and the CUs' L1s access by access (MI355X_MICROARCH.md, inter-workgroup
// visibility: "sc1 stores and loads both sides") -- and a flag word (relaxed agent-scope atomic) that goes out once the
// stores have completed (s_waitcnt vmcnt(0)).  Measured alternatives: agent-scope fences on every lane (__threadfence) write
// back / invalidate whole caches per workgroup: launch 0.54 -> 0.68 ms; one 4-byte agent atomic per word: as fast as this,
// but every word is a fabric write of its own (HBM counters 46 -> 124 MB per launch).
typedef float part_f4 __attribute__((ext_vector_type(4)));
#ifdef EMLOCO_EMU
__device__ __forceinline__ void part_st16(float *p, float a, float b, float c, float d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
__device__ __forceinline__ void part_ld16(const float *p, float *v) { for (int k = 0; k < 4; ++k) v[k] = p[k]; }
__device__ __forceinline__ void part_ld48(const float *p, float *v) { for (int g = 0; g < 3; ++g) for (int k = 0; k < 4; ++k) v[4 * g + k] = p[g * 4 * EMLOCO_NB + k]; }
__device__ __forceinline__ void part_stores_done() {}
__device__ __forceinline__ void part_flag_set(unsigned *f, unsigned v) { *f = v; }
__device__ __forceinline__ unsigned part_flag_get(const unsigned *f) { return *f; }
__device__ __forceinline__ void part_err_raise(unsigned *e, unsigned bit) { if (e) *e |= bit; }
#else
__device__ __forceinline__ void part_st16(float *p, float a, float b, float c, float d) {
    const part_f4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void part_ld16(const float *p, float *v) {
    part_f4 a;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(p) : "memory");
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void part_ld48(const float *p, float *v) {
    part_f4 a, b, c;
    static_assert(EMLOCO_NB * 16 == 384, "granule-major hand-over: a lane's three granules are 24 x 16 bytes apart");
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:384 sc1\n\t"
                 "global_load_dwordx4 %2, %3, off offset:768 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(p) : "memory");
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
}
__device__ __forceinline__ void part_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void part_flag_set(unsigned *f, unsigned v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned part_flag_get(const unsigned *f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void part_err_raise(unsigned *e, unsigned bit) { if (e) __hip_atomic_fetch_or(e, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
#endif

// 16-byte load of a model record: base is workgroup-uniform (scalar registers), off a 32-bit lane offset in words
#ifdef EMLOCO_EMU
__device__ __forceinline__ void ld4(const float *base, int off, float *v) { for (int k = 0; k < 4; ++k) v[k] = base[off + k]; }
#else
__device__ __forceinline__ void ld4(const float *base, int off, float *v) {
    const part_f4 a = *(const part_f4 *)(base + off);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
#endif

// pd_pack word (topology.h): parent | depth << 5 | index among the bodies of its depth << 9 | children << 12, 17, 22 (31: none)
#define PD_PARENT(w) ((w) & 31)
#define PD_DEPTH(w) (((w) >> 5) & 15)
#define PD_SLOT(w) (((w) >> 9) & 7)
#define PD_CHILD(w, i) (((w) >> (12 + 5 * (i))) & 31)

#ifdef EMLOCO_SIM_PROFILE
#define PSTAMP(i) do { if (d.prof && env0 == 0 && lane == 0) d.prof[sub * 16 + (i)] = (long long)wall_clock64(); } while (0)
#endif
#ifdef EMLOCO_SIM_PACC
// segment times summed over ALL pairs of the launches since the buffer was cleared (tools/sim_phase_profile.py --pair-all): slot 128 + i
// collects the ticks since the previous PACC of this wave, slot 160 + i how often
#define PACC(i) do { if (d.prof && lane == 0) { const long long t_ = (long long)wall_clock64(); atomicAdd((unsigned long long *)&d.prof[128 + (i)], (unsigned long long)(t_ - pacc_t)); \
                                                atomicAdd((unsigned long long *)&d.prof[160 + (i)], 1ull); pacc_t = t_; } } while (0)
#else
#define PACC(i) do { } while (0)
#endif
#ifndef EMLOCO_SIM_PROFILE
#define PSTAMP(i) do { } while (0)
#endif
#ifndef PAIR_FAST_MAXC
#define PAIR_FAST_MAXC 10   /* contacts per env up to which the row phases of both envs of a pair share the wave (3 x 10 rows <= 32 lanes) */
#endif

// Height-field ground under the world point (cx, cy): height zt of the cell triangle's plane there and its unit normal.
// Cell (i, j) holds the mesh triangles (v00, v10, v11) [u >= v] and (v00, v11, v01) [u < v] (u, v: position in the cell
// along x, y), as terrain_utils.convert_heightfield_to_trimesh lays them out; beyond the map the border cell's plane extends.
// `id` names the triangle that was used: (cell i, cell j, which half)
__device__ __forceinline__ void hf_plane(const EmlocoSimDev &d, float cx, float cy, float &zt, float n[3], int &id) {
    const float gx = (cx - d.hf_ox) * d.hf_inv_hs, gy = (cy - d.hf_oy) * d.hf_inv_hs;
    int i = (int)floorf(gx), j = (int)floorf(gy);
    i = i < 0 ? 0 : (i > d.hf_nx - 2 ? d.hf_nx - 2 : i);
    j = j < 0 ? 0 : (j > d.hf_ny - 2 ? d.hf_ny - 2 : j);
    const float u = gx - (float)i, v = gy - (float)j;
    id = ((i << 15) + j) * 2 + (u >= v ? 1 : 0);
    const short *c = d.hf + (long)i * d.hf_ny + j;
    const float h00 = d.hf_vs * (float)c[0], h01 = d.hf_vs * (float)c[1];
    const float h10 = d.hf_vs * (float)c[d.hf_ny], h11 = d.hf_vs * (float)c[d.hf_ny + 1];
    float zx, zy;
    if (u >= v) { zx = h10 - h00; zy = h11 - h10; } else { zy = h01 - h00; zx = h11 - h01; }
    zt = fmaf(v, zy, fmaf(u, zx, h00));
    const float sx = zx * d.hf_inv_hs, sy = zy * d.hf_inv_hs;
    const float inv = 1.0f / sqrtf(fmaf(sx, sx, fmaf(sy, sy, 1.0f)));
    n[0] = 0.0f - sx * inv; n[1] = 0.0f - sy * inv; n[2] = inv;
}
// the triangle under a point, without its plane
__device__ __forceinline__ int hf_triangle(const EmlocoSimDev &d, float cx, float cy) {
    const float gx = (cx - d.hf_ox) * d.hf_inv_hs, gy = (cy - d.hf_oy) * d.hf_inv_hs;
    int i = (int)floorf(gx), j = (int)floorf(gy);
    i = i < 0 ? 0 : (i > d.hf_nx - 2 ? d.hf_nx - 2 : i);
    j = j < 0 ? 0 : (j > d.hf_ny - 2 ? d.hf_ny - 2 : j);
    const float u = gx - (float)i, v = gy - (float)j;
    return ((i << 15) + j) * 2 + (u >= v ? 1 : 0);
}

// ---- the slope-corrected mesh (terrain_utils.convert_heightfield_to_trimesh with a slope threshold, built by the task at
// humanoid_pedestrain_terrain.py:859-881): where the step between two neighbouring samples exceeds the threshold the LOWER vertex
// sits one cell sideways, under the upper one -- the cell between them is a vertical face (a stair riser), the cell on the low
// side is stretched.  d.hf_mv carries the moves; xy in grid units (integers as floats), z in metres.
struct MeshV { float x, y, z; };
__device__ __forceinline__ MeshV mesh_vert(const EmlocoSimDev &d, int ci, int cj) {
    const long k = (long)ci * d.hf_ny + cj;
    const int b = d.hf_mv[k];
    MeshV v;
    v.x = (float)(ci + (b & 3) - 1); v.y = (float)(cj + ((b >> 2) & 3) - 1); v.z = d.hf_vs * (float)d.hf[k];
    return v;
}
// cell (ci, cj), triangle t: t = 0 (v00, v10, v11) [id half 1], t = 1 (v00, v11, v01) [id half 0]: the mesh's winding, normals out of the solid
__device__ __forceinline__ void mesh_tri(const EmlocoSimDev &d, int ci, int cj, int t, MeshV &A, MeshV &B, MeshV &Cc) {
    A = mesh_vert(d, ci, cj);
    if (t == 0) { B = mesh_vert(d, ci + 1, cj); Cc = mesh_vert(d, ci + 1, cj + 1); }
    else { B = mesh_vert(d, ci + 1, cj + 1); Cc = mesh_vert(d, ci, cj + 1); }
}
// The mesh surface under the world point (cx, cy): the highest of the (at most 18) triangles of the 3 x 3 cells around the point's
// regular cell that cover it.  Cells whose 4 x 4 vertex block carries no move take the regular-grid formula (bit-equal to the
// uncorrected height field), and so do points no triangle covers (beyond the map).
__device__ __noinline__ void mesh_plane(const EmlocoSimDev &d, float cx, float cy, float &zt, float n[3], int &id) {
    if (!d.hf_mv) { hf_plane(d, cx, cy, zt, n, id); return; }
    const float gx = (cx - d.hf_ox) * d.hf_inv_hs, gy = (cy - d.hf_oy) * d.hf_inv_hs;
    int i = (int)floorf(gx), j = (int)floorf(gy);
    i = i < 0 ? 0 : (i > d.hf_nx - 2 ? d.hf_nx - 2 : i);
    j = j < 0 ? 0 : (j > d.hf_ny - 2 ? d.hf_ny - 2 : j);
    if (!(d.hf_mv[(long)i * d.hf_ny + j] & 16)) { hf_plane(d, cx, cy, zt, n, id); return; }
    bool found = false;
    int bid = 0;
    float bz = 0.0f, bsx = 0.0f, bsy = 0.0f;
    for (int a = -1; a <= 1; ++a) {
        const int ci = i + a;
        if (ci < 0 || ci > d.hf_nx - 2) continue;
        for (int b = -1; b <= 1; ++b) {
            const int cj = j + b;
            if (cj < 0 || cj > d.hf_ny - 2) continue;
            for (int t = 0; t < 2; ++t) {
                MeshV A, B, Cc;
                mesh_tri(d, ci, cj, t, A, B, Cc);
                const float bx = B.x - A.x, by = B.y - A.y, qx = Cc.x - A.x, qy = Cc.y - A.y;
                const float ar = bx * qy - by * qx;
                if (ar == 0.0f) continue;                                   // collapsed: a vertical face, see mesh_walls
                const float px = gx - A.x, py = gy - A.y;
                const float eb = px * qy - py * qx, ec = bx * py - by * px;
                const bool in = ar > 0.0f ? (eb >= 0.0f && ec >= 0.0f && eb + ec <= ar) : (eb <= 0.0f && ec <= 0.0f && eb + ec >= ar);
                if (!in) continue;
                const float zb = B.z - A.z, zc = Cc.z - A.z;
                const float sx = (zb * qy - zc * by) / ar, sy = (zc * bx - zb * qx) / ar;
                const float z = fmaf(py, sy, fmaf(px, sx, A.z));
                if (!found || z > bz) { found = true; bz = z; bsx = sx; bsy = sy; bid = ((ci << 15) + cj) * 2 + (t == 0 ? 1 : 0); }
            }
        }
    }
    if (!found) { hf_plane(d, cx, cy, zt, n, id); return; }
    zt = bz; id = bid;
    const float sx = bsx * d.hf_inv_hs, sy = bsy * d.hf_inv_hs;
    const float inv = 1.0f / sqrtf(fmaf(sx, sx, fmaf(sy, sy, 1.0f)));
    n[0] = 0.0f - sx * inv; n[1] = 0.0f - sy * inv; n[2] = inv;
}
__device__ __forceinline__ float dot3f(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
// The vertical faces of the mesh near the world point P (centre of a contact sphere): every collapsed triangle of the 3 x 3 cells
// around P's regular cell, closest point by regions (vertex, edge, face).  dsel / nsel hold the signed distance of P to the nearest
// surface found so far and its normal (>= 0: P is outside the terrain solid).  Outside: a face P is in front of wins when it is
// nearer, normal from its closest point to P.  Inside: a face P is behind wins when the foot of P's perpendicular lies in it and
// it is nearer than the surface above, normal = the face's.
__device__ __noinline__ void mesh_walls(const EmlocoSimDev &d, const float P[3], float &dsel, float nsel[3]) {
    const float gx = (P[0] - d.hf_ox) * d.hf_inv_hs, gy = (P[1] - d.hf_oy) * d.hf_inv_hs;
    int i = (int)floorf(gx), j = (int)floorf(gy);
    i = i < 0 ? 0 : (i > d.hf_nx - 2 ? d.hf_nx - 2 : i);
    j = j < 0 ? 0 : (j > d.hf_ny - 2 ? d.hf_ny - 2 : j);
    if (!(d.hf_mv[(long)i * d.hf_ny + j] & 16)) return;
    for (int a = -1; a <= 1; ++a) {
        const int ci = i + a;
        if (ci < 0 || ci > d.hf_nx - 2) continue;
        for (int b = -1; b <= 1; ++b) {
            const int cj = j + b;
            if (cj < 0 || cj > d.hf_ny - 2) continue;
            for (int t = 0; t < 2; ++t) {
                MeshV A, B, Cc;
                mesh_tri(d, ci, cj, t, A, B, Cc);
                if ((B.x - A.x) * (Cc.y - A.y) - (B.y - A.y) * (Cc.x - A.x) != 0.0f) continue;
                // corners relative to P, metres
                const float va[3] = {fmaf(A.x, d.hf_hs, d.hf_ox) - P[0], fmaf(A.y, d.hf_hs, d.hf_oy) - P[1], A.z - P[2]};
                const float vb[3] = {fmaf(B.x, d.hf_hs, d.hf_ox) - P[0], fmaf(B.y, d.hf_hs, d.hf_oy) - P[1], B.z - P[2]};
                const float vc[3] = {fmaf(Cc.x, d.hf_hs, d.hf_ox) - P[0], fmaf(Cc.y, d.hf_hs, d.hf_oy) - P[1], Cc.z - P[2]};
                const float ab[3] = {vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]}, ac[3] = {vc[0] - va[0], vc[1] - va[1], vc[2] - va[2]};
                const float fn[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
                const float fn2 = dot3f(fn, fn);
                if (fn2 == 0.0f) continue;                                  // no area in space either
                // closest point q of the triangle to the origin (= P), by regions
                const float ap[3] = {0.0f - va[0], 0.0f - va[1], 0.0f - va[2]}, bp[3] = {0.0f - vb[0], 0.0f - vb[1], 0.0f - vb[2]};
                const float cp[3] = {0.0f - vc[0], 0.0f - vc[1], 0.0f - vc[2]};
                const float d1 = dot3f(ab, ap), d2 = dot3f(ac, ap), d3 = dot3f(ab, bp), d4 = dot3f(ac, bp), d5 = dot3f(ab, cp), d6 = dot3f(ac, cp);
                float q[3];
                bool face = false;
                const float vcc = d1 * d4 - d3 * d2, vbb = d5 * d2 - d1 * d6, vaa = d3 * d6 - d5 * d4;
                if (d1 <= 0.0f && d2 <= 0.0f) { q[0] = va[0]; q[1] = va[1]; q[2] = va[2]; }
                else if (d3 >= 0.0f && d4 <= d3) { q[0] = vb[0]; q[1] = vb[1]; q[2] = vb[2]; }
                else if (vcc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) { const float w = d1 / (d1 - d3); for (int k = 0; k < 3; ++k) q[k] = fmaf(w, ab[k], va[k]); }
                else if (d6 >= 0.0f && d5 <= d6) { q[0] = vc[0]; q[1] = vc[1]; q[2] = vc[2]; }
                else if (vbb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) { const float w = d2 / (d2 - d6); for (int k = 0; k < 3; ++k) q[k] = fmaf(w, ac[k], va[k]); }
                else if (vaa <= 0.0f && d4 - d3 >= 0.0f && d5 - d6 >= 0.0f) {
                    const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
                    for (int k = 0; k < 3; ++k) q[k] = fmaf(w, vc[k] - vb[k], vb[k]);
                } else {
                    const float den = 1.0f / (vaa + vbb + vcc), v = vbb * den, w = vcc * den;
                    for (int k = 0; k < 3; ++k) q[k] = fmaf(w, ac[k], fmaf(v, ab[k], va[k]));
                    face = true;
                }
                const float dv[3] = {0.0f - q[0], 0.0f - q[1], 0.0f - q[2]};
                const float side = dot3f(dv, fn);
                const float ifn = 1.0f / sqrtf(fn2);
                if (dsel >= 0.0f) {
                    const float dist = sqrtf(dot3f(dv, dv));
                    if (side >= 0.0f && dist < dsel) {
                        dsel = dist;
                        if (dist > 1.0e-6f) { const float id_ = 1.0f / dist; for (int k = 0; k < 3; ++k) nsel[k] = dv[k] * id_; }
                        else for (int k = 0; k < 3; ++k) nsel[k] = fn[k] * ifn;
                    }
                } else if (side < 0.0f && face) {
                    const float dw = side * ifn;
                    if (dw > dsel) { dsel = dw; for (int k = 0; k < 3; ++k) nsel[k] = fn[k] * ifn; }
                }
            }
        }
    }
}


#ifndef EMLOCO_SIM_WAVES_PER_SIMD
#define EMLOCO_SIM_WAVES_PER_SIMD 2   /* 20.0 KB of LDS per env PAIR: eight one-wave workgroups = 16 envs per CU, two resident waves per SIMD (register budget 256 per lane) */
#endif

// Sum over the lanes of one 32-lane half of the wave (round 6: two envs per wave, env 0's bodies in lanes 0..23, env 1's in lanes
// 32..55).  Bit-equal to wave_sum() of the env alone in a wave: there the four 16-lane row totals are added as ((r0 + r1) + r2) + r3
// with r2 = r3 = +0 (idle lanes contribute literal zeros); here a half's two rows are added and the two +0 terms follow explicitly
// (x + 0 is x except for x = -0, which is why they are spelled out).  The oracle's wave_sum_order is unchanged.
__device__ __forceinline__ float half_sum(float v, int half) {
    v += dpp_mov<0x140>(v);   // row_mirror
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    const float lo = lane_bcast(v, 0) + lane_bcast(v, 16), hi = lane_bcast(v, 32) + lane_bcast(v, 48);
    return ((half ? hi : lo) + 0.0f) + 0.0f;
}


// the 32 bits of a wave ballot that belong to one half of the wave
__device__ __forceinline__ unsigned half_bits(unsigned long long m, int half) { return (unsigned)(half ? (m >> 32) : m); }

// One step of an env PAIR: the body of the kernel below (one 64-lane wave, two envs; env1 or env0 may be -1: that half idles).
//
// Round 6 mapping.  The phases whose lane is a BODY (kinematics, momentum balances, drive, inertia / bias, the articulated-body
// factorisation and its tree passes, the impulse solve's tree passes, integration) run for both envs at once -- lane = 32 x env + body,
// ONE instruction stream -- which halves their instructions per env (they are 54 % of a substep; inside a tree level 1-5 of 64 lanes
// were live).  The phases whose lane is a contact candidate, a contact row or a limb-limb pair stay one env wide and run for the two
// envs one after the other (a loop of two, not unrolled: one copy of the code).  Same operations per value in the same order as one
// env per wave: the step stays on the oracle's bytes (tests/test_emu_kernels.py, tests/test_gpu_sim.py).
template <int HF>
__device__ __forceinline__ void sim_step_pair(const EmlocoSimParams &prm, const EmlocoSimDev &d, int env0, int env1, const int part, const int n_parts) {
    const int lane = threadIdx.x;
    const int half = lane >> 5, b = lane & 31;                // joint phases: env of the pair, body

    // ---------------------------------------------------------------- LDS: one blob per env PAIR, 5 032 words = 19.7 KB (eight per CU need <= 20 480 B)
    // Per env a persistent block P (1 172 words: root, momenta, tree constants, R | r, W | K, root factor, contact list, multipliers,
    // slot map) and the body-phase rows, packed without padding words: [V | pa] (12 words per body; V holds v_free from the end of
    // phase 4), pq (8), [a | Aacc] (12; the `a` half carries the limb-limb wrench from phase 1b to 2b, the `Aacc` half the body's bias
    // force from 2b on, for the rare second factorisation pass), Ia (24, phase 3 only).
    // Shared region of the pair, in this order:  B1 = [V|pa]_1 pq_1 | B0 = [V|pa]_0 pq_0 | [a|Aacc]_0 | Ia_0 | Ia_1 | [a|Aacc]_1
    // Contact matrices: on the fast path (both envs <= 10 contacts) every lane holds its row in registers; on the full-size path the
    // packed lower triangle (1 830 words) of the env being solved lies over its B block .. Ia (env 0: from B0, env 1: from B1 -- B1 is
    // still live while env 0 solves).  What the contact phases stage lies over rows that are dead then: the candidate staging of phase 5 and both envs' contact frames (height field) in
    // [a|Aacc]_0 .. Ia_1, the staged Jacobian rows of phase 7a in the B blocks, the limb-limb scratch of phase 1b in Ia_0 | Ia_1.
    enum { O_ROOT = 0, O_P = 16, O_V0 = 24, O_L = 36, O_PD = 48, O_R = O_PD + NB, O_W = O_R + NB * 12,
           O_L0 = O_W + NB * 24, O_CB = O_L0 + 44, O_CX = O_CB + MAXC / 4 + 3, O_CDIST = O_CX + 3 * MAXC, O_LAM = O_CDIST + MAXC,
           O_SLOT = O_LAM + MAXR, O_CRANGE = O_SLOT + 32, PW = O_CRANGE + 2 * NB / 4,
           SR = 2 * PW, BW = NB * 12 + NB * 8, O_B1 = SR, O_B0 = SR + BW,
           O_AA0 = SR + 2 * BW, O_IA0 = O_AA0 + NB * 12, O_IA1 = O_IA0 + NB * 24, O_AA1 = O_IA1 + NB * 24, LDS_WORDS = O_AA1 + NB * 12,
           AMAT = MAXR * (MAXR + 1) / 2,
           O_STAGE = O_AA0,                                    // candidate staging of phase 5 [2 envs][MAXCAND][7]
           O_CDIR = LDS_WORDS - 2 * 9 * MAXC,                  // contact frames [2 envs][MAXC][9] (height-field ground)
           O_ROWS = O_B1,                                      // fast path: staged Jacobian rows of phase 7a [2 envs][32][12]
           O_SCR = O_IA0 };                                    // limb-limb scratch of phase 1b
    static_assert(PW % 4 == 0 && O_R % 4 == 0 && O_W % 4 == 0 && SR % 4 == 0 && BW % 4 == 0, "rows must be 16-byte aligned");
    static_assert(O_STAGE + 2 * MAXCAND * 7 <= O_CDIR, "the candidate staging reaches the contact frames");
    static_assert(O_B0 + AMAT <= O_CDIR, "the full-size path's contact matrix (env 0: from B0 on; env 1: from B1 on) reaches the contact frames");
    static_assert(2 * 32 * 12 <= 2 * BW && 3 * PAIR_FAST_MAXC <= 32, "the fast path's staged rows do not fit the B blocks");
    static_assert(O_SCR + EMLOCO_SC_MAXSEG * 8 + EMLOCO_SC_MAXHITS * 8 + EMLOCO_SC_MAXPAIRS <= O_AA1, "limb-limb scratch does not fit Ia_0 | Ia_1");
    static_assert(LDS_WORDS * 4 <= 20480, "LDS per env pair above 160 KiB / 8 (two waves per SIMD, 16 envs per CU)");
    __shared__ __attribute__((aligned(16))) float lds[LDS_WORDS];
    // the arrays of ONE env, by the names the phases use; bound per lane (joint phases: the lane's own env) or per loop pass
    // (one-env-wide phases: env e)
#define ENV_VIEW(Pb, VPb, AAb, IAb)                                                                                                   \
    float *const sh_root = (Pb) + O_ROOT;                     /* p0[3] q0[4] V0[6] */                                                 \
    float *const sh_P = (Pb) + O_P;                           /* linear momentum: expected [0..2], of the current substep [3..5]; total mass [6] */ \
    float *const sh_V0 = (Pb) + O_V0;                         /* the root lane's hand-over between phases: free root twist [0..5], impulse change [6..11] */ \
    float *const sh_L = (Pb) + O_L;                           /* angular momentum about the centre of mass: expected [0..2], "a balance exists" [3], of the current substep [4..6]; centre of mass relative to O [8..10] */ \
    int *const sh_pd = (int *)((Pb) + O_PD);                  /* per body: tree constants, packed (PD_PARENT / PD_DEPTH / PD_SLOT / PD_CHILD) */ \
    float (*const sh_R)[12] = (float (*)[12])((Pb) + O_R);    /* rotation matrix [0..8] | position relative to O [9..11] */            \
    float (*const sh_W)[24] = (float (*)[24])((Pb) + O_W);    /* per joint: W = U K^T (6 x 3) [0..17] | K, the inverse Cholesky factor of D (packed lower) [18..23] */ \
    float *const sh_L0 = (Pb) + O_L0, *const sh_L0i = (Pb) + O_L0 + 36;   /* root Cholesky factor and 1 / its diagonal */              \
    unsigned char *const sh_cbody = (unsigned char *)((Pb) + O_CB);       /* body of each contact (bytes) */                          \
    float (*const sh_cx)[3] = (float (*)[3])((Pb) + O_CX);                                                                            \
    float *const sh_cdist = (Pb) + O_CDIST;                                                                                           \
    float *const sh_lam = (Pb) + O_LAM;                       /* per contact row: warm-start multiplier (6a), solved multiplier (after 6c) */ \
    signed char *const sh_crange = (signed char *)((Pb) + O_CRANGE);      /* per body (bytes): first [0..NB) and last [NB..2NB) contact of the current substep */ \
    unsigned char *const sh_slot = (unsigned char *)((Pb) + O_SLOT);      /* per candidate: its contact slot of the latest substep (255: none) */ \
    float (*const sh_V)[12] = (float (*)[12])(VPb);           /* twist about O; from the end of phase 4: v_free */                    \
    float (*const sh_pa)[12] = (float (*)[12])((VPb) + 6);                                                                            \
    float (*const sh_pq)[8] = (float (*)[8])((VPb) + NB * 12);            /* world position [0..2] | world rotation quaternion [4..7] */ \
    float (*const sh_a)[12] = (float (*)[12])(AAb);           /* phases 3-4 and 7; phases 1b-2b: the limb-limb wrench about O */      \
    float (*const sh_Aacc)[12] = (float (*)[12])((AAb) + 6);  /* velocity-product acceleration (phase 1 -> 2b), then the body's own bias force */ \
    float (*const sh_Ia)[24] = (float (*)[24])(IAb);
    ENV_VIEW(lds + half * PW, lds + (half ? O_B1 : O_B0), lds + (half ? O_AA1 : O_AA0), lds + (half ? O_IA1 : O_IA0))
    constexpr bool hf_on = HF != 0;          // compile time: the plane instantiation carries none of the height-field code

    // ---------------------------------------------------------------- per-lane constants
    const int *topo = d.topo;
    int my_env = half ? env1 : env0;
    bool on = b < NB && my_env >= 0;                          // this lane carries a body of a live env
    const int bb = b < NB ? b : 0;
    // rows that idle lanes may read (unconditional loads) come from an env of the pair that exists
    const int senv = my_env >= 0 ? my_env : (env0 >= 0 ? env0 : env1);
    const float *mdl = d.model + (size_t)senv * EMLOCO_MODEL_WORDS;                             /* this lane's env: model block, targets, dof state */
#define o_dyn (EMLOCO_MB_DYN + bb * 16)                                                         /* this body's dynamics record */
#define jdof ((b >= 1 && b < NB) ? (b - 1) * 3 : 0)                                             /* first dof of this body's joint */
#define o_drv (EMLOCO_MB_DRV + jdof * 4)
    const float *tgt_env = d.pd_target + (size_t)senv * NDOF;
    float *dofs_env = d.dof_state + (size_t)senv * NDOF * 2;
    if (b < NB) sh_pd[b] = topo[EMLOCO_TOPO_PDPACK + b];
    ((unsigned *)sh_slot)[b] = 0xffffffffu;                   // 128 candidate slots per env: none

    // ---------------------------------------------------------------- state -> registers
    // Split launch: the substeps [sub_lo, sub_hi) of this workgroup; part 0 starts from the state tensors like the whole
    // step, a later part continues from what its predecessor left (joint quaternions / rates / rotation vectors per lane,
    // root, momentum balance, multipliers and their slot map: plain copies, so the parts together are the fused step bit
    // for bit).  Parts are workgroups of ONE launch, the later ones at higher indices.
    const int sub_lo = (prm.n_sub * part) / n_parts, sub_hi = (prm.n_sub * (part + 1)) / n_parts;
    float *pst = d.part_state ? d.part_state + (long)senv * EMLOCO_PART_WORDS : nullptr;
    int work = 0;                                             // contact work of this lane's env: sum over substeps of (10 + contacts) where there are any
    float qj[4] = {0, 0, 0, 1}, wj[3] = {0, 0, 0}, edof[3] = {0, 0, 0};
    if (part == 0) {
        if (on && b >= 1) {
            const float *ds = dofs_env + jdof * 2;
            float e[3] = {ds[0], ds[2], ds[4]};
            rotvec2quat(e, qj);
            wj[0] = ds[1]; wj[1] = ds[3]; wj[2] = ds[5];
            quat2rotvec(qj, edof);
        }
        if (on && b == 0) {
            const float *rs = d.root_state + (long)my_env * 13;
            float q0[4] = {rs[3], rs[4], rs[5], rs[6]};
            qnormalize(q0);
            for (int k = 0; k < 3; ++k) { sh_root[k] = rs[k]; sh_root[7 + k] = rs[10 + k]; sh_root[10 + k] = rs[7 + k]; }
            for (int k = 0; k < 4; ++k) sh_root[3 + k] = q0[k];
        }
    } else {
        // The predecessor has published its state.  Parts are workgroups of one launch at ascending indices, so in dispatch
        // order the flag is set long before; the wait is bounded all the same (a lost flag must not hang the device) and a
        // wait that runs out is an ERROR, not a reason to go on from stale hand-over state: the workgroup raises the device
        // error word (emloco_sim_sync / the next step return EMLOCO_E_HIP) and abandons that env's step (its partner goes on).
        if (b == 0) {
            float ok = 0.0f;
            if (my_env >= 0) {
                // (the bound grows with the part index: a part that waited its bound out for one env of its pair delays the partner's
                // hand-over by that much, and the partner's later parts must not give up on it meanwhile)
                int spins = 0;
                const int spin_max = d.part_spin_max * part;
                const unsigned want = EMLOCO_PART_TAG(d.part_seq, part - 1);
                while (part_flag_get(d.part_flag + my_env) != want && ++spins < spin_max) __builtin_amdgcn_s_sleep(16);
                ok = spins < spin_max ? 1.0f : 0.0f;
                if (spins >= spin_max) part_err_raise(d.err, EMLOCO_ERR_PART_TIMEOUT);
            }
            sh_V0[0] = ok;
        }
        __syncthreads();
        if (lds[O_V0] == 0.0f) env0 = -1;
        if (lds[PW + O_V0] == 0.0f) env1 = -1;
        if (env0 < 0 && env1 < 0) return;
        my_env = half ? env1 : env0;
        on = b < NB && my_env >= 0;
        __syncthreads();
        if (on) {                   // granules b, 24 + b, 48 + b (granule-major: the lanes of one store / load instruction touch
            float v[12];            // one contiguous 384-byte run per env): joint quaternion | rates, e_0 | e_1, e_2
            part_ld48(pst + b * 4, v);
            for (int k = 0; k < 4; ++k) qj[k] = v[k];
            for (int k = 0; k < 3; ++k) wj[k] = v[4 + k];
            edof[0] = v[7]; edof[1] = v[8]; edof[2] = v[9];
        }
        if (my_env >= 0) {          // one granule of LDS state per lane of the half: root (4) | momentum (2) | multipliers (16) | slot map (8) | misc | angular momentum
            const int g = b;
            float v[4];
            part_ld16(pst + NB * 12 + 4 * g, v);
            float *dst = g < 4 ? sh_root + 4 * g : g < 6 ? sh_P + 4 * (g - 4) : g < 22 ? sh_lam + 4 * (g - 6) : g < 30 ? (float *)sh_slot + 4 * (g - 22) : sh_L;
            if (g != 30) for (int k = 0; k < 4; ++k) dst[k] = v[k];
            else work = __float_as_int(v[0]);                   // the work counter, handed to every lane of the half below
        }
        work = __shfl(work, 32 * half + 30);
    }
    __syncthreads();

    const float h = prm.h;
    // registers that persist across phases (lane = body of its env)
    // (R, r, V of a body live in LDS; the motion-subspace columns S = [R e_c ; r x R e_c] are re-formed where needed)
    float tau[3], dd[3];
    int sat[3];            // 0: implicit drive; +1 / -1: constant torque at +/- the effort limit; 2: effort drive (the command, clipped).
    float uh[3], qdd[3];

    for (int sub = sub_lo; sub < sub_hi + (part == n_parts - 1 ? 1 : 0); ++sub) {
        const bool final_pass = (sub == prm.n_sub);   // kinematics only, to write the body states
        const bool last = (sub == prm.n_sub - 1);

        PSTAMP(0);
#ifdef EMLOCO_SIM_PACC
        long long pacc_t = (long long)wall_clock64();
#endif
        // ============================================================ 1. kinematics + velocities (root -> leaves), both envs
        if (on && b == 0) {
            float q0[4] = {sh_root[3], sh_root[4], sh_root[5], sh_root[6]}, R[9];
            q2mat(q0, R);
            for (int k = 0; k < 3; ++k) { sh_pq[0][k] = sh_root[k]; sh_R[0][9 + k] = 0.0f; }
            for (int k = 0; k < 4; ++k) sh_pq[0][4 + k] = q0[k];
            for (int k = 0; k < 9; ++k) sh_R[0][k] = R[k];
            for (int k = 0; k < 6; ++k) { sh_V[0][k] = sh_root[7 + k]; sh_Aacc[0][k] = 0.0f; }
        }
        // joint offsets: re-read per substep (L2 hit) instead of held for the launch, but ahead of the level loop so that the
        // load's latency is not paid inside every level
        float jm[4];                                     // joint offset xyz | body mass
        ld4(mdl, o_dyn, jm);
        const float joff[3] = {jm[0], jm[1], jm[2]};
        __syncthreads();
        const int pd1 = sh_pd[bb];
        for (int lev = 1; lev <= d.max_depth; ++lev) {
            if (on && PD_DEPTH(pd1) == lev) {
                const int p = PD_PARENT(pd1);
                float Rp[9], o[3], qp[4], qw[4], pw[3], R[9], r[3], Sl[3][3], V[6];
                for (int k = 0; k < 9; ++k) Rp[k] = sh_R[p][k];
                for (int k = 0; k < 4; ++k) qp[k] = sh_pq[p][4 + k];
                matvec3(Rp, joff, o);
                for (int k = 0; k < 3; ++k) { pw[k] = sh_pq[p][k] + o[k]; r[k] = pw[k] - sh_root[k]; }
                qmul(qp, qj, qw);
                qnormalize(qw);
                q2mat(qw, R);
                float wv[3], Vp[6];
                for (int k = 0; k < 6; ++k) Vp[k] = sh_V[p][k];
                for (int c = 0; c < 3; ++c) {
                    float ax[3] = {R[c], R[3 + c], R[6 + c]};
                    cross3(r, ax, Sl[c]);
                }
                for (int k = 0; k < 3; ++k) {
                    wv[k] = SOP3(R[k * 3], wj[0], R[k * 3 + 1], wj[1], R[k * 3 + 2], wj[2]);
                    V[k] = Vp[k] + wv[k];
                    V[3 + k] = Vp[3 + k] + SOP3(Sl[0][k], wj[0], Sl[1][k], wj[1], Sl[2][k], wj[2]);
                }
                // velocity-product acceleration c = [w_p x wv ; vj x wv + r x (w_p x wv)]
                float t[3], vj[3], cc[6], t1[3], t2[3];
                cross3(V, r, t);
                for (int k = 0; k < 3; ++k) vj[k] = V[3 + k] + t[k];
                cross3(Vp, wv, cc);
                cross3(vj, wv, t1);
                cross3(r, cc, t2);
                for (int k = 0; k < 3; ++k) cc[3 + k] = t1[k] + t2[k];
                for (int k = 0; k < 3; ++k) { sh_pq[b][k] = pw[k]; sh_R[b][9 + k] = r[k]; }
                for (int k = 0; k < 4; ++k) sh_pq[b][4 + k] = qw[k];
                for (int k = 0; k < 9; ++k) sh_R[b][k] = R[k];
                for (int k = 0; k < 6; ++k) { sh_V[b][k] = V[k]; sh_Aacc[b][k] = sh_Aacc[p][k] + cc[k]; }
            }
            __syncthreads();
        }
        // Link angular-speed limit.  The reference caps link angular velocities (AssetOptions.max_angular_velocity = 100,
        // humanoid.py:685-688); here the cap is min(that, EMLOCO_WH_MAX / h).  When the fastest link of an env exceeds the cap
        // (rare; the branch is taken by the wave when either env of the pair needs it, and acts on that env's lanes only), root
        // angular velocity and joint rates are scaled down uniformly so that link just meets it, and the root's linear velocity is
        // shifted so the linear momentum is unchanged: V_i -> [sc w_i ; v_0' + sc (v_i - v_0)], v_0' = v_0 + (1 - sc)(v_com - v_0);
        // the velocity-product accelerations are quadratic in the angular rates.
        if (!final_pass) {
            float w2 = 0.0f;
            if (on) { const float *Vb = sh_V[b]; w2 = fmaf(Vb[0], Vb[0], fmaf(Vb[1], Vb[1], Vb[2] * Vb[2])); }
            for (int off = 16; off >= 1; off >>= 1) { const float o = __shfl_xor(w2, off); w2 = o > w2 ? o : w2; }     // within the half
            float wlim = EMLOCO_WH_MAX / h;
            if (prm.max_ang_vel < wlim) wlim = prm.max_ang_vel;
            const bool over_lim = w2 > wlim * wlim;               // uniform within a half
            if (__ballot(over_lim) != 0ull) {
                const float sc = wlim / sqrtf(w2), sc2 = sc * sc;
                float lm = 0.0f, lp[3] = {0.0f, 0.0f, 0.0f};
                if (on) {
                    float Rb[9], cm[4], cw[3], rc[3], wx[3], Vb[6];
                    for (int k = 0; k < 9; ++k) Rb[k] = sh_R[b][k];
                    ld4(mdl, o_dyn + 4, cm);
                    for (int k = 0; k < 6; ++k) Vb[k] = sh_V[b][k];
                    matvec3(Rb, cm, cw);
                    for (int k = 0; k < 3; ++k) rc[k] = sh_R[b][9 + k] + cw[k];
                    cross3(Vb, rc, wx);
                    lm = jm[3];
                    for (int k = 0; k < 3; ++k) lp[k] = lm * (Vb[3 + k] + wx[k]);
                }
                const float M = half_sum(lm, half);
                float v0[3], v0n[3];
                for (int k = 0; k < 3; ++k) { v0[k] = sh_root[10 + k]; v0n[k] = v0[k] + (1.0f - sc) * (half_sum(lp[k], half) / M - v0[k]); }
                __syncthreads();
                if (on && over_lim) {
                    for (int k = 0; k < 3; ++k) { sh_V[b][k] *= sc; sh_V[b][3 + k] = v0n[k] + sc * (sh_V[b][3 + k] - v0[k]); }
                    for (int k = 0; k < 6; ++k) sh_Aacc[b][k] *= sc2;
                    if (b >= 1) for (int k = 0; k < 3; ++k) wj[k] *= sc;
                    if (b == 0) for (int k = 0; k < 3; ++k) { sh_root[7 + k] *= sc; sh_root[10 + k] = v0n[k]; sh_L[k] *= sc; }   // (every angular rate scaled: so is L about c)
                }
                __syncthreads();
            }
        }

        // Linear-momentum balance.  The integrator is first order in the velocity products, so the velocity of the centre
        // of mass of a tumbling body would drift (measured: 5 % of g t in a tumbling free fall).  The total linear momentum is
        // therefore carried across the substeps of a launch -- P_exp = P + h (M g + sum of the contact forces), all known
        // exactly -- and the momentum the new generalized velocities actually have (known once the kinematics of the new
        // configuration are: here, and in the final pass that writes the body states) is shifted onto it by a uniform change
        // dv of the linear velocities.  The velocity-product accelerations see dv through v_i x S_i qd_i: + dv x (w_i - w_0).
        {
            float lm = 0.0f, lp[3] = {0.0f, 0.0f, 0.0f}, Vb[6] = {0, 0, 0, 0, 0, 0};
            if (on) {
                float Rb[9], cm[4], cw[3], rc[3], wx[3];
                for (int k = 0; k < 9; ++k) Rb[k] = sh_R[b][k];
                ld4(mdl, o_dyn + 4, cm);
                for (int k = 0; k < 6; ++k) Vb[k] = sh_V[b][k];
                matvec3(Rb, cm, cw);
                for (int k = 0; k < 3; ++k) rc[k] = sh_R[b][9 + k] + cw[k];
                cross3(Vb, rc, wx);
                lm = jm[3];
                for (int k = 0; k < 3; ++k) lp[k] = lm * (Vb[3 + k] + wx[k]);
            }
            const float Mtot = half_sum(lm, half);
            float Pact[3];
            for (int k = 0; k < 3; ++k) Pact[k] = half_sum(lp[k], half);
            if (sub > 0) {                                       // wave-uniform: a balance exists from the previous substep
                float dv[3];
                for (int k = 0; k < 3; ++k) dv[k] = (sh_P[k] - Pact[k]) / Mtot;
                if (on) {
                    const float wr[3] = {Vb[0] - sh_V[0][0], Vb[1] - sh_V[0][1], Vb[2] - sh_V[0][2]};
                    float t[3];
                    cross3(dv, wr, t);
                    for (int k = 0; k < 3; ++k) { sh_Aacc[b][3 + k] += t[k]; sh_V[b][3 + k] = Vb[3 + k] + dv[k]; }
                }
                if (on && b == 0) for (int k = 0; k < 3; ++k) { sh_root[10 + k] += dv[k]; sh_P[3 + k] = sh_P[k]; }
            } else if (on && b == 0) {
                for (int k = 0; k < 3; ++k) sh_P[3 + k] = Pact[k];
            }
            if (on && b == 0) sh_P[6] = Mtot;
            __syncthreads();
            // Angular-momentum balance (oracle_sim.c: project_angular_momentum).  Like the linear momentum, the angular momentum about
            // the centre of mass c is carried across the substeps of a launch -- L_exp = damp (L + sum of (x - c) x contact impulse) --
            // and what the new generalized velocities have in the new configuration is moved onto it by a rigid rotation rate dw of the
            // whole body about c: J dw = L_exp - L (J: composite inertia about c, joints locked), every body twist gains
            // [dw ; c x dw] about O -- the linear momentum is untouched.
            {
                // the twelve sums are taken about O (none waits for another) and moved to c afterwards:
                // L_c = L_O - c x P,  J_c = J_O - M ((c.c) E - c c^T)
                float lc[3] = {0.0f, 0.0f, 0.0f}, ll[3] = {0.0f, 0.0f, 0.0f}, lj[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, Vn[6] = {0, 0, 0, 0, 0, 0};
                if (on) {
                    float Rb[9], cm[4], cw[3], rcb[3], wx[3], vcb[3], in6[8], Rc[9], Ic[6], ru[3], Iw[3];
                    for (int k = 0; k < 9; ++k) Rb[k] = sh_R[b][k];
                    ld4(mdl, o_dyn + 4, cm);
                    ld4(mdl, o_dyn + 8, in6); ld4(mdl, o_dyn + 12, in6 + 4);
                    for (int k = 0; k < 6; ++k) Vn[k] = sh_V[b][k];
                    matvec3(Rb, cm, cw);
                    for (int k = 0; k < 3; ++k) rcb[k] = sh_R[b][9 + k] + cw[k];
                    cross3(Vn, rcb, wx);
                    for (int k = 0; k < 3; ++k) vcb[k] = Vn[3 + k] + wx[k];
                    const float Ib[9] = {in6[0], in6[3], in6[4], in6[3], in6[1], in6[5], in6[4], in6[5], in6[2]};
                    for (int a = 0; a < 3; ++a)
                        for (int q = 0; q < 3; ++q) Rc[a * 3 + q] = SOP3(Rb[a * 3], Ib[q], Rb[a * 3 + 1], Ib[3 + q], Rb[a * 3 + 2], Ib[6 + q]);
                    const int ja[6] = {0, 1, 2, 0, 0, 1}, jb[6] = {0, 1, 2, 1, 2, 2};          // upper triangle: 00 11 22 01 02 12
                    for (int e = 0; e < 6; ++e) {
                        const int a = ja[e], q = jb[e];
                        Ic[e] = SOP3(Rc[a * 3], Rb[q * 3], Rc[a * 3 + 1], Rb[q * 3 + 1], Rc[a * 3 + 2], Rb[q * 3 + 2]);
                    }
                    cross3(rcb, vcb, ru);
                    Iw[0] = SOP3(Ic[0], Vn[0], Ic[3], Vn[1], Ic[4], Vn[2]);
                    Iw[1] = SOP3(Ic[3], Vn[0], Ic[1], Vn[1], Ic[5], Vn[2]);
                    Iw[2] = SOP3(Ic[4], Vn[0], Ic[5], Vn[1], Ic[2], Vn[2]);
                    const float rr = dot3(rcb, rcb);
                    for (int k = 0; k < 3; ++k) { lc[k] = lm * rcb[k]; ll[k] = Iw[k] + lm * ru[k]; }
                    for (int e = 0; e < 6; ++e) {
                        const int a = ja[e], q = jb[e];
                        lj[e] = Ic[e] + lm * ((a == q ? rr : 0.0f) - rcb[a] * rcb[q]);
                    }
                }
                float Cc[3], LO[3], JO[6], Lact[3], Jc[6], cP[3];
                for (int k = 0; k < 3; ++k) Cc[k] = half_sum(lc[k], half) / Mtot;
                for (int k = 0; k < 3; ++k) LO[k] = half_sum(ll[k], half);
                for (int e = 0; e < 6; ++e) JO[e] = half_sum(lj[e], half);
                {
                    const float Pc[3] = {sh_P[3], sh_P[4], sh_P[5]};
                    cross3(Cc, Pc, cP);
                    for (int k = 0; k < 3; ++k) Lact[k] = LO[k] - cP[k];
                    const int ja[6] = {0, 1, 2, 0, 0, 1}, jb[6] = {0, 1, 2, 1, 2, 2};
                    const float cc = dot3(Cc, Cc);
                    for (int e = 0; e < 6; ++e) Jc[e] = JO[e] - Mtot * ((ja[e] == jb[e] ? cc : 0.0f) - Cc[ja[e]] * Cc[jb[e]]);
                }
                const bool havL = sub > 0 && sh_L[3] != 0.0f;        // uniform within a half
                float dw[3] = {0.0f, 0.0f, 0.0f}, dvO[3] = {0.0f, 0.0f, 0.0f};
                bool moved = false;
                if (havL) {
                    const float b0 = sh_L[0] - Lact[0], b1 = sh_L[1] - Lact[1], b2 = sh_L[2] - Lact[2];
                    const float c00 = Jc[1] * Jc[2] - Jc[5] * Jc[5], c01 = Jc[4] * Jc[5] - Jc[3] * Jc[2], c02 = Jc[3] * Jc[5] - Jc[4] * Jc[1];
                    const float c11 = Jc[0] * Jc[2] - Jc[4] * Jc[4], c12 = Jc[3] * Jc[4] - Jc[0] * Jc[5], c22 = Jc[0] * Jc[1] - Jc[3] * Jc[3];
                    const float det = SOP3(Jc[0], c00, Jc[3], c01, Jc[4], c02);
                    if (det > 1e-12f) {
                        dw[0] = SOP3(c00, b0, c01, b1, c02, b2) / det;
                        dw[1] = SOP3(c01, b0, c11, b1, c12, b2) / det;
                        dw[2] = SOP3(c02, b0, c12, b1, c22, b2) / det;
                        cross3(Cc, dw, dvO);
                        moved = true;
                    }
                }
                const float Lexp[3] = {sh_L[0], sh_L[1], sh_L[2]};
                __syncthreads();                                      // every lane has read sh_L / sh_P / sh_V
                if (moved && on)
                    for (int k = 0; k < 3; ++k) { sh_V[b][k] = Vn[k] + dw[k]; sh_V[b][3 + k] = Vn[3 + k] + dvO[k]; }
                if (on && b == 0) {
                    if (moved) for (int k = 0; k < 3; ++k) { sh_root[7 + k] += dw[k]; sh_root[10 + k] += dvO[k]; }
                    for (int k = 0; k < 3; ++k) { sh_L[4 + k] = moved ? Lexp[k] : Lact[k]; sh_L[8 + k] = Cc[k]; }
                }
                __syncthreads();
            }
        }
        if (final_pass) break;

        // ============================================================ 1b. limb-limb contacts (self-collision, penalty), one env at a time
        // lane = collision segment: world capsule -> LDS; lane = pair: closest points, spring-damper force; hits are compacted in
        // pair order (ballot) and every body adds the wrenches that act on it in that order.
        if (d.sc_n > 0) {
#pragma nounroll
            for (int e = 0; e < 2; ++e) {
                const int env = e ? env1 : env0;
                if (env < 0) continue;
                ENV_VIEW(lds + e * PW, lds + (e ? O_B1 : O_B0), lds + (e ? O_AA1 : O_AA0), lds + (e ? O_IA1 : O_IA0))
                const float *mdl = d.model + (size_t)env * EMLOCO_MODEL_WORDS;
                // lane = collision segment: one sphere-swept segment per body, and a second one for a box much wider than thick (the
                // ankle boxes: two capsules along the long edges); the segment's record names its body
                float *sh_seg = lds + O_SCR;                                      // [MAXSEG][8]   (Ia of both envs is free until phase 3)
                float *sh_hitw = sh_seg + EMLOCO_SC_MAXSEG * 8;                    // [MAXHITS][6] wrench on body i about O (body j gets the negative)
                int *sh_hitb = (int *)(sh_seg + EMLOCO_SC_MAXSEG * 8 + EMLOCO_SC_MAXHITS * 6);   // [MAXHITS][2]
                if (lane < d.sc_nseg) {
                    float R[9], r[3], ca[4], cb[4], pa[3], pb[3];
                    ld4(mdl, EMLOCO_MB_CAP + lane * 8, ca); ld4(mdl, EMLOCO_MB_CAP + lane * 8 + 4, cb);     // end a xyz, radius | end b xyz, body
                    const int sb = (int)cb[3];
                    for (int k = 0; k < 9; ++k) R[k] = sh_R[sb][k];
                    for (int k = 0; k < 3; ++k) r[k] = sh_R[sb][9 + k];
                    matvec3(R, ca, pa); matvec3(R, cb, pb);
                    for (int k = 0; k < 3; ++k) { sh_seg[lane * 8 + k] = r[k] + pa[k]; sh_seg[lane * 8 + 3 + k] = r[k] + pb[k]; }
                    sh_seg[lane * 8 + 6] = ca[3];
                    const float ax[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
                    sh_seg[lane * 8 + 7] = 0.5f * sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);      // half length of the capsule's segment
                }
                __syncthreads();
                // broad phase: two capsules can only touch when their segment midpoints are closer than the two radii plus the two
                // half lengths (triangle inequality; 1 mm of slack covers the rounding): the pairs that pass -- a few dozen of the
                // 245 -- are compacted in pair order and only they run the closest-point test, in one round instead of four.
                // A culled pair cannot hit, so the hit list (and every force) is what the full sweep gives.
                int *sh_cand = (int *)(sh_seg + EMLOCO_SC_MAXSEG * 8 + EMLOCO_SC_MAXHITS * 8);
                int ncand = 0;
                for (int q0 = 0; q0 < d.sc_n; q0 += 64) {
                    const int q = q0 + lane;
                    bool keep = false;
                    if (q < d.sc_n) {
                        const int pr = topo[EMLOCO_TOPO_SCPAIR + q], bi = pr & 0xff, bj = (pr >> 8) & 0xff;      // segments
                        float dm2 = 0.0f;
                        for (int k = 0; k < 3; ++k) {
                            const float dm = (sh_seg[bi * 8 + k] + sh_seg[bi * 8 + 3 + k]) - (sh_seg[bj * 8 + k] + sh_seg[bj * 8 + 3 + k]);   // 2 (m_i - m_j)
                            dm2 += dm * dm;
                        }
                        const float bound = (sh_seg[bi * 8 + 6] + sh_seg[bj * 8 + 6]) + (sh_seg[bi * 8 + 7] + sh_seg[bj * 8 + 7]) + 1e-3f;
                        keep = dm2 <= 4.0f * bound * bound;
                    }
                    const unsigned long long bal = __ballot(keep);
                    if (keep) sh_cand[ncand + __popcll(bal & ((1ull << lane) - 1ull))] = q;
                    ncand += __popcll(bal);
                }
                __syncthreads();
                int nh = 0;
                for (int c0 = 0; c0 < ncand; c0 += 64) {
                    const int q = (c0 + lane < ncand) ? sh_cand[c0 + lane] : d.sc_n;
                    bool hit = false;
                    float w6[6] = {0, 0, 0, 0, 0, 0};
                    int bi = 0, bj = 0;
                    if (q < d.sc_n) {
                        const int pr = topo[EMLOCO_TOPO_SCPAIR + q];
                        const int si = pr & 0xff, sj = (pr >> 8) & 0xff;                 // the two segments ...
                        bi = (pr >> 16) & 0xff; bj = (pr >> 24) & 0xff;                  // ... and the bodies they belong to
                        float p0[3], p1[3], g0[3], g1[3], c1[3], c2[3];
                        for (int k = 0; k < 3; ++k) { p0[k] = sh_seg[si * 8 + k]; p1[k] = sh_seg[si * 8 + 3 + k]; g0[k] = sh_seg[sj * 8 + k]; g1[k] = sh_seg[sj * 8 + 3 + k]; }
                        const float rsum = sh_seg[si * 8 + 6] + sh_seg[sj * 8 + 6];
                        seg_seg_closest(p0, p1, g0, g1, c1, c2);
                        const float dv[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
                        const float dist2 = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
                        if (dist2 < rsum * rsum && dist2 > 1e-12f) {
                            const float dist = sqrtf(dist2);
                            float pen = rsum - dist;
                            const float n[3] = {dv[0] / dist, dv[1] / dist, dv[2] / dist};
                            const float off = sh_seg[sj * 8 + 6] - 0.5f * pen;     // contact point: middle of the overlap
                            const float pt[3] = {c2[0] + n[0] * off, c2[1] + n[1] * off, c2[2] + n[2] * off};
                            float Vi[6], Vj[6], wi[3], wjx[3];
                            for (int k = 0; k < 6; ++k) { Vi[k] = sh_V[bi][k]; Vj[k] = sh_V[bj][k]; }
                            cross3(Vi, pt, wi); cross3(Vj, pt, wjx);
                            const float vn = ((Vi[3] + wi[0]) - (Vj[3] + wjx[0])) * n[0] + ((Vi[4] + wi[1]) - (Vj[4] + wjx[1])) * n[1]
                                           + ((Vi[5] + wi[2]) - (Vj[5] + wjx[2])) * n[2];
                            if (pen > d.sc_max_pen) pen = d.sc_max_pen;
                            float F = d.sc_k * pen - d.sc_c * vn;
                            if (F > 0.0f) {
                                float Fv[3] = {n[0] * F, n[1] * F, n[2] * F};
                                if (d.sc_mu > 0.0f) {      // Coulomb friction capped by the contact's damper: - min(mu F / |v_t|, c) v_t
                                    const float vr[3] = {(Vi[3] + wi[0]) - (Vj[3] + wjx[0]), (Vi[4] + wi[1]) - (Vj[4] + wjx[1]),
                                                         (Vi[5] + wi[2]) - (Vj[5] + wjx[2])};
                                    const float vt[3] = {vr[0] - vn * n[0], vr[1] - vn * n[1], vr[2] - vn * n[2]};
                                    const float vt2 = vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2];
                                    if (vt2 > 1e-12f) {
                                        float g = d.sc_mu * F / sqrtf(vt2);
                                        if (g > d.sc_c) g = d.sc_c;
                                        for (int k = 0; k < 3; ++k) Fv[k] -= g * vt[k];
                                    }
                                }
                                cross3(pt, Fv, w6);
                                w6[3] = Fv[0]; w6[4] = Fv[1]; w6[5] = Fv[2];
                                hit = true;
                            }
                        }
                    }
                    const unsigned long long bal = __ballot(hit);
                    const int slot = nh + __popcll(bal & ((1ull << lane) - 1ull));
                    if (hit && slot < EMLOCO_SC_MAXHITS) {
                        for (int k = 0; k < 6; ++k) sh_hitw[slot * 6 + k] = w6[k];
                        sh_hitb[slot * 2] = bi; sh_hitb[slot * 2 + 1] = bj;
                    }
                    nh += __popcll(bal);
                }
                if (nh > EMLOCO_SC_MAXHITS) nh = EMLOCO_SC_MAXHITS;
                __syncthreads();
                if (lane < NB) {                                        // the wrench rides in the `a` half of the env's [a | Aacc] rows until phase 2b
                    float fe[6] = {0, 0, 0, 0, 0, 0};
                    for (int hh = 0; hh < nh; ++hh) {
                        const int bi = sh_hitb[hh * 2], bj = sh_hitb[hh * 2 + 1];
                        if (bi == lane) for (int k = 0; k < 6; ++k) fe[k] += sh_hitw[hh * 6 + k];
                        else if (bj == lane) for (int k = 0; k < 6; ++k) fe[k] -= sh_hitw[hh * 6 + k];
                    }
                    for (int k = 0; k < 6; ++k) sh_a[lane][k] = fe[k];
                }
                __syncthreads();                                       // the scratch is reused by the partner env, then by phase 3
            }
        }

        PSTAMP(1); PACC(0);
        // ============================================================ 2. drive (both envs)
        if (on) {
            // implicit PD drive: tau~ = kp (q* - q) - (kd + h kp) qd, joint-space diagonal d = armature + h kd + h^2 kp
            for (int k = 0; k < 3; ++k) {
                float dr[4];                                     // kp, kd, armature, effort limit of this dof
                ld4(mdl, o_drv + 4 * k, dr);
                const float kp = b >= 1 ? dr[0] : 0.0f, kd = b >= 1 ? dr[1] : 0.0f;
                const float arm = b >= 1 ? dr[2] : 0.0f, tgt = b >= 1 ? tgt_env[jdof + k] : 0.0f;
                const float e = tgt - edof[k];
                if (prm.drive_mode == 1) {       // effort drive (gymapi.DOF_MODE_EFFORT): the given torque within the limit, nothing implicit
                    const float eff = b >= 1 ? dr[3] : 0.0f;
                    sat[k] = 2; tau[k] = tgt > eff ? eff : (tgt < -eff ? -eff : tgt); dd[k] = arm;
                } else {
                    sat[k] = 0; tau[k] = kp * e - (kd + h * kp) * wj[k]; dd[k] = arm + h * kd + h * h * kp;
                }
            }
        }

        PSTAMP(2); PACC(1);
        // Phases 3-4 run once with every drive implicit.  The torque such a drive delivers over the substep is
        // tau~ - (h kd + h^2 kp) qdd; where that exceeds the effort limit the drive becomes a constant torque at the limit
        // (no implicit terms) and the phases run once more (rare; when either env of the pair needs it -- the partner repeats
        // them on unchanged inputs and lands on the same bytes).
        float pA[6];
        for (int pass = 0; pass < 2; ++pass) {
        // ============================================================ 2b. inertia about O and bias force of every body (lane = body)
        // They stay in this lane's registers until its level of the up pass below takes them over in place -- no trip through
        // LDS.  The bias force is also left in the body's own Aacc slot (dead from here on): the second pass reloads it instead
        // of forming it again from the velocity-product acceleration and the limb-limb wrench, whose slots phases 3-4 reuse.
        float IA[21];
        for (int k = 0; k < 21; ++k) IA[k] = 0.0f;
        if (on) {
            float R[9], r[3];
            for (int k = 0; k < 9; ++k) R[k] = sh_R[b][k];
            for (int k = 0; k < 3; ++k) r[k] = sh_R[b][9 + k];
            float Rc[9], Ic[9], cw[3], c[3], in6[8], bcom[4];
            ld4(mdl, o_dyn + 8, in6); ld4(mdl, o_dyn + 12, in6 + 4);            // mass properties: re-read per substep (L2 hits)
            ld4(mdl, o_dyn + 4, bcom);
            const float bmass = mdl[o_dyn + 3];
            const float Ib[9] = {in6[0], in6[3], in6[4], in6[3], in6[1], in6[5], in6[4], in6[5], in6[2]};
            for (int a = 0; a < 3; ++a)
                for (int q = 0; q < 3; ++q) Rc[a * 3 + q] = SOP3(R[a * 3], Ib[q], R[a * 3 + 1], Ib[3 + q], R[a * 3 + 2], Ib[6 + q]);
            for (int a = 0; a < 3; ++a)
                for (int q = a; q < 3; ++q) {
                    Ic[a * 3 + q] = SOP3(Rc[a * 3], R[q * 3], Rc[a * 3 + 1], R[q * 3 + 1], Rc[a * 3 + 2], R[q * 3 + 2]);
                    Ic[q * 3 + a] = Ic[a * 3 + q];
                }
            matvec3(R, bcom, cw);
            for (int k = 0; k < 3; ++k) c[k] = r[k] + cw[k];
            const float ms = bmass, cc = dot3(c, c);
            for (int a = 0; a < 3; ++a)
                for (int q = a; q < 3; ++q) IA[sidx(a, q)] = Ic[a * 3 + q] + ms * ((a == q ? cc : 0.0f) - c[a] * c[q]);
            const float cx[9] = {0.0f, -c[2], c[1], c[2], 0.0f, -c[0], -c[1], c[0], 0.0f};
            for (int a = 0; a < 3; ++a)
                for (int q = 0; q < 3; ++q) IA[sidx(a, 3 + q)] = ms * cx[a * 3 + q];
            for (int a = 0; a < 3; ++a)
                for (int q = a; q < 3; ++q) IA[sidx(3 + a, 3 + q)] = (a == q) ? ms : 0.0f;
            if (pass == 0) {
                float V[6], Aa[6], hI[6], IAc[6], x1[3], x2[3];
                for (int k = 0; k < 6; ++k) V[k] = sh_V[b][k];
                for (int k = 0; k < 6; ++k) Aa[k] = sh_Aacc[b][k];
                for (int a = 0; a < 6; ++a) {
                    const float Ir[6] = {IA[sidx(a, 0)], IA[sidx(a, 1)], IA[sidx(a, 2)], IA[sidx(a, 3)], IA[sidx(a, 4)], IA[sidx(a, 5)]};
                    hI[a] = fdot6(Ir, V);
                    IAc[a] = fdot6(Ir, Aa);
                }
                cross3(V, hI, x1); cross3(V + 3, hI + 3, x2);
                for (int k = 0; k < 3; ++k) pA[k] = IAc[k] + x1[k] + x2[k];
                cross3(V, hI + 3, x1);
                for (int k = 0; k < 3; ++k) pA[3 + k] = IAc[3 + k] + x1[k];
                float fg[3] = {0.0f, 0.0f, ms * prm.gravity_z}, ng[3];
                cross3(c, fg, ng);
                for (int k = 0; k < 3; ++k) { pA[k] -= ng[k]; pA[3 + k] -= fg[k]; }
                if (d.sc_n > 0) for (int k = 0; k < 6; ++k) pA[k] -= sh_a[b][k];        // external wrench (phase 1b): f -= [p x F ; F]
                for (int k = 0; k < 6; ++k) sh_Aacc[b][k] = pA[k];                      // this lane's own row: kept for a second pass
            } else {
                for (int k = 0; k < 6; ++k) pA[k] = sh_Aacc[b][k];
            }
        }
        // ============================================================ 3. articulated-body factorisation + up pass (leaves -> root)
        const int pd3 = sh_pd[bb];
        for (int lev = d.max_depth; lev >= 0; --lev) {
            if (on && PD_DEPTH(pd3) == lev) {
                float Wm[18], Km[6];
                float R[9], r[3], Sl[3][3];
                for (int k = 0; k < 9; ++k) R[k] = sh_R[b][k];
                for (int k = 0; k < 3; ++k) r[k] = sh_R[b][9 + k];
                for (int c = 0; c < 3; ++c) { const float ax[3] = {R[c], R[3 + c], R[6 + c]}; cross3(r, ax, Sl[c]); }
                for (int ci = 0; ci < 3; ++ci) {     // children in descending body index
                    const int ch = PD_CHILD(pd3, ci);
                    if (ch != 31) {
                        for (int k = 0; k < 21; ++k) IA[k] += sh_Ia[ch][k];
                        for (int k = 0; k < 6; ++k) pA[k] += sh_pa[ch][k];
                    }
                }
                if (lev > 0) {
                    float U[18], D[9];
                    for (int c = 0; c < 3; ++c) {
                        const float Sc[6] = {R[c], R[3 + c], R[6 + c], Sl[c][0], Sl[c][1], Sl[c][2]};
                        for (int a = 0; a < 6; ++a)
                        { const float Ir[6] = {IA[sidx(a, 0)], IA[sidx(a, 1)], IA[sidx(a, 2)], IA[sidx(a, 3)], IA[sidx(a, 4)], IA[sidx(a, 5)]}; U[a * 3 + c] = fdot6(Ir, Sc); }
                    }
                    for (int a = 0; a < 3; ++a) {
                        const float Sa[6] = {R[a], R[3 + a], R[6 + a], Sl[a][0], Sl[a][1], Sl[a][2]};
                        for (int q = 0; q < 3; ++q) {
                            float acc = 0.0f;
                            for (int k = 0; k < 6; ++k) acc = fmaf(Sa[k], U[k * 3 + q], acc);
                            D[a * 3 + q] = acc + (a == q ? dd[a] : 0.0f);
                        }
                    }
                    // reciprocals of the pivots first: the off-diagonal entries multiply instead of divide
                    const float l00 = sqrtf(D[0]), k00 = 1.0f / l00, l10 = D[3] * k00, l20 = D[6] * k00;
                    const float l11 = sqrtf(fmaf(-l10, l10, D[4])), k11 = 1.0f / l11, l21 = fmaf(-l20, l10, D[7]) * k11;
                    const float l22 = sqrtf(fmaf(-l21, l21, fmaf(-l20, l20, D[8]))), k22 = 1.0f / l22;
                    const float k10 = -l10 * k00 * k11, k21 = -l21 * k11 * k22, k20 = -SOP2(l20, k00, l21, k10) * k22;
                    Km[0] = k00; Km[1] = k10; Km[2] = k11; Km[3] = k20; Km[4] = k21; Km[5] = k22;
                    for (int a = 0; a < 6; ++a) {
                        Wm[a * 3 + 0] = U[a * 3] * k00;
                        Wm[a * 3 + 1] = SOP2(U[a * 3], k10, U[a * 3 + 1], k11);
                        Wm[a * 3 + 2] = SOP3(U[a * 3], k20, U[a * 3 + 1], k21, U[a * 3 + 2], k22);
                    }
                    float u[3];
                    for (int c = 0; c < 3; ++c) {
                        const float Sc[6] = {R[c], R[3 + c], R[6 + c], Sl[c][0], Sl[c][1], Sl[c][2]};
                        u[c] = tau[c] - fdot6(Sc, pA);
                    }
                    uh[0] = Km[0] * u[0];
                    uh[1] = SOP2(Km[1], u[0], Km[2], u[1]);
                    uh[2] = SOP3(Km[3], u[0], Km[4], u[1], Km[5], u[2]);
                    for (int a = 0; a < 6; ++a)
                        for (int q = a; q < 6; ++q)
                            sh_Ia[b][sidx(a, q)] = SUB_SOP3(IA[sidx(a, q)], Wm[a * 3], Wm[q * 3], Wm[a * 3 + 1], Wm[q * 3 + 1], Wm[a * 3 + 2], Wm[q * 3 + 2]);
                    for (int k = 0; k < 6; ++k)
                        sh_pa[b][k] = ADD_SOP3(pA[k], Wm[k * 3], uh[0], Wm[k * 3 + 1], uh[1], Wm[k * 3 + 2], uh[2]);
                    for (int k = 0; k < 18; ++k) sh_W[b][k] = Wm[k];
                    for (int k = 0; k < 6; ++k) sh_W[b][18 + k] = Km[k];
                } else {
                    // root: Cholesky of the 6x6 articulated inertia, a0 = -IA0^-1 pA0
                    float L[21], Li[6];    // lower triangle, (a, q<=a) at a(a+1)/2 + q; reciprocal pivots
#define LT(a, q) L[(a) * ((a) + 1) / 2 + (q)]
                    for (int a = 0; a < 6; ++a)
                        for (int q = 0; q <= a; ++q) {
                            float acc = IA[sidx(a, q)];
                            for (int k = 0; k < q; ++k) acc = fmaf(-LT(a, k), LT(q, k), acc);
                            if (a == q) { LT(a, q) = sqrtf(acc); Li[a] = 1.0f / LT(a, q); }
                            else LT(a, q) = acc * Li[q];
                        }
                    float y[6], x[6];
                    for (int a = 0; a < 6; ++a) {
                        float acc = -pA[a];
                        for (int k = 0; k < a; ++k) acc = fmaf(-LT(a, k), y[k], acc);
                        y[a] = acc * Li[a];
                    }
                    for (int a = 5; a >= 0; --a) {
                        float acc = y[a];
                        for (int k = a + 1; k < 6; ++k) acc = fmaf(-LT(k, a), x[k], acc);
                        x[a] = acc * Li[a];
                    }
                    for (int a = 0; a < 6; ++a)
                        for (int q = 0; q <= a; ++q) sh_L0[a * 6 + q] = LT(a, q);
                    for (int a = 0; a < 6; ++a) sh_L0i[a] = Li[a];
#undef LT
                    for (int k = 0; k < 6; ++k) sh_a[0][k] = x[k];
                }
            }
            __syncthreads();
        }

        PSTAMP(3); PACC(2);
        // ============================================================ 4. down pass: joint accelerations, v_free
        const int pd4 = sh_pd[bb];
        for (int lev = 1; lev <= d.max_depth; ++lev) {
            if (on && PD_DEPTH(pd4) == lev) {
                float ap[6], t[3], a[6], Wm[18], Km[6];
                float R[9], r[3], Sl[3][3];
                for (int k = 0; k < 9; ++k) R[k] = sh_R[b][k];
                for (int k = 0; k < 3; ++k) r[k] = sh_R[b][9 + k];
                for (int c = 0; c < 3; ++c) { const float ax[3] = {R[c], R[3 + c], R[6 + c]}; cross3(r, ax, Sl[c]); }
                for (int k = 0; k < 18; ++k) Wm[k] = sh_W[b][k];
                for (int k = 0; k < 6; ++k) Km[k] = sh_W[b][18 + k];
                for (int k = 0; k < 6; ++k) ap[k] = sh_a[PD_PARENT(pd4)][k];
                for (int c = 0; c < 3; ++c) {
                    float acc = 0.0f;
                    for (int k = 0; k < 6; ++k) acc = fmaf(Wm[k * 3 + c], ap[k], acc);
                    t[c] = uh[c] - acc;
                }
                qdd[0] = SOP3(Km[0], t[0], Km[1], t[1], Km[3], t[2]);
                qdd[1] = SOP2(Km[2], t[1], Km[4], t[2]);
                qdd[2] = Km[5] * t[2];
                for (int k = 0; k < 3; ++k) {
                    a[k] = ADD_SOP3(ap[k], R[k * 3], qdd[0], R[k * 3 + 1], qdd[1], R[k * 3 + 2], qdd[2]);
                    a[3 + k] = ADD_SOP3(ap[3 + k], Sl[0][k], qdd[0], Sl[1][k], qdd[1], Sl[2][k], qdd[2]);
                }
                for (int k = 0; k < 6; ++k) sh_a[b][k] = a[k];
            }
            __syncthreads();
        }
        if (pass == 0) {
            bool over = false;
            if (on && b >= 1)
                for (int k = 0; k < 3; ++k) {
                    float dr[4];
                    ld4(mdl, o_drv + 4 * k, dr);
                    const float kp = dr[0], kd = dr[1], eff = dr[3];
                    const float ti = tau[k] - (h * kd + h * h * kp) * qdd[k];
                    if (sat[k] == 0 && fabsf(ti) > eff) { sat[k] = ti > 0.0f ? 1 : -1; tau[k] = ti > 0.0f ? eff : -eff; dd[k] = dr[2]; over = true; }
                }
            if (__ballot(over) == 0ull) break;
        }
        }   // pass
        if (on) {       // v_free takes V's place (V is not read again in this substep)
            for (int k = 0; k < 6; ++k) sh_V[b][k] = fmaf(h, sh_a[b][k], sh_V[b][k]);
            if (b == 0) for (int k = 0; k < 6; ++k) { sh_V0[k] = fmaf(h, sh_a[0][k], sh_root[7 + k]); sh_V0[6 + k] = 0.0f; }
        }
        __syncthreads();

        PSTAMP(4); PACC(3);
        // ============================================================ 5 - 7a: the contact phases
        // What the joint tree passes of phase 7 need from here stays in the lanes of the env's half: the impulse each body collects
        // from its contact rows (pin, phase 7a), the env's contact count and deepest contact level.
        float pin[6] = {0, 0, 0, 0, 0, 0};
        int my_nc = 0, my_dmax = 0;
        const bool live = my_env >= 0;
        // ============================================================ 5. ground-contact candidates of BOTH envs (lane = 32 env + candidate mod 32)
        // three candidates per lane (`b`, `b + 32`, `b + 64`); their body-frame points are re-derived from the model each substep
        // (L2 hits).  What a candidate leaves for the contact list -- its contact point and, on a height field, the ground normal there --
        // waits in LDS instead of registers across the ballots.
        float *const sh_stage = lds + O_STAGE + half * (MAXCAND * 7);                               // [MAXCAND][7] per env
        float (*const sh_cdir)[9] = (float (*)[9])(lds + O_CDIR + half * 9 * MAXC);               // contact frames (height-field ground)
        int nc = 0;
        {
            const float *lws_m = d.lambda_ws + (size_t)senv * MAXCAND * 3;
            int cb[3]; float cdist[3]; bool act[3];
            for (int s = 0; s < 3; ++s) {
                const int c = b + 32 * s;
                cb[s] = -1; act[s] = false; cdist[s] = 0.0f;
                if (live && c < d.n_cand) {
                    const int cp = topo[EMLOCO_TOPO_CAND + c], body = cp & 0xff, k = (cp >> 8) & 0xff, gt = cp >> 16;
                    float ga[4], gb[4], clp[3];                      // geom a xyz, radius | geom b xyz
                    ld4(mdl, EMLOCO_MB_GEO + body * 8, ga); ld4(mdl, EMLOCO_MB_GEO + body * 8 + 4, gb);
                    const float crad = ga[3];
                    cb[s] = body;
                    if (gt == EMLOCO_GEOM_SPHERE) { clp[0] = ga[0]; clp[1] = ga[1]; clp[2] = ga[2]; }
                    else if (gt == EMLOCO_GEOM_CAPSULE) {
                        const float *src = k == 0 ? ga : gb;
                        clp[0] = src[0]; clp[1] = src[1]; clp[2] = src[2];
                    } else {
                        clp[0] = ga[0] + ((k & 1) ? gb[0] : -gb[0]);
                        clp[1] = ga[1] + ((k & 2) ? gb[1] : -gb[1]);
                        clp[2] = ga[2] + ((k & 4) ? gb[2] : -gb[2]);
                    }
                    float Rb[9], wp[3], cxw[3];
                    for (int k2 = 0; k2 < 9; ++k2) Rb[k2] = sh_R[body][k2];
                    matvec3(Rb, clp, wp);
                    const float z = sh_pq[body][2] + wp[2];
                    float *stg = sh_stage + c * 7;
                    if (!hf_on) {
                        cdist[s] = (z - prm.ground_z) - crad;
                        cxw[0] = sh_R[body][9] + wp[0];
                        cxw[1] = sh_R[body][10] + wp[1];
                        cxw[2] = (sh_R[body][11] + wp[2]) - crad;
                    } else {
                        // sphere of the candidate against the plane of the terrain triangle under its centre -- and, for a sphere with a
                        // radius, against the triangles under four probes one radius out along +-x / +-y: a neighbouring face (the ramp
                        // of a stair riser) is met when the sphere's SURFACE reaches it, not when its centre has crossed into the face's
                        // cell.  A probed triangle counts when the foot of the centre's perpendicular lies in it (its plane is not the
                        // terrain elsewhere) and its plane is nearer than what has been found; one contact per candidate, the nearest.
                        float zt, cnrm[3];
                        int tid0;
                        const float pcx = sh_pq[body][0] + wp[0], pcy = sh_pq[body][1] + wp[1];
                        mesh_plane(d, pcx, pcy, zt, cnrm, tid0);
                        float dperp = (z - zt) * cnrm[2];
                        if (crad > 0.0f)
                            for (int q = 0; q < 4; ++q) {
                                const float ex = q == 0 ? crad : (q == 1 ? 0.0f - crad : 0.0f), ey = q == 2 ? crad : (q == 3 ? 0.0f - crad : 0.0f);
                                float ztq, nq[3];
                                int tidq;
                                mesh_plane(d, pcx + ex, pcy + ey, ztq, nq, tidq);
                                const float dq = fmaf(z - ztq, nq[2], 0.0f - fmaf(ex, nq[0], ey * nq[1]));
                                int tidf;
                                if (d.hf_mv) { float zf, nf[3]; mesh_plane(d, pcx - dq * nq[0], pcy - dq * nq[1], zf, nf, tidf); }
                                else tidf = hf_triangle(d, pcx - dq * nq[0], pcy - dq * nq[1]);
                                if (tidq != tid0 && tidf == tidq && dq < dperp) { dperp = dq; cnrm[0] = nq[0]; cnrm[1] = nq[1]; cnrm[2] = nq[2]; }
                            }
                        if (d.hf_mv) {                                   // the corrected mesh's vertical faces
                            const float P[3] = {pcx, pcy, z};
                            mesh_walls(d, P, dperp, cnrm);
                        }
                        cdist[s] = dperp - crad;
                        for (int k2 = 0; k2 < 3; ++k2) { cxw[k2] = (sh_R[body][9 + k2] + wp[k2]) - crad * cnrm[k2]; stg[3 + k2] = cnrm[k2]; }
                    }
                    for (int k2 = 0; k2 < 3; ++k2) stg[k2] = cxw[k2];
                    act[s] = cdist[s] < prm.contact_offset;
                }
            }
            unsigned mh[3];                                           // the half's candidate masks
            for (int s = 0; s < 3; ++s) mh[s] = half_bits(__ballot(act[s]), half);
            nc = (int)(__popc(mh[0]) + __popc(mh[1]) + __popc(mh[2]));
            while (__ballot(nc > MAXC) != 0ull) {   // rare: drop the shallowest candidate (largest dist; ties -> highest candidate id) of the env(s) over the limit
                float best = -3.0e38f; int bid = -1;
                if (nc > MAXC)
                    for (int s = 0; s < 3; ++s)
                        if (act[s] && (cdist[s] > best || (cdist[s] == best && b + 32 * s > bid))) { best = cdist[s]; bid = b + 32 * s; }
                for (int off = 16; off >= 1; off >>= 1) {            // within the half
                    const float ob = __shfl_xor(best, off); const int oi = __shfl_xor(bid, off);
                    if (ob > best || (ob == best && oi > bid)) { best = ob; bid = oi; }
                }
                if (nc > MAXC) for (int s = 0; s < 3; ++s) if (bid == b + 32 * s) act[s] = false;
                for (int s = 0; s < 3; ++s) mh[s] = half_bits(__ballot(act[s]), half);
                nc = (int)(__popc(mh[0]) + __popc(mh[1]) + __popc(mh[2]));
            }
            // Warm start: the multipliers a candidate carried at the end of the previous substep (from the previous launch for
            // substep 0).  Every lane first fetches the old values of its own candidates -- old slot map and old multipliers are
            // still in place -- then, behind a barrier, the new contact list, its slot map and the warm values are written.
            float wl[3][3];
            for (int s = 0; s < 3; ++s) {
                wl[s][0] = wl[s][1] = wl[s][2] = 0.0f;
                const int c = b + 32 * s;
                if (act[s]) {
                    if (sub == 0) {
                        for (int k = 0; k < 3; ++k) wl[s][k] = lws_m[c * 3 + k];
                    } else {
                        const int os = sh_slot[c];
                        if (os != 255) for (int k = 0; k < 3; ++k) wl[s][k] = sh_lam[3 * os + k];
                    }
                }
            }
            __syncthreads();
            if (live) {
                const unsigned below = (1u << b) - 1u;
                const int base[3] = {0, (int)__popc(mh[0]), (int)(__popc(mh[0]) + __popc(mh[1]))};
                for (int s = 0; s < 3; ++s) {
                    const int c = b + 32 * s, ci = base[s] + (int)__popc(mh[s] & below);
                    sh_slot[c] = act[s] ? (unsigned char)ci : (unsigned char)255;
                    if (act[s]) {
                        const float *stg = sh_stage + c * 7;              // this lane's own entry: no other lane touches it
                        sh_cbody[ci] = (unsigned char)cb[s]; sh_cdist[ci] = cdist[s];
                        for (int k = 0; k < 3; ++k) { sh_cx[ci][k] = stg[k]; sh_lam[3 * ci + k] = wl[s][k]; }
                        if (hf_on) {      // frame: normal, t1 = (y x n) / |y x n|, t2 = n x t1
                            const float n[3] = {stg[3], stg[4], stg[5]};
                            const float l2 = fmaf(n[2], n[2], n[0] * n[0]);
                            float *D = sh_cdir[ci];
                            D[0] = n[0]; D[1] = n[1]; D[2] = n[2];
                            if (l2 >= 0.1f) {
                                const float il = 1.0f / sqrtf(l2);
                                const float t1x = n[2] * il, t1z = 0.0f - n[0] * il;
                                D[3] = t1x; D[4] = 0.0f; D[5] = t1z;
                                D[6] = n[1] * t1z; D[7] = fmaf(n[2], t1x, -(n[0] * t1z)); D[8] = 0.0f - n[1] * t1x;
                            } else {      // a face looking along y (a riser across the y axis): t1 = (n x x) / |n x x|, t2 = n x t1
                                const float il = 1.0f / sqrtf(fmaf(n[2], n[2], n[1] * n[1]));
                                const float t1y = n[2] * il, t1z = 0.0f - n[1] * il;
                                D[3] = 0.0f; D[4] = t1y; D[5] = t1z;
                                D[6] = fmaf(n[1], t1z, -(n[2] * t1y)); D[7] = 0.0f - n[0] * t1z; D[8] = n[0] * t1y;
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (live) work += nc > 0 ? 10 + nc : 0;
        my_nc = nc;
        const int nc0 = __builtin_amdgcn_readlane(nc, 0), nc1 = __builtin_amdgcn_readlane(nc, 32);      // wave-uniform
        PSTAMP(5); PACC(4);

        if (nc0 <= PAIR_FAST_MAXC && nc1 <= PAIR_FAST_MAXC) {
            // ======================================================== FAST PATH (86 % of the env-substeps of the bench workload have <= 10
            // contacts, profiles/r06_contact_histogram.txt): an env's <= 30 rows fit the 32 lanes of its half, so the row phases run
            // for BOTH envs at once too -- lane = 32 env + row -- and a pair's substep costs about what one env's did.
            constexpr int FR = 3 * PAIR_FAST_MAXC;                   // rows per env on this path
            const int nr = 3 * nc;                                    // (uniform within a half)
            // ---------------------------------------------------- 6a. rows: Jacobian, rhs, chain propagation
            float J[6] = {0, 0, 0, 0, 0, 0}, rhs = 0.0f, lam = 0.0f;
            float ys[YLEN];
            if (b < NB) { sh_crange[b] = 0; sh_crange[NB + b] = -1; }
            __syncthreads();
            if (b < nc) {
                const int cb_ = sh_cbody[b];
                if (b == 0 || sh_cbody[b - 1] != cb_) sh_crange[cb_] = (signed char)b;
                if (b == nc - 1 || sh_cbody[b + 1] != cb_) sh_crange[NB + cb_] = (signed char)b;
            }
            const int myc = b / 3, myd = b - 3 * myc;
            int rbody = 0, rdep = 0;
            unsigned code = 0u;
            float p[6] = {0, 0, 0, 0, 0, 0};
            float dir[3] = {myd == 1 ? 1.0f : 0.0f, myd == 2 ? 1.0f : 0.0f, myd == 0 ? 1.0f : 0.0f};
            if (b < nr) {
                rbody = sh_cbody[myc];
                if (hf_on) for (int k = 0; k < 3; ++k) dir[k] = sh_cdir[myc][3 * myd + k];
                float x[3] = {sh_cx[myc][0], sh_cx[myc][1], sh_cx[myc][2]};
                cross3(x, dir, J);
                J[3] = dir[0]; J[4] = dir[1]; J[5] = dir[2];
                float Vb[6];
                for (int k = 0; k < 6; ++k) Vb[k] = sh_V[rbody][k];                  // v_free (end of phase 4)
                const float vel = dot6(J, Vb);
                float bias = 0.0f;
                if (myd == 0) {
                    const float dist = sh_cdist[myc];
                    if (dist > 0.0f) bias = dist / h;
                    else { bias = prm.erp * dist / h; if (bias < -prm.max_depen_vel) bias = -prm.max_depen_vel; }
                }
                rhs = vel + bias;
                for (int k = 0; k < 6; ++k) p[k] = -J[k];
                rdep = PD_DEPTH(sh_pd[rbody]);
            }
            int rdmax = rdep;
            for (int off = 16; off >= 1; off >>= 1) { const int o = __shfl_xor(rdmax, off); rdmax = o > rdmax ? o : rdmax; }   // within the half
            my_dmax = rdmax;                                         // deepest chain among the env's contact bodies
            const int dm0 = __builtin_amdgcn_readlane(rdmax, 0), dm1 = __builtin_amdgcn_readlane(rdmax, 32);
            const int dmax = dm0 > dm1 ? dm0 : dm1;                  // wave-uniform bound of the level loops
            {
                int ci = rbody;
#pragma unroll
                for (int lev = 7; lev >= 0; --lev) {
                    if (lev < dmax) {
                        if (b < nr && lev < rdep) {
                            const int i = ci;
                            float Ri[9], ri[3], u[3], uhh[3];
                            for (int k = 0; k < 9; ++k) Ri[k] = sh_R[i][k];
                            for (int k = 0; k < 3; ++k) ri[k] = sh_R[i][9 + k];
                            for (int a = 0; a < 3; ++a) {
                                float ax[3] = {Ri[a], Ri[3 + a], Ri[6 + a]}, sl[3];
                                cross3(ri, ax, sl);
                                const float Sa[6] = {ax[0], ax[1], ax[2], sl[0], sl[1], sl[2]};
                                u[a] = -dot6(Sa, p);
                            }
                            const float *W = sh_W[i], *K = W + 18;
                            uhh[0] = K[0] * u[0]; uhh[1] = SOP2(K[1], u[0], K[2], u[1]); uhh[2] = SOP3(K[3], u[0], K[4], u[1], K[5], u[2]);
                            const int pdi = sh_pd[i];
                            code |= (unsigned)(PD_SLOT(pdi) + 1) << (3 * lev);
                            ys[6 + 3 * lev] = uhh[0]; ys[6 + 3 * lev + 1] = uhh[1]; ys[6 + 3 * lev + 2] = uhh[2];
                            for (int k = 0; k < 6; ++k) p[k] = ADD_SOP3(p[k], W[k * 3], uhh[0], W[k * 3 + 1], uhh[1], W[k * 3 + 2], uhh[2]);
                            ci = PD_PARENT(pdi);
                        }
                    }
                }
            }
            if (b < nr) {
                for (int a = 0; a < 6; ++a) {   // L0 y = p
                    float acc = p[a];
                    for (int k = 0; k < a; ++k) acc = fmaf(-sh_L0[a * 6 + k], ys[k], acc);
                    ys[a] = acc * sh_L0i[a];
                }
                lam = prm.warm * sh_lam[b];
            }
            __syncthreads();
            PSTAMP(6); PACC(5);
            // ---------------------------------------------------- 6b. both contact matrices in one pass: env 0 = tile (0,0), env 1 = tile (1,1)
            // The operand swap hands lanes 0-31 env 0's rows (k even | odd) and lanes 32-63 env 1's: acc0 += Y0 Y0^T, acc1 += Y1 Y1^T.
            // The masked level blocks walk the chain bodies of BOTH envs together (one group of each per step; an env that has
            // run out of groups at a level adds exact zeros).  Rows to registers as in the full-size path: 16 swaps.
            float A[32];
            {
                typedef float sim_f32x16 __attribute__((vector_size(64)));
                const int own_dep = (b < nr) ? rdep : -1;        // -1: no row in this lane (all operands zero)
                const bool has_row = own_dep >= 0;
                sim_f32x16 acc0, acc1;
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
#define GRAM_STEP(V0, V1)                                                                                              \
                {                                                                                                      \
                    const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(V0), __float_as_uint(V1), false, false); \
                    const float op0_ = __uint_as_float(sw_[0]), op1_ = __uint_as_float(sw_[1]);                        \
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(op0_, op0_, acc0, 0, 0, 0);                            \
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(op1_, op1_, acc1, 0, 0, 0);                            \
                }
                GRAM_STEP(has_row ? ys[0] : 0.0f, has_row ? ys[1] : 0.0f) GRAM_STEP(has_row ? ys[2] : 0.0f, has_row ? ys[3] : 0.0f)
                GRAM_STEP(has_row ? ys[4] : 0.0f, has_row ? ys[5] : 0.0f)
#pragma unroll
                for (int lev = 0; lev < 8; ++lev) {
                    if (lev < dmax) {                               // wave-uniform
                        // this row's chain body at the level, as its index within the level + 1 (0: the chain ends above it)
                        const int gid = (lev < own_dep) ? (int)((code >> (3 * lev)) & 7u) : 0;
                        unsigned long long rem = __ballot(gid != 0);
                        while (rem != 0ull) {
                            const unsigned lo = (unsigned)rem, hi = (unsigned)(rem >> 32);
                            const int g0 = lo ? __builtin_amdgcn_readlane(gid, __builtin_ctz(lo)) : 0;
                            const int g1 = hi ? __builtin_amdgcn_readlane(gid, 32 + __builtin_ctz(hi)) : 0;
                            const bool in_g = gid != 0 && gid == (half ? g1 : g0);
                            GRAM_STEP(in_g ? ys[6 + 3 * lev] : 0.0f, in_g ? ys[7 + 3 * lev] : 0.0f) GRAM_STEP(in_g ? ys[8 + 3 * lev] : 0.0f, 0.0f)
                            rem &= ~__ballot(in_g);
                        }
                    }
                }
#undef GRAM_STEP
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c0 = (r & 3) + 8 * (r >> 2);
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[r]), __float_as_uint(acc1[r]), false, false);
                    A[c0] = __uint_as_float(sw[0]); A[c0 + 4] = __uint_as_float(sw[1]);
                }
            }
            PSTAMP(7); PACC(6);
            // ---------------------------------------------------- 6c. projected Gauss-Seidel, both envs in lock step (lane 32 e + s holds row s of env e)
            // The leader-lane sweep of the full-size path; a broadcast from row r is one v_readlane per half and a select.
            const int lb = 32 * half;
#define HB(x, r) (half ? lane_bcast((x), 32 + (r)) : lane_bcast((x), (r)))
            const int ncmax = nc0 > nc1 ? nc0 : nc1;
            const int lr0 = (b < nr) ? b - myd : 0;                  // first row of this lane's contact (idle lanes shadow row 0)
            float w = rhs;
            float gl0 = __shfl(lam, lb + lr0), gl1 = __shfl(lam, lb + lr0 + 1), gl2 = __shfl(lam, lb + lr0 + 2);
            float diag = 1.0f, gA10 = 0.0f, gA20 = 0.0f, gA21 = 0.0f;
#pragma unroll
            for (int c = 0; c < PAIR_FAST_MAXC; ++c)
                if (c < ncmax) {                                     // warm start (rows beyond an env's own carry zero multipliers: skipped)
                    const int r0 = 3 * c;
                    const float a0 = A[r0], a1 = A[r0 + 1], a2 = A[r0 + 2];
                    const float l0 = HB(lam, r0), l1 = HB(lam, r0 + 1), l2 = HB(lam, r0 + 2);
                    const float u0 = fmaf(a0, l0, w);
                    w = (l0 != 0.0f) ? u0 : w;
                    const float u1 = fmaf(a1, l1, w);
                    w = (l1 != 0.0f) ? u1 : w;
                    const float u2 = fmaf(a2, l2, w);
                    w = (l2 != 0.0f) ? u2 : w;
                    diag = b == r0 ? a0 : (b == r0 + 1 ? a1 : (b == r0 + 2 ? a2 : diag));
                    const float t21 = HB(a2, r0 + 1);                // A[r0 + 1][r0 + 2]
                    if (b == r0) { gA10 = a1; gA20 = a2; gA21 = t21; }
                }
            const float ainv = (b < nr) ? 1.0f / (diag * (1.0f + prm.cfm)) : 0.0f;
            const float gi1 = __shfl(ainv, lb + lr0 + 1), gi2 = __shfl(ainv, lb + lr0 + 2);
            PSTAMP(11); PACC(7);
            for (int it = 0; it < prm.n_iter; ++it) {
#pragma unroll
                for (int c = 0; c < PAIR_FAST_MAXC; ++c)
                    if (c < ncmax) {
                        const int r0 = 3 * c;
                        const bool act = c < nc;                     // (uniform within a half)
                        const float a0 = A[r0], a1 = A[r0 + 1], a2 = A[r0 + 2];
                        const float w1s = HB(w, r0 + 1), w2s = HB(w, r0 + 2);
                        // branch-free: every lane runs the leader's chain on its own contact's values, only row r0's results are read
                        float nl0 = fmaf(-w, ainv, gl0);
                        if (nl0 < 0.0f) nl0 = 0.0f;
                        const float d0 = nl0 - gl0;
                        const float w1 = fmaf(gA10, d0, w1s);
                        const float nl1 = fmaf(-w1, gi1, gl1);
                        const float d1 = nl1 - gl1;
                        const float w2 = fmaf(gA21, d1, fmaf(gA20, d0, w2s));
                        const float nl2 = fmaf(-w2, gi2, gl2);
                        const float d2 = nl2 - gl2;
                        const bool me = act && b == r0;
                        gl0 = me ? nl0 : gl0; gl1 = me ? nl1 : gl1; gl2 = me ? nl2 : gl2;
                        const float lim = prm.mu * nl0;
                        const float m2 = fmaf(nl1, nl1, nl2 * nl2);
                        float wn = fmaf(a0, HB(d0, r0), w);
                        wn = fmaf(a1, HB(d1, r0), wn);
                        wn = fmaf(a2, HB(d2, r0), wn);
                        w = act ? wn : w;
                        const bool out = me && m2 > lim * lim;
                        const unsigned long long cone = __ballot(out);
                        if (__builtin_expect(cone != 0ull, 0)) {     // wave-uniform: a contact outside its friction cone (in either env)
                            const bool mine_out = half_bits(cone, half) != 0u;
                            const float sc = lim / sqrtf(m2);
                            const float n1 = nl1 * sc, n2 = nl2 * sc;
                            gl1 = out ? n1 : gl1; gl2 = out ? n2 : gl2;
                            const float e1 = HB(n1 - nl1, r0), e2 = HB(n2 - nl2, r0);
                            const float wc = fmaf(a2, e2, fmaf(a1, e1, w));
                            w = mine_out ? wc : w;
                        }
                    }
                if (it == 0) PSTAMP(12);
            }
#undef HB
            if (b < nr && myd == 0) { sh_lam[b] = gl0; sh_lam[b + 1] = gl1; sh_lam[b + 2] = gl2; }
            else if (b >= nr) sh_lam[b] = 0.0f;
            if (b + 32 < MAXR) sh_lam[b + 32] = 0.0f;                 // (nr <= 30: the rows beyond the half hold no contact)
            __syncthreads();
            lam = (b < nr) ? sh_lam[b] : 0.0f;
            PSTAMP(8); PACC(8);
            // ---------------------------------------------------- 7a. impulses: what each body collects from its contact rows
            float *cf_m = d.contact_force + (size_t)senv * NB * 3;
            if (live && b < NB && last && nc == 0)
                for (int k = 0; k < 3; ++k) cf_m[b * 3 + k] = 0.0f;
            float (*sh_row)[12] = (float (*)[12])(lds + O_ROWS + half * (32 * 12));
            if (b < nr) {
                const float x[3] = {sh_cx[myc][0], sh_cx[myc][1], sh_cx[myc][2]};
                float Jr[3];
                cross3(x, dir, Jr);
                for (int k = 0; k < 3; ++k) { sh_row[b][k] = Jr[k]; sh_row[b][3 + k] = dir[k]; }
                if (last) for (int k = 0; k < 3; ++k) sh_row[b][6 + k] = dir[k] * lam / h;
            }
            __syncthreads();
            if (live && b < NB && nc > 0) {
                float cf[3] = {0, 0, 0};
                const int r_end = 3 * sh_crange[NB + b] + 3;
                for (int r = 3 * sh_crange[b]; r < r_end; ++r) {
                    const float l = sh_lam[r];
                    for (int k = 0; k < 6; ++k) pin[k] = fmaf(-sh_row[r][k], l, pin[k]);
                    if (last) for (int k = 0; k < 3; ++k) cf[k] += sh_row[r][6 + k];
                }
                if (last) for (int k = 0; k < 3; ++k) cf_m[b * 3 + k] = cf[k];
            }
            {   // momentum the system must have after this substep: gravity and the contact impulses are the only external ones
                float imp[3];
                for (int k = 0; k < 3; ++k) imp[k] = half_sum(b < nr ? dir[k] * lam : 0.0f, half);
                if (live && b == 0) {
                    for (int k = 0; k < 3; ++k) sh_P[k] = sh_P[3 + k] + (nc > 0 ? imp[k] : 0.0f);
                    sh_P[2] = fmaf(sh_P[6] * prm.gravity_z, h, sh_P[2]);
                }
            }
            {   // angular momentum about the centre of mass after this substep: the moments of the contact impulses, then the damping
                const float damp_ = 1.0f / (1.0f + h * prm.ang_damping);
                float t[3] = {0.0f, 0.0f, 0.0f}, tq[3];
                if (b < nr) {
                    float arm[3], ip[3];
                    for (int k = 0; k < 3; ++k) { arm[k] = sh_cx[myc][k] - sh_L[8 + k]; ip[k] = dir[k] * lam; }
                    cross3(arm, ip, t);
                }
                for (int k = 0; k < 3; ++k) tq[k] = half_sum(t[k], half);
                if (live && b == 0) for (int k = 0; k < 3; ++k) sh_L[k] = (sh_L[4 + k] + (nc > 0 ? tq[k] : 0.0f)) * damp_;
            }
            __syncthreads();
            PACC(9);
        } else {
            // ======================================================== FULL-SIZE PATH: an env with more than 10 contacts needs the whole wave for
            // its rows; the two envs go one after the other (a loop of two, not unrolled), the matrix in LDS as the packed lower triangle
#pragma nounroll
            for (int e = 0; e < 2; ++e) {
                const int env = e ? env1 : env0;
                if (env < 0) continue;
                ENV_VIEW(lds + e * PW, lds + (e ? O_B1 : O_B0), lds + (e ? O_AA1 : O_AA0), lds + (e ? O_IA1 : O_IA0))
                float *cf_env = d.contact_force + (size_t)env * NB * 3;
                float *const sh_A = lds + (e ? O_B1 : O_B0);           // contact matrix, lower triangle: (r, s<=r) at r(r+1)/2 + s
                float (*const sh_cdir)[9] = (float (*)[9])(lds + O_CDIR + e * 9 * MAXC);
                const bool mine = half == e;                          // this lane's half carries env e's bodies in the joint phases
                const int nc = e ? nc1 : nc0, nr = 3 * nc;
                // ------------------------------------------------ 6a. rows: Jacobian, rhs, chain propagation (lane = row)
                float J[6] = {0, 0, 0, 0, 0, 0}, rhs = 0.0f, lam = 0.0f;
                float ys[YLEN];
                if (lane < NB) { sh_crange[lane] = 0; sh_crange[NB + lane] = -1; }
                __syncthreads();
                if (lane < nc) {
                    const int cb_ = sh_cbody[lane];
                    if (lane == 0 || sh_cbody[lane - 1] != cb_) sh_crange[cb_] = (signed char)lane;
                    if (lane == nc - 1 || sh_cbody[lane + 1] != cb_) sh_crange[NB + cb_] = (signed char)lane;
                }
                const int myc = lane / 3, myd = lane - 3 * myc;
                int rbody = 0, rdep = 0;
                unsigned code = 0u;
                float p[6] = {0, 0, 0, 0, 0, 0};
                float dir[3] = {myd == 1 ? 1.0f : 0.0f, myd == 2 ? 1.0f : 0.0f, myd == 0 ? 1.0f : 0.0f};
                if (lane < nr) {
                    rbody = sh_cbody[myc];
                    if (hf_on) for (int k = 0; k < 3; ++k) dir[k] = sh_cdir[myc][3 * myd + k];
                    float x[3] = {sh_cx[myc][0], sh_cx[myc][1], sh_cx[myc][2]};
                    cross3(x, dir, J);
                    J[3] = dir[0]; J[4] = dir[1]; J[5] = dir[2];
                    float Vb[6];
                    for (int k = 0; k < 6; ++k) Vb[k] = sh_V[rbody][k];                  // v_free (end of phase 4)
                    const float vel = dot6(J, Vb);
                    float bias = 0.0f;
                    if (myd == 0) {
                        const float dist = sh_cdist[myc];
                        if (dist > 0.0f) bias = dist / h;
                        else { bias = prm.erp * dist / h; if (bias < -prm.max_depen_vel) bias = -prm.max_depen_vel; }
                    }
                    rhs = vel + bias;
                    for (int k = 0; k < 6; ++k) p[k] = -J[k];
                    rdep = PD_DEPTH(sh_pd[rbody]);
                }
                int rdmax = rdep;
                for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(rdmax, off); rdmax = o > rdmax ? o : rdmax; }
                const int dmax = __builtin_amdgcn_readfirstlane(rdmax);              // deepest chain among the contact bodies
                if (mine) my_dmax = dmax;
                {
                    int ci = rbody;
#pragma unroll
                    for (int lev = 7; lev >= 0; --lev) {
                        if (lev < dmax) {
                            if (lane < nr && lev < rdep) {
                                const int i = ci;
                                float Ri[9], ri[3], u[3], uhh[3];
                                for (int k = 0; k < 9; ++k) Ri[k] = sh_R[i][k];
                                for (int k = 0; k < 3; ++k) ri[k] = sh_R[i][9 + k];
                                for (int a = 0; a < 3; ++a) {
                                    float ax[3] = {Ri[a], Ri[3 + a], Ri[6 + a]}, sl[3];
                                    cross3(ri, ax, sl);
                                    const float Sa[6] = {ax[0], ax[1], ax[2], sl[0], sl[1], sl[2]};
                                    u[a] = -dot6(Sa, p);
                                }
                                const float *W = sh_W[i], *K = W + 18;
                                uhh[0] = K[0] * u[0]; uhh[1] = SOP2(K[1], u[0], K[2], u[1]); uhh[2] = SOP3(K[3], u[0], K[4], u[1], K[5], u[2]);
                                const int pdi = sh_pd[i];
                                code |= (unsigned)(PD_SLOT(pdi) + 1) << (3 * lev);
                                ys[6 + 3 * lev] = uhh[0]; ys[6 + 3 * lev + 1] = uhh[1]; ys[6 + 3 * lev + 2] = uhh[2];
                                for (int k = 0; k < 6; ++k) p[k] = ADD_SOP3(p[k], W[k * 3], uhh[0], W[k * 3 + 1], uhh[1], W[k * 3 + 2], uhh[2]);
                                ci = PD_PARENT(pdi);
                            }
                        }
                    }
                }
                if (lane < nr) {
                    for (int a = 0; a < 6; ++a) {   // L0 y = p
                        float acc = p[a];
                        for (int k = 0; k < a; ++k) acc = fmaf(-sh_L0[a * 6 + k], ys[k], acc);
                        ys[a] = acc * sh_L0i[a];
                    }
                    lam = prm.warm * sh_lam[lane];
                }
                __syncthreads();
                // ------------------------------------------------ 6b. contact matrix: the three tiles in one pass, to LDS
                {
                    typedef float sim_f32x16 __attribute__((vector_size(64)));
                    const int hh = lane >> 5, j31 = lane & 31;
                    const int own_dep = (lane < nr) ? rdep : -1;   // -1: no row in this lane (all operands zero)
                    const bool has_row = own_dep >= 0;
                    const bool big = nr > 32;                      // wave-uniform: rows beyond the first tile
                    sim_f32x16 acc00, acc10, acc11;
                    for (int r = 0; r < 16; ++r) { acc00[r] = 0.0f; acc10[r] = 0.0f; acc11[r] = 0.0f; }
#define GRAM_STEP(V0, V1)                                                                                              \
                    {                                                                                                      \
                        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(V0), __float_as_uint(V1), false, false); \
                        const float op0_ = __uint_as_float(sw_[0]), op1_ = __uint_as_float(sw_[1]);                        \
                        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(op0_, op0_, acc00, 0, 0, 0);                          \
                        if (big) {                                                                                         \
                            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(op1_, op0_, acc10, 0, 0, 0);                      \
                            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(op1_, op1_, acc11, 0, 0, 0);                      \
                        }                                                                                                  \
                    }
                    GRAM_STEP(has_row ? ys[0] : 0.0f, has_row ? ys[1] : 0.0f) GRAM_STEP(has_row ? ys[2] : 0.0f, has_row ? ys[3] : 0.0f)
                    GRAM_STEP(has_row ? ys[4] : 0.0f, has_row ? ys[5] : 0.0f)
#pragma unroll
                    for (int lev = 0; lev < 8; ++lev) {
                        if (lev < dmax) {                               // wave-uniform
                            const int gid = (lev < own_dep) ? (int)((code >> (3 * lev)) & 7u) : 0;
                            unsigned long long rem = __ballot(gid != 0);
                            while (rem != 0ull) {
                                const int first = __builtin_ctzll(rem);
                                const int g = __builtin_amdgcn_readlane(gid, first);
                                const bool in_g = gid == g;
                                GRAM_STEP(in_g ? ys[6 + 3 * lev] : 0.0f, in_g ? ys[7 + 3 * lev] : 0.0f) GRAM_STEP(in_g ? ys[8 + 3 * lev] : 0.0f, 0.0f)
                                rem &= ~__ballot(in_g);
                            }
                        }
                    }
#undef GRAM_STEP
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (row < nr && j31 <= row) sh_A[row * (row + 1) / 2 + j31] = acc00[r];
                        if (big) {
                            const int row1 = 32 + row;
                            if (row1 < nr) sh_A[row1 * (row1 + 1) / 2 + j31] = acc10[r];
                            if (row1 < nr && 32 + j31 <= row1) sh_A[row1 * (row1 + 1) / 2 + 32 + j31] = acc11[r];
                        }
                    }
                }
                __syncthreads();
                // ------------------------------------------------ 6c. projected Gauss-Seidel (lane s holds lambda_s), leader-lane sweep
                const int ls = lane < nr ? lane : 0;                      // idle lanes shadow lane 0 (their w is never used)
                const int tri_s = ls * (ls + 1) / 2;
                const float ainv = (lane < nr) ? 1.0f / (sh_A[tri_s + ls] * (1.0f + prm.cfm)) : 0.0f;
#define A_OF(rr) sh_A[tri_index(ls, (rr))]
                float w = rhs;
                {                                                          // warm start (matrix entries read one contact ahead)
                    float n0 = 0.0f, n1a = 0.0f, n2a = 0.0f;
                    if (nc > 0) { n0 = A_OF(0); n1a = A_OF(1); n2a = A_OF(2); }
                    for (int c = 0; c < nc; ++c) {
                        const int r0 = 3 * c;
                        const float a0 = n0, a1 = n1a, a2 = n2a;
                        if (c + 1 < nc) { n0 = A_OF(r0 + 3); n1a = A_OF(r0 + 4); n2a = A_OF(r0 + 5); }
                        const float l0 = lane_bcast(lam, r0), l1 = lane_bcast(lam, r0 + 1), l2 = lane_bcast(lam, r0 + 2);
                        const float u0 = fmaf(a0, l0, w);
                        w = (l0 != 0.0f) ? u0 : w;
                        const float u1 = fmaf(a1, l1, w);
                        w = (l1 != 0.0f) ? u1 : w;
                        const float u2 = fmaf(a2, l2, w);
                        w = (l2 != 0.0f) ? u2 : w;
                    }
                }
                const int lr0 = ls - (lane < nr ? myd : 0);               // first row of this lane's contact
                float gl0 = __shfl(lam, lr0), gl1 = __shfl(lam, lr0 + 1), gl2 = __shfl(lam, lr0 + 2);
                const float gi1 = __shfl(ainv, lr0 + 1), gi2 = __shfl(ainv, lr0 + 2);
                const float gA10 = sh_A[tri_index(lr0 + 1, lr0)], gA20 = sh_A[tri_index(lr0 + 2, lr0)], gA21 = sh_A[tri_index(lr0 + 2, lr0 + 1)];
                const bool leader = lane < nr && myd == 0;
                for (int it = 0; it < prm.n_iter; ++it) {
                    float n0 = 0.0f, n1a = 0.0f, n2a = 0.0f;
                    if (nc > 0) { n0 = A_OF(0); n1a = A_OF(1); n2a = A_OF(2); }
                    for (int c = 0; c < nc; ++c) {
                        const int r0 = 3 * c;
                        const float a0 = n0, a1 = n1a, a2 = n2a;
                        if (c + 1 < nc) { n0 = A_OF(r0 + 3); n1a = A_OF(r0 + 4); n2a = A_OF(r0 + 5); }
                        const float w1s = lane_bcast(w, r0 + 1), w2s = lane_bcast(w, r0 + 2);
                        float nl0 = fmaf(-w, ainv, gl0);
                        if (nl0 < 0.0f) nl0 = 0.0f;
                        const float d0 = nl0 - gl0;
                        const float w1 = fmaf(gA10, d0, w1s);
                        const float nl1 = fmaf(-w1, gi1, gl1);
                        const float d1 = nl1 - gl1;
                        const float w2 = fmaf(gA21, d1, fmaf(gA20, d0, w2s));
                        const float nl2 = fmaf(-w2, gi2, gl2);
                        const float d2 = nl2 - gl2;
                        const bool me = lane == r0;
                        gl0 = me ? nl0 : gl0; gl1 = me ? nl1 : gl1; gl2 = me ? nl2 : gl2;
                        const float lim = prm.mu * nl0;
                        const float m2 = fmaf(nl1, nl1, nl2 * nl2);
                        w = fmaf(a0, lane_bcast(d0, r0), w);
                        w = fmaf(a1, lane_bcast(d1, r0), w);
                        w = fmaf(a2, lane_bcast(d2, r0), w);
                        if (__builtin_expect(__ballot(me && m2 > lim * lim) != 0ull, 0)) {   // wave-uniform: outside the friction cone
                            const float sc = lim / sqrtf(m2);
                            const float n1 = nl1 * sc, n2 = nl2 * sc;
                            gl1 = me ? n1 : gl1; gl2 = me ? n2 : gl2;
                            w = fmaf(a1, lane_bcast(n1 - nl1, r0), w);
                            w = fmaf(a2, lane_bcast(n2 - nl2, r0), w);
                        }
                    }
                }
#undef A_OF
                if (leader) { sh_lam[lane] = gl0; sh_lam[lane + 1] = gl1; sh_lam[lane + 2] = gl2; }
                else if (lane >= nr && lane < MAXR) sh_lam[lane] = 0.0f;
                __syncthreads();
                lam = (lane < nr) ? sh_lam[lane] : 0.0f;
                // ------------------------------------------------ 7a. impulses: what each body collects from its contact rows
                if ((lane < NB) && last && nc == 0)
                    for (int k = 0; k < 3; ++k) cf_env[lane * 3 + k] = 0.0f;
                if (nc > 0) {
                    float (*sh_row)[12] = (float (*)[12])sh_A;        // (the matrix's place: dead by now)
                    if (lane < nr) {
                        const float x[3] = {sh_cx[myc][0], sh_cx[myc][1], sh_cx[myc][2]};
                        float Jr[3];
                        cross3(x, dir, Jr);
                        for (int k = 0; k < 3; ++k) { sh_row[lane][k] = Jr[k]; sh_row[lane][3 + k] = dir[k]; }
                        if (last) for (int k = 0; k < 3; ++k) sh_row[lane][6 + k] = dir[k] * lam / h;
                    }
                    __syncthreads();
                    if (mine && b < NB) {
                        float cf[3] = {0, 0, 0};
                        const int r_end = 3 * sh_crange[NB + b] + 3;
                        for (int r = 3 * sh_crange[b]; r < r_end; ++r) {
                            const float l = sh_lam[r];
                            for (int k = 0; k < 6; ++k) pin[k] = fmaf(-sh_row[r][k], l, pin[k]);
                            if (last) for (int k = 0; k < 3; ++k) cf[k] += sh_row[r][6 + k];
                        }
                        if (last) for (int k = 0; k < 3; ++k) cf_env[b * 3 + k] = cf[k];
                    }
                }
                {   // momentum the system must have after this substep
                    float imp[3] = {0.0f, 0.0f, 0.0f};
                    if (nc > 0) for (int k = 0; k < 3; ++k) imp[k] = wave_sum(lane < nr ? dir[k] * lam : 0.0f);
                    if (lane == 0) {
                        for (int k = 0; k < 3; ++k) sh_P[k] = sh_P[3 + k] + imp[k];
                        sh_P[2] = fmaf(sh_P[6] * prm.gravity_z, h, sh_P[2]);
                    }
                }
                {   // angular momentum about the centre of mass after this substep
                    const float damp_ = 1.0f / (1.0f + h * prm.ang_damping);
                    float tq[3] = {0.0f, 0.0f, 0.0f};
                    if (nc > 0) {
                        float t[3] = {0.0f, 0.0f, 0.0f};
                        if (lane < nr) {
                            float arm[3], ip[3];
                            for (int k = 0; k < 3; ++k) { arm[k] = sh_cx[myc][k] - sh_L[8 + k]; ip[k] = dir[k] * lam; }
                            cross3(arm, ip, t);
                        }
                        for (int k = 0; k < 3; ++k) tq[k] = wave_sum(t[k]);
                    }
                    if (lane == 0) for (int k = 0; k < 3; ++k) sh_L[k] = (sh_L[4 + k] + tq[k]) * damp_;
                }
                __syncthreads();                                       // the matrix's place is reused by the partner env / phase 7
            }
            PACC(10);
        }

        // ============================================================ 7. impulses -> velocity change (second solve): tree passes, both envs
        float dq[3] = {0, 0, 0};
        {
            const bool act7 = on && my_nc > 0;                      // this lane's env has contacts (uniform within a half)
            // bodies deeper than every contact body carry no impulse and have no loaded descendant: their share of the up pass
            // is exactly zero (uh = +0, pa = +0), so an env's pass starts at its deepest contact level
            const int top7 = act7 ? (d.max_depth < my_dmax ? d.max_depth : my_dmax) : -1;
            const int t0 = __builtin_amdgcn_readlane(top7, 0), t1 = __builtin_amdgcn_readlane(top7, 32);
            const int lev_hi = t0 > t1 ? t0 : t1;                   // wave-uniform
            if (lev_hi >= 0) {
                const int pd7 = sh_pd[bb];
                if (act7 && PD_DEPTH(pd7) > my_dmax) { uh[0] = uh[1] = uh[2] = 0.0f; for (int k = 0; k < 6; ++k) sh_pa[b][k] = 0.0f; }
                __syncthreads();
                for (int lev = lev_hi; lev >= 0; --lev) {
                    if (act7 && PD_DEPTH(pd7) == lev && lev <= top7) {
                        for (int k = 0; k < 6; ++k) pA[k] = pin[k];
                        for (int ci = 0; ci < 3; ++ci) {
                            const int ch = PD_CHILD(pd7, ci);
                            if (ch != 31) for (int k = 0; k < 6; ++k) pA[k] += sh_pa[ch][k];
                        }
                        if (lev > 0) {
                            float u[3], Wm[18], Km[6];
                            float R[9], r[3], Sl[3][3];
                            for (int k = 0; k < 9; ++k) R[k] = sh_R[b][k];
                            for (int k = 0; k < 3; ++k) r[k] = sh_R[b][9 + k];
                            for (int c = 0; c < 3; ++c) { const float ax[3] = {R[c], R[3 + c], R[6 + c]}; cross3(r, ax, Sl[c]); }
                            for (int k = 0; k < 18; ++k) Wm[k] = sh_W[b][k];
                            for (int k = 0; k < 6; ++k) Km[k] = sh_W[b][18 + k];
                            for (int c = 0; c < 3; ++c) {
                                const float Sc[6] = {R[c], R[3 + c], R[6 + c], Sl[c][0], Sl[c][1], Sl[c][2]};
                                u[c] = 0.0f - fdot6(Sc, pA);
                            }
                            uh[0] = Km[0] * u[0];
                            uh[1] = SOP2(Km[1], u[0], Km[2], u[1]);
                            uh[2] = SOP3(Km[3], u[0], Km[4], u[1], Km[5], u[2]);
                            for (int k = 0; k < 6; ++k)
                                sh_pa[b][k] = ADD_SOP3(pA[k], Wm[k * 3], uh[0], Wm[k * 3 + 1], uh[1], Wm[k * 3 + 2], uh[2]);
                        } else {
                            float y[6], x[6];
                            for (int a = 0; a < 6; ++a) {
                                float acc = -pA[a];
                                for (int k = 0; k < a; ++k) acc = fmaf(-sh_L0[a * 6 + k], y[k], acc);
                                y[a] = acc * sh_L0i[a];
                            }
                            for (int a = 5; a >= 0; --a) {
                                float acc = y[a];
                                for (int k = a + 1; k < 6; ++k) acc = fmaf(-sh_L0[k * 6 + a], x[k], acc);
                                x[a] = acc * sh_L0i[a];
                            }
                            for (int k = 0; k < 6; ++k) { sh_a[0][k] = x[k]; sh_V0[6 + k] = x[k]; }
                        }
                    }
                    __syncthreads();
                }
                for (int lev = 1; lev <= d.max_depth; ++lev) {
                    if (act7 && PD_DEPTH(pd7) == lev) {
                        float ap[6], t[3], a[6], Wm[18], Km[6];
                        float R[9], r[3], Sl[3][3];
                        for (int k = 0; k < 9; ++k) R[k] = sh_R[b][k];
                        for (int k = 0; k < 3; ++k) r[k] = sh_R[b][9 + k];
                        for (int c = 0; c < 3; ++c) { const float ax[3] = {R[c], R[3 + c], R[6 + c]}; cross3(r, ax, Sl[c]); }
                        for (int k = 0; k < 18; ++k) Wm[k] = sh_W[b][k];
                        for (int k = 0; k < 6; ++k) Km[k] = sh_W[b][18 + k];
                        for (int k = 0; k < 6; ++k) ap[k] = sh_a[PD_PARENT(pd7)][k];
                        for (int c = 0; c < 3; ++c) {
                            float acc = 0.0f;
                            for (int k = 0; k < 6; ++k) acc = fmaf(Wm[k * 3 + c], ap[k], acc);
                            t[c] = uh[c] - acc;
                        }
                        dq[0] = SOP3(Km[0], t[0], Km[1], t[1], Km[3], t[2]);
                        dq[1] = SOP2(Km[2], t[1], Km[4], t[2]);
                        dq[2] = Km[5] * t[2];
                        for (int k = 0; k < 3; ++k) {
                            a[k] = ADD_SOP3(ap[k], R[k * 3], dq[0], R[k * 3 + 1], dq[1], R[k * 3 + 2], dq[2]);
                            a[3 + k] = ADD_SOP3(ap[3 + k], Sl[0][k], dq[0], Sl[1][k], dq[1], Sl[2][k], dq[2]);
                        }
                        for (int k = 0; k < 6; ++k) sh_a[b][k] = a[k];
                    }
                    __syncthreads();
                }
            }
        }
        const float damp = 1.0f / (1.0f + h * prm.ang_damping);
        PSTAMP(9); PACC(11);
        // ============================================================ 8. integrate (both envs)
        bool clamped = false;       // a rate clamped to max_ang_vel changes the angular momentum in a way the balance does not predict: it is skipped once
        if (on && b >= 1) {
            float wn[3];
            for (int k = 0; k < 3; ++k) {
                wn[k] = fmaf(h, qdd[k], wj[k]) + dq[k];      // free joint rate of phase 4 + the impulses' share
                if (last) {
                    float dr[4];
                    ld4(mdl, o_drv + 4 * k, dr);
                    const float kpk = dr[0], kdk = dr[1], tgk = tgt_env[jdof + k];
                    // torque applied over this substep (the contact impulses moved the implicit drive along; reported within the limit)
                    const float effk = dr[3];
                    float tq = sat[k] == 0 ? kpk * (tgk - edof[k] - h * wn[k]) - kdk * wn[k]
                             : (sat[k] == 2 ? tgk : (sat[k] > 0 ? effk : -effk));         // (clipped to the limit just below)
                    tq = tq > effk ? effk : (tq < -effk ? -effk : tq);
                    d.dof_force[(size_t)my_env * NDOF + jdof + k] = tq;
                }
                wj[k] = wn[k] * damp;
            }
            const float nj = sqrtf(dot3(wj, wj));
            if (nj > prm.max_ang_vel) { const float sc = prm.max_ang_vel / nj; wj[0] *= sc; wj[1] *= sc; wj[2] *= sc; clamped = true; }
            float e[3] = {h * wj[0], h * wj[1], h * wj[2]}, dqt[4], qn[4];
            rotvec2quat(e, dqt);
            qmul(qj, dqt, qn); qnormalize(qn);
            for (int k = 0; k < 4; ++k) qj[k] = qn[k];
            quat2rotvec(qj, edof);
        }
        if (on && b == 0) {
            float V0[6];
            for (int k = 0; k < 6; ++k) V0[k] = sh_V0[k] + sh_V0[6 + k];
            for (int k = 0; k < 3; ++k) V0[k] *= damp;
            const float n = sqrtf(dot3(V0, V0));
            if (n > prm.max_ang_vel) { const float sc = prm.max_ang_vel / n; V0[0] *= sc; V0[1] *= sc; V0[2] *= sc; clamped = true; }
            float e[3], dqt[4], qn[4], q0[4] = {sh_root[3], sh_root[4], sh_root[5], sh_root[6]};
            for (int k = 0; k < 3; ++k) { sh_root[k] += h * V0[3 + k]; e[k] = h * V0[k]; }
            rotvec2quat(e, dqt);
            qmul(dqt, q0, qn); qnormalize(qn);
            for (int k = 0; k < 4; ++k) sh_root[3 + k] = qn[k];
            // the reference point O of the spatial quantities moves with the root origin: re-base the root twist from O to
            // O + h v (velocity of the body-fixed point there: v + w x (h v)); without it the root's linear velocity would not
            // turn with the body and linear momentum would not be conserved
            float wxv[3];
            cross3(V0, V0 + 3, wxv);
            for (int k = 0; k < 3; ++k) V0[3 + k] = fmaf(h, wxv[k], V0[3 + k]);
            for (int k = 0; k < 6; ++k) sh_root[7 + k] = V0[k];
        }
        {
            const unsigned long long cl = __ballot(clamped);
            const bool any_clamped = (half ? (cl >> 32) : (cl & 0xffffffffull)) != 0ull;
            if (on && b == 0) sh_L[3] = any_clamped ? 0.0f : 1.0f;
        }
        __syncthreads();
        PSTAMP(10); PACC(12);
    }

    if (part < n_parts - 1) {         // hand over to the next part (see above) and publish
        if (on) {
            float *ps = pst + b * 4;
            part_st16(ps, qj[0], qj[1], qj[2], qj[3]);
            part_st16(ps + 4 * NB, wj[0], wj[1], wj[2], edof[0]);
            part_st16(ps + 8 * NB, edof[1], edof[2], 0.0f, 0.0f);
        }
        if (my_env >= 0) {
            const int g = b;
            const float *src = g < 4 ? sh_root + 4 * g : g < 6 ? sh_P + 4 * (g - 4) : g < 22 ? sh_lam + 4 * (g - 6) : g < 30 ? (const float *)sh_slot + 4 * (g - 22) : sh_L;
            if (g != 30) part_st16(pst + NB * 12 + 4 * g, src[0], src[1], src[2], src[3]);
            else part_st16(pst + NB * 12 + 4 * g, __int_as_float(work), 0.0f, 0.0f, 0.0f);
        }
        part_stores_done();           // the stores above have completed ...
        __syncthreads();
        if (b == 0 && my_env >= 0 && !(part == 0 && my_env == d.part_poison)) part_flag_set(d.part_flag + my_env, EMLOCO_PART_TAG(d.part_seq, part));      // ... before the flag goes out
        return;
    }
    // ---------------------------------------------------------------- write back (state after the final kinematics pass)
    if (on) {
        float *o = d.rb_state + (size_t)my_env * NB * 13 + b * 13;
        float t[3], V[6], r[3];
        for (int k = 0; k < 6; ++k) V[k] = sh_V[b][k];
        for (int k = 0; k < 3; ++k) r[k] = sh_R[b][9 + k];
        cross3(V, r, t);
        for (int k = 0; k < 3; ++k) { o[k] = sh_pq[b][k]; o[7 + k] = V[3 + k] + t[k]; o[10 + k] = V[k]; }
        for (int k = 0; k < 4; ++k) o[3 + k] = sh_pq[b][4 + k];
        if (b >= 1) {
            float *ds = dofs_env + jdof * 2;
            for (int k = 0; k < 3; ++k) { ds[2 * k] = edof[k]; ds[2 * k + 1] = wj[k]; }
        }
        if (b == 0) {
            float *rs = d.root_state + (long)my_env * 13;
            for (int k = 0; k < 3; ++k) { rs[k] = sh_root[k]; rs[7 + k] = sh_root[10 + k]; rs[10 + k] = sh_root[7 + k]; }
            for (int k = 0; k < 4; ++k) rs[3 + k] = sh_root[3 + k];
        }
    }
    if (my_env >= 0) {
        float *lws_env = d.lambda_ws + (size_t)my_env * MAXCAND * 3;
        for (int s = 0; s < (MAXCAND + 31) / 32; ++s) {      // warm-start multipliers per candidate for the next launch
            const int c = b + 32 * s;
            if (c < MAXCAND) {
                const int os = sh_slot[c];
                for (int k = 0; k < 3; ++k) lws_env[c * 3 + k] = os != 255 ? sh_lam[3 * os + k] : 0.0f;
            }
        }
        if (d.step_ticks && b == 0) d.step_ticks[my_env] = (unsigned)work;     // the key of the next launch's order
    }
}
#undef o_dyn
#undef jdof
#undef o_drv
#undef ENV_VIEW

// The step of every env: one workgroup = one wave per env PAIR and part.  Subset launches (emloco_sim_step_subset): with skip
// flags the flagged envs' halves idle (a workgroup whose two envs are flagged leaves at once); with a device-compacted id list
// (valid ids first, -1 after them) workgroup i steps list entries 2 i and 2 i + 1.  Both are THIS kernel: a second instantiation of the
// body for the list launch ran beside the big launch out of a different code object, the two thrashed the instruction cache the
// CUs share (measured in round 3: the big launch 0.58 -> 0.64 ms, the 25-env list launch 0.22 -> 0.44 ms).
// HF = 0: ground plane, HF = 1: height-field ground (d.hf).  Two instantiations, a simulator uses one of them for its lifetime.
template <int HF>
__global__ void __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(EMLOCO_SIM_WAVES_PER_SIMD, EMLOCO_SIM_WAVES_PER_SIMD)))
sim_step_kernel(EmlocoSimParams prm, EmlocoSimDev d) {
    const int n_parts = d.n_parts > 1 ? d.n_parts : 1;
    const int n_wg = (d.n_slots + 1) >> 1;                           // workgroups per part: all first parts, then all second parts
    const int part = (int)blockIdx.x / n_wg, pw = (int)blockIdx.x - part * n_wg;
    if (part >= n_parts) return;
    int envs[2];
    for (int s = 0; s < 2; ++s) {
        const int slot = 2 * pw + s;
        int env = -1;
        if (slot < d.n_slots) {
            env = d.step_order ? d.step_order[slot] : slot;
            if (d.step_ids) env = d.step_ids[slot];                   // list launch: slot i steps list entry i (-1: padding)
            if (env >= 0 && d.step_skip && d.step_skip[env] != 0) env = -1;     // flagged envs are stepped elsewhere
        }
        envs[s] = __builtin_amdgcn_readfirstlane(env);               // workgroup-uniform
    }
    if (envs[0] < 0 && envs[1] < 0) return;
    const long long t0 = d.step_start ? (long long)wall_clock64() : 0ll;
    sim_step_pair<HF>(prm, d, envs[0], envs[1], part, n_parts);
    if (d.step_start && (threadIdx.x & 31) == 0 && part == n_parts - 1) {
        const int env = envs[threadIdx.x >> 5];
        if (env >= 0) { d.step_start[env] = (unsigned long long)t0; d.step_start[d.n_env + env] = (unsigned long long)wall_clock64(); }
    }
}


// Dispatch order of the next full launch: env ids sorted by the contact work of their last step (sum over its substeps of
// 10 + number of contacts where there were any: the contact phases are ~45 % of a substep and grow with the contact count),
// most first (counting sort, one 1024-thread workgroup).  The measured duration of the last step is the worse key: it is
// dominated by what the env's wave shared its SIMD and CU with (correlation between consecutive steps 0.2).  Workgroups are handed to the CUs in index order and the launch is two
// resident rounds of waves (4096 envs on 256 CUs x 8), so its length is set by what the LAST workgroups cost: with the
// expensive envs (many contacts) first and the cheap ones (airborne) last, the slots that free up late receive short work.
// The order within a bucket is whatever the LDS atomics give -- envs are independent, results do not depend on it.
__global__ void __launch_bounds__(1024)
sim_order_kernel(const unsigned *ticks, int n, int *order, unsigned char *bucket_ws) {
    order_sort(ticks, n, order, bucket_ws);             // order_device.h
}

// Forward kinematics only (used after state writes through the *_indexed setters): fills rb_state of
// the listed envs from root_state / dof_state.  One wave per env; lanes walk the tree level by level (fk_device.h).
__global__ void __launch_bounds__(64)
sim_fk_kernel(EmlocoSimDev d, const int *env_ids, int n_ids) {
    const int lane = threadIdx.x;
    // grid-stride over the id list (one pass when the grid covers it); a device-compacted list ends at its first -1
    for (int bi = blockIdx.x; bi < n_ids; bi += gridDim.x) {
        const int env = env_ids ? env_ids[bi] : bi;
        if (env < 0) break;
        __shared__ float sm[FK_SM_FLOATS];
        fk_env(d, env, lane, sm);
        __syncthreads();                              // LDS is reused by the next list entry
    }
}

}  // namespace emloco

#undef cross3
#undef dot3
#undef dot6
#undef qmul
#undef qnormalize
#undef q2mat
#undef matvec3
#undef rotvec2quat
#undef quat2rotvec
