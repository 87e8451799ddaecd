// fk_device.h -- forward kinematics of one env by one wave (lanes = bodies, level by level), as a device function: sim_fk_kernel
// (sim_kernels.hip) and the reset role of reset_obs_kernel (chain_kernels.hip: sample -> kinematics -> finish -> observations in
// ONE launch) run the same body, with the rigid-body kernels' fused helper set (sim_math.h) on both sides.
#pragma once
#include <hip/hip_runtime.h>
#include "dev_math.h"
#include "sim_math.h"
#include "emloco_types.h"

namespace emloco {

// rb_state of env from its root_state / dof_state.  All 64 lanes of the wave call it (barriers inside); the caller puts a barrier
// behind it before the LDS arrays are reused.
#define FK_SM_FLOATS (EMLOCO_NB * 22)      /* LDS workspace of the calling workgroup */
__device__ __forceinline__ void fk_env(const EmlocoSimDev &d, int env, int lane, float *sm) {
    float (*sh_pw)[3] = (float (*)[3])sm;
    float (*sh_qw)[4] = (float (*)[4])(sm + EMLOCO_NB * 3);
    float (*sh_R)[9] = (float (*)[9])(sm + EMLOCO_NB * 7);
    float (*sh_V)[6] = (float (*)[6])(sm + EMLOCO_NB * 16);
    const int b = (lane < EMLOCO_NB) ? lane : 0;
    const int parent = d.topo[EMLOCO_TOPO_PARENT + b], depth = d.topo[EMLOCO_TOPO_DEPTH + b];
    float off[3], qj[4] = {0, 0, 0, 1}, wj[3] = {0, 0, 0}, V[6] = {0, 0, 0, 0, 0, 0}, r[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) off[k] = d.model[(size_t)env * EMLOCO_MODEL_WORDS + EMLOCO_MB_DYN + b * 16 + k];
    if ((lane < EMLOCO_NB) && lane >= 1) {
        const float *ds = d.dof_state + ((long)env * EMLOCO_NDOF + (lane - 1) * 3) * 2;
        float e[3] = {ds[0], ds[2], ds[4]};
        frotvec2quat(e, qj);
        wj[0] = ds[1]; wj[1] = ds[3]; wj[2] = ds[5];
    }
    const float *rs = d.root_state + (long)env * 13;
    const float p0[3] = {rs[0], rs[1], rs[2]};
    if (lane == 0) {
        float q0[4] = {rs[3], rs[4], rs[5], rs[6]}, R[9];
        fqnormalize(q0);
        fq2mat(q0, R);
        for (int k = 0; k < 3; ++k) { sh_pw[0][k] = p0[k]; V[k] = rs[10 + k]; V[3 + k] = rs[7 + k]; }
        for (int k = 0; k < 4; ++k) sh_qw[0][k] = q0[k];
        for (int k = 0; k < 9; ++k) sh_R[0][k] = R[k];
        for (int k = 0; k < 6; ++k) sh_V[0][k] = V[k];
    }
    __syncthreads();
    for (int lev = 1; lev <= d.max_depth; ++lev) {
        if ((lane < EMLOCO_NB) && depth == lev) {
            float Rp[9], o[3], qp[4], qw[4], pw[3], R[9];
            for (int k = 0; k < 9; ++k) Rp[k] = sh_R[parent][k];
            for (int k = 0; k < 4; ++k) qp[k] = sh_qw[parent][k];
            fmatvec3(Rp, off, o);
            for (int k = 0; k < 3; ++k) { pw[k] = sh_pw[parent][k] + o[k]; r[k] = pw[k] - p0[k]; }
            fqmul(qp, qj, qw); fqnormalize(qw); fq2mat(qw, R);
            float Sl[3][3];
            for (int c = 0; c < 3; ++c) { float ax[3] = {R[c], R[3 + c], R[6 + c]}; fcross3(r, ax, Sl[c]); }
            for (int k = 0; k < 3; ++k) {
                V[k] = sh_V[parent][k] + SOP3(R[k * 3], wj[0], R[k * 3 + 1], wj[1], R[k * 3 + 2], wj[2]);
                V[3 + k] = sh_V[parent][3 + k] + SOP3(Sl[0][k], wj[0], Sl[1][k], wj[1], Sl[2][k], wj[2]);
            }
            for (int k = 0; k < 3; ++k) sh_pw[lane][k] = pw[k];
            for (int k = 0; k < 4; ++k) sh_qw[lane][k] = qw[k];
            for (int k = 0; k < 9; ++k) sh_R[lane][k] = R[k];
            for (int k = 0; k < 6; ++k) sh_V[lane][k] = V[k];
        }
        __syncthreads();
    }
    if ((lane < EMLOCO_NB)) {
        float *o = d.rb_state + ((long)env * EMLOCO_NB + lane) * 13, t[3];
        fcross3(V, r, t);
        for (int k = 0; k < 3; ++k) { o[k] = sh_pw[lane][k]; o[7 + k] = V[3 + k] + t[k]; o[10 + k] = V[k]; }
        for (int k = 0; k < 4; ++k) o[3 + k] = sh_qw[lane][k];
    }
}

}  // namespace emloco
