// cclqr_newton.h -- device-only orchestration of the Newton solve shared by the rollout and linearisation kernels
// (tolerances and line search: SURVEY 8a-bis; the same sequence of values as oracle/cclqr_oracle.c newton()).
#pragma once
#include "cclqr_dev.h"

namespace cclqr {

#define NEWTON_EPS 1e-10
#define NEWTON_MAXIT 100
#define LINE_MAXIT 10

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// residual + Jacobians at the point stored at (s_off, lam_off); returns the group's ||f||_2
template <int G>
__device__ __forceinline__ double eval_point(int t, int nb, const Lay& Y, double* L, const LaneRegs& r, const MechDev* M, double dt,
                                             int s_off, int lam_off, bool active) {
    if (active) ph_body_eval(t, nb, Y, L, r, dt, s_off);
    __syncthreads();
    if (active) ph_joint_eval(t, nb, Y, L, r, dt);
    __syncthreads();
    double part = active ? ph_force_map_norm(t, G, nb, Y, L, M, lam_off) : 0.0;
    return sqrt(group_sum<G>(part));
}


// newton! on the instance held in LDS (S/LAM = guess in, solution out; XQ/NB/DINV/GV* = values at the solution).
// valid=false groups only keep the wave's control flow uniform.  Returns iterations used; *converged reports success.
template <int G>
__device__ __forceinline__ int newton_solve(int t, int nb, const Lay& Y, double* L, LaneRegs& r, const MechDev* M, double dt, bool valid,
                                            bool* converged) {
    double normf0 = eval_point<G>(t, nb, Y, L, r, M, dt, Y.S, Y.LAM, valid);
    bool done = !valid;
    int its = 0;
    for (int iter = 1; iter <= NEWTON_MAXIT; iter++) {
        if (!__any(!done)) break;
        // Schur complement on the multipliers, block-tridiagonal solve along each chain, body back-substitution
        if (!done) ph_schur_w(t, G, nb, Y, L, M);
        __syncthreads();
        if (!done) ph_schur_s(t, G, nb, Y, L, M);
        __syncthreads();
        for (int l = nb - 1; l >= 0; l--) {
            double lu[5];
            if (!done) ph_tri_fwd(t, l, Y, L, M, lu);
            __syncthreads();
            if (!done) ph_tri_store(t, l, Y, L, lu);
            __syncthreads();
        }
        for (int l = 0; l < nb; l++) {
            if (!done) ph_tri_bwd(t, l, Y, L, M);
            __syncthreads();
        }
        if (!done) ph_body_solve(t, G, nb, Y, L, M);
        __syncthreads();
        // line search: halve while ||f|| grows
        double alpha = 1.0, normf1 = 0.0, nd = 0.0;
        bool ls_done = done;
        for (int ls = 0; ls <= LINE_MAXIT; ls++) {
            if (!__any(!ls_done)) break;
            double pd = ls_done ? 0.0 : ph_trial(t, G, nb, Y, L, alpha);
            double nd2 = group_sum<G>(pd);
            if (ls == 0) nd = sqrt(nd2);
            __syncthreads();
            double nf = eval_point<G>(t, nb, Y, L, r, M, dt, Y.ST, Y.LT, !ls_done);
            if (!ls_done) {
                normf1 = nf;
                if (normf1 > normf0 && ls < LINE_MAXIT) alpha *= 0.5; else ls_done = true;
            }
            __syncthreads();
        }
        if (!done) {
            ph_accept(t, G, nb, Y, L);
            its = iter;
            if (normf1 < NEWTON_EPS && alpha * nd < NEWTON_EPS) done = true;
            normf0 = normf1;
        }
        __syncthreads();
    }
    *converged = done;
    return its;
}

}  // namespace cclqr
