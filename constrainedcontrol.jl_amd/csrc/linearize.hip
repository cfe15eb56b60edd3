// linearize.hip -- batched linearsystem(): one knot (setpoint) per wavefront, LDS-resident, reusing the rollout kernel's
// Newton solve for (z+, lambda*) and then assembling A, Bu, Bl, G (cclqr_lin_dev.h).
// Replaces: linearsystem(mechanism, xd, vd, qd, ωd, Fτd, bodyids, eqcids) at src/control/lqr.jl:63 and, once per backward
// step, at src/control/lqr_tracking.jl:88 (there the nk knots are linearised in ONE launch).
#include "cclqr_dev.h"
#include "cclqr_internal.h"
#include "cclqr_newton.h"
#include "cclqr_lin_dev.h"

namespace cclqr {

template <bool TREE>
__global__ __launch_bounds__(64) void linearize_kernel(LinArgs a) {
    extern __shared__ double lds[];
    constexpr int G = 64;
    const int t = threadIdx.x;
    const int knot = blockIdx.x;
    const MechDev* M = a.M;
    const int nb = M->nb, nz = 13 * nb, mx = 12 * nb, ml = 5 * nb, mu = a.mu;
    const double dt = M->dt;
    const Lay Y = make_layout(nb, TREE ? 2 * M->npairs : 0);
    const int JB = Y.total;
    double* L = lds;
    LinOut O;
    O.A = a.A + (size_t)knot * mx * mx; O.Bu = a.Bu + (size_t)knot * mx * mu; O.Bl = a.Bl + (size_t)knot * mx * ml;
    O.G = a.G + (size_t)knot * ml * mx; O.mx = mx; O.mu = mu; O.ml = ml;
    for (int e = t; e < mx * mx; e += G) O.A[e] = 0.0;
    for (int e = t; e < mx * mu; e += G) O.Bu[e] = 0.0;
    for (int e = t; e < mx * ml; e += G) O.Bl[e] = 0.0;
    for (int e = t; e < ml * mx; e += G) O.G[e] = 0.0;

    LaneRegs r;
    const int NL = newton_level_groups(G, nb), lg = t / nb, tl = t - lg * nb;   // lane groups of the level-parallel line search
    lane_load_consts(r, M, lg < NL ? tl : 0);
    for (int e = t; e < nz; e += G) { int l = e / 13, c = e - 13 * l; L[Y.Z + e] = a.zd[(size_t)knot * nz + M->perm[l] * 13 + c]; }
    for (int e = t; e < 5 * nb; e += G) L[Y.LAM + e] = 0.0;
    for (int e = t; e < nb; e += G) L[Y.UJ + e] = 0.0;
    __syncthreads();
    if (t == 0)
        for (int i = 0; i < mu; i++) L[Y.UJ + a.cj[i]] += a.Fd ? a.Fd[(size_t)knot * mu + i] : 0.0;
    __syncthreads();
    if (lg < NL) ph_forces<TREE>(tl, nb, Y, L, r, M, lg == 0);
    ph_knot_jac(t, nb, Y, L, r);
    __syncthreads();
    if (TREE) ph_force_map_tree(t, G, nb, Y, L, M);
    else ph_force_map(t, G, nb, Y, L, M->end_mask);
    __syncthreads();
    bool done = false;
#ifdef CCLQR_PROFILE
    Prof prof;
    prof.start();
#endif
    int its = newton_solve<G, TREE>(t, nb, Y, L, r, M, dt, true, &done PROF_PASS);
    __syncthreads();
    // D_R^-1 and the next pose at the converged solution (the last line-search trial may have been residual-only)
    ph_body_eval<true>(t, nb, Y, L, r, dt, Y.S, 0.0);
    __syncthreads();
    ph_lin_joint(t, nb, Y, JB, L, r);
    __syncthreads();
    ph_lin_rows_A(t, nb, Y, JB, L, r, M, O);
    ph_lin_rows_B(t, nb, Y, L, r, M, a.cj, O);
    if (t == 0 && a.status) a.status[knot] = done ? its : -its;
}

size_t linearize_lds_bytes(int nb, int tree, int npairs) { return (size_t)(make_layout(nb, tree ? 2 * npairs : 0).total + LJB * nb) * sizeof(double); }

hipError_t launch_linearize(const LinArgs& a, int nb, int tree, int npairs, hipStream_t stream) {
    if (a.nk <= 0) return hipSuccess;
    const size_t lds = linearize_lds_bytes(nb, tree, npairs);
    const void* fn = tree ? (const void*)linearize_kernel<true> : (const void*)linearize_kernel<false>;
    hipError_t e = set_max_dynamic_lds_once(fn, lds);
    if (e != hipSuccess) return e;
    if (tree) hipLaunchKernelGGL(linearize_kernel<true>, dim3(a.nk), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL(linearize_kernel<false>, dim3(a.nk), dim3(64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace cclqr
