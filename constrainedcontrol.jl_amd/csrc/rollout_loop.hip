// rollout_loop.hip -- simulate! with the LQR feedback law for mechanisms with closed kinematic loops (examples/lqr_deltabot.jl:25-53),
// one instance per wavefront, persistent over the horizon.  Phases and the singular dense solve: cclqr_loop.h.
//
// Replaces: ConstrainedDynamics.simulate!/newton! on a Mechanism whose constraint graph has cycles, driven by control_lqr!
// (src/control/lqr.jl:89-139).  Not a throughput kernel: loop mechanisms are small (the reference's only one has five bodies) and
// the solve is a 5 nj-step pivoted elimination; the chain / tree kernels keep every mechanism without loops.
#include "cclqr_loop.h"
#include "cclqr_internal.h"
#include "cclqr_newton.h"

namespace cclqr {

// residual (+ Jacobians) at the point s_off with multipliers lambda - alpha dlambda already folded into C - alpha CD
template <bool JAC>
__device__ __forceinline__ double loop_eval(int t, const Lay& Y, double* L, const LaneRegs& r, const MechDev* M, int s_off, double alpha) {
    double part = ph_body_eval<JAC>(t, M->nb, Y, L, r, M->dt, s_off, alpha);
    __syncthreads();
    part += lp_joint_eval<JAC>(t, Y, L, r, M);
    __syncthreads();
    return sqrt(group_sum<64>(part));
}

// S dl = r by elimination with complete pivoting up to the numerical rank; dl of the free (redundant) directions is 0
__device__ __forceinline__ void loop_solve(int t, const Lay& Y, double* L, const MechDev* M) {
    const int mr = 5 * M->nj, stride = loop_row_stride(M->nj);
    lp_schur_row(t, Y, L, M);
    if (t < mr) L[Y.R + t] = (double)t;
    __syncthreads();
    int rank = mr;
    double first = 0.0;
    for (int k = 0; k < mr; k++) {
        double best; int bcol, brow = t;
        lp_pivot_search(t, k, mr, stride, Y, L, &best, &bcol);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {      // wavefront arg-max, ties to the smaller (row, column): every lane ends with the same triple
            const double ov = __shfl_xor(best, o, 64);
            const int orow = __shfl_xor(brow, o, 64), ocol = __shfl_xor(bcol, o, 64);
            if (ov > best || (ov == best && (orow < brow || (orow == brow && ocol < bcol)))) { best = ov; brow = orow; bcol = ocol; }
        }
        if (k == 0) first = best;
        if (!(best > LOOP_RANK_TOL * first) || !(best > 0.0)) { rank = k; break; }      // uniform: every lane holds the same pivot
        lp_swap_rows(t, k, brow, mr, stride, Y, L);
        __syncthreads();
        lp_swap_cols(t, k, bcol, mr, stride, Y, L);
        __syncthreads();
        lp_eliminate(t, k, mr, stride, Y, L);
        __syncthreads();
    }
    for (int k = rank - 1; k >= 0; k--) {
        lp_back_step(t, k, mr, stride, Y, L);
        __syncthreads();
    }
    lp_scatter(t, rank, mr, stride, Y, L);
    __syncthreads();
}

__global__ __launch_bounds__(64) void rollout_loop_kernel(RolloutArgs a) {
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    const int64_t inst = blockIdx.x;
    const MechDev* M = a.M;
    const CtrlDev* C = a.C;
    const int nb = M->nb, nj = M->nj, nz = 13 * nb;
    const Lay Y = make_loop_layout(nb, nj);
    double* L = lds;
    LaneRegs r;
    loop_load_consts(r, M, t);
    for (int e = t; e < Y.total; e += 64) L[e] = 0.0;
    __syncthreads();
    for (int e = t; e < nz; e += 64) L[Y.Z + e] = a.z0[inst * nz + e];
    if (a.lam && a.k0 > 1)
        for (int e = t; e < 5 * nj; e += 64) L[Y.LAM + e] = a.lam[inst * 5 * nj + e];
    __syncthreads();

    int worst = 0;
    bool bad = false, dead = false;
    const long long ginst = a.inst0 + inst;
    for (int kk = 0; kk < a.steps; kk++) {
        const int k = a.k0 + kk;
        if (a.traj)
            for (int e = t; e < nz; e += 64) a.traj[((size_t)inst * a.steps + kk) * nz + e] = L[Y.Z + e];
        // ---------------- feedback law (lqr.jl:89-139)
        const bool gate = (C->N <= 0) || (k < C->N);
        const int ksp = (C->nsp > 1) ? ((k - 1 < C->nsp) ? k - 1 : C->nsp - 1) : 0;
        const int kidx = (C->N <= 0) ? 0 : ((k - 1 < C->nK) ? k - 1 : C->nK - 1);
        if (t < nj) L[Y.UJ + t] = 0.0;
        if (gate) ph_control_error(t, nb, Y, L, r, C, C->zd + ginst * C->zd_stride + (size_t)ksp * nz);
        __syncthreads();
        if (t < nj) L[Y.UJ + t] = 0.0;          // ph_control_error leaves the (absent) friction term of joint t < nb there
        __syncthreads();
        if (gate) {
            for (int i = 0; i < C->mu; i++) {
                double part = 0.0;
                if (C->K) part = ph_gain_partial(t, 64, nb, Y, L, C->K + ginst * C->K_stride + ((size_t)kidx * C->mu + i) * 12 * nb);
                const double s = group_sum<64>(part);
                if (t == 0) L[Y.UJ + C->cj[i]] += (C->Fd ? C->Fd[ginst * C->Fd_stride + (size_t)ksp * C->mu + i] : 0.0) - s;
                __syncthreads();
            }
        }
        // ---------------- per-step invariants
        lp_forces(t, Y, L, r, M);
        lp_knot_jac(t, Y, L, r, M);
        __syncthreads();
        lp_force_map(t, Y, L, M);
        __syncthreads();
        // ---------------- newton! (tolerances and line search: SURVEY 8a-bis)
        bool done = dead, failed = false;
        int its = 0;
        double normf0 = loop_eval<true>(t, Y, L, r, M, Y.S, 0.0);
        for (int iter = 1; iter <= NEWTON_MAXIT && !done; iter++) {
            loop_solve(t, Y, L, M);
            lp_body_solve(t, Y, L, M);
            __syncthreads();
            double alpha = 1.0, normf1 = 0.0;
            const double nd = sqrt(group_sum<64>(lp_trial(t, Y, L, M, alpha)));
            __syncthreads();
            for (int ls = 0; ls <= LINE_MAXIT; ls++) {     // halve while ||f|| grows; level LINE_MAXIT is taken as it is
                normf1 = loop_eval<false>(t, Y, L, r, M, Y.ST, alpha);
                if (!(normf1 > normf0) || ls == LINE_MAXIT) break;
                alpha *= 0.5;
                lp_trial(t, Y, L, M, alpha);
                __syncthreads();
            }
            lp_accept(t, Y, L, M, alpha);
            __syncthreads();
            its = iter;
            if (normf1 < NEWTON_EPS && alpha * nd < NEWTON_EPS) done = true;
            if (!(normf1 < 1e300)) { done = true; failed = true; }
            if (!done) normf0 = loop_eval<true>(t, Y, L, r, M, Y.S, 0.0);     // Jacobians at the accepted point
        }
        if (!dead) {
            const bool conv = done && !failed;
            if (!conv) bad = true;
            if (its > worst) worst = its;
            if (!conv && its < NEWTON_MAXIT) {       // non-finite residual: freeze the instance at its last pose, at rest
                dead = true;
                for (int e = t; e < nb; e += 64)
                    for (int i = 0; i < 6; i++) L[Y.Z + 13 * e + 7 + i] = 0.0;
            } else {
                ph_update(t, nb, Y, L);       // XQ = next pose of the accepted solution (the last evaluation was at that point)
            }
        }
        __syncthreads();
    }
    for (int e = t; e < nz; e += 64) a.zT[inst * nz + e] = L[Y.Z + e];
    if (a.lam) for (int e = t; e < 5 * nj; e += 64) a.lam[inst * 5 * nj + e] = L[Y.LAM + e];
    if (a.status && t == 0) a.status[inst] = bad ? -worst : worst;
}

size_t loop_lds_bytes(int nb, int nj) { return (size_t)make_loop_layout(nb, nj).total * sizeof(double); }

hipError_t launch_rollout_loop(const RolloutArgs& a, int nb, int nj, hipStream_t stream) {
    if (a.n_inst <= 0) return hipSuccess;
    const size_t lds = loop_lds_bytes(nb, nj);
    hipError_t e = hipFuncSetAttribute((const void*)rollout_loop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rollout_loop_kernel, dim3((unsigned)a.n_inst), dim3(64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace cclqr
