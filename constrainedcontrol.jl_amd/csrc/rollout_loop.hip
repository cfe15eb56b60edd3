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
__device__ __forceinline__ double loop_eval(int t, const Lay& Y, double* L, const LaneRegs& r, const MechDev* M, int s_off, double alpha PROF_ARG) {
    double part = ph_body_eval<JAC>(t, M->nb, Y, L, r, M->dt, s_off, alpha);
    __syncthreads();
    STAMP(PF_EVAL_BODY);
    part += lp_joint_eval<JAC>(t, Y, L, r, M);
    __syncthreads();
    STAMP(PF_EVAL_JOINT);
    PCOUNT(PF_EVALS);
    return sqrt(group_sum<64>(part));
}

// maximum of one 64-bit key per lane over the wavefront, through the DPP crossbar (no LDS round trips): running maximum along
// each 16-lane row, row 0 -> 1 and 2 -> 3, rows {0,1} -> {2,3}; lane 63 holds the result, read back as a wavefront-uniform value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long key_dpp_max(unsigned long long v) {
    const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long v) {
    v = key_dpp_max<0x111, 0xf>(v);      // row_shr:1
    v = key_dpp_max<0x112, 0xf>(v);      // row_shr:2
    v = key_dpp_max<0x114, 0xf>(v);      // row_shr:4
    v = key_dpp_max<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of each row holds the row's maximum
    v = key_dpp_max<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = key_dpp_max<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wavefront's maximum
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// S dl = r by elimination with complete pivoting up to the numerical rank; dl of the free (redundant) directions is 0
__device__ __forceinline__ void loop_solve(int t, const Lay& Y, double* L, const MechDev* M PROF_ARG) {
    const int mr = 5 * M->nj, stride = loop_row_stride(M->nj);
    lp_schur_row(t, Y, L, M);
    if (t < mr) { L[Y.R + t] = (double)t; L[Y.DL + t] = 0.0; }
    __syncthreads();
    STAMP(PF_SCHUR_S);
    LoopRow R;
    lp_row_init(R, t, mr, stride, Y, L);
    int rank = 0;
    double first = 0.0;
    for (int k = 0; k < mr; k++) {
        const unsigned long long key = wave_max_key(R.key);            // uniform: every lane holds the same pivot
        const double best = lp_key_value(key);
        if (k == 0) first = best;
        if (!(best > LOOP_RANK_TOL * first) || !(best > 0.0)) break;
        const int prow = lp_key_row(key), pcol = lp_key_col(key);
        lp_col_swap(t, k, pcol, mr, stride, Y, L);
        __syncthreads();
        lp_elim_search(R, t, k, prow, mr, stride, Y, L);
        rank = k + 1;
    }
    __syncthreads();
    STAMP(PF_TRI_FWD);
    for (int k = rank - 1; k >= 0; k--) {
        lp_back_step(R, t, k, (int)L[Y.R + mr + k], mr, stride, Y, L);
        __syncthreads();
    }
    STAMP(PF_TRI_BWD);
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

#ifdef CCLQR_PROFILE
    Prof prof;
    prof.start();
#endif
    int worst = 0;
    bool bad = false, dead = false;
    const long long ginst = a.inst0 + inst;
    for (int kk = 0; kk < a.steps; kk++) {
        const int k = a.k0 + kk;
        if (a.traj)
            for (int e = t; e < nz; e += 64) a.traj[((size_t)inst * a.steps + kk) * nz + e] = L[Y.Z + e];
        STAMP(PF_IO);
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
        STAMP(PF_CONTROL);
        // ---------------- per-step invariants
        lp_forces(t, Y, L, r, M);
        lp_knot_jac(t, Y, L, r, M);
        __syncthreads();
        lp_force_map(t, Y, L, M);
        __syncthreads();
        STAMP(PF_FORCES);
        PCOUNT(PF_STEPS);
        // ---------------- newton! (tolerances and line search: SURVEY 8a-bis)
        bool done = dead, failed = false;
        int its = 0;
        double normf0 = loop_eval<true>(t, Y, L, r, M, Y.S, 0.0 PROF_PASS);
        for (int iter = 1; iter <= NEWTON_MAXIT && !done; iter++) {
            PCOUNT(PF_NEWTON_ITERS);
            loop_solve(t, Y, L, M PROF_PASS);
            lp_body_solve(t, Y, L, M);
            __syncthreads();
            STAMP(PF_BODY_SOLVE);
            double alpha = 1.0, normf1 = 0.0;
            const double nd = sqrt(group_sum<64>(lp_trial(t, Y, L, M, alpha)));
            __syncthreads();
            STAMP(PF_TRIAL);
            for (int ls = 0; ls <= LINE_MAXIT; ls++) {     // halve while ||f|| grows; level LINE_MAXIT is taken as it is
                normf1 = loop_eval<false>(t, Y, L, r, M, Y.ST, alpha PROF_PASS);
                if (!(normf1 > normf0) || ls == LINE_MAXIT) break;
                alpha *= 0.5;
                lp_trial(t, Y, L, M, alpha);
                __syncthreads();
            }
            lp_accept(t, Y, L, M, alpha);
            __syncthreads();
            STAMP(PF_ACCEPT);
            its = iter;
            if (normf1 < NEWTON_EPS && alpha * nd < NEWTON_EPS) done = true;
            if (!(normf1 < 1e300)) { done = true; failed = true; }
            if (!done) normf0 = loop_eval<true>(t, Y, L, r, M, Y.S, 0.0 PROF_PASS);     // Jacobians at the accepted point
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
#ifdef CCLQR_PROFILE
    prof.stamp(PF_IO);
    prof.flush();
#endif
}

#ifdef CCLQR_PROFILE
extern "C" int cclqr_prof_read_loop(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * PF_N);
    if (e == hipSuccess && reset) { unsigned long long z[PF_N] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return e == hipSuccess ? PF_N : -1;
}
#endif

size_t loop_lds_bytes(int nb, int nj) { return (size_t)make_loop_layout(nb, nj).total * sizeof(double); }

hipError_t launch_rollout_loop(const RolloutArgs& a, int nb, int nj, hipStream_t stream) {
    if (a.n_inst <= 0) return hipSuccess;
    const size_t lds = loop_lds_bytes(nb, nj);
    hipError_t e = set_max_dynamic_lds_once((const void*)rollout_loop_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rollout_loop_kernel, dim3((unsigned)a.n_inst), dim3(64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace cclqr
