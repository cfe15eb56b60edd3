// rollout_chain.hip -- the LQR-controlled rollout kernel for forests of chains (all five BASELINE configs), persistent over
// the whole horizon and REGISTER-resident: lane t of an instance's lane group owns link t and keeps its state, multipliers,
// Jacobians and Newton iterate in registers; neighbour links talk through whole-wave DPP shifts; LDS holds only the 5x5
// Schur blocks of the block-tridiagonal solve (cclqr_chain.h).  HBM sees one state load, one gain row per step, one
// trajectory row per step (if recorded) and the final state.
//
// Replaces: ConstrainedDynamics.simulate!/newton! as driven by the reference (examples/lqr_cartpole.jl:44) with
//           control_lqr! (src/control/lqr.jl:89-139) / control_trackinglqr! (src/control/lqr_tracking.jl:46-71).
#include "cclqr_chain.h"
#include "cclqr_internal.h"
#include "cclqr_newton.h"

namespace cclqr {

// lane i <- lane i-1 / lane i+1 of the wavefront (DPP wave shifts; lanes shifted in from outside read 0)
__device__ __forceinline__ double wave_from_prev(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_from_next(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ void from_prev(const double* in, double* out) {
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = wave_from_prev(in[i]);
}
template <int N>
__device__ __forceinline__ void from_next(const double* in, double* out) {
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = wave_from_next(in[i]);
}

// dynamic per-lane data of the owned link.  The velocity part of the state is s itself: the solution (v+, w+) of one step is
// the state's (v, w) at the next knot and the Newton start of the next step.
struct LinkS {
    double z[7], s[6];
    double ds[6], cd[6], d[6];
};

// residual (+ Jacobians when JAC) at the point s - alpha ds with constraint forces C - alpha cd; returns the group's ||f||_2.
// With JAC the Schur complement rows of the point go straight to LDS: W = G_v D^-1 only lives inside this function.
// (The full-step trial is evaluated with JAC on the speculation that it is accepted; if it is not, the accepted point is
// evaluated again, which overwrites these rows.)
template <int G, bool JAC>
__device__ __forceinline__ double chain_eval(const LinkC& c, LinkS& S, int t, const Lay& Y, double* L, double alpha, bool active, double dt PROF_ARG) {
    double part = 0.0;
    double NB[9], g[5], xq[7];
#pragma unroll
    for (int k = 0; k < 7; k++) xq[k] = S.z[k];
    if (active) {
        double cf[6], sv[6], cTR[6], DINV[9];
#pragma unroll
        for (int k = 0; k < 6; k++) { cf[k] = L[Y.C + 6 * t + k] - alpha * S.cd[k]; sv[k] = S.s[k] - alpha * S.ds[k]; cTR[k] = L[Y.D + 6 * t + k]; }
        part = ck_body_eval<JAC>(c, S.z, sv, cf, cTR, cTR + 3, dt, xq, S.d, DINV, NB);
        if (JAC) {
#pragma unroll
            for (int k = 0; k < 9; k++) L[Y.DINV + 9 * t + k] = DINV[k];
        }
    }
    STAMP(PF_EVAL_BODY);
    double pxq[7], pNB[9];
    from_prev<7>(xq, pxq);
    if (JAC) from_prev<9>(NB, pNB);
    if (!c.has_a) {
#pragma unroll
        for (int i = 0; i < 7; i++) pxq[i] = (i == 3) ? 1.0 : 0.0;
    }
    double wXT[3][3], wPB[5][3], wPA[5][3];
    if (active) {
        joint_eval_sparse<JAC>(c, pxq, pxq + 3, xq, xq + 3, pNB, NB, g, wXT, wPB, wPA);
#pragma unroll
        for (int i = 0; i < 5; i++) part += g[i] * g[i];
    }
    STAMP(PF_EVAL_JOINT);
    if (JAC) {
        double pd[6];
        from_prev<6>(S.d, pd);
        ck_schur_rows(c, t, active, Y, L, wXT, wPB, wPA, g, S.d, pd);
        STAMP(PF_SCHUR_S);
    }
    const double nrm = sqrt(group_sum<G>(part));
    STAMP(PF_EVAL_MAP);
    PCOUNT(PF_EVALS);
    return nrm;
}

template <int G>
__global__ __launch_bounds__(64) void rollout_chain_kernel(RolloutArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, t = lane % G, grp = lane / G;
    const int64_t inst = (int64_t)blockIdx.x * (64 / G) + grp;
    const bool valid = inst < a.n_inst;
    const MechDev* M = a.M;
    const CtrlDev* C = a.C;
    const int nb = M->nb;
    const double dt = M->dt;
    const Lay Y = make_chain_layout(nb);
    double* L = lds + grp * Y.total;
    const int nz = 13 * nb;

    LinkC c;
    link_load_consts(c, M, t, nb, dt);
    if (C->has_fric && c.on) c.fric = C->fric[t];
    const bool on = c.on && valid;
    const int ut = c.on ? M->perm[t] : 0;      // user body index of the owned link

    LinkS S;
    double pid_int = 0.0, pid_last = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) S.z[i] = on ? a.z0[inst * nz + ut * 13 + i] : ((i == 3) ? 1.0 : 0.0);
#pragma unroll
    for (int i = 0; i < 6; i++) S.s[i] = on ? a.z0[inst * nz + ut * 13 + 7 + i] : 0.0;
    if (a.pid_state && a.k0 > 1 && on) { pid_int = a.pid_state[(inst * nb + t) * 2]; pid_last = a.pid_state[(inst * nb + t) * 2 + 1]; }
#pragma unroll
    for (int i = 0; i < 6; i++) { S.cd[i] = 0.0; S.d[i] = 0.0; S.ds[i] = 0.0; }
    for (int e = t; e < Y.total; e += G) L[e] = 0.0;
    __syncthreads();
    if (on && a.lam && a.k0 > 1) {
#pragma unroll
        for (int i = 0; i < 5; i++) L[Y.LAM + 5 * t + i] = a.lam[inst * 5 * nb + 5 * t + i];
    }

#ifdef CCLQR_PROFILE
    Prof prof;
    prof.start();
#endif
    int worst = 0;
    bool bad = false, dead = false;    // dead: a step produced a non-finite residual; the instance is frozen from then on
    for (int kk = 0; kk < a.steps; kk++) {
        const int k = a.k0 + kk;
        if (a.traj) {     // Storage row of this step, staged through LDS in user body order so that the HBM stores coalesce
            if (on) {
#pragma unroll
                for (int i = 0; i < 7; i++) L[Y.Z + 13 * ut + i] = S.z[i];
#pragma unroll
                for (int i = 0; i < 6; i++) L[Y.Z + 13 * ut + 7 + i] = S.s[i];
            }
            __syncthreads();
            if (valid) {
                double* dst = a.traj + ((size_t)inst * a.steps + kk) * nz;
                for (int e = t; e < nz; e += G) dst[e] = L[Y.Z + e];
            }
            __syncthreads();
        }
        STAMP(PF_IO);
        // ---------------- feedback law (lqr.jl:89-139 / lqr_tracking.jl:46-71)
        const bool gate = (C->N <= 0) || (k < C->N);
        const int ksp = (C->nsp > 1) ? ((k - 1 < C->nsp) ? k - 1 : C->nsp - 1) : 0;
        const int kidx = (C->N <= 0) ? 0 : ((k - 1 < C->nK) ? k - 1 : C->nK - 1);
        double uj = 0.0;
        double zf[13], za[13];
#pragma unroll
        for (int i = 0; i < 7; i++) zf[i] = S.z[i];
#pragma unroll
        for (int i = 0; i < 6; i++) zf[7 + i] = S.s[i];
        from_prev<13>(zf, za);
        if (!c.has_a) {
#pragma unroll
            for (int i = 0; i < 13; i++) za[i] = (i == 3) ? 1.0 : 0.0;
        }
        if (gate) {
            if (on) {
                double dz[12];
                ck_control_error(zf, C->zd + (size_t)ksp * nz + 13 * t, dz);
#pragma unroll
                for (int i = 0; i < 12; i++) L[Y.DZ + 12 * t + i] = dz[i];
                if (C->has_fric && c.fric != 0.0) uj = ck_friction(c, zf, za);
            }
            __syncthreads();
            for (int i = 0; i < C->mu; i++) {
                double part = 0.0;
                if (C->K && valid) {
                    const double* Krow = C->K + ((size_t)kidx * C->mu + i) * 12 * nb;
                    for (int e = t; e < 12 * nb; e += G) part += Krow[e] * L[Y.DZ + e];
                }
                const double s = group_sum<G>(part);
                double u = (C->Fd ? C->Fd[(size_t)ksp * C->mu + i] : 0.0) - s;
                if (C->noise_scale != 0.0 && valid) {
                    if (a.noise) u += C->noise_scale * a.noise[(size_t)inst * a.noise_stride + (k - 1)];
                    else if (C->noise_philox) u += C->noise_scale * philox_normal(C->noise_key0, (unsigned long long)(a.inst0 + inst), k);
                }
                if (t == C->cj[i]) uj += u;
            }
            __syncthreads();
        }
        if (C->has_pid) {
            if (on && C->pid_on[t]) uj += ck_pid(c, zf, za, C->pid_P[t], C->pid_I[t], C->pid_D[t], C->pid_goal[t], dt, k == 1, pid_int, pid_last);
        }
        STAMP(PF_CONTROL);
        // ---------------- joint inputs -> wrenches, per-step invariants, constraint Jacobians at the current knot, force map
        {
            double F[3], tau[3], W6[6], cW6[6];
            ck_joint_wrench(c, uj, zf + 3, za + 3, F, tau, W6, W6 + 3);
            from_next<6>(W6, cW6);
            if (c.has_c) {
#pragma unroll
                for (int i = 0; i < 3; i++) { F[i] += cW6[i]; tau[i] += cW6[3 + i]; }
            }
            double cTR[6];
            ck_step_invariants(c, zf, F, tau, dt, M->g, cTR, cTR + 3);
            double gk[5], kXT[3][3], kPB[5][3], kPA[5][3], lam[5];
            joint_eval_sparse<true>(c, za, za + 3, zf, zf + 3, nullptr, nullptr, gk, kXT, kPB, kPA);
#pragma unroll
            for (int i = 0; i < 5; i++) lam[i] = L[Y.LAM + 5 * t + i];
            if (on) {
                gk_store(t, Y, L, kXT, kPB, kPA);
#pragma unroll
                for (int i = 0; i < 6; i++) L[Y.D + 6 * t + i] = cTR[i];
            }
            double own[6], par[6], cpar[6];
            jac_t_apply(c, kXT, kPB, kPA, lam, own, par);
            from_next<6>(par, cpar);
#pragma unroll
            for (int i = 0; i < 6; i++) S.cd[i] = 0.0;
            if (on) {
#pragma unroll
                for (int i = 0; i < 6; i++) L[Y.C + 6 * t + i] = own[i] + (c.has_c ? cpar[i] : 0.0);
            }
        }
        __syncthreads();

        STAMP(PF_FORCES);
        PCOUNT(PF_STEPS);
        // ---------------- newton! (tolerances and line search: SURVEY 8a-bis)
        const bool go = valid && !dead;
        bool done = !go, failed = false;
        int its = 0;
        double normf0 = chain_eval<G, true>(c, S, t, Y, L, 0.0, on && !done, dt PROF_PASS);
        __syncthreads();
        const int nchains = M->nchains;
        for (int iter = 1; iter <= NEWTON_MAXIT; iter++) {
            if (!__any(!done)) break;
            PCOUNT(PF_NEWTON_ITERS);
            const bool active = on && !done;
            // block-tridiagonal solve along each chain, swept from both ends (cclqr_dev.h S3)
            for (int ci = 0; ci < nchains; ci++) {
                const TriPlan P = tri_plan(M->chain_start[ci], M->chain_len[ci]);
                for (int i = 0; i < P.steps; i++) {
                    double zy[5];
                    int l = 0;
                    const bool act = !done && ph_tri_elim(t, i, P, Y, L, zy, &l);
                    if (act) ph_tri_store(t, l, Y, L, zy);
                    __syncthreads();
                }
                STAMP(PF_TRI_FWD);
                if (!done) ph_tri_mid(t, P, Y, L);
                __syncthreads();
                for (int j = 0; j < P.steps; j++) {
                    if (!done) ph_tri_back(t, j, P, Y, L);
                    __syncthreads();
                }
                STAMP(PF_TRI_BWD);
            }
            double nd;
            {   // multiplier step from LDS, body solve
                double own[6], par[6], cpar[6], dl[5], pdn = 0.0;
#pragma unroll
                for (int r = 0; r < 5; r++) dl[r] = L[Y.DL + 5 * t + r];
                gk_t_apply(c, t, Y, L, dl, own, par);
                from_next<6>(par, cpar);
                if (active) {
                    double DINV[9];
#pragma unroll
                    for (int i = 0; i < 9; i++) DINV[i] = L[Y.DINV + 9 * t + i];
#pragma unroll
                    for (int i = 0; i < 6; i++) S.cd[i] = own[i] + (c.has_c ? cpar[i] : 0.0);
                    ck_body_solve(c, S.d, S.cd, DINV, S.ds);
#pragma unroll
                    for (int i = 0; i < 6; i++) pdn += S.ds[i] * S.ds[i];
#pragma unroll
                    for (int i = 0; i < 5; i++) pdn += dl[i] * dl[i];
                }
                nd = sqrt(group_sum<G>(pdn));
            }
            __syncthreads();
            STAMP(PF_BODY_SOLVE);
            // line search: halve while ||f|| grows.  The first (full-step) trial also evaluates the Jacobians, speculating that
            // it is accepted; later trials evaluate the residual only.
            double alpha = 1.0, normf1 = 0.0;
            bool ls_done = done, jac_ok = true;
            {
                const double nf = chain_eval<G, true>(c, S, t, Y, L, 1.0, active, dt PROF_PASS);
                if (!ls_done) {
                    normf1 = nf;
                    if (!(normf1 > normf0)) ls_done = true;
                }
            }
            for (int lv = 1; lv <= LINE_MAXIT; lv++) {
                if (!__any(!ls_done)) break;
                const double a_l = ldexp(1.0, -lv);
                const double nf = chain_eval<G, false>(c, S, t, Y, L, a_l, on && !ls_done, dt PROF_PASS);
                if (!ls_done) {
                    normf1 = nf; alpha = a_l; jac_ok = false;
                    if (!(nf > normf0) || lv == LINE_MAXIT) ls_done = true;
                }
            }
            bool need_jac = false;
            if (!done) {
                if (on) {
#pragma unroll
                    for (int i = 0; i < 6; i++) { S.s[i] -= alpha * S.ds[i]; L[Y.C + 6 * t + i] -= alpha * S.cd[i]; S.cd[i] = 0.0; S.ds[i] = 0.0; }
#pragma unroll
                    for (int i = 0; i < 5; i++) L[Y.LAM + 5 * t + i] -= alpha * L[Y.DL + 5 * t + i];
                }
                its = iter;
                if (normf1 < NEWTON_EPS && alpha * nd < NEWTON_EPS) done = true;
                if (!(normf1 < 1e300)) { done = true; failed = true; }   // non-finite residual: the instance has left the domain of the integrator
                normf0 = normf1;
                need_jac = !done && !jac_ok;
            }
            STAMP(PF_ACCEPT);
            if (__any(need_jac)) chain_eval<G, true>(c, S, t, Y, L, 0.0, on && need_jac, dt PROF_PASS);
            __syncthreads();
        }
        const bool conv = done && !failed;
        if (go) {
            if (!conv) bad = true;
            if (its > worst) worst = its;
            if (!conv && its < NEWTON_MAXIT) {   // stopped early on a non-finite residual: freeze the instance at its last pose, at rest
                dead = true;
#pragma unroll
                for (int i = 0; i < 6; i++) S.s[i] = 0.0;
            } else if (on) {
                double xq[7];
                ck_next_pose(S.z, S.s, dt, xq);
#pragma unroll
                for (int i = 0; i < 7; i++) S.z[i] = xq[i];
            }
        }
    }
#ifdef CCLQR_PROFILE
    prof.stamp(PF_IO);
    prof.flush();
#endif
    // ---------------- final state, multipliers, status
    __syncthreads();
    if (on) {
#pragma unroll
        for (int i = 0; i < 7; i++) L[Y.Z + 13 * ut + i] = S.z[i];
#pragma unroll
        for (int i = 0; i < 6; i++) L[Y.Z + 13 * ut + 7 + i] = S.s[i];
    }
    __syncthreads();
    if (valid) {
        for (int e = t; e < nz; e += G) a.zT[inst * nz + e] = L[Y.Z + e];
        if (a.status && t == 0) a.status[inst] = bad ? -worst : worst;
    }
    if (on) {
        if (a.lam) {
#pragma unroll
            for (int i = 0; i < 5; i++) a.lam[inst * 5 * nb + 5 * t + i] = L[Y.LAM + 5 * t + i];
        }
        if (a.pid_state) { a.pid_state[(inst * nb + t) * 2] = pid_int; a.pid_state[(inst * nb + t) * 2 + 1] = pid_last; }
    }
}

#ifdef CCLQR_PROFILE
extern "C" int cclqr_prof_read_chain(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * PF_N);
    if (e == hipSuccess && reset) { unsigned long long z[PF_N] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return e == hipSuccess ? PF_N : -1;
}
#endif

// 16 lanes per instance up to 8 links (the two elimination fronts need 14), 32 beyond: with 16 lanes a 9..16-link instance would
// fill LDS with two wavefronts per CU
int chain_lanes_per_instance(int nb) { return nb <= 8 ? 16 : 32; }

size_t chain_lds_bytes(int nb) { return (size_t)(64 / chain_lanes_per_instance(nb)) * make_chain_layout(nb).total * sizeof(double); }

template <int G>
static hipError_t launch_chain_one(const RolloutArgs& a, unsigned grid, size_t lds, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)rollout_chain_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((rollout_chain_kernel<G>), dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_rollout_chain(const RolloutArgs& a, int nb, hipStream_t stream) {
    const int G = chain_lanes_per_instance(nb);
    const int per_wg = 64 / G;
    const size_t lds = chain_lds_bytes(nb);
    const unsigned grid = (unsigned)((a.n_inst + per_wg - 1) / per_wg);
    if (grid == 0) return hipSuccess;
    return G == 16 ? launch_chain_one<16>(a, grid, lds, stream) : launch_chain_one<32>(a, grid, lds, stream);
}

}  // namespace cclqr
