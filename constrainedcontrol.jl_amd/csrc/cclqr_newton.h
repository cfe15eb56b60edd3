// cclqr_newton.h -- device-only orchestration of the Newton solve shared by the rollout and linearisation kernels
// (tolerances and line search: SURVEY 8a-bis; the same sequence of values as oracle/cclqr_oracle.c newton()).
#pragma once
#include "cclqr_dev.h"

namespace cclqr {

// ---- optional in-kernel phase stamps (diagnostic build only: -DCCLQR_PROFILE; the shipped library contains none) ----
#ifdef CCLQR_PROFILE
enum { PF_CONTROL, PF_FORCES, PF_EVAL_BODY, PF_EVAL_JOINT, PF_EVAL_MAP, PF_SCHUR_W, PF_SCHUR_S, PF_TRI_FWD, PF_TRI_BWD, PF_BODY_SOLVE, PF_TRIAL,
       PF_ACCEPT, PF_IO, PF_NEWTON_ITERS, PF_EVALS, PF_STEPS, PF_N };
static __device__ unsigned long long g_prof[PF_N];
struct Prof {
    unsigned long long t0, acc[PF_N];
    __device__ void start() { for (int i = 0; i < PF_N; i++) acc[i] = 0; t0 = __builtin_readcyclecounter(); }
    __device__ void stamp(int c) { unsigned long long t1 = __builtin_readcyclecounter(); acc[c] += t1 - t0; t0 = t1; }
    __device__ void count(int c, int n = 1) { acc[c] += n; }
    __device__ void flush() { if ((threadIdx.x & 63) == 0) for (int i = 0; i < PF_N; i++) atomicAdd(&g_prof[i], acc[i]); }
};
#define PROF_ARG , Prof& prof
#define PROF_PASS , prof
#define STAMP(c) prof.stamp(c)
#define PCOUNT(c) prof.count(c)
#else
#define PROF_ARG
#define PROF_PASS
#define STAMP(c)
#define PCOUNT(c)
#endif

#define NEWTON_EPS 1e-10
#define NEWTON_MAXIT 100
#define LINE_MAXIT 10

// 16-lane row rotation through the DPP crossbar (no LDS round trip)
template <int N>
__device__ __forceinline__ double dpp_row_ror(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sum over the G lanes of a group, result in every lane of the group: rotate-and-add inside each 16-lane row (DPP), then
// combine rows (G = 32: one cross-row permute; G = 64: four scalar lane reads)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
    if (G == 8) {     // 8-lane groups (half rows): mirror inside the half row, then the two quad exchanges; every lane adds the same
                      // pairs (in either operand order), so all eight end up with the same bits
        v += dpp_f64<0x141>(v);     // row_half_mirror: lane i <-> 7 - i
        v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
        v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
        return v;
    }
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<1>(v);
    // rows 0+1 and 2+3 through the LDS crossbar (ds_bpermute).  gfx950's v_permlane16_swap does the same exchange in the vector
    // ALU, but its two result pairs cost the 17-link instantiations 4 vector registers they do not have (VGPR spills; round 2,
    // 6ef5c0b) for no measured gain
    if (G == 32) v += __shfl_xor(v, 16, 64);
    if (G == 64) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
        double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
        double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
        double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
        v = (r0 + r1) + (r2 + r3);
    }
    return v;
}

// The LDS layout is 26 offsets that are all functions of nb.  Held as scalars through the kernel they cost 26 SGPRs for its whole
// lifetime (the tree kernels spilled 74-80 scalars); recomputed where a phase is called -- a few scalar multiply-adds on a value the
// optimiser cannot see through -- they live only inside that phase.  (SS, the sibling blocks, does not depend on their number.)
__device__ __forceinline__ Lay fresh_layout(int nb) {
    asm volatile("" : "+s"(nb));
    return make_layout(nb, 0);
}
#define FRESH_Y fresh_layout(nb)
// The same for per-lane predicates (t < nb, t in an elimination front, ...): every phase starts from a lane index the optimiser
// cannot relate to the one of the previous phase, so its lane masks (64-bit scalars) are formed inside the phase and die with it
// instead of being hoisted out of the step loop and kept -- or spilled -- for the whole launch.
__device__ __forceinline__ int fresh_lane(int t) {
    asm volatile("" : "+v"(t));
    return t;
}
#define FRESH_T fresh_lane(t)
// ... and for the predicates on the owned link's integers (joint type, parent, child, row kinds) and the addresses of the mechanism's
// tables (M + constant): re-derived per phase instead of held
#define FRESH_PHASE(M, r) asm volatile("" : "+s"(M), "+v"((r).parent), "+v"((r).childl), "+v"((r).rotmask), "+v"((r).type))

// residual (+ Jacobians when JAC) at the point s_off with multipliers lambda - alpha dlambda; returns the group's ||f||_2
template <int G, bool JAC>
__device__ __forceinline__ double eval_point(int t, int nb, const Lay& Y, double* L, LaneRegs& r, const MechDev* M, double dt,
                                             int s_off, double alpha, bool active PROF_ARG) {
    FRESH_PHASE(M, r);
    double part = active ? ph_body_eval<JAC>(FRESH_T, nb, FRESH_Y, L, r, dt, s_off, alpha) : 0.0;
    __syncthreads();
    STAMP(PF_EVAL_BODY);
    FRESH_PHASE(M, r);
    part += active ? ph_joint_eval<JAC>(FRESH_T, nb, FRESH_Y, L, r, dt) : 0.0;
    __syncthreads();
    STAMP(PF_EVAL_JOINT);
    double nrm = sqrt(group_sum<G>(part));
    STAMP(PF_EVAL_MAP);
    PCOUNT(PF_EVALS);
    return nrm;
}

// newton! on the instance held in LDS (S/LAM = guess in, solution out; XQ = next pose of the solution).
// valid=false groups only keep the wave's control flow uniform.  Returns iterations used; *converged reports success.
// After a successful return DINV/NB/GV* hold the values at the solution only if the last accepted trial was a full step
// (callers that need them -- the linearisation -- re-evaluate).
template <int G, bool TREE = false>
__device__ __forceinline__ int newton_solve(int t, int nb, const Lay& Y, double* L, LaneRegs& r, const MechDev* M, double dt, bool valid,
                                            bool* converged PROF_ARG) {
    double normf0 = eval_point<G, true>(t, nb, FRESH_Y, L, r, M, dt, Y.S, 0.0, valid PROF_PASS);
    bool done = !valid, failed = false;
    int its = 0;
    const unsigned long long smask = M->start_mask, emask = M->end_mask;
    const int nchains = M->nchains;
    int s_cur = Y.S, s_try = Y.ST, l_cur = Y.LAM, l_try = Y.LT;   // accepted / trial buffers swap roles on every acceptance
    // lane groups of the level-parallel line search: group lg works on link tl (the caller has loaded link tl's constants and
    // per-step invariants into r for every lane with lg < NL)
    const int NL = newton_level_groups(G, nb), lg = t / nb, tl = t - lg * nb;
    for (int iter = 1; iter <= NEWTON_MAXIT; iter++) {
        if (!__any(!done)) break;
        PCOUNT(PF_NEWTON_ITERS);
        FRESH_PHASE(M, r);
        if (TREE) {
            // general tree: sibling-coupled Schur complement, table-driven elimination leaves -> roots, back substitution roots -> leaves
            if (!done) ph_schur_s_tree(FRESH_T, G, nb, FRESH_Y, L, M);
            __syncthreads();
            STAMP(PF_SCHUR_S);
            FRESH_PHASE(M, r);
            {   // offsets and lane index re-derived once for the whole sweep (not per link)
                const Lay Ye = FRESH_Y;
                const int te = FRESH_T;
                for (int l = nb - 1; l >= 0; l--) {
                    if (!done) ph_tree_elim(te, l, Ye, L, M);
                    __syncthreads();
                }
            }
            STAMP(PF_TRI_FWD);
            FRESH_PHASE(M, r);
            {
                const Lay Yb = FRESH_Y;
                const int tb = FRESH_T;
                for (int l = 0; l < nb; l++) {
                    if (!done) ph_tree_back(tb, l, Yb, L, M);
                    __syncthreads();
                }
            }
            STAMP(PF_TRI_BWD);
            FRESH_PHASE(M, r);
            if (!done) ph_body_solve_tree(FRESH_T, G, nb, FRESH_Y, L, M);
            __syncthreads();
            STAMP(PF_BODY_SOLVE);
        } else {
        // Schur complement on the multipliers
        if (!done) ph_schur_s(FRESH_T, G, nb, FRESH_Y, L, smask);
        __syncthreads();
        STAMP(PF_SCHUR_S);
        // block-tridiagonal solve along each chain, swept from both ends
        for (int c = 0; c < nchains; c++) {
            const TriPlan P = tri_plan(M->chain_start[c], M->chain_len[c]);
            for (int i = 0; i < P.steps; i++) {
                double lu[5];
                int l = 0;
                bool act = !done && ph_tri_elim(FRESH_T, i, P, FRESH_Y, L, lu, &l);
                // no barrier here: the store overwrites what the same wavefront has already loaded (LDS is in order per wavefront)
                if (act) ph_tri_store(FRESH_T, l, FRESH_Y, L, lu);
                __syncthreads();
            }
            STAMP(PF_TRI_FWD);
            if (!done) ph_tri_mid(FRESH_T, P, FRESH_Y, L);
            __syncthreads();
            for (int j = 0; j < P.steps; j++) {
                if (!done) ph_tri_back(FRESH_T, j, P, FRESH_Y, L);
                __syncthreads();
            }
            STAMP(PF_TRI_BWD);
        }
        FRESH_PHASE(M, r);
        if (!done) ph_body_solve(FRESH_T, G, nb, FRESH_Y, L, emask);
        __syncthreads();
        STAMP(PF_BODY_SOLVE);
        }
        // line search: halve while ||f|| grows.  The first (full-step) trial also evaluates the Jacobians, speculating that it
        // is accepted; later trials evaluate the residual only.
        double alpha = 1.0, normf1 = 0.0, nd = 0.0;
        bool ls_done = done, jac_ok = true;
        {   // full step (ls = 0), with the Jacobians
            double pd = ls_done ? 0.0 : ph_trial(FRESH_T, G, nb, FRESH_Y, L, alpha, s_cur, s_try, l_cur, l_try);
            nd = sqrt(group_sum<G>(pd));
            __syncthreads();
            STAMP(PF_TRIAL);
            double nf = eval_point<G, true>(t, nb, FRESH_Y, L, r, M, dt, s_try, alpha, !ls_done PROF_PASS);
            if (!ls_done) {
                normf1 = nf;
                if (!(normf1 > normf0)) ls_done = true;
            }
            __syncthreads();
        }
        // halvings, NL levels per pass (see level_layout): lane group lg tests alpha = 2^-(lv + lg)
        for (int lv = 1; lv <= LINE_MAXIT; lv += NL) {
            if (!__any(!ls_done)) break;
            FRESH_PHASE(M, r);
            const int mylv = lv + lg;
            const bool lane_on = !ls_done && lg < NL && mylv <= LINE_MAXIT;
            const double a_l = ldexp(1.0, -mylv);
            const Lay V = level_layout(FRESH_Y, nb, lg < NL ? lg : 0);
            if (lane_on) ph_trial_level(fresh_lane(tl), nb, FRESH_Y, V, L, a_l, s_cur, l_cur);
            __syncthreads();
            STAMP(PF_TRIAL);
            double part = lane_on ? ph_body_eval<false>(fresh_lane(tl), nb, V, L, r, dt, V.ST, a_l) : 0.0;
            __syncthreads();
            STAMP(PF_EVAL_BODY);
            part += lane_on ? ph_joint_eval<false>(fresh_lane(tl), nb, V, L, r, dt) : 0.0;
            __syncthreads();
            STAMP(PF_EVAL_JOINT);
            double nfq[LEVEL_SLOTS];
#pragma unroll
            for (int q = 0; q < LEVEL_SLOTS; q++) nfq[q] = (q < NL) ? sqrt(group_sum<G>(lg == q ? part : 0.0)) : 0.0;
            STAMP(PF_EVAL_MAP);
            PCOUNT(PF_EVALS);
            int chosen = -1;
            if (!ls_done) {
#pragma unroll
                for (int q = 0; q < LEVEL_SLOTS; q++)
                    if (chosen < 0 && q < NL && lv + q <= LINE_MAXIT && (!(nfq[q] > normf0) || lv + q == LINE_MAXIT)) { chosen = q; normf1 = nfq[q]; alpha = ldexp(1.0, -(lv + q)); }
                if (chosen < 0 && lv + NL > LINE_MAXIT) chosen = -2;   // cannot happen: level LINE_MAXIT always accepts
                if (chosen >= 0) { ls_done = true; jac_ok = false; }
            }
            if (chosen >= 0) ph_level_commit(FRESH_T, nb, Y, level_layout(FRESH_Y, nb, chosen), L, s_try, l_try);
            __syncthreads();
        }
        bool need_jac = false;
        // all groups of the wavefront swap together: groups that are already done copy nothing and keep their solution where
        // it is, so the swap is only applied to a group's own view
        FRESH_PHASE(M, r);
        if (!done) {
            ph_accept(FRESH_T, G, nb, FRESH_Y, L, alpha);
            { int q = s_cur; s_cur = s_try; s_try = q; q = l_cur; l_cur = l_try; l_try = q; }
            its = iter;
            if (normf1 < NEWTON_EPS && alpha * nd < NEWTON_EPS) done = true;
            if (!(normf1 < 1e300)) { done = true; failed = true; }   // non-finite residual: the instance has left the domain of the integrator
            normf0 = normf1;
            need_jac = !done && !jac_ok;
        }
        __syncthreads();
        STAMP(PF_ACCEPT);
        if (__any(need_jac)) eval_point<G, true>(t, nb, FRESH_Y, L, r, M, dt, s_cur, 0.0, need_jac PROF_PASS);
    }
    ph_copy_solution(FRESH_T, G, nb, FRESH_Y, L, s_cur, l_cur);
    __syncthreads();
    *converged = done && !failed;
    return its;
}

}  // namespace cclqr
