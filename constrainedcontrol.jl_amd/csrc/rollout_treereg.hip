// rollout_treereg.hip -- the LQR-controlled rollout kernel for BRANCHING trees (a body with several child joints), persistent over the whole
// horizon and REGISTER-resident like the chain kernel (rollout_chain.hip): lane t of an instance's lane group owns link t and keeps its state,
// multipliers, Jacobians and Newton iterate in registers.  The parent link is any earlier lane, so parent data arrive by ds_bpermute instead of
// a wave shift and child contributions are summed over the child list; LDS holds the 5x5 Schur blocks (one more pair per pair of sibling joints),
// eliminated without fill in a host-made schedule (cclqr_treereg.h, cclqr_treereg_tables.h).
//
// Replaces: ConstrainedDynamics.simulate!/newton! as driven by the reference (examples/lqr_cartpole.jl:44) with
//           control_lqr! (src/control/lqr.jl:89-139) / control_trackinglqr! (src/control/lqr_tracking.jl:46-71), on mechanisms the reference
//           reads from URDF files (examples/examples_files/*.urdf) that branch.
#include "cclqr_treereg.h"
#include "cclqr_internal.h"
#include "cclqr_newton.h"

namespace cclqr {

// value of lane `addr / 4` of the wavefront (ds_bpermute: the LDS crossbar, no memory access)
__device__ __forceinline__ double lane_read(double v, int addr) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(addr, lo);
    hi = __builtin_amdgcn_ds_bpermute(addr, hi);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ void from_lane(const double* in, double* out, int addr) {
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = lane_read(in[i], addr);
}
// sum over the child links of the owned body of what their lanes hold in `in` (gb4 = 4 x first lane of the group)
template <int N>
__device__ __forceinline__ void child_sum(const TreeL& T, int gb4, int maxchild, const double* in, double* out) {
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = 0.0;
    int c0 = T.child[0], c1 = T.child[1], c2 = T.child[2], c3 = T.child[3];      // the list rotates through fixed registers (cclqr_treereg.h tr_schur_rows)
#pragma unroll 1
    for (int k = 0; k < maxchild; k++) {
        const int addr = gb4 + 4 * c0;
        const bool has = k < T.nchild;
#pragma unroll
        for (int i = 0; i < N; i++) { const double v = lane_read(in[i], addr); out[i] += has ? v : 0.0; }
        c0 = c1; c1 = c2; c2 = c3;
    }
}

struct TLinkS {
    double z[7], s[6];
    double ds[6], cd[6], d[6];
};

__device__ __forceinline__ void load_rec(TrRec& K, const TrRec* p) {
    const int4* q = (const int4*)p;
    const int4 a = q[0], b = q[1], c = q[2];
    K.o0 = a.x; K.o1 = a.y; K.ctl = a.z; K.pad = a.w;
    K.a[0] = b.x; K.a[1] = b.y; K.a[2] = b.z; K.a[3] = b.w;
    K.b[0] = c.x; K.b[1] = c.y; K.b[2] = c.z; K.b[3] = c.w;
}

// residual (+ Jacobians when JAC) at the point s - alpha ds with constraint forces C - alpha cd; returns the group's ||f||_2.
// With JAC the Schur complement rows of the point go straight to LDS (tr_schur_rows).
template <int G, bool JAC>
__device__ __forceinline__ double tree_eval(LinkC& c, const TreeL& T, TLinkS& S, int t, int pa4, int maxchild, int maxsib, const Lay& Y, double* L, double alpha,
                                            bool active, double dt PROF_ARG) {
    double part = 0.0;
    double NB[9], g[5], xq[7];
    LINK_FLAGS_FRESH(c);
#pragma unroll
    for (int k = 0; k < 7; k++) xq[k] = S.z[k];
#pragma unroll
    for (int k = 0; k < 9; k++) NB[k] = 0.0;
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
    from_lane<7>(xq, pxq, pa4);
    if (JAC) from_lane<9>(NB, pNB, pa4);
    if (!c.has_a()) {
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
    LINK_FLAGS_FRESH(c);
    if (JAC) {
        double pd[6];
        from_lane<6>(S.d, pd, pa4);
        tr_schur_rows(c, T, t, active, maxchild, maxsib, Y, L, wXT, wPB, wPA, g, S.d, pd);
        STAMP(PF_SCHUR_S);
    }
    const double nrm = sqrt(group_sum<G>(part));
    STAMP(PF_EVAL_MAP);
    PCOUNT(PF_EVALS);
    return nrm;
}

// line search of the 32-lane instantiations: two step lengths per pass in a group's own lanes, and the wavefront's other group helps when
// it has nothing to search itself (rollout_chain.hip chain_eval2 / the accept sequence there: unchanged)
struct TTrialIn { double z[7], s[6], ds[6], cd[6]; };
__device__ __forceinline__ double tr_other_half(double v) { return __shfl_xor(v, 32, 64); }
template <int G>
__device__ __forceinline__ void tree_eval2(LinkC& c, const TTrialIn& T, const double* Lc, int t, int pa4, const Lay& Y, double a1, double a2, bool active, double dt,
                                           double& n1, double& n2) {
    double part1 = 0.0, part2 = 0.0, xq1[7], xq2[7];
    LINK_FLAGS_FRESH(c);
#pragma unroll
    for (int k = 0; k < 7; k++) { xq1[k] = T.z[k]; xq2[k] = T.z[k]; }
    if (active) {
        double cf1[6], cf2[6], sv1[6], sv2[6], cTR[6], d1[6], d2[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double cc = Lc[Y.C + 6 * t + k];
            cTR[k] = Lc[Y.D + 6 * t + k];
            cf1[k] = cc - a1 * T.cd[k]; cf2[k] = cc - a2 * T.cd[k];
            sv1[k] = T.s[k] - a1 * T.ds[k]; sv2[k] = T.s[k] - a2 * T.ds[k];
        }
        part1 = ck_body_eval<false>(c, T.z, sv1, cf1, cTR, cTR + 3, dt, xq1, d1, nullptr, nullptr);
        part2 = ck_body_eval<false>(c, T.z, sv2, cf2, cTR, cTR + 3, dt, xq2, d2, nullptr, nullptr);
    }
    double p1[7], p2[7];
    from_lane<7>(xq1, p1, pa4);
    from_lane<7>(xq2, p2, pa4);
    if (!c.has_a()) {
#pragma unroll
        for (int i = 0; i < 7; i++) { p1[i] = (i == 3) ? 1.0 : 0.0; p2[i] = p1[i]; }
    }
    if (active) {
        double g1[5], g2[5];
        joint_eval_sparse<false>(c, p1, p1 + 3, xq1, xq1 + 3, nullptr, nullptr, g1, (double(*)[3]) nullptr, (double(*)[3]) nullptr, (double(*)[3]) nullptr);
        joint_eval_sparse<false>(c, p2, p2 + 3, xq2, xq2 + 3, nullptr, nullptr, g2, (double(*)[3]) nullptr, (double(*)[3]) nullptr, (double(*)[3]) nullptr);
#pragma unroll
        for (int i = 0; i < 5; i++) { part1 += g1[i] * g1[i]; part2 += g2[i] * g2[i]; }
    }
    n1 = sqrt(group_sum<G>(part1));
    n2 = sqrt(group_sum<G>(part2));
}

// G lanes per instance (16 / 32), NBP links the LDS image is laid out for, EXTRA / RELAX as in rollout_chain_kernel
template <int G, int NBP, int EXTRA, bool RELAX = false>
__global__ __launch_bounds__(64) void rollout_treereg_kernel(RolloutArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, t = lane % G, grp = lane / G;
    const int gb4 = 4 * (lane - t);
    const int64_t inst = (int64_t)blockIdx.x * a.ipw + grp;      // (a.ipw instances per wavefront: cclqr_internal.h)
    const MechDev* M = a.M;
    const TreeRegDev* R = treereg_of(M);
    const CtrlDev* C = a.C;
    const int nb = M->nb;
    const double dt = M->dt;
    const int nss = R->nss;
    Lay Y = make_treereg_layout(NBP, 0);
    const int total = (Y.SS + 25 * nss) | 1;
    Y.total = total;
    double* L = lds + grp * total;
    const int nz = 13 * nb;
    const int maxchild = R->maxchild, maxsib = R->maxsib;

    LinkC c;
    link_load_consts(c, M, t, nb, dt);
    TreeL T;
    tree_load(T, M, R, t, nb);
    const int pa4 = gb4 + 4 * T.par;
    if (T.nchild > 0) c.flags |= 4;          // "has a child link" (link_load_consts reads the chains' single-child table)
    if (EXTRA && C->has_fric && c.on()) { c.fric = C->fric[t]; if (c.fric != 0.0) c.flags |= LinkC::FRIC; }
    c.set_valid(grp < a.ipw && inst < a.n_inst);
    const long long ginst = a.inst0 + inst;
    const int ut = c.on() ? M->perm[t] : 0;

    TLinkS S;
    double pid_int = 0.0, pid_last = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) S.z[i] = c.live() ? a.z0[inst * nz + ut * 13 + i] : ((i == 3) ? 1.0 : 0.0);
#pragma unroll
    for (int i = 0; i < 6; i++) S.s[i] = c.live() ? a.z0[inst * nz + ut * 13 + 7 + i] : 0.0;
    if (EXTRA >= 2 && a.pid_state && a.k0 > 1 && c.live()) { pid_int = a.pid_state[(inst * nb + t) * 2]; pid_last = a.pid_state[(inst * nb + t) * 2 + 1]; }
#pragma unroll
    for (int i = 0; i < 6; i++) { S.cd[i] = 0.0; S.d[i] = 0.0; S.ds[i] = 0.0; }
    // as in rollout_chain.hip: the multiplier block is all a Newton phase reads before writing it (tests/emu/emu_treereg.cpp poisons the rest of the image with
    // signalling NaNs; on the hardware tests/test_gpu_lds_poison.py does), so the image is not cleared -- zero, or the caller's warm start
    if (c.on()) {
        const bool warm = c.live() && a.lam && a.k0 > 1;
#pragma unroll
        for (int i = 0; i < 5; i++) L[Y.LAM + 5 * t + i] = warm ? a.lam[inst * 5 * nb + 5 * t + i] : 0.0;
    }

#ifdef CCLQR_PROFILE
    Prof prof;
    prof.start();
#endif
    int worst = 0;
    if (a.carry && a.status && c.valid()) {      // CCLQR_ROLLOUT_CARRY_STATUS (rollout_chain.hip)
        const int carried = a.status[inst];
        worst = carried < 0 ? -carried : carried;
        if (carried < 0) c.flags |= LinkC::BAD;
        if (carried < 0 && carried > -NEWTON_MAXIT) c.flags |= LinkC::DEAD;
    }
    typedef const __attribute__((address_space(4))) RolloutArgs* KernArgs;
    KernArgs ap = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const int k0 = a.k0;
    int nsteps = a.steps;
    for (int kk = 0; kk < nsteps; kk++) {
        const int k = k0 + kk;
        asm volatile("" : "+s"(ap));
        LINK_FLAGS_FRESH(c);
        double* const traj_out = ap->traj;
        if (traj_out) {     // Storage row of this step, staged through LDS in user body order so that the HBM stores coalesce
            if (c.live()) {
#pragma unroll
                for (int i = 0; i < 7; i++) L[Y.Z + 13 * ut + i] = S.z[i];
#pragma unroll
                for (int i = 0; i < 6; i++) L[Y.Z + 13 * ut + 7 + i] = S.s[i];
            }
            __syncthreads();
            if (c.valid()) {
                int kr = kk, nzr = nz;
                asm volatile("" : "+s"(kr), "+s"(nzr));
                double* dst = traj_out + ((size_t)inst * ap->steps + kr) * nzr;
                int e0 = t;
                asm volatile("" : "+v"(e0));
                for (int e = e0; e < nz; e += G) dst[e] = L[Y.Z + e];
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
        from_lane<7>(zf, za, pa4);                      // parent pose (every joint evaluation needs it)
        if (EXTRA) from_lane<6>(zf + 7, za + 7, pa4);   // parent velocities (friction, PID)
        if (!c.has_a()) {
#pragma unroll
            for (int i = 0; i < 13; i++) za[i] = (i == 3) ? 1.0 : 0.0;
        }
        if (gate) {
            if (c.live()) {
                double dz[12];
                ck_control_error(zf, C->zd + ginst * C->zd_stride + (size_t)ksp * nz + 13 * t, dz);
#pragma unroll
                for (int i = 0; i < 12; i++) L[Y.DZ + 12 * t + i] = dz[i];
                if (EXTRA && C->has_fric && c.has_fric()) uj = ck_friction(c, zf, za);
            }
            __syncthreads();
            {
                // u_i = Fd_i - K_i . dz: the gain entries of a lane are all requested before the first is used, CH inputs at a time (the whole
                // arm's on the 16-lane kernels), entries past a row's end meet a zero in dz (the tables are padded: CCLQR_K_PAD) -- the control
                // phase of rollout_chain.hip, where the measurement and the reasons are
                constexpr int NE = (12 * NBP + G - 1) / G;
                constexpr int CH = (G == 16) ? 8 : 1;
                const long long gi = c.valid() ? ginst : a.inst0;
                const int ne = 12 * nb;
                double unoise = 0.0;
                if (EXTRA) {
                    const double* noise = ap->noise;
                    if (C->noise_scale != 0.0 && c.valid() && noise) unoise = C->noise_scale * noise[(size_t)inst * ap->noise_stride + (k - 1)];
                }
                double dzv[NE];
                int tf = t;
                asm volatile("" : "+v"(tf));            // the entries' range tests are made here, every step -- not once per launch and kept as NE lane masks
#pragma unroll
                for (int q = 0; q < NE; q++) { const int e = tf + q * G; dzv[q] = (c.valid() && e < ne) ? L[Y.DZ + e] : 0.0; }
                const int mu = C->mu;
                const double* Fp = C->Fd ? C->Fd + gi * C->Fd_stride + (size_t)ksp * mu : nullptr;
                if (C->K) {                                  // (uniform) LQR / TrackingLQR
                    const double* Kp = C->K + gi * C->K_stride + (size_t)kidx * mu * ne + t;
                    for (int i0 = 0; i0 < mu; i0 += CH) {
                        double kv[CH][NE], fd[CH];
                        int cjv[CH];
#pragma unroll
                        for (int j = 0; j < CH; j++) {
                            const bool ok = i0 + j < mu;         // (uniform)
                            const int ij = ok ? i0 + j : i0;
#pragma unroll
                            for (int q = 0; q < NE; q++) kv[j][q] = Kp[(size_t)ij * ne + q * G];
                            fd[j] = Fp ? Fp[ij] : 0.0;
                            cjv[j] = ok ? C->cj[ij] : -1;
                        }
#pragma unroll
                        for (int j = 0; j < CH; j++) {
                            double part = 0.0;
#pragma unroll
                            for (int q = 0; q < NE; q++) part += kv[j][q] * dzv[q];
                            const double s = group_sum<G>(part);
                            double u = fd[j] - s;
                            if (EXTRA) u += unoise;
                            if (t == cjv[j]) uj += u;
                        }
                    }
                } else {                                     // feed-forward only (OpenLoop, a host closure's inputs)
                    for (int i = 0; i < mu; i++) {
                        double u = Fp ? Fp[i] : 0.0;
                        if (EXTRA) u += unoise;
                        if (t == C->cj[i]) uj += u;
                    }
                }
            }
            __syncthreads();
        }
        if (EXTRA >= 2 && C->has_pid) {
            if (c.live() && C->pid_on[t]) uj += ck_pid(c, zf, za, C->pid_P[t], C->pid_I[t], C->pid_D[t], C->pid_goal[t], dt, k == 1, pid_int, pid_last);
        }
        STAMP(PF_CONTROL);
        LINK_FLAGS_FRESH(c);
        // ---------------- joint inputs -> wrenches, per-step invariants, constraint Jacobians at the current knot, force map
        {
            double F[3], tau[3], W6[6], cW6[6];
            ck_joint_wrench(c, uj, zf + 3, za + 3, F, tau, W6, W6 + 3);
            child_sum<6>(T, gb4, maxchild, W6, cW6);
#pragma unroll
            for (int i = 0; i < 3; i++) { F[i] += cW6[i]; tau[i] += cW6[3 + i]; }
            double cTR[6];
            ck_step_invariants(c, zf, F, tau, dt, M->g, cTR, cTR + 3);
            double gk[5], kXT[3][3], kPB[5][3], kPA[5][3], lam[5];
            joint_eval_sparse<true>(c, za, za + 3, zf, zf + 3, nullptr, nullptr, gk, kXT, kPB, kPA);
#pragma unroll
            for (int i = 0; i < 5; i++) lam[i] = L[Y.LAM + 5 * t + i];
            if (c.live()) {
                gk_store(t, Y, L, kXT, kPB, kPA);
#pragma unroll
                for (int i = 0; i < 6; i++) L[Y.D + 6 * t + i] = cTR[i];
            }
            double own[6], par[6], cpar[6];
            jac_t_apply(c, kXT, kPB, kPA, lam, own, par);
            child_sum<6>(T, gb4, maxchild, par, cpar);
#pragma unroll
            for (int i = 0; i < 6; i++) S.cd[i] = 0.0;
            if (c.live()) {
#pragma unroll
                for (int i = 0; i < 6; i++) L[Y.C + 6 * t + i] = own[i] + cpar[i];
            }
        }
        __syncthreads();

        STAMP(PF_FORCES);
        PCOUNT(PF_STEPS);
        // ---------------- newton! (tolerances and line search: SURVEY 8a-bis)
        const bool go = c.valid() && !c.dead();
        bool done = !go, failed = false;
        int its = 0;
        double normf0 = tree_eval<G, true>(c, T, S, t, pa4, maxchild, maxsib, Y, L, 0.0, c.live() && !done, dt PROF_PASS);
        __syncthreads();
        const int ne_steps = R->ne_steps, nb_steps = R->nb_steps;
        for (int iter = 1; iter <= NEWTON_MAXIT; iter++) {
            if (!__any(!done)) break;
            PCOUNT(PF_NEWTON_ITERS);
            const bool active = c.live() && !done;
            // scheduled no-fill elimination (leaves towards the roots), then the multiplier steps (roots towards the leaves)
            {   // the record of a step is fetched from the tables one step AHEAD (an L2 round trip costs as much as the step's arithmetic)
                const TrRec* rec = &R->el[0][t];
                TrRec K, Kn;
                load_rec(K, rec);
                for (int s = 0; s < ne_steps; s++) {
                    load_rec(Kn, s + 1 < ne_steps ? rec + (s + 1) * TR_LANES : &R->bk[0][t]);
                    if (done) K.ctl = 0;
                    tr_elim(K, L);
                    __syncthreads();
                    K = Kn;
                }
                STAMP(PF_TRI_FWD);
                rec = &R->bk[0][t];
                for (int s = 0; s < nb_steps; s++) {
                    load_rec(Kn, rec + (s + 1 < nb_steps ? s + 1 : s) * TR_LANES);
                    if (done) K.ctl = 0;
                    tr_back(K, L);
                    __syncthreads();
                    K = Kn;
                }
                STAMP(PF_TRI_BWD);
            }
            double nd;
            LINK_FLAGS_FRESH(c);
            {   // multiplier step from LDS, body solve
                double own[6], par[6], cpar[6], dl[5], pdn = 0.0;
#pragma unroll
                for (int r = 0; r < 5; r++) dl[r] = L[Y.DL + 5 * t + r];
                gk_t_apply(c, t, Y, L, dl, own, par);
                child_sum<6>(T, gb4, maxchild, par, cpar);
                if (active) {
                    double DINV[9];
#pragma unroll
                    for (int i = 0; i < 9; i++) DINV[i] = L[Y.DINV + 9 * t + i];
#pragma unroll
                    for (int i = 0; i < 6; i++) S.cd[i] = own[i] + cpar[i];
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
            // line search: halve while ||f|| grows; the full-step trial also evaluates the Jacobians and the Schur blocks (rollout_chain.hip)
            double alpha = 1.0, normf1 = 0.0;
            bool ls_done = done, jac_ok = true;
            {
                const double nf = tree_eval<G, true>(c, T, S, t, pa4, maxchild, maxsib, Y, L, 1.0, active, dt PROF_PASS);
                if (!ls_done) {
                    normf1 = nf;
                    if (!(normf1 > normf0)) ls_done = true;
                }
            }
            if (G == 32) {
                for (int lv = 1; lv <= LINE_MAXIT;) {
                    if (!__any(!ls_done)) break;
                    const bool mine = !ls_done;
                    const bool other = __shfl_xor(mine ? 1 : 0, 32, 64) != 0;
                    const bool helping = !mine && other;
                    TTrialIn Tr;
#pragma unroll
                    for (int i = 0; i < 7; i++) Tr.z[i] = S.z[i];
#pragma unroll
                    for (int i = 0; i < 6; i++) { Tr.s[i] = S.s[i]; Tr.ds[i] = S.ds[i]; Tr.cd[i] = S.cd[i]; }
                    if (mine != other) {
#pragma unroll
                        for (int i = 0; i < 7; i++) { const double o = tr_other_half(S.z[i]); Tr.z[i] = helping ? o : Tr.z[i]; }
#pragma unroll
                        for (int i = 0; i < 6; i++) {
                            const double o1 = tr_other_half(S.s[i]), o2 = tr_other_half(S.ds[i]), o3 = tr_other_half(S.cd[i]);
                            Tr.s[i] = helping ? o1 : Tr.s[i]; Tr.ds[i] = helping ? o2 : Tr.ds[i]; Tr.cd[i] = helping ? o3 : Tr.cd[i];
                        }
                    }
                    const double* Lc = helping ? lds + (1 - grp) * total : L;
                    const int l0 = helping ? lv + 2 : lv;
                    double n1, n2;
                    tree_eval2<G>(c, Tr, Lc, t, pa4, Y, ldexp(1.0, -l0), ldexp(1.0, -(l0 + 1)), c.on() && (mine || helping) && l0 <= LINE_MAXIT, dt, n1, n2);
                    PCOUNT(PF_EVALS);
                    const double h1 = tr_other_half(n1), h2 = tr_other_half(n2);
                    if (mine) {
                        const double cand[4] = {n1, n2, h1, h2};
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int l = lv + i;
                            if (!ls_done && l <= LINE_MAXIT && (i < 2 || !other)) {
                                normf1 = cand[i]; alpha = ldexp(1.0, -l); jac_ok = false;
                                if (!(cand[i] > normf0) || l == LINE_MAXIT) ls_done = true;
                            }
                        }
                    }
                    lv += (mine && other) ? 2 : 4;
                }
            } else {
                for (int lv = 1; lv <= LINE_MAXIT; lv++) {
                    if (!__any(!ls_done)) break;
                    const double a_l = ldexp(1.0, -lv);
                    const double nf = tree_eval<G, false>(c, T, S, t, pa4, maxchild, maxsib, Y, L, a_l, c.live() && !ls_done, dt PROF_PASS);
                    if (!ls_done) {
                        normf1 = nf; alpha = a_l; jac_ok = false;
                        if (!(nf > normf0) || lv == LINE_MAXIT) ls_done = true;
                    }
                }
            }
            bool need_jac = false;
            if (!done) {
                if (c.live()) {
#pragma unroll
                    for (int i = 0; i < 6; i++) { S.s[i] -= alpha * S.ds[i]; L[Y.C + 6 * t + i] -= alpha * S.cd[i]; S.cd[i] = 0.0; S.ds[i] = 0.0; }
#pragma unroll
                    for (int i = 0; i < 5; i++) L[Y.LAM + 5 * t + i] -= alpha * L[Y.DL + 5 * t + i];
                }
                its = iter;
                if (normf1 < NEWTON_EPS && alpha * nd < NEWTON_EPS) done = true;
                if (RELAX && normf1 < ap->eps_alone) done = true;      // measured-error mode: the residual alone
                if (!(normf1 < 1e300)) { done = true; failed = true; }
                normf0 = normf1;
                need_jac = !done && !jac_ok;
            }
            STAMP(PF_ACCEPT);
            if (__any(need_jac)) tree_eval<G, true>(c, T, S, t, pa4, maxchild, maxsib, Y, L, 0.0, c.live() && need_jac, dt PROF_PASS);
            __syncthreads();
        }
        const bool conv = done && !failed;
        if (go) {
            if (!conv) c.flags |= LinkC::BAD;
            if (its > worst) worst = its;
            if (!conv && its < NEWTON_MAXIT) {   // stopped early on a non-finite residual: freeze the instance at its last pose, at rest
                c.flags |= LinkC::DEAD;
#pragma unroll
                for (int i = 0; i < 6; i++) S.s[i] = 0.0;
            } else if (c.live()) {
                double xq[7];
                ck_next_pose(S.z, S.s, dt, xq);
#pragma unroll
                for (int i = 0; i < 7; i++) S.z[i] = xq[i];
            }
        }
        asm volatile("" : "+s"(ap));
        nsteps = ap->steps;
    }
#ifdef CCLQR_PROFILE
    prof.stamp(PF_IO);
    prof.flush();
#endif
    // ---------------- final state, multipliers, status
    __syncthreads();
    if (c.live()) {
#pragma unroll
        for (int i = 0; i < 7; i++) L[Y.Z + 13 * ut + i] = S.z[i];
#pragma unroll
        for (int i = 0; i < 6; i++) L[Y.Z + 13 * ut + 7 + i] = S.s[i];
    }
    __syncthreads();
    asm volatile("" : "+s"(ap));
    LINK_FLAGS_FRESH(c);
    if (c.valid()) {
        double* zT = ap->zT;
        int* status = ap->status;
        for (int e = t; e < nz; e += G) zT[inst * nz + e] = L[Y.Z + e];
        if (status && t == 0) status[inst] = c.bad() ? -((ap->carry && c.dead() && worst >= NEWTON_MAXIT) ? NEWTON_MAXIT - 1 : worst) : worst;
    }
    if (c.live()) {
        const int nbT = ap->M->nb;
        double* lam = ap->lam;
        if (lam) {
#pragma unroll
            for (int i = 0; i < 5; i++) lam[inst * 5 * nbT + 5 * t + i] = L[Y.LAM + 5 * t + i];
        }
        if (EXTRA >= 2) {
            double* pid_state = ap->pid_state;
            if (pid_state) { pid_state[(inst * nbT + t) * 2] = pid_int; pid_state[(inst * nbT + t) * 2 + 1] = pid_last; }
        }
    }
}

#ifdef CCLQR_PROFILE
extern "C" int cclqr_prof_read_treereg(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * PF_N);
    if (e == hipSuccess && reset) { unsigned long long z[PF_N] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return e == hipSuccess ? PF_N : -1;
}
#endif

size_t treereg_lds_bytes(int nb, int tree8, int npairs) {
    return (size_t)(64 / treereg_lanes(nb, tree8)) * make_treereg_layout(treereg_layout_links(nb, tree8), 2 * npairs).total * sizeof(double);
}

template <int G, int NBP>
static hipError_t launch_treereg_one(const RolloutArgs& a, int extra, int newton_mode, unsigned grid, size_t lds, hipStream_t stream) {
    const bool relax = newton_mode != 0 && extra == 0;
    const void* f = relax ? (const void*)rollout_treereg_kernel<G, NBP, 0, true>
                          : (extra == 0 ? (const void*)rollout_treereg_kernel<G, NBP, 0> : (extra == 1 ? (const void*)rollout_treereg_kernel<G, NBP, 1> : (const void*)rollout_treereg_kernel<G, NBP, 2>));
    hipError_t e = set_max_dynamic_lds_once(f, lds);
    if (e != hipSuccess) return e;
    if (relax) hipLaunchKernelGGL((rollout_treereg_kernel<G, NBP, 0, true>), dim3(grid), dim3(64), lds, stream, a);
    else if (extra == 0) hipLaunchKernelGGL((rollout_treereg_kernel<G, NBP, 0>), dim3(grid), dim3(64), lds, stream, a);
    else if (extra == 1) hipLaunchKernelGGL((rollout_treereg_kernel<G, NBP, 1>), dim3(grid), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((rollout_treereg_kernel<G, NBP, 2>), dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

// a.M must be the device image [MechDev | TreeRegDev] of capi.hip (treereg_of)
hipError_t launch_rollout_treereg(const RolloutArgs& a_in, int nb, int tree8, int npairs, int extra, int newton_mode, hipStream_t stream) {
    const int G = treereg_lanes(nb, tree8), nbp = treereg_layout_links(nb, tree8);
    const int per_wg = spread_instances_per_wavefront(64 / G, a_in.n_inst, a_in.steps, a_in.ipw != 0);
    RolloutArgs a = a_in;
    a.ipw = per_wg;
    const size_t lds = treereg_lds_bytes(nb, tree8, npairs);
    const unsigned grid = (unsigned)((a.n_inst + per_wg - 1) / per_wg);
    if (grid == 0) return hipSuccess;
    if (nb > G || nb > nbp) return hipErrorInvalidValue;
    if (G == 16) {
#if TR_G16_MAXLINKS > 8
        if (nbp == 14) return launch_treereg_one<16, 14>(a, extra, newton_mode, grid, lds, stream);
        if (nbp > 8) return hipErrorInvalidValue;
#endif
        return nbp == 4 ? launch_treereg_one<16, 4>(a, extra, newton_mode, grid, lds, stream) : launch_treereg_one<16, 8>(a, extra, newton_mode, grid, lds, stream);
    }
    if (G == 64) return nbp == 48 ? launch_treereg_one<64, 48>(a, extra, newton_mode, grid, lds, stream) : launch_treereg_one<64, 64>(a, extra, newton_mode, grid, lds, stream);
    switch (nbp) {
        case 8: return launch_treereg_one<32, 8>(a, extra, newton_mode, grid, lds, stream);
        case 10: return launch_treereg_one<32, 10>(a, extra, newton_mode, grid, lds, stream);
        case 12: return launch_treereg_one<32, 12>(a, extra, newton_mode, grid, lds, stream);
        case 14: return launch_treereg_one<32, 14>(a, extra, newton_mode, grid, lds, stream);
        case 16: return launch_treereg_one<32, 16>(a, extra, newton_mode, grid, lds, stream);
        case 24: return launch_treereg_one<32, 24>(a, extra, newton_mode, grid, lds, stream);
        default: return launch_treereg_one<32, 32>(a, extra, newton_mode, grid, lds, stream);
    }
}

}  // namespace cclqr
