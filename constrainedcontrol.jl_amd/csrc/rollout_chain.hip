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
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_from_next(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
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
// KL > 1 (several lanes per link, cclqr_chain.h): t is the LINK of the lane (its sub-lane is Q.w); with Jacobians the lane evaluates the rows of its slots only
// and builds their Schur rows, the body's share of the norm comes from the primary sub-lane alone.  KL = 1: the code of rounds 2-4, unchanged.
template <int G, bool JAC, int KL = 1>
__device__ __forceinline__ double chain_eval(LinkC& c, LinkS& S, int t, const Lay& Y, double* L, double alpha, bool active, double dt, const SubSel& Q PROF_ARG) {
    double part = 0.0;
    double NB[9], g[5], xq[7];
    LINK_FLAGS_FRESH(c);
#pragma unroll
    for (int k = 0; k < 7; k++) xq[k] = S.z[k];
    if (active) {
        double cf[6], sv[6], cTR[6], DINV[9];
#pragma unroll
        for (int k = 0; k < 6; k++) { cf[k] = L[Y.C + 6 * t + k] - alpha * S.cd[k]; sv[k] = S.s[k] - alpha * S.ds[k]; cTR[k] = L[Y.D + 6 * t + k]; }
        part = ck_body_eval<JAC>(c, S.z, sv, cf, cTR, cTR + 3, dt, xq, S.d, DINV, NB);
        if (KL > 1 && !c.prim()) part = 0.0;
        if (JAC && (KL == 1 || c.prim())) {
#pragma unroll
            for (int k = 0; k < 9; k++) L[Y.DINV + 9 * t + k] = DINV[k];
        }
    }
    STAMP(PF_EVAL_BODY);
    double pxq[7], pNB[9];
    from_prev<7>(xq, pxq);
    if (JAC) from_prev<9>(NB, pNB);
    if (!c.has_a()) {
#pragma unroll
        for (int i = 0; i < 7; i++) pxq[i] = (i == 3) ? 1.0 : 0.0;
    }
    if (JAC && KL > 1) {
        constexpr int NR = SubRows<KL>::NR;
        double gs[NR], sXT[NR][3], sPB[NR][3], sPA[NR][3];
        if (active) {
            joint_eval_rows<KL>(c, Q, pxq, pxq + 3, xq, xq + 3, pNB, NB, gs, sXT, sPB, sPA);
#pragma unroll
            for (int i = 0; i < NR; i++) part += gs[i] * gs[i];
        }
        STAMP(PF_EVAL_JOINT);
        LINK_FLAGS_FRESH(c);
        double pd[6];
        from_prev<6>(S.d, pd);
        // (the next column's operands are requested ahead on the three-lane shape, which has the registers: cartpole 160.0 -> 163.0 M; measured 0 / - 0.3 % on the
        // one-lane 8- and 16-lane kernels -- tracking cfg5, Sawyer -- which therefore keep the plain form)
        ck_schur_rows_sub<KL, (KL == 3)>(c, Q, t, active, Y, L, gs, sXT, sPB, sPA, S.d, pd);
        STAMP(PF_SCHUR_S);
    } else {
        double wXT[3][3], wPB[5][3], wPA[5][3];
        if (active) {
            joint_eval_sparse<JAC>(c, pxq, pxq + 3, xq, xq + 3, pNB, NB, g, wXT, wPB, wPA);
            if (KL == 1) {
#pragma unroll
                for (int i = 0; i < 5; i++) part += g[i] * g[i];
            } else {                                     // (residual-only evaluation with several lanes per link: every lane computes all rows, one counts)
                double pj = 0.0;
#pragma unroll
                for (int i = 0; i < 5; i++) pj += g[i] * g[i];
                part += c.prim() ? pj : 0.0;
            }
        }
        STAMP(PF_EVAL_JOINT);
        LINK_FLAGS_FRESH(c);
        if (JAC) {
            double pd[6];
            from_prev<6>(S.d, pd);
            ck_schur_rows(c, t, active, Y, L, wXT, wPB, wPA, g, S.d, pd);
            STAMP(PF_SCHUR_S);
        }
    }
    const double nrm = sqrt(group_sum<G>(part));
    STAMP(PF_EVAL_MAP);
    PCOUNT(PF_EVALS);
    return nrm;
}

// ---- line search of the 32-lane instantiations (two instances per wavefront): TWO step lengths per pass in a group's own lanes, and when
// only ONE of the wavefront's two instances is still searching the other group's idle lanes evaluate two more for it.  The noise-floor
// searches of the exact stopping rule are heavy-tailed (9 % of them run to the 10th halving), and a wavefront pays the longer of its two:
// the accept sequence -- first level that does not grow, level LINE_MAXIT at the latest -- and every bit of the result are unchanged.
struct TrialIn { double z[7], s[6], ds[6], cd[6]; };
__device__ __forceinline__ double other_half(double v) { return __shfl_xor(v, 32, 64); }
// ||f|| of the owning group at the trial points s - a ds (constraint forces C - a cd) for a = a1 and a = a2; Lc = LDS image of the instance
template <int G, int KL = 1>
__device__ __forceinline__ void chain_eval2(LinkC& c, const TrialIn& T, const double* Lc, int t, const Lay& Y, double a1, double a2, bool active, double dt,
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
    from_prev<7>(xq1, p1);
    from_prev<7>(xq2, p2);
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
    if (KL > 1 && !c.prim()) { part1 = 0.0; part2 = 0.0; }      // (several lanes per link: every lane of a link has evaluated the same residual, one counts)
    n1 = sqrt(group_sum<G>(part1));
    n2 = sqrt(group_sum<G>(part2));
}

// ---- line search of the 8- and 16-lane instantiations (eight / four instances per wavefront): a wavefront pays the LONGEST of its instances'
// searches, and with eight heavy-tailed searches per wavefront somebody nearly always runs deep (53 % of the noise-floor iterations of a
// wavefront of eight see a search go to the 10th halving) while the groups whose search has ended sit idle.  So the idle groups evaluate
// further step lengths of the instances still searching (group_assist below): ||f|| at ONE step length, for any group's lanes, of the
// instance whose trial state is T and whose LDS image is Lc.  Same functions, same order as chain_eval<G, false>.
template <int G, int KL = 1>
__device__ __forceinline__ double chain_eval1(LinkC& c, const TrialIn& T, const double* Lc, int t, const Lay& Y, double a1, bool active, double dt) {
    double part1 = 0.0, xq1[7];
    LINK_FLAGS_FRESH(c);
#pragma unroll
    for (int k = 0; k < 7; k++) xq1[k] = T.z[k];
    if (active) {
        double cf1[6], sv1[6], cTR[6], d1[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            cTR[k] = Lc[Y.D + 6 * t + k];
            cf1[k] = Lc[Y.C + 6 * t + k] - a1 * T.cd[k];
            sv1[k] = T.s[k] - a1 * T.ds[k];
        }
        part1 = ck_body_eval<false>(c, T.z, sv1, cf1, cTR, cTR + 3, dt, xq1, d1, nullptr, nullptr);
    }
    double p1[7];
    from_prev<7>(xq1, p1);
    if (!c.has_a()) {
#pragma unroll
        for (int i = 0; i < 7; i++) p1[i] = (i == 3) ? 1.0 : 0.0;
    }
    if (active) {
        double g1[5];
        joint_eval_sparse<false>(c, p1, p1 + 3, xq1, xq1 + 3, nullptr, nullptr, g1, (double(*)[3]) nullptr, (double(*)[3]) nullptr, (double(*)[3]) nullptr);
#pragma unroll
        for (int i = 0; i < 5; i++) part1 += g1[i] * g1[i];
    }
    if (KL > 1 && !c.prim()) part1 = 0.0;
    return sqrt(group_sum<G>(part1));
}
// value of lane `addr / 4` of the wavefront (ds_bpermute: the LDS crossbar, no memory access)
__device__ __forceinline__ double lane_fetch(double v, int addr) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}

// Counter-based noise (noise_philox): the samples of a launch are generated by this kernel into a workspace and the rollout
// reads them like an injected array -- sqrt/log/cos inside the persistent kernel cost it ~20 scalar registers of polynomial
// constants for its whole lifetime.  sample (instance n, step k) = Box-Muller of Philox-4x32-10 keyed by the GLOBAL instance index.
// Short launches (the step-per-launch / hipGraph form of BASELINE configs[4], cclqr.h CCLQR_PHILOX_INKERNEL_STEPS) generate the sample inside the
// rollout kernel instead (EXTRA = 3): no fill launch in front of every step, no workspace.  ONE compiled body serves both (not inlined), so
// that a sample has the same bits wherever it is generated.
__device__ __attribute__((noinline)) double philox_normal_dev(unsigned key0, unsigned long long instance, int k) { return philox_normal(key0, instance, k); }
__global__ __launch_bounds__(256) void philox_fill_kernel(double* out, unsigned key0, long long inst0, long long n_inst, int k0, int steps) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_inst * steps) return;
    const long long n = i / steps;
    const int kk = (int)(i - n * steps);
    out[i] = philox_normal_dev(key0, (unsigned long long)(inst0 + n), k0 + kk);
}
hipError_t launch_philox_fill(double* out, unsigned key0, long long inst0, long long n_inst, int k0, int steps, hipStream_t stream) {
    const long long total = n_inst * steps;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, out, key0, inst0, n_inst, k0, steps);
    return hipGetLastError();
}

// NBP: links the LDS image is laid out for (>= nb): a compile-time layout turns every LDS offset into an immediate instead
// of a live scalar register and index arithmetic.
// EXTRA: 0 = plain LQR / TrackingLQR feedback; 1 = + joint friction and noise (examples/trackingLQR_triple_cartpole.jl:93-111);
// 2 = + the PID law of src/control/pid.jl; 3 = 1 with the noise sample of (instance, step) generated HERE (Philox, philox_normal_dev) instead of read
// from an array: the launches of a few steps that a hipGraph replays.  The plain instantiations carry none of that code (nor its registers).
// RELAX: the measured-error Newton mode of cclqr_rollout_opts.newton_mode = 1 -- a solve also stops on ||f|| < eps_alone, whatever the step.  A template
// parameter (plain law only), not a launch argument: the exact-rule kernels sit at 486-502 of 512 registers and any extra live value
// moves them across the line (a scalar mode flag read in the accept phase cost the 17-link instantiations 2 VGPR spills).
// KL, NL: several lanes per link (cclqr_chain.h "SEVERAL LANES PER LINK"): lane t of a group is sub-lane w = t / NL of link tl = t % NL; a mechanism of at most NL
// links whose lane group has KL NL <= G lanes.  KL = 1 (NL = G): one lane per link, the kernels of rounds 2-4, bit for bit.
template <int G, int NBP, int EXTRA, bool RELAX = false, int KL = 1, int NL = G>
__global__ __launch_bounds__(64) void rollout_chain_kernel(RolloutArgs a) {
    extern __shared__ double lds[];
    static_assert(KL >= 1 && KL <= 3 && KL * NL <= G && (KL > 1 || NL == G) && (KL == 1 || NL <= NBP), "lane group too small for KL lanes per link");
#if defined(CHAIN_DIAG_EXIT) && CHAIN_DIAG_EXIT == 1
    if (a.steps == 0) return;                                // (diagnostic build, tools/gpu_launch_overhead.py: what a launch costs with NO kernel body)
#endif
    const int lane = threadIdx.x, t = lane % G, grp = lane / G;
    const int w = KL > 1 ? t / NL : 0;                      // sub-lane of the lane's link
    const int tl = KL > 1 ? t - NL * w : t;                 // the link the lane works for (LDS slots, tables)
    const int tc = KL > 1 ? (w < KL ? tl : -2) : t;         // ... for comparisons with a link number (no match on a lane without a link)
    const int64_t inst = (int64_t)blockIdx.x * a.ipw + grp;
    const MechDev* M = a.M;
    const CtrlDev* C = a.C;
    const int nb = M->nb;
    const double dt = M->dt;
    const Lay Y = make_chain_layout(NBP);
    double* L = lds + grp * Y.total;
    const int nz = 13 * nb;

    LinkC c;
    link_load_consts(c, M, (KL > 1 && w >= KL) ? CCLQR_MAXL : tl, nb, dt);
    if (KL == 1 || w == 0) c.flags |= LinkC::PRIM;
    SubSel Q;
    if (KL > 1) sub_setup<KL>(c, w < KL ? w : 0, Q);
    if (EXTRA && C->has_fric && c.on()) { c.fric = C->fric[tl]; if (c.fric != 0.0) c.flags |= LinkC::FRIC; }
    c.set_valid(grp < a.ipw && inst < a.n_inst);
#if defined(CHAIN_DIAG_EXIT) && CHAIN_DIAG_EXIT == 3
    if (a.steps == 0) { if (t == 0 && c.valid()) a.status[inst] = (int)(c.m + c.J[8] + c.sxa + c.qoc[3] + c.V12[5] + c.p1[2] + c.p2[2] + c.axis[2]); return; }      // (diagnostic: link constants only)
#endif
    const long long ginst = a.inst0 + inst;     // global instance index: selects the controller table when there is one per instance
    const int ut = c.on() ? M->perm[tl] : 0;      // user body index of the owned link

    LinkS S;
    double pid_int = 0.0, pid_last = 0.0;
#pragma unroll
    for (int i = 0; i < 7; i++) S.z[i] = c.live() ? a.z0[inst * nz + ut * 13 + i] : ((i == 3) ? 1.0 : 0.0);
#pragma unroll
    for (int i = 0; i < 6; i++) S.s[i] = c.live() ? a.z0[inst * nz + ut * 13 + 7 + i] : 0.0;
    if (EXTRA == 2 && a.pid_state && a.k0 > 1 && c.live()) { pid_int = a.pid_state[(inst * nb + tl) * 2]; pid_last = a.pid_state[(inst * nb + tl) * 2 + 1]; }
#pragma unroll
    for (int i = 0; i < 6; i++) { S.cd[i] = 0.0; S.d[i] = 0.0; S.ds[i] = 0.0; }
#if defined(CHAIN_DIAG_EXIT) && CHAIN_DIAG_EXIT == 4
    if (a.steps == 0) { if (t == 0 && c.valid()) a.status[inst] = (int)(S.z[0] + S.s[5] + c.m); return; }      // (diagnostic: constants + state)
#endif
#ifdef CHAIN_LDS_ZERO_FILL
    for (int e = t; e < Y.total; e += G) L[e] = 0.0;
    __syncthreads();
#endif
    // What the Newton phases read before they have written it is the multiplier block only (tests/emu/emu_chain.cpp runs every phase function on an
    // LDS image poisoned with signalling NaNs but for LAM): zero, or the caller's warm start.  Everything else in the image is discarded by a
    // select wherever a phase reads past what was produced (ck_tri_back / cr_back), so it is not cleared: at one step per launch (configs[4]'s
    // graph mode) clearing 601 doubles per instance was 2.3 of the launch's 6.5 us (profiles/r05/launch_overhead.txt).
    if (c.on() && (KL == 1 || c.prim())) {
        const bool warm = c.live() && a.lam && a.k0 > 1;
#pragma unroll
        for (int i = 0; i < 5; i++) L[Y.LAM + 5 * tl + i] = warm ? a.lam[inst * 5 * nb + 5 * tl + i] : 0.0;
    }
    __syncthreads();

#if defined(CHAIN_DIAG_EXIT) && CHAIN_DIAG_EXIT == 2
    if (a.steps == 0) { if (t == 0 && c.valid()) a.status[inst] = (int)(S.z[0] + L[Y.LAM + 5 * tl]); return; }      // (diagnostic: prologue only)
#endif
#ifdef CCLQR_PROFILE
    Prof prof;
    prof.start();
#endif
    int worst = 0;
    // "bad" (a step did not converge) and "dead" (a step produced a non-finite residual; the instance is frozen from then on) are
    // bits of the lane flags, not 64-bit lane masks held in scalar registers through the launch
    if (a.carry && a.status && c.valid()) {      // CCLQR_ROLLOUT_CARRY_STATUS: the launches before this one count (step-per-launch chains)
        const int carried = a.status[inst];
        worst = carried < 0 ? -carried : carried;
        if (carried < 0) c.flags |= LinkC::BAD;
        if (carried < 0 && carried > -NEWTON_MAXIT) c.flags |= LinkC::DEAD;      // lost in an earlier launch (stopped before NEWTON_MAXIT): it stays frozen
    }
    // Launch arguments that are only needed once per step or at the end are read from the kernel-argument segment where they are
    // used, through a pointer the optimiser cannot see through, instead of sitting in scalar registers for the whole launch.
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
            if (c.live() && (KL == 1 || c.prim())) {
#pragma unroll
                for (int i = 0; i < 7; i++) L[Y.Z + 13 * ut + i] = S.z[i];
#pragma unroll
                for (int i = 0; i < 6; i++) L[Y.Z + 13 * ut + 7 + i] = S.s[i];
            }
            __syncthreads();
            if (c.valid()) {
                int kr = kk, nzr = nz;
                asm volatile("" : "+s"(kr), "+s"(nzr));   // keeps the row address a product computed here (not a running pointer + a 64-bit stride kept in registers)
                double* dst = traj_out + ((size_t)inst * ap->steps + kr) * nzr;
                int e0 = t;
                asm volatile("" : "+v"(e0));      // the loop's entry test is made here, not once per launch and kept as a lane mask
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
        from_prev<7>(zf, za);                   // parent pose (every joint evaluation needs it)
        if (EXTRA) from_prev<6>(zf + 7, za + 7);   // parent velocities (friction, PID)
        if (!c.has_a()) {
#pragma unroll
            for (int i = 0; i < 13; i++) za[i] = (i == 3) ? 1.0 : 0.0;
        }
        if (gate) {
            if (c.live()) {
                double dz[12];
                ck_control_error(zf, C->zd + ginst * C->zd_stride + (size_t)ksp * nz + 13 * tl, dz);
                if (KL == 1 || c.prim()) {
#pragma unroll
                    for (int i = 0; i < 12; i++) L[Y.DZ + 12 * tl + i] = dz[i];
                }
                if (EXTRA && C->has_fric && c.has_fric()) uj = ck_friction(c, zf, za);
            }
            __syncthreads();
            {
                // u_i = Fd_i - K_i . dz for the mu inputs.  The gain entries of a lane -- NE per input, read from HBM / L2 -- are ALL requested
                // before the first one is used, CH inputs at a time, together with the inputs' feed-forward values and joint numbers: as a loop
                // "load, multiply-add, next entry" every entry paid its own memory round trip (6 round trips per input and lane; 25 % of a
                // step of the seven-input Sawyer arm, 2.4 % of the headline's).  Same products, same order of summation.
                constexpr int NE = (12 * NBP + G - 1) / G;
                constexpr int CH = (G == 16) ? 8 : 1;       // (16 lanes = 5 .. 8 links: the multi-input arms, all their inputs at once; the others usually have one input)
                // No predicate lives across the loads (each would be a 64-bit lane mask in scalar registers): an entry past the end of a row is
                // fetched all the same -- it is the next row's, or the zero padding behind the table (CCLQR_K_PAD) -- and meets a zero in dz;
                // a lane of an instance that does not exist reads the first instance's tables, and its result is never used.
                const long long gi = c.valid() ? ginst : a.inst0;
                const int ne = 12 * nb;
                double unoise = 0.0;                    // noise: injected by the caller, or generated for this launch by philox_fill_kernel
                if (EXTRA == 3) {
                    if (C->noise_scale != 0.0) unoise = C->noise_scale * philox_normal_dev(C->noise_key0, (unsigned long long)gi, k);
                } else if (EXTRA) {
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
                    const double* Kp = C->K + gi * C->K_stride + (size_t)kidx * mu * ne + t;      // the lane's first entry of the step's first row
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
                            if (tc == cjv[j]) uj += u;
                        }
                    }
                } else {                                     // feed-forward only (OpenLoop, a host closure's inputs)
                    for (int i = 0; i < mu; i++) {
                        double u = Fp ? Fp[i] : 0.0;
                        if (EXTRA) u += unoise;
                        if (tc == C->cj[i]) uj += u;
                    }
                }
            }
            __syncthreads();
        }
        if (EXTRA == 2 && C->has_pid) {
            if (c.live() && C->pid_on[tl]) uj += ck_pid(c, zf, za, C->pid_P[tl], C->pid_I[tl], C->pid_D[tl], C->pid_goal[tl], dt, k == 1, pid_int, pid_last);
        }
        STAMP(PF_CONTROL);
        LINK_FLAGS_FRESH(c);
        // ---------------- joint inputs -> wrenches, per-step invariants, constraint Jacobians at the current knot, force map
        {
            double F[3], tau[3], W6[6], cW6[6];
            ck_joint_wrench(c, uj, zf + 3, za + 3, F, tau, W6, W6 + 3);
            from_next<6>(W6, cW6);
            if (c.has_c()) {
#pragma unroll
                for (int i = 0; i < 3; i++) { F[i] += cW6[i]; tau[i] += cW6[3 + i]; }
            }
            double cTR[6];
            ck_step_invariants(c, zf, F, tau, dt, M->g, cTR, cTR + 3);
            double gk[5], kXT[3][3], kPB[5][3], kPA[5][3], lam[5];
            joint_eval_sparse<true>(c, za, za + 3, zf, zf + 3, nullptr, nullptr, gk, kXT, kPB, kPA);
#pragma unroll
            for (int i = 0; i < 5; i++) lam[i] = L[Y.LAM + 5 * tl + i];
            if (c.live() && (KL == 1 || c.prim())) {
                gk_store(tl, Y, L, kXT, kPB, kPA);
#pragma unroll
                for (int i = 0; i < 6; i++) L[Y.D + 6 * tl + i] = cTR[i];
            }
            double own[6], par[6], cpar[6];
            jac_t_apply(c, kXT, kPB, kPA, lam, own, par);
            from_next<6>(par, cpar);
#pragma unroll
            for (int i = 0; i < 6; i++) S.cd[i] = 0.0;
            if (c.live() && (KL == 1 || c.prim())) {
#pragma unroll
                for (int i = 0; i < 6; i++) L[Y.C + 6 * tl + i] = own[i] + (c.has_c() ? cpar[i] : 0.0);
            }
        }
        __syncthreads();

        STAMP(PF_FORCES);
        PCOUNT(PF_STEPS);
        // ---------------- newton! (tolerances and line search: SURVEY 8a-bis)
        const bool go = c.valid() && !c.dead();
        bool done = !go, failed = false;
        int its = 0;
        double normf0 = chain_eval<G, true, KL>(c, S, tl, Y, L, 0.0, c.live() && !done, dt, Q PROF_PASS);
        __syncthreads();
        const int nchains = M->nchains;
        for (int iter = 1; iter <= NEWTON_MAXIT; iter++) {
            if (!__any(!done)) break;
            PCOUNT(PF_NEWTON_ITERS);
            const bool active = c.live() && !done;
            // block-tridiagonal solve along each chain, swept from both ends (cclqr_chain.h)
            for (int ci = 0; ci < nchains; ci++) {
                const int cs = M->chain_start[ci], cn = M->chain_len[ci];
                // long chains: every second link is eliminated first, all at once (cr_level, cclqr_chain.h), and the two-front sweep
                // runs over the half that is left
                constexpr int CRW = 4;
                constexpr bool CR = G == 32 && NBP <= 17;    // 4 lanes for each of up to 8 odd links
                const bool cr = CR && cn >= CR_MIN_LINKS;
                if (CR && cr) {
                    CrLane<CRW> CK;
                    double tc[CrLane<CRW>::NS][5];
                    cr_setup<CRW>(CK, t, cs, cn, 1, Y, done);
                    CR_FLAGS_FRESH(CK);
                    cr_phase_a<CRW>(CK, L, tc);
                    CR_FLAGS_FRESH(CK);
                    cr_store_a<CRW>(CK, L, tc);
                    __syncthreads();
                    CR_FLAGS_FRESH(CK);
                    cr_phase_b<CRW>(CK, L, tc);
                    CR_FLAGS_FRESH(CK);
                    cr_store_b<CRW>(CK, L, tc);
                    __syncthreads();
                }
                STAMP(PF_SCHUR_W);
                const TriPlanB PB = cr ? tri_plan_balanced(cs, (cn + 1) / 2, 2) : tri_plan_balanced(cs, cn, 1, G >= 16 ? 2 : 1);
                const TriPlan& P = PB.P;
                TriCur K = tri_cursor(t, PB, Y);
                if (done) K.n = 0;
                for (int i = 0; i < P.steps; i++) {
                    double tg[5], zy[5];
                    int otg = 0, oout = 0;
                    if (tri_step(K, i, L, tg, zy, &otg, &oout)) tri_step_store(L, otg, oout, tg, zy);
                    __syncthreads();
                }
                STAMP(PF_TRI_FWD);
                if (!done) ck_tri_mid(t, PB, Y, L);
                __syncthreads();
                for (int j = 0; j < P.steps; j++) {
                    if (!done) ck_tri_back(t, j, PB, Y, L);
                    __syncthreads();
                }
                if (CR && cr) {
                    cr_back<CRW>(t, cs, cn, 1, Y, L, done);
                    __syncthreads();
                }
                STAMP(PF_TRI_BWD);
            }
            double nd;
            LINK_FLAGS_FRESH(c);
#ifdef CHAIN_RELOAD_CONSTS
            {   // experiment: the link constants (33 doubles per lane) re-read behind the linear solve through a pointer the optimiser cannot see
                // through, so that they are dead -- 66 registers free -- while the solve runs
                const MechDev* Mq = ap->M;
                asm volatile("" : "+s"(Mq));
                link_reload_consts(c, Mq, tl, nb, dt);
            }
#endif
            {   // multiplier step from LDS, body solve
                double own[6], par[6], cpar[6], dl[5], pdn = 0.0;
#pragma unroll
                for (int r = 0; r < 5; r++) dl[r] = L[Y.DL + 5 * tl + r];
                gk_t_apply(c, tl, Y, L, dl, own, par);
                from_next<6>(par, cpar);
                if (active) {
                    double DINV[9];
#pragma unroll
                    for (int i = 0; i < 9; i++) DINV[i] = L[Y.DINV + 9 * tl + i];
#pragma unroll
                    for (int i = 0; i < 6; i++) S.cd[i] = own[i] + (c.has_c() ? cpar[i] : 0.0);
                    ck_body_solve(c, S.d, S.cd, DINV, S.ds);
#pragma unroll
                    for (int i = 0; i < 6; i++) pdn += S.ds[i] * S.ds[i];
#pragma unroll
                    for (int i = 0; i < 5; i++) pdn += dl[i] * dl[i];
                    if (KL > 1 && !c.prim()) pdn = 0.0;
                }
                nd = sqrt(group_sum<G>(pdn));
            }
            __syncthreads();
            STAMP(PF_BODY_SOLVE);
            // line search: halve while ||f|| grows.  The first (full-step) trial also evaluates the Jacobians and the Schur blocks,
            // speculating that it is accepted; later trials evaluate the residual only.  (Speculating the other way round once
            // ||f|| < eps -- residual only, Jacobians afterwards if the instance goes on -- was measured: no gain.)
            double alpha = 1.0, normf1 = 0.0;
            bool ls_done = done, jac_ok = true;
            {
                const double nf = chain_eval<G, true, KL>(c, S, tl, Y, L, 1.0, active, dt, Q PROF_PASS);
                if (!ls_done) {
                    normf1 = nf;
                    if (!(normf1 > normf0)) ls_done = true;
                }
            }
            if (G == 32) {
                for (int lv = 1; lv <= LINE_MAXIT;) {
                    if (!__any(!ls_done)) break;
                    const bool mine = !ls_done;                                   // uniform over the group
                    const bool other = __shfl_xor(mine ? 1 : 0, 32, 64) != 0;     // the wavefront's other instance is searching too
                    const bool helping = !mine && other;                          // this group's lanes evaluate two more levels of the other's search
                    TrialIn T;
#pragma unroll
                    for (int i = 0; i < 7; i++) T.z[i] = S.z[i];
#pragma unroll
                    for (int i = 0; i < 6; i++) { T.s[i] = S.s[i]; T.ds[i] = S.ds[i]; T.cd[i] = S.cd[i]; }
                    if (mine != other) {       // (symmetric in the two groups: a wavefront-uniform branch) one searches, one helps: hand the trial over
#pragma unroll
                        for (int i = 0; i < 7; i++) { const double o = other_half(S.z[i]); T.z[i] = helping ? o : T.z[i]; }
#pragma unroll
                        for (int i = 0; i < 6; i++) {
                            const double o1 = other_half(S.s[i]), o2 = other_half(S.ds[i]), o3 = other_half(S.cd[i]);
                            T.s[i] = helping ? o1 : T.s[i]; T.ds[i] = helping ? o2 : T.ds[i]; T.cd[i] = helping ? o3 : T.cd[i];
                        }
                    }
                    const double* Lc = helping ? lds + (1 - grp) * Y.total : L;
                    const int l0 = helping ? lv + 2 : lv;
                    double n1, n2;
                    chain_eval2<G, KL>(c, T, Lc, tl, Y, ldexp(1.0, -l0), ldexp(1.0, -(l0 + 1)), c.on() && (mine || helping) && l0 <= LINE_MAXIT, dt, n1, n2);
                    PCOUNT(PF_EVALS);
                    const double h1 = other_half(n1), h2 = other_half(n2);
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
                // group assist: the NG = 64 / G groups of the wavefront are dealt to the `count` instances that are still searching -- group g
                // works for the (g mod count)-th of them at level lv + g / count -- so that a pass covers NG / count levels of every search
                // (all of them searching: one level each, as before; one straggler: NG levels at once).  Same accept sequence: the first
                // level that does not grow, level LINE_MAXIT at the latest.
                constexpr int NG = 64 / G;
                for (int lv = 1; lv <= LINE_MAXIT;) {
                    if (!__any(!ls_done)) break;
                    const bool mine = !ls_done;                                                   // uniform over the group
                    const unsigned long long heads = __ballot(mine && t == 0);                    // bit g G: group g is searching
                    const int count = __builtin_popcountll(heads);                                // >= 1 behind the vote above
                    const int per = NG / count;                                                   // levels of every search this pass (>= 1)
                    const int kq = grp % count, off = grp / count;                                // this group works for the kq-th searcher, level lv + off
                    unsigned long long hm = heads;
                    for (int i = 0; i < kq; i++) hm &= hm - 1;                                    // (kq < count: the kq-th set bit exists)
                    const int src = __builtin_ctzll(hm) / G;                                      // that searcher's group
                    int myrank = 0;                                                               // rank of this group among the searchers
                    for (int g2 = 0; g2 < NG; g2++) myrank += (g2 < grp && ((heads >> (g2 * G)) & 1ull)) ? 1 : 0;
                    const int l0 = lv + off;
                    const bool work = off < per && l0 <= LINE_MAXIT;
                    const int fa = (src * G + t) * 4;
                    TrialIn T;
#pragma unroll
                    for (int i = 0; i < 7; i++) T.z[i] = S.z[i];
#pragma unroll
                    for (int i = 0; i < 6; i++) { T.s[i] = S.s[i]; T.ds[i] = S.ds[i]; T.cd[i] = S.cd[i]; }
                    if (count < NG) {                                                             // (uniform) somebody has lanes to spare: hand the trials over
#pragma unroll
                        for (int i = 0; i < 7; i++) T.z[i] = lane_fetch(S.z[i], fa);
#pragma unroll
                        for (int i = 0; i < 6; i++) { T.s[i] = lane_fetch(S.s[i], fa); T.ds[i] = lane_fetch(S.ds[i], fa); T.cd[i] = lane_fetch(S.cd[i], fa); }
                    }
                    const double nf = chain_eval1<G, KL>(c, T, lds + src * Y.total, tl, Y, ldexp(1.0, -l0), c.on() && work, dt);
                    PCOUNT(PF_EVALS);
                    if (count == NG) {                                                            // (uniform) everybody searches: one level each, the group's own result
                        if (mine) {
                            normf1 = nf; alpha = ldexp(1.0, -lv); jac_ok = false;
                            if (!(nf > normf0) || lv == LINE_MAXIT) ls_done = true;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NG; i++) {                                            // the searcher collects its levels in order
                            const double ci = lane_fetch(nf, ((myrank + i * count) % NG) * G * 4);
                            const int l = lv + i;
                            if (mine && !ls_done && i < per && l <= LINE_MAXIT) {
                                normf1 = ci; alpha = ldexp(1.0, -l); jac_ok = false;
                                if (!(ci > normf0) || l == LINE_MAXIT) ls_done = true;
                            }
                        }
                    }
                    lv += per;
                }
            }
            bool need_jac = false;
            if (!done) {
                if (c.live()) {
                    if (KL == 1) {
#pragma unroll
                        for (int i = 0; i < 6; i++) { S.s[i] -= alpha * S.ds[i]; L[Y.C + 6 * t + i] -= alpha * S.cd[i]; S.cd[i] = 0.0; S.ds[i] = 0.0; }
#pragma unroll
                        for (int i = 0; i < 5; i++) L[Y.LAM + 5 * t + i] -= alpha * L[Y.DL + 5 * t + i];
                    } else {      // every lane of a link moves its own copy of the iterate; the link's LDS slots are the primary lane's to update
#pragma unroll
                        for (int i = 0; i < 6; i++) { S.s[i] -= alpha * S.ds[i]; if (c.prim()) L[Y.C + 6 * tl + i] -= alpha * S.cd[i]; S.cd[i] = 0.0; S.ds[i] = 0.0; }
                        if (c.prim()) {
#pragma unroll
                            for (int i = 0; i < 5; i++) L[Y.LAM + 5 * tl + i] -= alpha * L[Y.DL + 5 * tl + i];
                        }
                    }
                }
                its = iter;
                if (normf1 < NEWTON_EPS && alpha * nd < NEWTON_EPS) done = true;
                if (RELAX && normf1 < ap->eps_alone) done = true;      // measured-error mode: the residual alone
                if (!(normf1 < 1e300)) { done = true; failed = true; }   // non-finite residual: the instance has left the domain of the integrator
                normf0 = normf1;
                need_jac = !done && !jac_ok;
            }
            STAMP(PF_ACCEPT);
            if (__any(need_jac)) chain_eval<G, true, KL>(c, S, tl, Y, L, 0.0, c.live() && need_jac, dt, Q PROF_PASS);
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
        nsteps = ap->steps;      // read again rather than kept in a scalar register through the step
    }
#ifdef CCLQR_PROFILE
    prof.stamp(PF_IO);
    prof.flush();
#endif
    // ---------------- final state, multipliers, status
#ifdef CHAIN_DIRECT_FINAL_STORE      // (rejected variant, kept buildable for A/B: 13 one-double stores per lane straight from registers; 0-step launch 4.75 -> 5.23 us)
    asm volatile("" : "+s"(ap));
    LINK_FLAGS_FRESH(c);
    if (c.live() && (KL == 1 || c.prim())) {
        double* zT = ap->zT;
#pragma unroll
        for (int i = 0; i < 7; i++) zT[inst * nz + 13 * ut + i] = S.z[i];
#pragma unroll
        for (int i = 0; i < 6; i++) zT[inst * nz + 13 * ut + 7 + i] = S.s[i];
    }
    if (c.valid()) {
        int* status = ap->status;
        // (carried statuses mark a lost instance by a count below NEWTON_MAXIT: one that had also failed to converge earlier reports NEWTON_MAXIT - 1)
        if (status && t == 0) status[inst] = c.bad() ? -((ap->carry && c.dead() && worst >= NEWTON_MAXIT) ? NEWTON_MAXIT - 1 : worst) : worst;
    }
#else
    __syncthreads();
    if (c.live() && (KL == 1 || c.prim())) {
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
        // (carried statuses mark a lost instance by a count below NEWTON_MAXIT: one that had also failed to converge earlier reports NEWTON_MAXIT - 1)
        if (status && t == 0) status[inst] = c.bad() ? -((ap->carry && c.dead() && worst >= NEWTON_MAXIT) ? NEWTON_MAXIT - 1 : worst) : worst;
    }
#endif
    if (c.live() && (KL == 1 || c.prim())) {
        const int nbT = ap->M->nb;      // read again here rather than kept in a scalar register through the launch
        double* lam = ap->lam;
        if (lam) {
#pragma unroll
            for (int i = 0; i < 5; i++) lam[inst * 5 * nbT + 5 * tl + i] = L[Y.LAM + 5 * tl + i];
        }
        if (EXTRA == 2) {
            double* pid_state = ap->pid_state;
            if (pid_state) { pid_state[(inst * nbT + tl) * 2] = pid_int; pid_state[(inst * nbT + tl) * 2 + 1] = pid_last; }
        }
    }
}

#ifdef CCLQR_PROFILE
extern "C" int cclqr_prof_read_chain(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * PF_N);
    if (e == hipSuccess && reset) { unsigned long long z[PF_N] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return e == hipSuccess ? PF_N : -1;
}
#endif

// 8 lanes per instance up to 4 links (one elimination front of 6 lanes; eight instances per wavefront: the cartpole and triple-cartpole
// configs), 16 up to 8 links (the two fronts need 14), 32 up to 32 links: with 16 lanes a 9..16-link instance would fill LDS with two
// wavefronts per CU; 33..64 links: the whole wavefront is one instance (76.8 KB of LDS: two workgroups per CU)
int chain_lanes_per_instance(int nb) { return nb <= 4 ? 8 : (nb <= 8 ? 16 : (nb <= 32 ? 32 : 64)); }
// links the LDS image is laid out for: the instantiations below (17 = the headline mechanism: exactly four workgroups per CU)
int chain_layout_links(int nb) { return nb <= 4 ? 4 : (nb <= 8 ? 8 : (nb <= 16 ? 16 : (nb == 17 ? 17 : (nb <= 32 ? 32 : 64)))); }

// Instances per wavefront of a launch.  A wavefront's step is latency -- a chain of dependent 5 x 5 stages -- not lanes, and a lane group without an
// instance is not idle: it evaluates further step lengths of its neighbours' line searches (group_assist / the partner group).  So a batch that would
// leave SIMDs without a wavefront when packed 64 / G to a wavefront is spread: the fewest instances per wavefront that still fit the batch into
// `slots` wavefronts -- slots = every SIMD of the device for a persistent launch (steps >= 8: it has the device to itself), a quarter of them for
// short launches (step-per-launch chains run several to a device, bench.py::_graph_captured_steps: spreading each over the whole device would
// queue them behind one another).  Measured (tools/gpu_batch_density.py, ms per 1000 steps, packed -> spread): 256 tracking triple cartpoles
// 43.7 -> 34.7, 256 cartpoles 24.6 -> 21.5, 256 17-body chains (300 steps) 22.0 -> 20.2, 4096 cartpoles (configs[1]) 24.9 -> 24.0; a batch that
// fills the device is packed as before.  Same arithmetic in the same order either way: results are bitwise those of the packed launch
// (tests/test_gpu_rollout.py::test_spread_and_packed_launches_agree_bitwise).  packed: CCLQR_ROLLOUT_PACK_WAVEFRONTS.  `full` = 64 / lanes per
// instance; the branching-tree kernel (rollout_treereg.hip) spreads by the same rule.
int spread_instances_per_wavefront(int full, int64_t n_inst, int steps, bool packed) {
    if (packed || full == 1) return full;
    static int simds = 0;
    if (simds == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        simds = 4 * cus;
    }
    const int64_t slots = steps >= 8 ? simds : simds / 4;
    const int64_t ipw = (n_inst + slots - 1) / slots;
    return ipw < 1 ? 1 : (ipw > full ? full : (int)ipw);
}
int chain_instances_per_wavefront(int nb, int64_t n_inst, int steps, bool packed) {
    return spread_instances_per_wavefront(64 / chain_lanes_per_instance(nb), n_inst, steps, packed);
}

size_t chain_lds_bytes(int nb) { return (size_t)(64 / chain_lanes_per_instance(nb)) * make_chain_layout(chain_layout_links(nb)).total * sizeof(double); }

template <int G, int NBP, int KL = 1, int NL = G>
static hipError_t launch_chain_one(const RolloutArgs& a, int extra, int newton_mode, unsigned grid, size_t lds, hipStream_t stream) {
    const bool relax = newton_mode != 0 && extra == 0;
    const void* f = relax ? (const void*)rollout_chain_kernel<G, NBP, 0, true, KL, NL>
                          : (extra == 0 ? (const void*)rollout_chain_kernel<G, NBP, 0, false, KL, NL> : (extra == 1 ? (const void*)rollout_chain_kernel<G, NBP, 1, false, KL, NL>
                          : (extra == 2 ? (const void*)rollout_chain_kernel<G, NBP, 2, false, KL, NL> : (const void*)rollout_chain_kernel<G, NBP, 3, false, KL, NL>)));
    hipError_t e = set_max_dynamic_lds_once(f, lds);
    if (e != hipSuccess) return e;
    if (relax) hipLaunchKernelGGL((rollout_chain_kernel<G, NBP, 0, true, KL, NL>), dim3(grid), dim3(64), lds, stream, a);
    else if (extra == 0) hipLaunchKernelGGL((rollout_chain_kernel<G, NBP, 0, false, KL, NL>), dim3(grid), dim3(64), lds, stream, a);
    else if (extra == 1) hipLaunchKernelGGL((rollout_chain_kernel<G, NBP, 1, false, KL, NL>), dim3(grid), dim3(64), lds, stream, a);
    else if (extra == 2) hipLaunchKernelGGL((rollout_chain_kernel<G, NBP, 2, false, KL, NL>), dim3(grid), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((rollout_chain_kernel<G, NBP, 3, false, KL, NL>), dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

// lanes per link of the instantiation a mechanism of nb links runs on.  Only the 1- and 2-link mechanisms (pendulum, cartpole, acrobot: 2 of 8 lanes own a link)
// get several -- three.  TWO lanes per link were built for every group with lanes to spare (3-4 links in 8 lanes, 5-8 in 16, 9-16 in 32), measured and NOT
// shipped: the joint rows get cheaper (11.0 k -> 9.6 k cycles per step on the tracking triple cartpole) but the Schur rows do not (25.7 k -> 25.5 k) -- that phase is
// bound by its ~180 LDS instructions per evaluation, which every lane issues whatever rows it keeps, not by its multiply-adds -- and the whole step moves by
// < 0.5 % while the order of summation (hence noise-floor decisions of the stopping rule) changes (profiles/r05/lanes_per_link_*).
// -DCHAIN_ONE_LANE_PER_LINK: one lane per link everywhere, -DCHAIN_TWO_LANES_PER_LINK: the measured-and-rejected shapes, both for A/B timing
int chain_lanes_per_link(int nb) {
#if defined(CHAIN_ONE_LANE_PER_LINK)
    (void)nb;
    return 1;
#elif defined(CHAIN_TWO_LANES_PER_LINK)
    return nb <= 2 ? 3 : (nb <= 16 ? 2 : 1);
#else
    return nb <= 2 ? 3 : 1;
#endif
}

hipError_t launch_rollout_chain(const RolloutArgs& a_in, int nb, int extra, int newton_mode, hipStream_t stream) {
    const int per_wg = chain_instances_per_wavefront(nb, a_in.n_inst, a_in.steps, a_in.ipw != 0);
    const size_t lds = chain_lds_bytes(nb);
    RolloutArgs a = a_in;
    a.ipw = per_wg;
    const unsigned grid = (unsigned)((a.n_inst + per_wg - 1) / per_wg);
    if (grid == 0) return hipSuccess;
    {   // the reduction level's back substitution reads DL / R of one link past the chain and selects the value away (cclqr_chain.h cr_back): that
        // read must stay inside the instance's image whatever order a later re-cut of the layout puts the arrays in
        const int nbp = chain_layout_links(nb);
        const Lay Y = make_chain_layout(nbp);
        if (Y.DL + 5 * (nbp + 1) > Y.total || Y.R + 5 * (nbp + 1) > Y.total) return hipErrorInvalidValue;
    }
    const int kl = chain_lanes_per_link(nb);
    switch (chain_layout_links(nb)) {
#if defined(CHAIN_TWO_LANES_PER_LINK)
        case 4: return nb <= 2 ? launch_chain_one<8, 4, 3, 2>(a, extra, newton_mode, grid, lds, stream) : launch_chain_one<8, 4, 2, 4>(a, extra, newton_mode, grid, lds, stream);
        case 8: return launch_chain_one<16, 8, 2, 8>(a, extra, newton_mode, grid, lds, stream);
        case 16: return launch_chain_one<32, 16, 2, 16>(a, extra, newton_mode, grid, lds, stream);
#elif defined(CHAIN_ONE_LANE_PER_LINK)
        case 4: return launch_chain_one<8, 4>(a, extra, newton_mode, grid, lds, stream);
        case 8: return launch_chain_one<16, 8>(a, extra, newton_mode, grid, lds, stream);
        case 16: return launch_chain_one<32, 16>(a, extra, newton_mode, grid, lds, stream);
#else
        case 4: return nb <= 2 ? launch_chain_one<8, 4, 3, 2>(a, extra, newton_mode, grid, lds, stream) : launch_chain_one<8, 4>(a, extra, newton_mode, grid, lds, stream);
        case 8: return launch_chain_one<16, 8>(a, extra, newton_mode, grid, lds, stream);
        case 16: return launch_chain_one<32, 16>(a, extra, newton_mode, grid, lds, stream);
#endif
        case 17: return launch_chain_one<32, 17>(a, extra, newton_mode, grid, lds, stream);
        case 32: return launch_chain_one<32, 32>(a, extra, newton_mode, grid, lds, stream);
        default: (void)kl; return launch_chain_one<64, 64>(a, extra, newton_mode, grid, lds, stream);
    }
}

}  // namespace cclqr
