// emu_loop.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libcclqr.so).
// Runs the closed-loop rollout kernel's __host__ __device__ phase functions (csrc/cclqr_loop.h and the body phases of
// csrc/cclqr_dev.h it shares with the tree kernel) serially on the CPU, lane by lane, in the phase order of csrc/rollout_loop.hip,
// so that the dense singular solve, the incidence bookkeeping and the LDS layout can be checked against oracle/loops.py without a
// GPU.  The LDS image has exactly the kernel's size (an out-of-range offset is an out-of-bounds access for a sanitizer build).
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_tables.h"
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_loop.h"
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_lin_loop.h"
#include <math.h>
#include <string>
#include <vector>

using namespace cclqr;

namespace {
struct LoopInst {
    const MechDev* M;
    Lay Y;
    std::vector<double> lds;
    double* L;
    std::vector<LaneRegs> r;
};

template <bool JAC>
double loop_eval(LoopInst& I, int s_off, double alpha) {
    const MechDev* M = I.M;
    double acc = 0.0;
    for (int t = 0; t < 64; t++) acc += ph_body_eval<JAC>(t, M->nb, I.Y, I.L, I.r[t], M->dt, s_off, alpha);
    for (int t = 0; t < 64; t++) acc += lp_joint_eval<JAC>(t, I.Y, I.L, I.r[t], M);
    return sqrt(acc);
}

template <int NCB, int U>
void loop_solve_step(std::vector<LoopRowR<NCB>>& R, int kb, int mr, double tol) {
    const int col = 8 * kb + U;
    double own[64];
    unsigned key = 0u;
    for (int t = 0; t < 64; t++) {
        own[t] = lpr_entry<NCB, U>(R[t], kb);
        const unsigned kt = lpr_key(R[t], t, mr, own[t]);
        if (kt > key) key = kt;                                    // the wavefront's maximum key
    }
    const int prow = lp_key_row(key);
    if (!(fabs(own[prow]) > tol)) return;
    const LoopRowR<NCB> P = R[prow];                               // the lane reads of the kernel (the pivot row before this step)
    const double ip = 1.0 / own[prow];
    for (int t = 0; t < 64; t++) {
        const double f = lpr_step(R[t], t, col, prow, mr, own[t], ip);
        R[t].rhs -= f * P.rhs;
        for (int B = kb; B < NCB; B++)
            for (int u = 0; u < 8; u++) R[t].a[8 * B + u] -= f * P.a[8 * B + u];
    }
}
template <int NCB>
void loop_solve_t(LoopInst& I) {
    const MechDev* M = I.M;
    const Lay& Y = I.Y;
    double* L = I.L;
    const int mr = 5 * M->nj;
    for (int t = 0; t < mr; t++) L[Y.DL + t] = 0.0;
    std::vector<LoopRowR<NCB>> R(64);
    double amax = 0.0;
    for (int t = 0; t < 64; t++) amax = fmax(amax, lpr_assemble(R[t], t, Y, L, M));
    const double tol = LOOP_RANK_TOL * amax;
    for (int kb = 0; kb < NCB; kb++) {
        loop_solve_step<NCB, 0>(R, kb, mr, tol); loop_solve_step<NCB, 1>(R, kb, mr, tol);
        loop_solve_step<NCB, 2>(R, kb, mr, tol); loop_solve_step<NCB, 3>(R, kb, mr, tol);
        loop_solve_step<NCB, 4>(R, kb, mr, tol); loop_solve_step<NCB, 5>(R, kb, mr, tol);
        loop_solve_step<NCB, 6>(R, kb, mr, tol); loop_solve_step<NCB, 7>(R, kb, mr, tol);
    }
    for (int t = 0; t < 64; t++) lpr_solution(R[t], t, mr, Y, L);
}
void loop_solve(LoopInst& I) {
    switch (loop_col_blocks(I.M->nj)) {
        case 1: loop_solve_t<1>(I); break;
        case 2: loop_solve_t<2>(I); break;
        case 3: loop_solve_t<3>(I); break;
        case 4: loop_solve_t<4>(I); break;
        case 5: loop_solve_t<5>(I); break;
        case 6: loop_solve_t<6>(I); break;
        case 7: loop_solve_t<7>(I); break;
        default: loop_solve_t<8>(I); break;
    }
}
}  // namespace

extern "C" int emu_loop_rollout(const cclqr_mech_desc* md, const cclqr_ctrl_desc* cd, int64_t n_inst, int steps, int k0, const double* z0,
                                double* lam, double* traj, double* zT, int* status) {
    cclqr_mech m;
    std::string err;
    int rc = build_mech_tables(md, &m, err);
    if (rc) return rc;
    if (!m.host.loop) return CCLQR_EUNSUPPORTED;
    CtrlHostTables Tb;
    rc = build_ctrl_tables(&m, cd, Tb, err);
    if (rc) return rc;
    Tb.H.K = Tb.K.empty() ? nullptr : Tb.K.data();
    Tb.H.zd = Tb.zd.data();
    Tb.H.Fd = Tb.Fd.empty() ? nullptr : Tb.Fd.data();
    const MechDev* M = &m.host;
    const CtrlDev* C = &Tb.H;
    const int nb = M->nb, nj = M->nj, nz = 13 * nb;
    LoopInst I;
    I.M = M;
    I.Y = make_loop_layout(nb, nj);
    I.lds.assign(I.Y.total, 0.0);
    I.L = I.lds.data();
    I.r.resize(64);
    const Lay& Y = I.Y;
    double* L = I.L;
    for (int t = 0; t < 64; t++) loop_load_consts(I.r[t], M, t);
    for (int64_t inst = 0; inst < n_inst; inst++) {
        for (int e = 0; e < Y.total; e++) L[e] = 0.0;
        for (int e = 0; e < nz; e++) L[Y.Z + e] = z0[inst * nz + e];
        if (lam && k0 > 1) for (int e = 0; e < 5 * nj; e++) L[Y.LAM + e] = lam[inst * 5 * nj + e];
        for (int t = 0; t < 64; t++) { I.r[t].pid_int = 0.0; I.r[t].pid_last = 0.0; }
        int worst = 0;
        bool bad = false, dead = false;
        for (int kk = 0; kk < steps; kk++) {
            const int k = k0 + kk;
            if (traj) for (int e = 0; e < nz; e++) traj[((size_t)inst * steps + kk) * nz + e] = L[Y.Z + e];
            const bool gate = (C->N <= 0) || (k < C->N);
            const int ksp = (C->nsp > 1) ? ((k - 1 < C->nsp) ? k - 1 : C->nsp - 1) : 0;
            const int kidx = (C->N <= 0) ? 0 : ((k - 1 < C->nK) ? k - 1 : C->nK - 1);
            if (gate) for (int t = 0; t < 64; t++) ph_control_error(t, nb, Y, L, I.r[t], C, C->zd + inst * C->zd_stride + (size_t)ksp * nz);
            for (int t = 0; t < nj; t++) L[Y.UJ + t] = (gate && C->has_fric) ? lp_friction(t, Y, L, I.r[t], M, C->fric[t]) : 0.0;      // (noise: GPU tests only)
            if (C->has_pid) for (int t = 0; t < nj; t++) lp_pid(t, Y, L, I.r[t], M, C, k == 1);
            if (gate)
                for (int i = 0; i < C->mu; i++) {
                    double s = 0.0;
                    if (C->K) for (int t = 0; t < 64; t++) s += ph_gain_partial(t, 64, nb, Y, L, C->K + inst * C->K_stride + ((size_t)kidx * C->mu + i) * 12 * nb);
                    L[Y.UJ + C->cj[i]] += (C->Fd ? C->Fd[inst * C->Fd_stride + (size_t)ksp * C->mu + i] : 0.0) - s;
                }
            for (int t = 0; t < 64; t++) lp_forces(t, Y, L, I.r[t], M);
            for (int t = 0; t < 64; t++) lp_knot_jac(t, Y, L, I.r[t], M);
            for (int t = 0; t < 64; t++) lp_force_map(t, Y, L, M);
            bool done = dead, failed = false;
            int its = 0;
            double normf0 = loop_eval<true>(I, Y.S, 0.0);
            for (int iter = 1; iter <= 100 && !done; iter++) {
                loop_solve(I);
                for (int t = 0; t < 64; t++) lp_body_solve(t, Y, L, M);
                double alpha = 1.0, normf1 = 0.0, pd = 0.0;
                for (int t = 0; t < 64; t++) pd += lp_trial(t, Y, L, M, alpha);
                const double nd = sqrt(pd);
                for (int ls = 0; ls <= 10; ls++) {
                    normf1 = loop_eval<false>(I, Y.ST, alpha);
                    if (!(normf1 > normf0) || ls == 10) break;
                    alpha *= 0.5;
                    for (int t = 0; t < 64; t++) lp_trial(t, Y, L, M, alpha);
                }
                for (int t = 0; t < 64; t++) lp_accept(t, Y, L, M, alpha);
                its = iter;
                if (normf1 < 1e-10 && alpha * nd < 1e-10) done = true;
                if (!(normf1 < 1e300)) { done = true; failed = true; }
                if (!done) normf0 = loop_eval<true>(I, Y.S, 0.0);
            }
            if (!dead) {
                const bool conv = done && !failed;
                if (!conv) bad = true;
                if (its > worst) worst = its;
                if (!conv && its < 100) {
                    dead = true;
                    for (int b = 0; b < nb; b++) for (int i = 0; i < 6; i++) L[Y.Z + 13 * b + 7 + i] = 0.0;
                } else {
                    for (int t = 0; t < 64; t++) ph_update(t, nb, Y, L);
                }
            }
        }
        for (int e = 0; e < nz; e++) zT[inst * nz + e] = L[Y.Z + e];
        if (lam) for (int e = 0; e < 5 * nj; e++) lam[inst * 5 * nj + e] = L[Y.LAM + e];
        if (status) status[inst] = bad ? -worst : worst;
    }
    return CCLQR_OK;
}


// linearsystem on the closed-loop layout (csrc/cclqr_lin_loop.h) in the phase order of linearize_loop_kernel (csrc/rollout_loop.hip): one
// converged Newton step at the setpoint, then A, Bu, Bl, G with the multipliers exogenous (row major, caller's body / joint order).
// force_loop != 0 sends a TREE through the closed-loop tables too, so that the result can be compared with the tree linearisation.
extern "C" int emu_loop_linearize(const cclqr_mech_desc* md, int force_loop, const double* zd, int mu, const int32_t* ctrl_joint, const double* Fd,
                                  double* A, double* Bu, double* Bl, double* G, int* status) {
    cclqr_mech m;
    std::string err;
    int rc = force_loop ? build_loop_tables(md, &m, err) : build_mech_tables(md, &m, err);
    if (rc) return rc;
    if (!m.host.loop) return CCLQR_EUNSUPPORTED;
    const MechDev* M = &m.host;
    const int nb = M->nb, nj = M->nj, nz = 13 * nb, mx = 12 * nb, ml = 5 * nj;
    LoopInst I;
    I.M = M;
    I.Y = make_loop_layout(nb, nj);
    const int JB = I.Y.total;
    I.lds.assign(I.Y.total + LJB * nj, 0.0);
    I.L = I.lds.data();
    I.r.resize(64);
    const Lay& Y = I.Y;
    double* L = I.L;
    for (int t = 0; t < 64; t++) loop_load_consts(I.r[t], M, t);
    for (int e = 0; e < nz; e++) L[Y.Z + e] = zd[e];
    std::vector<int> cj(mu > 0 ? mu : 1, 0);
    for (int i = 0; i < mu; i++) { cj[i] = ctrl_joint[i]; L[Y.UJ + cj[i]] += Fd ? Fd[i] : 0.0; }
    for (int t = 0; t < 64; t++) lp_forces(t, Y, L, I.r[t], M);
    for (int t = 0; t < 64; t++) lp_knot_jac(t, Y, L, I.r[t], M);
    for (int t = 0; t < 64; t++) lp_force_map(t, Y, L, M);
    bool done = false, failed = false;
    int its = 0;
    double normf0 = loop_eval<true>(I, Y.S, 0.0);
    for (int iter = 1; iter <= 100 && !done; iter++) {
        loop_solve(I);
        for (int t = 0; t < 64; t++) lp_body_solve(t, Y, L, M);
        double alpha = 1.0, normf1 = 0.0, pd = 0.0;
        for (int t = 0; t < 64; t++) pd += lp_trial(t, Y, L, M, alpha);
        const double nd = sqrt(pd);
        for (int ls = 0; ls <= 10; ls++) {
            normf1 = loop_eval<false>(I, Y.ST, alpha);
            if (!(normf1 > normf0) || ls == 10) break;
            alpha *= 0.5;
            for (int t = 0; t < 64; t++) lp_trial(t, Y, L, M, alpha);
        }
        for (int t = 0; t < 64; t++) lp_accept(t, Y, L, M, alpha);
        its = iter;
        if (normf1 < 1e-10 && alpha * nd < 1e-10) done = true;
        if (!(normf1 < 1e300)) { done = true; failed = true; }
        if (!done) normf0 = loop_eval<true>(I, Y.S, 0.0);
    }
    loop_eval<true>(I, Y.S, 0.0);        // D_R^-1, N D_R^-1 and the next pose at the converged solution
    if (status) *status = (done && !failed) ? its : -its;
    for (int e = 0; e < mx * mx; e++) A[e] = 0.0;
    for (int e = 0; e < mx * mu; e++) Bu[e] = 0.0;
    for (int e = 0; e < mx * ml; e++) Bl[e] = 0.0;
    for (int e = 0; e < ml * mx; e++) G[e] = 0.0;
    LinOut O;
    O.A = A; O.Bu = Bu; O.Bl = Bl; O.G = G; O.mx = mx; O.mu = mu; O.ml = ml;
    for (int t = 0; t < 64; t++) lp_lin_joint(t, Y, JB, L, I.r[t], M);
    for (int t = 0; t < 64; t++) lp_lin_rows_A(t, Y, JB, L, I.r[t], M, O);
    for (int t = 0; t < 64; t++) lp_lin_rows_B(t, Y, L, I.r[t], M, cj.data(), O);
    return CCLQR_OK;
}
