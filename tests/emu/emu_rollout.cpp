// emu_rollout.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libcclqr.so).
// Runs the rollout kernel's __host__ __device__ phase functions (csrc/cclqr_dev.h) serially on the CPU with the same
// phase order as the round-1 LDS-resident rollout kernel (now only linearize.hip runs these phases), lane by lane, so that the kernel's arithmetic, LDS layout and indexing can be checked
// against the oracle without a GPU.  It is not a fallback: nothing in the product can reach it.
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_tables.h"
#include <vector>
#include <string>
#include <math.h>

using namespace cclqr;

template <bool JAC>
static double eval_point(int G, int nb, const Lay& Y, double* L, std::vector<LaneRegs>& R, const MechDev* M, double dt, int s_off, double alpha) {
    double acc = 0.0;
    for (int t = 0; t < G; t++) acc += ph_body_eval<JAC>(t, nb, Y, L, R[t], dt, s_off, alpha);
    for (int t = 0; t < G; t++) acc += ph_joint_eval<JAC>(t, nb, Y, L, R[t], dt);
    return sqrt(acc);
}

// serial twin of cclqr_newton.h newton_solve<G> (one group)
static int emu_newton(int G, int nb, const Lay& Y, double* L, std::vector<LaneRegs>& R, const MechDev* M, double dt, bool* converged) {
    double normf0 = eval_point<true>(G, nb, Y, L, R, M, dt, Y.S, 0.0);
    bool done = false;
    int its = 0;
    int s_cur = Y.S, s_try = Y.ST, l_cur = Y.LAM, l_try = Y.LT;
    for (int iter = 1; iter <= 100 && !done; iter++) {
        if (M->tree) {
            for (int t = 0; t < G; t++) ph_schur_s_tree(t, G, nb, Y, L, M);
            for (int l = nb - 1; l >= 0; l--) for (int t = 0; t < G; t++) ph_tree_elim(t, l, Y, L, M);
            for (int l = 0; l < nb; l++) for (int t = 0; t < G; t++) ph_tree_back(t, l, Y, L, M);
            for (int t = 0; t < G; t++) ph_body_solve_tree(t, G, nb, Y, L, M);
        } else {
        for (int t = 0; t < G; t++) ph_schur_s(t, G, nb, Y, L, M->start_mask);
        for (int c = 0; c < M->nchains; c++) {
            const TriPlan P = tri_plan(M->chain_start[c], M->chain_len[c]);
            for (int i = 0; i < P.steps; i++) {
                double lu[64][5];
                int l[64];
                bool act[64];
                for (int t = 0; t < G; t++) act[t] = ph_tri_elim(t, i, P, Y, L, lu[t], &l[t]);
                for (int t = 0; t < G; t++) if (act[t]) ph_tri_store(t, l[t], Y, L, lu[t]);
            }
            for (int t = 0; t < G; t++) ph_tri_mid(t, P, Y, L);
            for (int j = 0; j < P.steps; j++) for (int t = 0; t < G; t++) ph_tri_back(t, j, P, Y, L);
        }
        for (int t = 0; t < G; t++) ph_body_solve(t, G, nb, Y, L, M->end_mask);
        }
        double alpha = 1.0, normf1 = 0.0, nd = 0.0;
        bool jac_ok = true, ls_done = false;
        {
            double pd = 0.0;
            for (int t = 0; t < G; t++) pd += ph_trial(t, G, nb, Y, L, alpha, s_cur, s_try, l_cur, l_try);
            nd = sqrt(pd);
            normf1 = eval_point<true>(G, nb, Y, L, R, M, dt, s_try, alpha);
            if (!(normf1 > normf0)) ls_done = true;
        }
        // serial twin of the level-parallel halvings: lane group lg = t / nb evaluates level lv + lg in its own slot of the Schur blocks
        const int NL = newton_level_groups(G, nb);
        for (int lv = 1; lv <= 10 && !ls_done; lv += NL) {
            double part[LEVEL_SLOTS] = {0, 0, 0};
            for (int ph = 0; ph < 3; ph++)
                for (int t = 0; t < G; t++) {
                    const int lg = t / nb, tl = t - lg * nb;
                    if (!(lg < NL && lv + lg <= 10)) continue;
                    const double a_l = ldexp(1.0, -(lv + lg));
                    const Lay V = level_layout(Y, nb, lg);
                    if (ph == 0) ph_trial_level(tl, nb, Y, V, L, a_l, s_cur, l_cur);
                    if (ph == 1) part[lg] += ph_body_eval<false>(tl, nb, V, L, R[t], dt, V.ST, a_l);
                    if (ph == 2) part[lg] += ph_joint_eval<false>(tl, nb, V, L, R[t], dt);
                }
            int chosen = -1;
            for (int q = 0; q < NL; q++)
                if (chosen < 0 && lv + q <= 10 && (!(sqrt(part[q]) > normf0) || lv + q == 10)) { chosen = q; normf1 = sqrt(part[q]); alpha = ldexp(1.0, -(lv + q)); }
            if (chosen >= 0) {
                ls_done = true; jac_ok = false;
                for (int t = 0; t < G; t++) ph_level_commit(t, nb, Y, level_layout(Y, nb, chosen), L, s_try, l_try);
            }
        }
        for (int t = 0; t < G; t++) ph_accept(t, G, nb, Y, L, alpha);
        { int q = s_cur; s_cur = s_try; s_try = q; q = l_cur; l_cur = l_try; l_try = q; }
        its = iter;
        if (normf1 < 1e-10 && alpha * nd < 1e-10) done = true;
        normf0 = normf1;
        if (!done && !jac_ok) eval_point<true>(G, nb, Y, L, R, M, dt, s_cur, 0.0);
    }
    for (int t = 0; t < G; t++) ph_copy_solution(t, G, nb, Y, L, s_cur, l_cur);
    *converged = done;
    return its;
}

extern "C" int emu_rollout(const cclqr_mech_desc* md, const cclqr_ctrl_desc* cd, int64_t n_inst, int steps, int k0, const double* z0,
                           const double* noise, double* traj, double* zT, int* status, int G_override) {
    cclqr_mech m;
    std::string err;
    int rc = build_mech_tables(md, &m, err);
    if (rc) return rc;
    CtrlHostTables T;
    rc = build_ctrl_tables(&m, cd, T, err);
    if (rc) return rc;
    T.H.K = T.K.empty() ? nullptr : T.K.data();
    T.H.zd = T.zd.data();
    T.H.Fd = T.Fd.empty() ? nullptr : T.Fd.data();
    const MechDev* M = &m.host;
    const CtrlDev* C = &T.H;
    const int nb = M->nb, nz = 13 * nb;
    const double dt = M->dt;
    const int G = G_override > 0 ? G_override : [&] { const int g = nb <= 4 ? 16 : (nb <= 8 ? 32 : 64); return M->tree > g ? (M->tree <= 16 ? 16 : (M->tree <= 32 ? 32 : 64)) : g; }();
    const Lay Y = make_layout(nb, M->tree ? 2 * M->npairs : 0);
    std::vector<double> lds(Y.total);
    std::vector<LaneRegs> R(G);
    for (int64_t inst = 0; inst < n_inst; inst++) {
        double* L = lds.data();
        for (int e = 0; e < Y.total; e++) L[e] = 0.0;
        const int NLg = newton_level_groups(G, nb);
        for (int t = 0; t < G; t++) lane_load_consts(R[t], M, t / nb < NLg ? t % nb : 0);
        for (int e = 0; e < nz; e++) { int l = e / 13, c = e - 13 * l; L[Y.Z + e] = z0[inst * nz + M->perm[l] * 13 + c]; }
        int worst = 0; bool bad = false;
        for (int kk = 0; kk < steps; kk++) {
            const int k = k0 + kk;
            if (traj) for (int e = 0; e < nz; e++) { int l = e / 13, c = e - 13 * l; traj[((size_t)inst * steps + kk) * nz + M->perm[l] * 13 + c] = L[Y.Z + e]; }
            const bool gate = (C->N <= 0) || (k < C->N);
            const int ksp = (C->nsp > 1) ? ((k - 1 < C->nsp) ? k - 1 : C->nsp - 1) : 0;
            const int kidx = (C->N <= 0) ? 0 : ((k - 1 < C->nK) ? k - 1 : C->nK - 1);
            for (int t = 0; t < G; t++) {
                if (gate) ph_control_error(t, nb, Y, L, R[t], C, C->zd + (size_t)ksp * nz);
                else if (t < nb) L[Y.UJ + t] = 0.0;
            }
            if (gate)
                for (int i = 0; i < C->mu; i++) {
                    double s = 0.0;
                    if (C->K) for (int t = 0; t < G; t++) s += ph_gain_partial(t, G, nb, Y, L, C->K + ((size_t)kidx * C->mu + i) * 12 * nb);
                    double u = (C->Fd ? C->Fd[(size_t)ksp * C->mu + i] : 0.0) - s;
                    if (C->noise_scale != 0.0) {
                        if (noise) u += C->noise_scale * noise[(size_t)inst * steps + (k - k0)];
                        else if (C->noise_philox) u += C->noise_scale * philox_normal(C->noise_key0, (unsigned long long)inst, k);
                    }
                    L[Y.UJ + C->cj[i]] += u;
                }
            if (C->has_pid) for (int t = 0; t < G; t++) ph_pid(t, nb, Y, L, R[t], C, dt, k == 1);
            for (int t = 0; t < G; t++) {
                const int lg = t / nb, tl = t - lg * nb;
                if (lg < NLg) { if (M->tree) ph_forces<true>(tl, nb, Y, L, R[t], M, lg == 0); else ph_forces<false>(tl, nb, Y, L, R[t], M, lg == 0); }
                ph_knot_jac(t, nb, Y, L, R[t]);
            }
            for (int t = 0; t < G; t++) { if (M->tree) ph_force_map_tree(t, G, nb, Y, L, M); else ph_force_map(t, G, nb, Y, L, M->end_mask); }
            bool done = false;
            int its = emu_newton(G, nb, Y, L, R, M, dt, &done);
            if (!done) bad = true;
            if (its > worst) worst = its;
            for (int t = 0; t < G; t++) ph_update(t, nb, Y, L);
        }
        for (int e = 0; e < nz; e++) { int l = e / 13, c = e - 13 * l; zT[inst * nz + M->perm[l] * 13 + c] = L[Y.Z + e]; }
        if (status) status[inst] = bad ? -worst : worst;
    }
    return 0;
}

#include "../../constrainedcontrol.jl_amd/csrc/cclqr_lin_dev.h"
// serial twin of linearize.hip's kernel body for one knot
extern "C" int emu_linearize(const cclqr_mech_desc* md, const double* zd, int mu, const int* ctrl_joint, const double* Fd, double* A, double* Bu,
                             double* Bl, double* Gm) {
    cclqr_mech m;
    std::string err;
    int rc = build_mech_tables(md, &m, err);
    if (rc) return rc;
    const MechDev* M = &m.host;
    const int nb = M->nb, nz = 13 * nb, mx = 12 * nb, ml = 5 * nb, G = 64;
    const double dt = M->dt;
    const Lay Y = make_layout(nb, M->tree ? 2 * M->npairs : 0);
    const int JB = Y.total;
    std::vector<double> lds(Y.total + LJB * nb, 0.0);
    double* L = lds.data();
    std::vector<LaneRegs> R(G);
    int cj[CCLQR_MAXL];
    for (int i = 0; i < mu; i++) cj[i] = m.link_of_joint[ctrl_joint[i]];
    LinOut O;
    O.A = A; O.Bu = Bu; O.Bl = Bl; O.G = Gm; O.mx = mx; O.mu = mu; O.ml = ml;
    for (int e = 0; e < mx * mx; e++) A[e] = 0;
    for (int e = 0; e < mx * mu; e++) Bu[e] = 0;
    for (int e = 0; e < mx * ml; e++) Bl[e] = 0;
    for (int e = 0; e < ml * mx; e++) Gm[e] = 0;
    const int NLg = newton_level_groups(G, nb);
    for (int t = 0; t < G; t++) lane_load_consts(R[t], M, t / nb < NLg ? t % nb : 0);
    for (int e = 0; e < nz; e++) { int l = e / 13, c = e - 13 * l; L[Y.Z + e] = zd[M->perm[l] * 13 + c]; }
    for (int i = 0; i < mu; i++) L[Y.UJ + cj[i]] += Fd ? Fd[i] : 0.0;
    for (int t = 0; t < G; t++) {
        const int lg = t / nb, tl = t - lg * nb;
        if (lg < NLg) { if (M->tree) ph_forces<true>(tl, nb, Y, L, R[t], M, lg == 0); else ph_forces<false>(tl, nb, Y, L, R[t], M, lg == 0); }
        ph_knot_jac(t, nb, Y, L, R[t]);
    }
    for (int t = 0; t < G; t++) { if (M->tree) ph_force_map_tree(t, G, nb, Y, L, M); else ph_force_map(t, G, nb, Y, L, M->end_mask); }
    bool done = false;
    emu_newton(G, nb, Y, L, R, M, dt, &done);
    if (!done) return -3;
    for (int t = 0; t < G; t++) ph_body_eval<true>(t, nb, Y, L, R[t], dt, Y.S, 0.0);
    for (int t = 0; t < G; t++) ph_lin_joint(t, nb, Y, JB, L, R[t]);
    for (int t = 0; t < G; t++) { ph_lin_rows_A(t, nb, Y, JB, L, R[t], M, O); ph_lin_rows_B(t, nb, Y, L, R[t], M, cj, O); }
    return 0;
}
