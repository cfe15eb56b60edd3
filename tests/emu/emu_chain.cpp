// emu_chain.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libcclqr.so).
// Runs the chain rollout kernel's __host__ __device__ phase functions (csrc/cclqr_chain.h + the block-tridiagonal sweep of
// csrc/cclqr_dev.h) serially on the CPU, lane by lane, in the phase order of csrc/rollout_chain.hip, so that the kernel's
// arithmetic, the sparse Jacobian algebra, the LDS layout and the indexing can be checked against the oracle without a GPU.
// What the kernel moves by DPP wave shifts is read here from the neighbour lane's struct.  The LDS image is filled with
// signalling NaNs before every instance, so a read of anything the kernel would not have written poisons the result
// (the GPU does not start from zeros).  Build with -fsanitize=address,undefined to catch out-of-range offsets.
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_tables.h"
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_chain.h"
#include <limits>
#include <math.h>
#include <string>
#include <array>
#include <vector>

using namespace cclqr;

namespace {
// one cyclic-reduction level as the kernel runs it: every lane's loads and arithmetic of a pass, then every lane's stores
template <int W>
static void cr_level_emu(int G, int cs, int cn, const cclqr::Lay& Y, double* L) {
    using namespace cclqr;
    std::vector<CrLane<W>> K(G);
    std::vector<std::array<std::array<double, 5>, CrLane<W>::NS>> tc(G);
    for (int t = 0; t < G; t++) cr_setup<W>(K[t], t, cs, cn, 1, Y, false);
    for (int t = 0; t < G; t++) cr_phase_a<W>(K[t], L, reinterpret_cast<double(*)[5]>(tc[t].data()));
    for (int t = 0; t < G; t++) if (K[t].act()) cr_store_a<W>(K[t], L, reinterpret_cast<const double(*)[5]>(tc[t].data()));
    for (int t = 0; t < G; t++) cr_phase_b<W>(K[t], L, reinterpret_cast<double(*)[5]>(tc[t].data()));
    for (int t = 0; t < G; t++) if (K[t].act()) cr_store_b<W>(K[t], L, reinterpret_cast<const double(*)[5]>(tc[t].data()));
}

int g_lanes_per_link = 1;      // emu_chain_set_lanes_per_link: 2 / 3 = the evaluations with Jacobians go through the row-split forms of cclqr_chain.h
struct LinkS { double z[7], s[6], ds[6], cd[6], d[6]; };
struct LaneTmp { double xq[7], NB[9], g[5], wXT[3][3], wPB[5][3], wPA[5][3]; double part; };

struct Inst {
    int G, nb;
    double dt;
    Lay Y;
    std::vector<double> lds;
    std::vector<LinkC> c;
    std::vector<LinkS> S;
    std::vector<LaneTmp> T;
    double* L;
};

const double ORIGIN13[13] = {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0};

template <bool JAC>
double chain_eval(Inst& I, double alpha, bool active) {
    const int G = I.G;
    double* L = I.L;
    const Lay& Y = I.Y;
    for (int t = 0; t < G; t++) {
        LaneTmp& T = I.T[t];
        T.part = 0.0;
        for (int k = 0; k < 7; k++) T.xq[k] = I.S[t].z[k];
        if (active && I.c[t].on()) {
            double cf[6], sv[6], cTR[6], DINV[9];
            for (int k = 0; k < 6; k++) { cf[k] = L[Y.C + 6 * t + k] - alpha * I.S[t].cd[k]; sv[k] = I.S[t].s[k] - alpha * I.S[t].ds[k]; cTR[k] = L[Y.D + 6 * t + k]; }
            T.part = ck_body_eval<JAC>(I.c[t], I.S[t].z, sv, cf, cTR, cTR + 3, I.dt, T.xq, I.S[t].d, DINV, T.NB);
            if (JAC) for (int k = 0; k < 9; k++) L[Y.DINV + 9 * t + k] = DINV[k];
        }
    }
    if (JAC && g_lanes_per_link > 1) {
        // several lanes per link (cclqr_chain.h): every sub-lane w of a link evaluates the rows of its slots and writes their Schur rows; together they
        // must cover every entry the one-lane form writes (the LDS image starts as signalling NaNs)
        for (int t = 0; t < G; t++) {
            const LinkC& c = I.c[t];
            if (!(active && c.on())) continue;
            const double* pxq = c.has_a() ? I.T[t - 1].xq : ORIGIN13;
            const double* pNB = c.has_a() ? I.T[t - 1].NB : I.T[t].NB;
            const double* pd = c.has_a() ? I.S[t - 1].d : I.S[t].d;
            for (int w = 0; w < g_lanes_per_link; w++) {
                SubSel Q;
                double g3[3], XT[3][3], PB[3][3], PA[3][3];
                if (g_lanes_per_link == 3) {
                    sub_setup<3>(c, w, Q);
                    joint_eval_rows<3>(c, Q, pxq, pxq + 3, I.T[t].xq, I.T[t].xq + 3, pNB, I.T[t].NB, g3, XT, PB, PA);
                    for (int i = 0; i < SubRows<3>::NR; i++) I.T[t].part += g3[i] * g3[i];
                } else {
                    sub_setup<2>(c, w, Q);
                    joint_eval_rows<2>(c, Q, pxq, pxq + 3, I.T[t].xq, I.T[t].xq + 3, pNB, I.T[t].NB, g3, XT, PB, PA);
                    for (int i = 0; i < SubRows<2>::NR; i++) I.T[t].part += g3[i] * g3[i];
                }
            }
        }
        for (int t = 0; t < G; t++) {     // (all evaluations first: a link's Schur rows read the parent's residual, as the kernel's wave shift does)
            const LinkC& c = I.c[t];
            if (!(active && c.on())) continue;
            const double* pxq = c.has_a() ? I.T[t - 1].xq : ORIGIN13;
            const double* pNB = c.has_a() ? I.T[t - 1].NB : I.T[t].NB;
            const double* pd = c.has_a() ? I.S[t - 1].d : I.S[t].d;
            for (int w = 0; w < g_lanes_per_link; w++) {
                SubSel Q;
                double g3[3], XT[3][3], PB[3][3], PA[3][3];
                if (g_lanes_per_link == 3) {
                    sub_setup<3>(c, w, Q);
                    joint_eval_rows<3>(c, Q, pxq, pxq + 3, I.T[t].xq, I.T[t].xq + 3, pNB, I.T[t].NB, g3, XT, PB, PA);
                    ck_schur_rows_sub<3>(c, Q, t, true, Y, L, g3, XT, PB, PA, I.S[t].d, pd);
                } else {
                    sub_setup<2>(c, w, Q);
                    joint_eval_rows<2>(c, Q, pxq, pxq + 3, I.T[t].xq, I.T[t].xq + 3, pNB, I.T[t].NB, g3, XT, PB, PA);
                    ck_schur_rows_sub<2>(c, Q, t, true, Y, L, g3, XT, PB, PA, I.S[t].d, pd);
                }
            }
        }
    } else {
    for (int t = 0; t < G; t++) {
        LaneTmp& T = I.T[t];
        const LinkC& c = I.c[t];
        if (!(active && c.on())) continue;
        const double* pxq = c.has_a() ? I.T[t - 1].xq : ORIGIN13;
        const double* pNB = c.has_a() ? I.T[t - 1].NB : I.T[t].NB;
        joint_eval_sparse<JAC>(c, pxq, pxq + 3, T.xq, T.xq + 3, pNB, T.NB, T.g, T.wXT, T.wPB, T.wPA);
        for (int i = 0; i < 5; i++) T.part += T.g[i] * T.g[i];
    }
    if (JAC)
        for (int t = 0; t < G; t++) {
            const LinkC& c = I.c[t];
            if (!(active && c.on())) continue;
            const double* pd = c.has_a() ? I.S[t - 1].d : I.S[t].d;
            ck_schur_rows(c, t, true, Y, L, I.T[t].wXT, I.T[t].wPB, I.T[t].wPA, I.T[t].g, I.S[t].d, pd);
        }
    }
    double acc = 0.0;
    for (int t = 0; t < G; t++) acc += I.T[t].part;
    return sqrt(acc);
}
}  // namespace

extern "C" void emu_chain_set_lanes_per_link(int kl) { g_lanes_per_link = (kl == 2 || kl == 3) ? kl : 1; }

extern "C" int emu_chain_rollout(const cclqr_mech_desc* md, const cclqr_ctrl_desc* cd, int64_t n_inst, int steps, int k0, const double* z0,
                                 const double* noise, double* traj, double* zT, int* status, int G_override) {
    cclqr_mech m;
    std::string err;
    int rc = build_mech_tables(md, &m, err);
    if (rc) return rc;
    if (m.host.tree) return CCLQR_EUNSUPPORTED;
    CtrlHostTables Tb;
    rc = build_ctrl_tables(&m, cd, Tb, err);
    if (rc) return rc;
    Tb.H.K = Tb.K.empty() ? nullptr : Tb.K.data();
    Tb.H.zd = Tb.zd.data();
    Tb.H.Fd = Tb.Fd.empty() ? nullptr : Tb.Fd.data();
    const MechDev* M = &m.host;
    const CtrlDev* C = &Tb.H;
    const int nb = M->nb, nz = 13 * nb;
    const double dt = M->dt;
    Inst I;
    I.G = G_override > 0 ? G_override : (nb <= 4 ? 8 : (nb <= 8 ? 16 : (nb <= 32 ? 32 : 64)));
    if (I.G < nb) return CCLQR_EINVAL;
    I.nb = nb; I.dt = dt;
    I.Y = make_chain_layout(nb <= 4 ? 4 : (nb <= 8 ? 8 : (nb <= 16 ? 16 : (nb == 17 ? 17 : (nb <= 32 ? 32 : 64)))));     // = chain_layout_links(nb) of rollout_chain.hip
    I.lds.resize(I.Y.total);          // exact size: an out-of-range offset is an out-of-bounds access for the sanitizer
    I.L = I.lds.data();
    I.c.resize(I.G); I.S.resize(I.G); I.T.resize(I.G);
    const Lay& Y = I.Y;
    double* L = I.L;
    const int G = I.G;
    for (int t = 0; t < G; t++) {
        link_load_consts(I.c[t], M, t, nb, dt);
        if (C->has_fric && I.c[t].on()) I.c[t].fric = C->fric[t];
    }
    std::vector<double> pid_int(G), pid_last(G);
    for (int64_t inst = 0; inst < n_inst; inst++) {
        for (int e = 0; e < Y.total; e++) L[e] = std::numeric_limits<double>::signaling_NaN();
        // what the kernel initialises: the multipliers (zero, or the caller's warm start)
        for (int t = 0; t < nb; t++) for (int i = 0; i < 5; i++) L[Y.LAM + 5 * t + i] = 0.0;
        for (int t = 0; t < G; t++) {
            const bool on = I.c[t].on();
            const int ut = on ? M->perm[t] : 0;
            for (int i = 0; i < 7; i++) I.S[t].z[i] = on ? z0[inst * nz + ut * 13 + i] : (i == 3 ? 1.0 : 0.0);
            for (int i = 0; i < 6; i++) { I.S[t].s[i] = on ? z0[inst * nz + ut * 13 + 7 + i] : 0.0; I.S[t].cd[i] = 0; I.S[t].d[i] = 0; I.S[t].ds[i] = 0; }
            pid_int[t] = 0; pid_last[t] = 0;
        }
        int worst = 0;
        bool bad = false, dead = false;
        for (int kk = 0; kk < steps; kk++) {
            const int k = k0 + kk;
            if (traj)
                for (int t = 0; t < nb; t++) {
                    double* dst = traj + ((size_t)inst * steps + kk) * nz + 13 * M->perm[t];
                    for (int i = 0; i < 7; i++) dst[i] = I.S[t].z[i];
                    for (int i = 0; i < 6; i++) dst[7 + i] = I.S[t].s[i];
                }
            const bool gate = (C->N <= 0) || (k < C->N);
            const int ksp = (C->nsp > 1) ? ((k - 1 < C->nsp) ? k - 1 : C->nsp - 1) : 0;
            const int kidx = (C->N <= 0) ? 0 : ((k - 1 < C->nK) ? k - 1 : C->nK - 1);
            std::vector<double> uj(G, 0.0);
            std::vector<double> zf(13 * G), za(13 * G);
            for (int t = 0; t < G; t++) {
                for (int i = 0; i < 7; i++) zf[13 * t + i] = I.S[t].z[i];
                for (int i = 0; i < 6; i++) zf[13 * t + 7 + i] = I.S[t].s[i];
            }
            for (int t = 0; t < G; t++)
                for (int i = 0; i < 13; i++) za[13 * t + i] = I.c[t].has_a() ? zf[13 * (t - 1) + i] : ORIGIN13[i];
            if (gate) {
                for (int t = 0; t < nb; t++) {
                    double dz[12];
                    ck_control_error(&zf[13 * t], C->zd + (size_t)ksp * nz + 13 * t, dz);
                    for (int i = 0; i < 12; i++) L[Y.DZ + 12 * t + i] = dz[i];
                    if (C->has_fric && I.c[t].fric != 0.0) uj[t] = ck_friction(I.c[t], &zf[13 * t], &za[13 * t]);
                }
                for (int i = 0; i < C->mu; i++) {
                    double s = 0.0;
                    if (C->K) {
                        const double* Krow = C->K + ((size_t)kidx * C->mu + i) * 12 * nb;
                        for (int e = 0; e < 12 * nb; e++) s += Krow[e] * L[Y.DZ + e];
                    }
                    double u = (C->Fd ? C->Fd[(size_t)ksp * C->mu + i] : 0.0) - s;
                    if (C->noise_scale != 0.0) {
                        if (noise) u += C->noise_scale * noise[(size_t)inst * steps + (k - k0)];
                        else if (C->noise_philox) u += C->noise_scale * philox_normal(C->noise_key0, (unsigned long long)inst, k);
                    }
                    uj[C->cj[i]] += u;
                }
            }
            if (C->has_pid)
                for (int t = 0; t < nb; t++)
                    if (C->pid_on[t]) uj[t] += ck_pid(I.c[t], &zf[13 * t], &za[13 * t], C->pid_P[t], C->pid_I[t], C->pid_D[t], C->pid_goal[t], dt, k == 1, pid_int[t], pid_last[t]);
            {
                std::vector<double> F(3 * G), tau(3 * G), W6(6 * G), par(6 * G), own(6 * G);
                for (int t = 0; t < G; t++) ck_joint_wrench(I.c[t], uj[t], &zf[13 * t + 3], &za[13 * t + 3], &F[3 * t], &tau[3 * t], &W6[6 * t], &W6[6 * t + 3]);
                for (int t = 0; t < G; t++) {
                    const LinkC& c = I.c[t];
                    if (c.has_c()) for (int i = 0; i < 3; i++) { F[3 * t + i] += W6[6 * (t + 1) + i]; tau[3 * t + i] += W6[6 * (t + 1) + 3 + i]; }
                    double cTR[6], gk[5], kXT[3][3], kPB[5][3], kPA[5][3], lam[5];
                    ck_step_invariants(c, &zf[13 * t], &F[3 * t], &tau[3 * t], dt, M->g, cTR, cTR + 3);
                    joint_eval_sparse<true>(c, &za[13 * t], &za[13 * t + 3], &zf[13 * t], &zf[13 * t + 3], nullptr, nullptr, gk, kXT, kPB, kPA);
                    if (!c.on()) continue;
                    for (int i = 0; i < 5; i++) lam[i] = L[Y.LAM + 5 * t + i];
                    gk_store(t, Y, L, kXT, kPB, kPA);
                    for (int i = 0; i < 6; i++) L[Y.D + 6 * t + i] = cTR[i];
                    jac_t_apply(c, kXT, kPB, kPA, lam, &own[6 * t], &par[6 * t]);
                }
                for (int t = 0; t < nb; t++)
                    for (int i = 0; i < 6; i++) { L[Y.C + 6 * t + i] = own[6 * t + i] + (I.c[t].has_c() ? par[6 * (t + 1) + i] : 0.0); I.S[t].cd[i] = 0.0; }
            }
            // ---- newton
            bool done = dead, failed = false;
            int its = 0;
            double normf0 = chain_eval<true>(I, 0.0, !done);
            for (int iter = 1; iter <= 100 && !done; iter++) {
                for (int ci = 0; ci < M->nchains; ci++) {
                    const int cs = M->chain_start[ci], cn = M->chain_len[ci];
                    const bool cr = G == 32 && nb <= 17 && cn >= CR_MIN_LINKS;     // = CR && cr of the kernel
                    const bool w2 = false;
                    if (cr) {
                        if (w2) cr_level_emu<2>(G, cs, cn, Y, L); else cr_level_emu<4>(G, cs, cn, Y, L);
                    }
                    const TriPlanB PB = cr ? tri_plan_balanced(cs, (cn + 1) / 2, 2) : tri_plan_balanced(cs, cn, 1, G >= 16 ? 2 : 1);
                    const TriPlan& P = PB.P;
                    std::vector<TriCur> K(G);
                    for (int t = 0; t < G; t++) K[t] = tri_cursor(t, PB, Y);
                    for (int i = 0; i < P.steps; i++) {
                        double tg[64][5], zy[64][5];
                        int otg[64], oout[64];
                        bool act[64];
                        for (int t = 0; t < G; t++) act[t] = tri_step(K[t], i, L, tg[t], zy[t], &otg[t], &oout[t]);     // every lane's loads come before any lane's stores
                        for (int t = 0; t < G; t++) if (act[t]) tri_step_store(L, otg[t], oout[t], tg[t], zy[t]);
                    }
                    for (int t = 0; t < G; t++) ck_tri_mid(t, PB, Y, L);
                    for (int j = 0; j < P.steps; j++) for (int t = 0; t < G; t++) ck_tri_back(t, j, PB, Y, L);
                    if (cr) for (int t = 0; t < G; t++) { if (w2) cr_back<2>(t, cs, cn, 1, Y, L, false); else cr_back<4>(t, cs, cn, 1, Y, L, false); }
                }
                double pdn = 0.0;
                {
                    std::vector<double> own(6 * G), par(6 * G);
                    for (int t = 0; t < nb; t++) gk_t_apply(I.c[t], t, Y, L, L + Y.DL + 5 * t, &own[6 * t], &par[6 * t]);
                    for (int t = 0; t < nb; t++) {
                        double DINV[9];
                        for (int i = 0; i < 9; i++) DINV[i] = L[Y.DINV + 9 * t + i];
                        for (int i = 0; i < 6; i++) I.S[t].cd[i] = own[6 * t + i] + (I.c[t].has_c() ? par[6 * (t + 1) + i] : 0.0);
                        ck_body_solve(I.c[t], I.S[t].d, I.S[t].cd, DINV, I.S[t].ds);
                        for (int i = 0; i < 6; i++) pdn += I.S[t].ds[i] * I.S[t].ds[i];
                        for (int i = 0; i < 5; i++) pdn += L[Y.DL + 5 * t + i] * L[Y.DL + 5 * t + i];
                    }
                }
                const double nd = sqrt(pdn);
                double alpha = 1.0, normf1 = chain_eval<true>(I, 1.0, true);
                bool jac_ok = true;
                if (normf1 > normf0)
                    for (int lv = 1; lv <= 10; lv++) {
                        alpha = ldexp(1.0, -lv);
                        normf1 = chain_eval<false>(I, alpha, true);
                        jac_ok = false;
                        if (!(normf1 > normf0)) break;
                    }
                for (int t = 0; t < nb; t++) {
                    for (int i = 0; i < 6; i++) { I.S[t].s[i] -= alpha * I.S[t].ds[i]; L[Y.C + 6 * t + i] -= alpha * I.S[t].cd[i]; I.S[t].cd[i] = 0.0; I.S[t].ds[i] = 0.0; }
                    for (int i = 0; i < 5; i++) L[Y.LAM + 5 * t + i] -= alpha * L[Y.DL + 5 * t + i];
                }
                its = iter;
                if (normf1 < 1e-10 && alpha * nd < 1e-10) done = true;
                if (!(normf1 < 1e300)) { done = true; failed = true; }
                normf0 = normf1;
                if (!done && !jac_ok) chain_eval<true>(I, 0.0, true);
            }
            if (!dead) {
                const bool conv = done && !failed;
                if (!conv) bad = true;
                if (its > worst) worst = its;
                if (!conv && its < 100) {
                    dead = true;
                    for (int t = 0; t < nb; t++) for (int i = 0; i < 6; i++) I.S[t].s[i] = 0.0;
                } else
                    for (int t = 0; t < nb; t++) {
                        double xq[7];
                        ck_next_pose(I.S[t].z, I.S[t].s, dt, xq);
                        for (int i = 0; i < 7; i++) I.S[t].z[i] = xq[i];
                    }
            }
        }
        for (int t = 0; t < nb; t++) {
            double* dst = zT + inst * nz + 13 * M->perm[t];
            for (int i = 0; i < 7; i++) dst[i] = I.S[t].z[i];
            for (int i = 0; i < 6; i++) dst[7 + i] = I.S[t].s[i];
        }
        if (status) status[inst] = bad ? -worst : worst;
    }
    return 0;
}
