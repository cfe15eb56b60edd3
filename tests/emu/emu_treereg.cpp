// emu_treereg.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libcclqr.so).
// Runs the register-resident tree kernel's __host__ __device__ phase functions (csrc/cclqr_treereg.h on top of csrc/cclqr_chain.h) and its
// host-made schedule (csrc/cclqr_treereg_tables.h) serially on the CPU, lane by lane, in the phase order of csrc/rollout_treereg.hip, so that
// the kernel's arithmetic, the sibling blocks, the elimination records, the LDS layout and the indexing can be checked against the oracle
// without a GPU.  What the kernel moves by ds_bpermute is read here from the other lane's struct.  The LDS image is filled with
// signalling NaNs before every instance, so a read of anything the kernel would not have written poisons the result
// (the GPU does not start from zeros).  Build with -fsanitize=address,undefined to catch out-of-range offsets.
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_tables.h"
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_treereg_tables.h"
#include <limits>
#include <math.h>
#include <string>
#include <array>
#include <vector>

using namespace cclqr;

namespace {
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
    std::vector<TreeL> tr;
    const TreeRegDev* R;
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
    for (int t = 0; t < G; t++) {
        LaneTmp& T = I.T[t];
        const LinkC& c = I.c[t];
        if (!(active && c.on())) continue;
        const double* pxq = c.has_a() ? I.T[I.tr[t].par].xq : ORIGIN13;
        const double* pNB = I.T[I.tr[t].par].NB;
        joint_eval_sparse<JAC>(c, pxq, pxq + 3, T.xq, T.xq + 3, pNB, T.NB, T.g, T.wXT, T.wPB, T.wPA);
        for (int i = 0; i < 5; i++) T.part += T.g[i] * T.g[i];
    }
    if (JAC)
        for (int t = 0; t < G; t++) {
            const LinkC& c = I.c[t];
            if (!(active && c.on())) continue;
            const double* pd = I.S[I.tr[t].par].d;
            tr_schur_rows(c, I.tr[t], t, true, I.R->maxchild, I.R->maxsib, Y, L, I.T[t].wXT, I.T[t].wPB, I.T[t].wPA, I.T[t].g, I.S[t].d, pd);
        }
    double acc = 0.0;
    for (int t = 0; t < G; t++) acc += I.T[t].part;
    return sqrt(acc);
}
}  // namespace

extern "C" int emu_treereg_rollout(const cclqr_mech_desc* md, const cclqr_ctrl_desc* cd, int64_t n_inst, int steps, int k0, const double* z0,
                                 const double* noise, double* traj, double* zT, int* status, int G_override) {
    cclqr_mech m;
    std::string err;
    int rc = build_mech_tables(md, &m, err);
    if (rc) return rc;
    if (!m.host.tree) return CCLQR_EUNSUPPORTED;
    static TreeRegDev Rt;
    if (!build_treereg_tables(m.host, Rt, err)) return CCLQR_EUNSUPPORTED;
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
    (void)G_override;
    I.G = Rt.lanes;
    if (I.G < nb) return CCLQR_EINVAL;
    I.nb = nb; I.dt = dt; I.R = &Rt;
    I.Y = make_treereg_layout(Rt.nbp, Rt.nss);
    I.lds.resize(I.Y.total);          // exact size: an out-of-range offset is an out-of-bounds access for the sanitizer
    I.L = I.lds.data();
    I.c.resize(I.G); I.S.resize(I.G); I.T.resize(I.G); I.tr.resize(I.G);
    const Lay& Y = I.Y;
    double* L = I.L;
    const int G = I.G;
    for (int t = 0; t < G; t++) {
        link_load_consts(I.c[t], M, t, nb, dt);
        tree_load(I.tr[t], M, &Rt, t, nb);
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
                for (int i = 0; i < 13; i++) za[13 * t + i] = I.c[t].has_a() ? zf[13 * I.tr[t].par + i] : ORIGIN13[i];
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
                    for (int k = 0; k < I.tr[t].nchild; k++) for (int i = 0; i < 3; i++) { F[3 * t + i] += W6[6 * I.tr[t].child[k] + i]; tau[3 * t + i] += W6[6 * I.tr[t].child[k] + 3 + i]; }
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
                    for (int i = 0; i < 6; i++) {
                        double v = own[6 * t + i];
                        for (int k = 0; k < I.tr[t].nchild; k++) v += par[6 * I.tr[t].child[k] + i];
                        L[Y.C + 6 * t + i] = v; I.S[t].cd[i] = 0.0;
                    }
            }
            // ---- newton
            bool done = dead, failed = false;
            int its = 0;
            double normf0 = chain_eval<true>(I, 0.0, !done);
            for (int iter = 1; iter <= 100 && !done; iter++) {
                for (int s = 0; s < Rt.ne_steps; s++) for (int t = 0; t < G; t++) tr_elim(Rt.el[s][t], L);
                for (int s = 0; s < Rt.nb_steps; s++) for (int t = 0; t < G; t++) tr_back(Rt.bk[s][t], L);
                double pdn = 0.0;
                {
                    std::vector<double> own(6 * G), par(6 * G);
                    for (int t = 0; t < nb; t++) gk_t_apply(I.c[t], t, Y, L, L + Y.DL + 5 * t, &own[6 * t], &par[6 * t]);
                    for (int t = 0; t < nb; t++) {
                        double DINV[9];
                        for (int i = 0; i < 9; i++) DINV[i] = L[Y.DINV + 9 * t + i];
                        for (int i = 0; i < 6; i++) {
                            double v = own[6 * t + i];
                            for (int k = 0; k < I.tr[t].nchild; k++) v += par[6 * I.tr[t].child[k] + i];
                            I.S[t].cd[i] = v;
                        }
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

// lengths of the elimination / back substitution schedules and lanes per instance (tests: the schedule is shorter than one step per link)
extern "C" int emu_treereg_schedule(const cclqr_mech_desc* md, int* ne_steps, int* nb_steps, int* lanes) {
    cclqr_mech m;
    std::string err;
    int rc = build_mech_tables(md, &m, err);
    if (rc) return rc;
    if (!m.host.tree) return CCLQR_EUNSUPPORTED;
    static TreeRegDev Rt;
    if (!build_treereg_tables(m.host, Rt, err)) return CCLQR_EUNSUPPORTED;
    *ne_steps = Rt.ne_steps; *nb_steps = Rt.nb_steps; *lanes = Rt.lanes;
    return 0;
}
