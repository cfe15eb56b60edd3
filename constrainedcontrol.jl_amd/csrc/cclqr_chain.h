// cclqr_chain.h -- register-resident phases of the chain rollout kernel (rollout_chain.hip).
//
// Execution model: one mechanism instance per group of G lanes (G = 16 / 32) of ONE wavefront, lane t < nb
// owns link t (= body t and the joint that hangs it off its parent) and keeps EVERYTHING that belongs to the link in its
// own registers: constants, state, multipliers, the sparse constraint Jacobians at knot k (G_k) and k+1 (W = G_v D^-1), the
// Newton iterate.  Links of a chain are numbered root to leaf, so the parent link is lane t-1 and the child link lane t+1:
// neighbour vectors move by whole-wave DPP shifts (wave_shr:1 / wave_shl:1).  LDS holds what is gathered across lanes (the
// 5x5 Schur blocks of the block-tridiagonal solve, G_k of every joint) plus a few private per-lane slots that relieve the
// register file: 150 doubles per link, 20.4 KB for the 17-body chain instead of 40.8 KB (make_chain_layout).
//
// Sparse form of a joint's 5x6 Jacobian pair (parent side Ba, child side Bb):  row r = sel_r' [X | Phi] with
//   translational row:  Bb = [ xt_r * sxb , pb_r ]   Ba = [ -xt_r * sxa , pa_r ]     xt_r = (Ra sel_r)'
//   rotational row:     Bb = [ 0          , pb_r ]   Ba = [ 0           , pa_r ]
// so a 3x3 (XT: rows 3, 4 are always rotational) and two 5x3 arrays (PB, PA) plus two scalars replace two 5x6 blocks, and a
// product with a Jacobian costs 6 to 9 instead of 12 multiply-adds per entry.
//
// Every function is __host__ __device__ so that tests/emu_chain can run the identical arithmetic lane by lane on the CPU
// (test infrastructure only; the product never executes these on the host).  Neighbour inputs are explicit arguments:
// the kernel fills them by DPP, the emulator by reading the neighbour lane's struct.
#pragma once
#include "cclqr_dev.h"

// the instruction scheduler must not move code across this point (device code only; the host build of tests/emu ignores it)
#if defined(__HIP_DEVICE_COMPILE__)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SCHED_FENCE() ((void)0)
#endif

namespace cclqr {

// ---- constants of the owned link.  The five constraint rows are (Revolute) e0 e1 e2 | V1 V2 or (Prismatic) V1 V2 | e0 e1 e2
// (translational | rotational part; V1, V2 span the plane normal to the joint axis -- cclqr_tables.h), so the row selectors
// are rebuilt from V12 and the joint type instead of being kept as fifteen registers.
struct LinkC {
    double m, J[9], p1[3], p2[3], V12[6], qoc[4], axis[3];
    double dtm;        // dt / m
    double sxb, sxa;   // dt^2 / m of the own body / of the parent body (0 when the parent is the origin)
    double fric;       // viscous friction of the own joint
    // lane flags: bit 0 = the lane owns a link, bit 1 = the parent is a link (not the origin), bit 2 = a child link exists,
    // bit 3 = the joint is revolute (else prismatic), bit 4 = the lane's instance exists, bit 5 = bits 0 and 4.  Kept as ONE
    // vector register and tested where needed (LINK_FLAGS_FRESH makes the compiler re-test instead of holding six 64-bit lane
    // masks in scalar registers for the whole kernel)
    int flags;
    HD bool on() const { return (flags & 1) != 0; }
    HD bool has_a() const { return (flags & 2) != 0; }
    HD bool has_c() const { return (flags & 4) != 0; }
    HD bool rev() const { return (flags & 8) != 0; }
    HD bool valid() const { return (flags & 16) != 0; }
    HD bool live() const { return (flags & 32) != 0; }
    static const int DEAD = 64, BAD = 128;     // set by the rollout loop (see there)
    static const int FRIC = 256;               // the own joint has viscous friction (friction/noise variant)
    HD bool has_fric() const { return (flags & FRIC) != 0; }
    static const int PRIM = 512;               // the lane is the PRIMARY one of its link's KL lanes (several lanes per link: below); always set when KL = 1
    HD bool prim() const { return (flags & PRIM) != 0; }
    HD bool dead() const { return (flags & DEAD) != 0; }
    HD bool bad() const { return (flags & BAD) != 0; }
    HD void set_valid(bool v) { flags |= v ? (16 | ((flags & 1) << 5)) : 0; }
};
#if defined(__HIP_DEVICE_COMPILE__)
#define LINK_FLAGS_FRESH(c) asm volatile("" : "+v"((c).flags))
#define LANE_INT_FRESH(x) asm volatile("" : "+v"(x))       // the same for any per-lane integer a predicate is formed from
#else
#define LINK_FLAGS_FRESH(c) ((void)0)
#define LANE_INT_FRESH(x) ((void)0)
#endif

HD void link_load_consts(LinkC& c, const MechDev* M, int t, int nb, double dt) {
    const bool on = t < nb;
    const int l = on ? t : 0;

    c.m = M->m[l];
#pragma unroll
    for (int i = 0; i < 9; i++) c.J[i] = M->J[l][i];
#pragma unroll
    for (int i = 0; i < 3; i++) { c.p1[i] = M->p1[l][i]; c.p2[i] = M->p2[l][i]; c.axis[i] = M->axis[l][i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) c.qoc[i] = M->qoc[l][i];
    const int type = M->type[l];              // 0 revolute, 1 prismatic
    const int vrow = type == 0 ? 3 : 0;       // where V1, V2 sit among the rows
#pragma unroll
    for (int i = 0; i < 3; i++) { c.V12[i] = M->sel[l][vrow][i]; c.V12[3 + i] = M->sel[l][vrow + 1][i]; }
    const int pa = M->parent[l];
    const bool has_a = on && pa >= 0;
    c.flags = (on ? 1 : 0) | (has_a ? 2 : 0) | ((on && M->childl[l] >= 0) ? 4 : 0) | (type == 0 ? 8 : 0);
    c.dtm = dt / c.m;
    c.sxb = dt * c.dtm;
    c.sxa = has_a ? dt * dt / M->m[pa >= 0 ? pa : 0] : 0.0;
    c.fric = 0.0;
}
// the link constants again (not the flags, not the friction coefficient): an experiment switch of rollout_chain.hip (CHAIN_RELOAD_CONSTS)
HD void link_reload_consts(LinkC& c, const MechDev* M, int t, int nb, double dt) {
    const bool on = t < nb;
    const int l = on ? t : 0;
    c.m = M->m[l];
#pragma unroll
    for (int i = 0; i < 9; i++) c.J[i] = M->J[l][i];
#pragma unroll
    for (int i = 0; i < 3; i++) { c.p1[i] = M->p1[l][i]; c.p2[i] = M->p2[l][i]; c.axis[i] = M->axis[l][i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) c.qoc[i] = M->qoc[l][i];
    const int vrow = c.rev() ? 3 : 0;
#pragma unroll
    for (int i = 0; i < 3; i++) { c.V12[i] = M->sel[l][vrow][i]; c.V12[3 + i] = M->sel[l][vrow + 1][i]; }
    const int pa = M->parent[l];
    c.dtm = dt / c.m;
    c.sxb = dt * c.dtm;
    c.sxa = c.has_a() ? dt * dt / M->m[pa >= 0 ? pa : 0] : 0.0;
}
// selector of constraint row `row` (compile-time row index)
HD void row_sel(const LinkC& c, int row, double* s) {
    const bool rev = c.rev();
    const int e = rev ? row : row - 2;          // unit vector index when the row is a unit row
    const bool unit = rev ? row < 3 : row >= 2;
    const int v = rev ? row - 3 : row;          // 0 / 1: V1 / V2 when it is not
#pragma unroll
    for (int i = 0; i < 3; i++) s[i] = unit ? (i == e ? 1.0 : 0.0) : c.V12[3 * (v & 1) + i];
}

// ---- joint: g and the sparse Jacobians.  Same formulas as joint_eval (cclqr_dev.h), outputs in sparse form:
// XT[3][3] (rows 0..2; rows 3, 4 are always rotational and have no x part; row 2 is zero for a prismatic joint),
// PB[5][3], PA[5][3].  Na / Nb: 3x3 applied to the rotational columns of the parent / child side (nullptr = identity).
template <bool JAC>
HD void joint_eval_sparse(const LinkC& c, const double* xa, const double* qa, const double* xb, const double* qb,
                          const double* Na, const double* Nb, double* g, double (*XT)[3], double (*PB)[3], double (*PA)[3]) {
    double Ra[9], Rb[9], rp[3], w[3], RaTw[3], gT[3];
    rotmat(qa, Ra); rotmat(qb, Rb);
    mv3(Rb, c.p2, rp);
#pragma unroll
    for (int i = 0; i < 3; i++) w[i] = xb[i] + rp[i] - xa[i];
    mtv3(Ra, w, RaTw);
#pragma unroll
    for (int i = 0; i < 3; i++) gT[i] = RaTw[i] - c.p1[i];
    double qac[4] = {qa[0], -qa[1], -qa[2], -qa[3]}, rel[4], e[4];
    qmul(qac, qb, rel);
    qmul(rel, c.qoc, e);
    const bool rot2 = !c.rev();     // row 2 is rotational for a prismatic joint
    if (!JAC) {
#pragma unroll
        for (int row = 0; row < 5; row++) {
            double s[3];
            row_sel(c, row, s);
            const double vT = s[0] * gT[0] + s[1] * gT[1] + s[2] * gT[2], vR = s[0] * e[1] + s[1] * e[2] + s[2] * e[3];
            g[row] = row < 2 ? vT : (row > 2 ? vR : (rot2 ? vR : vT));
        }
        return;
    }
    double RaTRb[9], PTb[9], PRb[9];
    mtm3(Ra, Rb, RaTRb);
    {
        const double* p = c.p2;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double a = RaTRb[i * 3], b = RaTRb[i * 3 + 1], cc = RaTRb[i * 3 + 2];
            PTb[i * 3 + 0] = -2.0 * (b * p[2] - cc * p[1]);
            PTb[i * 3 + 1] = -2.0 * (cc * p[0] - a * p[2]);
            PTb[i * 3 + 2] = -2.0 * (a * p[1] - b * p[0]);
        }
    }
    {
        const double s = rel[0], x = rel[1], y = rel[2], z = rel[3];
        const double os = c.qoc[0], ox = c.qoc[1], oy = c.qoc[2], oz = c.qoc[3];
        const double Lr[3][4] = {{x, s, -z, y}, {y, z, s, -x}, {z, -y, x, s}};
        const double Rc[4][3] = {{-ox, -oy, -oz}, {os, oz, -oy}, {-oz, os, ox}, {oy, -ox, os}};
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) PRb[i * 3 + j] = Lr[i][0] * Rc[0][j] + Lr[i][1] * Rc[1][j] + Lr[i][2] * Rc[2][j] + Lr[i][3] * Rc[3][j];
    }
    const double PTa[9] = {0, -2 * RaTw[2], 2 * RaTw[1], 2 * RaTw[2], 0, -2 * RaTw[0], -2 * RaTw[1], 2 * RaTw[0], 0};
    const double PRa[9] = {-e[0], -e[3], e[2], e[3], -e[0], -e[1], -e[2], e[1], -e[0]};
#pragma unroll
    for (int row = 0; row < 5; row++) {
        double s[3];
        row_sel(c, row, s);
        const double s0 = s[0], s1 = s[1], s2 = s[2];
        double pb3[3], pa3[3];
        if (row < 3) {   // translational, or (row 2) either kind
            const double vT = s0 * gT[0] + s1 * gT[1] + s2 * gT[2];
            double vR = 0.0;
            if (row == 2) vR = s0 * e[1] + s1 * e[2] + s2 * e[3];
            const bool rot = row == 2 && rot2;
            g[row] = rot ? vR : vT;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double xt = s0 * Ra[k * 3] + s1 * Ra[k * 3 + 1] + s2 * Ra[k * 3 + 2];
                XT[row][k] = rot ? 0.0 : xt;
                const double ptb = s0 * PTb[k] + s1 * PTb[3 + k] + s2 * PTb[6 + k];
                const double pta = s0 * PTa[k] + s1 * PTa[3 + k] + s2 * PTa[6 + k];
                if (row == 2) {
                    const double prb = s0 * PRb[k] + s1 * PRb[3 + k] + s2 * PRb[6 + k], pra = s0 * PRa[k] + s1 * PRa[3 + k] + s2 * PRa[6 + k];
                    pb3[k] = rot ? prb : ptb; pa3[k] = rot ? pra : pta;
                } else { pb3[k] = ptb; pa3[k] = pta; }
            }
        } else {         // rotational
            g[row] = s0 * e[1] + s1 * e[2] + s2 * e[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                pb3[k] = s0 * PRb[k] + s1 * PRb[3 + k] + s2 * PRb[6 + k];
                pa3[k] = s0 * PRa[k] + s1 * PRa[3 + k] + s2 * PRa[6 + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            PB[row][k] = Nb ? (pb3[0] * Nb[k] + pb3[1] * Nb[3 + k] + pb3[2] * Nb[6 + k]) : pb3[k];
            const double na = Na ? (pa3[0] * Na[k] + pa3[1] * Na[3 + k] + pa3[2] * Na[6 + k]) : pa3[k];
            PA[row][k] = c.has_a() ? na : 0.0;
        }
    }
}

// B' y for the child-side block (own[6]) and the parent-side block (par[6]) of a sparse Jacobian pair with unit x scales
// (the constraint-force map uses G_k, whose x scales are 1)
HD void jac_t_apply(const LinkC& c, const double (*XT)[3], const double (*PB)[3], const double (*PA)[3], const double* y, double* own, double* par) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double x = 0.0, pb = 0.0, pa = 0.0;
#pragma unroll
        for (int r = 0; r < 3; r++) x += XT[r][k] * y[r];
#pragma unroll
        for (int r = 0; r < 5; r++) { pb += PB[r][k] * y[r]; pa += PA[r][k] * y[r]; }
        own[k] = x; own[3 + k] = pb;
        par[k] = c.has_a() ? -x : 0.0; par[3 + k] = pa;
    }
}

// ---- control error of the owned body (order x, v, qtilde, w: lqr.jl:92-95; raw vector part, no sign fix: lqr.jl:101-102)
HD void ck_control_error(const double* z, const double* d, double* dz) {
    const double qdc[4] = {d[3], -d[4], -d[5], -d[6]};
    double qe[4];
    qmul(qdc, z + 3, qe);
#pragma unroll
    for (int i = 0; i < 3; i++) { dz[i] = z[i] - d[i]; dz[3 + i] = z[7 + i] - d[7 + i]; dz[6 + i] = qe[1 + i]; dz[9 + i] = z[10 + i] - d[10 + i]; }
}
// passive joint friction (trackingLQR_triple_cartpole.jl:93-101): -fric * relative joint velocity; za = parent state (13) or the origin's
HD double ck_friction(const LinkC& c, const double* z, const double* za) {
    double rel;
    if (c.rev()) {
        rel = c.axis[0] * z[10] + c.axis[1] * z[11] + c.axis[2] * z[12];
        if (c.has_a()) rel -= c.axis[0] * za[10] + c.axis[1] * za[11] + c.axis[2] * za[12];
    } else {
        double dv[3], dva[3], Ra[9];
#pragma unroll
        for (int i = 0; i < 3; i++) dv[i] = z[7 + i] - (c.has_a() ? za[7 + i] : 0.0);
        rotmat(za + 3, Ra);
        mtv3(Ra, dv, dva);
        rel = c.axis[0] * dva[0] + c.axis[1] * dva[1] + c.axis[2] * dva[2];
    }
    return -c.fric * rel;
}
// joint angle of a revolute from the relative quaternion: 2 atan2(axis . e_v, e_s).  Not inlined on the device: atan2's polynomial
// constants would otherwise sit in ~20 scalar registers for the whole persistent kernel
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline)) double pid_angle(double sn, double cs) { return 2.0 * atan2(sn, cs); }
#else
inline double pid_angle(double sn, double cs) { return 2.0 * atan2(sn, cs); }
#endif
// control_pid! for the owned joint (pid.jl:69-88); za = parent state or the origin's
HD double ck_pid(const LinkC& c, const double* z, const double* za, double P, double I, double D, double goal, double dt, bool first,
                 double& pid_int, double& pid_last) {
    double th;
    if (c.rev()) {
        const double qac[4] = {za[3], -za[4], -za[5], -za[6]};
        double rel[4], e[4];
        qmul(qac, z + 3, rel);
        qmul(rel, c.qoc, e);
        th = pid_angle(c.axis[0] * e[1] + c.axis[1] * e[2] + c.axis[2] * e[3], e[0]);
    } else {
        double Ra[9], Rb[9], rp[3], w[3], gT[3];
        rotmat(za + 3, Ra); rotmat(z + 3, Rb);
        mv3(Rb, c.p2, rp);
#pragma unroll
        for (int i = 0; i < 3; i++) w[i] = z[i] + rp[i] - za[i];
        mtv3(Ra, w, gT);
        th = c.axis[0] * (gT[0] - c.p1[0]) + c.axis[1] * (gT[1] - c.p1[1]) + c.axis[2] * (gT[2] - c.p1[2]);
    }
    const double PI = 3.14159265358979323846;
    double e = goal - th;
    if (c.rev()) { if (e > PI) e -= 2 * PI; else if (e < -PI) e += 2 * PI; }
    if (first) pid_last = e;
    pid_int += e * dt;
    const double de = (e - pid_last) / dt;
    pid_last = e;
    return P * e + I * pid_int + D * de;
}

// ---- joint input u of the own joint -> wrench on the own body (F world, tau body frame) and the reaction on the parent
// body (Fp world, taup in the parent's frame), SURVEY 8a-bis 'Joint input'.  qa = parent orientation (identity for the origin).
HD void ck_joint_wrench(const LinkC& c, double u, const double* q, const double* qa, double* F, double* tau, double* Fp, double* taup) {
    double Ra[9], Rb[9];
    rotmat(qa, Ra); rotmat(q, Rb);
    const double f[3] = {c.axis[0] * u, c.axis[1] * u, c.axis[2] * u};
    double fw[3], fb[3];
    mv3(Ra, f, fw); mtv3(Rb, fw, fb);
    if (!c.rev()) {
        double cb[3], cp[3];
        cross3(c.p2, fb, cb); cross3(c.p1, f, cp);
#pragma unroll
        for (int i = 0; i < 3; i++) { F[i] = fw[i]; tau[i] = cb[i]; Fp[i] = -fw[i]; taup[i] = -cp[i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) { F[i] = 0.0; tau[i] = fb[i]; Fp[i] = 0.0; taup[i] = -f[i]; }
    }
}
// per-step invariants of the owned body: cT = m(-v/dt + ez g') - F ; cR = -(sq1 I - [w1]x) J w1 - 2 tau
HD void ck_step_invariants(const LinkC& c, const double* z, const double* F, const double* tau, double dt, double g, double* cT, double* cR) {
    const double* v1 = z + 7; const double* w1 = z + 10;
    const double sq1 = sqrt(4.0 / (dt * dt) - (w1[0] * w1[0] + w1[1] * w1[1] + w1[2] * w1[2]));
    double Jw1[3], c1[3];
    mv3(c.J, w1, Jw1); cross3(w1, Jw1, c1);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        cT[i] = c.m * (-v1[i] / dt + (i == 2 ? -g : 0.0)) - F[i];
        cR[i] = -(sq1 * Jw1[i] - c1[i]) - 2.0 * tau[i];
    }
}

// ---- body at the trial solution s: next pose xq, residual d = dyn(s) - cf, with JAC also D_R^-1 and N D_R^-1; returns |d|^2
template <bool JAC>
HD double ck_body_eval(const LinkC& c, const double* z, const double* s, const double* cf, const double* cT, const double* cR, double dt,
                       double* xq, double* d, double* DINV, double* NB) {
    const double* w2 = s + 3;
#pragma unroll
    for (int i = 0; i < 3; i++) xq[i] = z[i] + s[i] * dt;
    const double inv_dt = fast_rcp(dt), m_dt = c.m * inv_dt;
    const double sq2 = sqrt(4.0 * inv_dt * inv_dt - (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2]));
    const double wb[4] = {0.5 * dt * sq2, 0.5 * dt * w2[0], 0.5 * dt * w2[1], 0.5 * dt * w2[2]};
    qmul(z + 3, wb, xq + 3);
    double Jw2[3], c2[3];
    mv3(c.J, w2, Jw2); cross3(w2, Jw2, c2);
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double dT = m_dt * s[i] + cT[i] - cf[i], dR = sq2 * Jw2[i] + c2[i] + cR[i] - cf[3 + i];
        d[i] = dT; d[3 + i] = dR;
        acc += dT * dT + dR * dR;
    }
    if (!JAC) return acc;
    const double S[9] = {sq2, -w2[2], w2[1], w2[2], sq2, -w2[0], -w2[1], w2[0], sq2};
    double SJ[9], Dr[9], N[9];
    mm3(S, c.J, SJ);
    const double isq = fast_rcp(sq2);
    const double Sj[9] = {0, -Jw2[2], Jw2[1], Jw2[2], 0, -Jw2[0], -Jw2[1], Jw2[0], 0};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Dr[i * 3 + j] = SJ[i * 3 + j] - Sj[i * 3 + j] - Jw2[i] * w2[j] * isq;
    inv3(Dr, DINV);
    make_N(w2, sq2, dt, N);
    mm3(N, DINV, NB);
    return acc;
}

// LDS slot of a joint's G_k (39 doubles): rows 0..2 = XT[3] PB[3] PA[3], rows 3, 4 = PB[3] PA[3]
#define GKSZ 39
HD int gk_row(int q) { return q < 3 ? 9 * q : 27 + 6 * (q - 3); }

// next pose of the owned body for the solution s = (v+, w+): x+ = x + v+ dt ; q+ = q (dt/2)(sqrt(4/dt^2 - w+'w+), w+)  -- the same
// expressions as in ck_body_eval, so the recomputed pose equals the one the residual was evaluated at
HD void ck_next_pose(const double* z, const double* s, double dt, double* xq) {
    const double* w2 = s + 3;
#pragma unroll
    for (int i = 0; i < 3; i++) xq[i] = z[i] + s[i] * dt;
    const double inv_dt = fast_rcp(dt);
    const double sq2 = sqrt(4.0 * inv_dt * inv_dt - (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2]));
    const double wb[4] = {0.5 * dt * sq2, 0.5 * dt * w2[0], 0.5 * dt * w2[1], 0.5 * dt * w2[2]};
    qmul(z + 3, wb, xq + 3);
}

// ---- Schur complement blocks built by lane j (the owner of link j) straight into LDS, from its own W = G_v D^-1 =
// (wXT, wPB, wPA; x scales sxb, sxa), which was just evaluated and lives in registers, and the G_k rows of its own, its
// parent's and its child's joint, read from their LDS slots:
//   S_jj = W_b[j] Gk_b[j]' + W_a[j] Gk_a[j]'   -> SJJ[j]
//   S_jp = W_a[j] Gk_b[p]'                      -> SJP[j]      (p = j-1, when has_a)
//   S_jc = W_b[j] Gk_a[c]'                      -> SPJ[c]      (c = j+1, when has_c: "S_{parent,child}" of the child's slot)
//   r_j  = g_j - W_b d_b - W_a d_a              -> R[j]        (pd = residual of the parent body, from the parent lane)
// One column q of the three blocks at a time.
// the G_k operands of column q of the three blocks (own, parent's and child's joint): 21 doubles (12 for the rotational columns q = 3, 4)
struct SchurCol { double kx[3], kpx[3], kcx[3], kb[3], ka[3], kpb[3], kca[3]; };
HD void schur_col_load(SchurCol& K, int q, int j, int jp, int jc, const Lay& Y, const double* L) {
    const int o = gk_row(q), ob = q < 3 ? 3 : 0;   // offset of PB inside the row
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (q < 3) { K.kx[i] = L[Y.GKA + GKSZ * j + o + i]; K.kpx[i] = L[Y.GKA + GKSZ * jp + o + i]; K.kcx[i] = L[Y.GKA + GKSZ * jc + o + i]; }
        else { K.kx[i] = 0.0; K.kpx[i] = 0.0; K.kcx[i] = 0.0; }
        K.kb[i] = L[Y.GKA + GKSZ * j + o + ob + i]; K.ka[i] = L[Y.GKA + GKSZ * j + o + ob + 3 + i];
        K.kpb[i] = L[Y.GKA + GKSZ * jp + o + ob + i]; K.kca[i] = L[Y.GKA + GKSZ * jc + o + ob + 3 + i];
    }
}
// PF: the operands of column q + 1 are requested before column q is computed (two operand sets live: + 42 registers, which the 8- and 16-lane instantiations
// have and the 17-link one does not -- there it was measured slower, DESIGN_HISTORY 9b).  The phase is a chain of five load -> compute -> store stages at one
// wavefront per SIMD; without the prefetch every stage pays its LDS round trip in full.  Same arithmetic either way.
template <bool PF = false>
HD void ck_schur_rows(const LinkC& c, int j, bool store, const Lay& Y, double* L, const double (*wXT)[3], const double (*wPB)[3],
                      const double (*wPA)[3], const double* g, const double* d, const double* pd) {
    const double sx = c.sxb + c.sxa;
    const int jp = c.has_a() ? j - 1 : j, jc = c.has_c() ? j + 1 : j;
    SchurCol KA, KB;
    if (PF) schur_col_load(KA, 0, j, jp, jc, Y, L);
#pragma unroll
    for (int q = 0; q < 5; q++) {
        SchurCol& K = PF ? ((q & 1) ? KB : KA) : KA;
        if (!PF) schur_col_load(K, q, j, jp, jc, Y, L);
        else if (q < 4) { schur_col_load((q & 1) ? KA : KB, q + 1, j, jp, jc, Y, L); SCHED_FENCE(); }
        const double *kx = K.kx, *kpx = K.kpx, *kcx = K.kcx, *kb = K.kb, *ka = K.ka, *kpb = K.kpb, *kca = K.kca;
        double ojj[5], ojp[5], ojc[5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const double bb = wPB[r][0] * kb[0] + wPB[r][1] * kb[1] + wPB[r][2] * kb[2];
            const double aa = wPA[r][0] * ka[0] + wPA[r][1] * ka[1] + wPA[r][2] * ka[2];
            const double ajp = wPA[r][0] * kpb[0] + wPA[r][1] * kpb[1] + wPA[r][2] * kpb[2];
            const double bjc = wPB[r][0] * kca[0] + wPB[r][1] * kca[1] + wPB[r][2] * kca[2];
            if (r < 3 && q < 3) {
                const double xx = wXT[r][0] * kx[0] + wXT[r][1] * kx[1] + wXT[r][2] * kx[2];
                const double xjp = wXT[r][0] * kpx[0] + wXT[r][1] * kpx[1] + wXT[r][2] * kpx[2];
                const double xjc = wXT[r][0] * kcx[0] + wXT[r][1] * kcx[1] + wXT[r][2] * kcx[2];
                ojj[r] = sx * xx + bb + aa; ojp[r] = ajp - c.sxa * xjp; ojc[r] = bjc - c.sxb * xjc;
            } else { ojj[r] = bb + aa; ojp[r] = ajp; ojc[r] = bjc; }
        }
        if (store) {     // blocks are stored column by column (element (r, q) at 5 q + r): a column is five consecutive doubles
#pragma unroll
            for (int r = 0; r < 5; r++) L[Y.SJJ + 25 * j + 5 * q + r] = ojj[r];
            if (c.has_a()) {
#pragma unroll
                for (int r = 0; r < 5; r++) L[Y.SJP + 25 * j + 5 * q + r] = ojp[r];
            }
            if (c.has_c()) {
#pragma unroll
                for (int r = 0; r < 5; r++) L[Y.SPJ + 25 * jc + 5 * q + r] = ojc[r];
            }
        }
        SCHED_FENCE();     // keep the next column's loads behind this column's arithmetic: hoisting all five columns' loads costs ~90 registers
    }
    if (store) {
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const double bd = wPB[r][0] * d[3] + wPB[r][1] * d[4] + wPB[r][2] * d[5];
            const double ad = wPA[r][0] * pd[3] + wPA[r][1] * pd[4] + wPA[r][2] * pd[5];
            double rr = g[r] - bd - ad;
            if (r < 3) {
                const double xd = wXT[r][0] * d[0] + wXT[r][1] * d[1] + wXT[r][2] * d[2];
                const double xa = wXT[r][0] * pd[0] + wXT[r][1] * pd[1] + wXT[r][2] * pd[2];
                rr = g[r] - (c.sxb * xd + bd) - (ad - c.sxa * xa);
            }
            L[Y.R + 5 * j + r] = rr;
        }
    }
}

// ================================================================== SEVERAL LANES PER LINK (round 5)
// A mechanism that leaves lanes of its lane group idle (2 links in 8 lanes, 4 in 8, 7 in 16, 16 in 32) gives every link KL = 3 or 2 lanes at
// UNCHANGED occupancy -- the same instances per wavefront, the same LDS image -- and deals the row-separable work of an evaluation with Jacobians
// to them: the five rows of the joint's (g, G_v D^-1) and the Schur rows / right-hand side built from them.  Lane = w NL + l (sub-lane w of
// link l, NL links per sub-lane group), so that the parent link still is one lane below and every neighbour vector moves by the same wave shifts.
// Everything that is not row-separable -- the body evaluation, rotation matrices, relative quaternion, the 3 x 3 products -- is computed by all
// lanes of a link (sharing it would cost ~40 doubles of exchange per evaluation), so no exchange between sub-lanes is needed at all; only the
// PRIMARY sub-lane (w = 0) writes the link's private LDS slots and contributes the body's residual to the norm.
// Row slots of a lane (uniform code, per-lane selectors -- rows 0, 1 are translational-kind, 3, 4 rotational-kind, row 2 either, by joint type):
//   KL = 3:  w < 2: slot A = row w (translational kind), slot B = row 3 + w (rotational kind);  w = 2: row 2, computed in BOTH kinds through the two
//            slots' code and merged into slot A (slot B then is empty)                                              -> 2 row units per lane (5 rows: 6)
//   KL = 2:  slots A, B as above, and slot C = row 2 in both kinds on sub-lane 0 (empty on sub-lane 1)                -> 4 row units per lane
// Measured on the 17-body chain (where it does NOT pay, because there the split costs occupancy: profiles/r05/stageA_*): same norms and Schur blocks
// to 4e-16 relative, 941 instead of 1563 vector instructions per evaluation and wavefront.
template <int KL> struct SubRows { static const int NR = KL == 3 ? 2 : 3; };
struct SubSel {
    double sa[3], sb[3], s2[3];     // selectors of slot A (translational kind), slot B (rotational kind), row 2
    int w;                          // sub-lane
    int ia, ib, ic;                 // row numbers of the slots inside a 5-row block (-1: the slot is empty on this lane)
};
// selector of constraint row `row` of the owned joint, row a RUN-TIME number (row_sel above wants a compile-time one)
HD void row_sel_rt(const LinkC& c, int row, double* s) {
    const bool rev = c.rev();
    const bool unit = rev ? row < 3 : row >= 2;
    const int e = rev ? row : row - 2, v = rev ? row - 3 : row;
#pragma unroll
    for (int i = 0; i < 3; i++) s[i] = unit ? (i == e ? 1.0 : 0.0) : c.V12[3 * (v & 1) + i];
}
template <int KL>
HD void sub_setup(const LinkC& c, int w, SubSel& Q) {
    Q.w = w;
    const bool mid = KL == 3 && w == 2;                  // the lane of row 2 (KL = 3)
    row_sel_rt(c, 2, Q.s2);
    row_sel_rt(c, mid ? 2 : (w & 1), Q.sa);
    row_sel_rt(c, mid ? 2 : 3 + (w & 1), Q.sb);
    Q.ia = mid ? 2 : w;
    Q.ib = mid ? -1 : 3 + w;
    Q.ic = (KL == 2 && w == 0) ? 2 : -1;
}
// the lane's rows of the joint's g and of W = G_v D^-1 in sparse form (slot s: g[s], XT[s] -- zero for a rotational row --, PB[s], PA[s]); same
// formulas as joint_eval_sparse<true>.  Na / Nb: N D_R^-1 of the parent / own body.
template <int KL>
HD void joint_eval_rows(const LinkC& c, const SubSel& Q, const double* xa, const double* qa, const double* xb, const double* qb, const double* Na, const double* Nb,
                        double* g, double (*XT)[3], double (*PB)[3], double (*PA)[3]) {
    constexpr int NR = SubRows<KL>::NR;
    double Ra[9], Rb[9], rp[3], wv[3], RaTw[3], gT[3];
    rotmat(qa, Ra); rotmat(qb, Rb);
    mv3(Rb, c.p2, rp);
#pragma unroll
    for (int i = 0; i < 3; i++) wv[i] = xb[i] + rp[i] - xa[i];
    mtv3(Ra, wv, RaTw);
#pragma unroll
    for (int i = 0; i < 3; i++) gT[i] = RaTw[i] - c.p1[i];
    double qac[4] = {qa[0], -qa[1], -qa[2], -qa[3]}, rel[4], e[4];
    qmul(qac, qb, rel);
    qmul(rel, c.qoc, e);
    double RaTRb[9], PTb[9], PRb[9];
    mtm3(Ra, Rb, RaTRb);
    {
        const double* p = c.p2;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double a = RaTRb[i * 3], b = RaTRb[i * 3 + 1], cc = RaTRb[i * 3 + 2];
            PTb[i * 3 + 0] = -2.0 * (b * p[2] - cc * p[1]);
            PTb[i * 3 + 1] = -2.0 * (cc * p[0] - a * p[2]);
            PTb[i * 3 + 2] = -2.0 * (a * p[1] - b * p[0]);
        }
    }
    {
        const double s = rel[0], x = rel[1], y = rel[2], z = rel[3];
        const double os = c.qoc[0], ox = c.qoc[1], oy = c.qoc[2], oz = c.qoc[3];
        const double Lr[3][4] = {{x, s, -z, y}, {y, z, s, -x}, {z, -y, x, s}};
        const double Rc[4][3] = {{-ox, -oy, -oz}, {os, oz, -oy}, {-oz, os, ox}, {oy, -ox, os}};
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) PRb[i * 3 + j] = Lr[i][0] * Rc[0][j] + Lr[i][1] * Rc[1][j] + Lr[i][2] * Rc[2][j] + Lr[i][3] * Rc[3][j];
    }
    const double PTa[9] = {0, -2 * RaTw[2], 2 * RaTw[1], 2 * RaTw[2], 0, -2 * RaTw[0], -2 * RaTw[1], 2 * RaTw[0], 0};
    const double PRa[9] = {-e[0], -e[3], e[2], e[3], -e[0], -e[1], -e[2], e[1], -e[0]};
    const bool rot2 = !c.rev();                          // row 2 is rotational for a prismatic joint
    double pb3[NR][3], pa3[NR][3];
    {   // slot A as a translational row, slot B as a rotational row
        const double a0 = Q.sa[0], a1 = Q.sa[1], a2 = Q.sa[2], b0 = Q.sb[0], b1 = Q.sb[1], b2 = Q.sb[2];
        const double vT = a0 * gT[0] + a1 * gT[1] + a2 * gT[2], vR = b0 * e[1] + b1 * e[2] + b2 * e[3];
        int wq = Q.w;
        LANE_INT_FRESH(wq);                               // (the predicate is formed here, per evaluation -- not once per launch and kept as a lane mask)
        const bool merge = KL == 3 && wq == 2;           // the lane of row 2: slot A keeps the kind the joint type selects, slot B is empty
        const bool rot = merge && rot2;
        g[0] = rot ? vR : vT;
        g[1] = merge ? 0.0 : vR;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double xt = a0 * Ra[k * 3] + a1 * Ra[k * 3 + 1] + a2 * Ra[k * 3 + 2];
            const double ptb = a0 * PTb[k] + a1 * PTb[3 + k] + a2 * PTb[6 + k], pta = a0 * PTa[k] + a1 * PTa[3 + k] + a2 * PTa[6 + k];
            const double prb = b0 * PRb[k] + b1 * PRb[3 + k] + b2 * PRb[6 + k], pra = b0 * PRa[k] + b1 * PRa[3 + k] + b2 * PRa[6 + k];
            XT[0][k] = rot ? 0.0 : xt;
            pb3[0][k] = rot ? prb : ptb; pa3[0][k] = rot ? pra : pta;
            XT[1][k] = 0.0;
            pb3[1][k] = merge ? 0.0 : prb; pa3[1][k] = merge ? 0.0 : pra;
        }
    }
    if (KL == 2) {   // slot C: row 2 in both kinds (sub-lane 0; zeroed elsewhere)
        int icq = Q.ic;
        LANE_INT_FRESH(icq);
        const bool have = icq >= 0;
        const double s0 = Q.s2[0], s1 = Q.s2[1], s2 = Q.s2[2];
        const double vT = s0 * gT[0] + s1 * gT[1] + s2 * gT[2], vR = s0 * e[1] + s1 * e[2] + s2 * e[3];
        g[NR - 1] = have ? (rot2 ? vR : vT) : 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double xt = s0 * Ra[k * 3] + s1 * Ra[k * 3 + 1] + s2 * Ra[k * 3 + 2];
            const double ptb = s0 * PTb[k] + s1 * PTb[3 + k] + s2 * PTb[6 + k], pta = s0 * PTa[k] + s1 * PTa[3 + k] + s2 * PTa[6 + k];
            const double prb = s0 * PRb[k] + s1 * PRb[3 + k] + s2 * PRb[6 + k], pra = s0 * PRa[k] + s1 * PRa[3 + k] + s2 * PRa[6 + k];
            XT[NR - 1][k] = (have && !rot2) ? xt : 0.0;
            pb3[NR - 1][k] = have ? (rot2 ? prb : ptb) : 0.0; pa3[NR - 1][k] = have ? (rot2 ? pra : pta) : 0.0;
        }
    }
#pragma unroll
    for (int s = 0; s < NR; s++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            PB[s][k] = pb3[s][0] * Nb[k] + pb3[s][1] * Nb[3 + k] + pb3[s][2] * Nb[6 + k];
            const double na = pa3[s][0] * Na[k] + pa3[s][1] * Na[3 + k] + pa3[s][2] * Na[6 + k];
            PA[s][k] = c.has_a() ? na : 0.0;
        }
}
// the lane's rows of S_jj, S_jp, S_jc and r_j (ck_schur_rows for the rows of its slots; element (r, q) of a block at 5 q + r)
template <int KL, bool PF = false>
HD void ck_schur_rows_sub(const LinkC& c, const SubSel& Q, int j, bool store, const Lay& Y, double* L, const double* g, const double (*XT)[3], const double (*PB)[3],
                          const double (*PA)[3], const double* d, const double* pd) {
    constexpr int NR = SubRows<KL>::NR;
    const double sx = c.sxb + c.sxa;
    const int jp = c.has_a() ? j - 1 : j, jc = c.has_c() ? j + 1 : j;
    int idx[3] = {Q.ia, Q.ib, Q.ic};
    SchurCol KA, KB;
    if (PF) schur_col_load(KA, 0, j, jp, jc, Y, L);
#pragma unroll
    for (int q = 0; q < 5; q++) {
        // (the "slot is empty" tests are made per column, where the stores are: hoisted out of the loop they are three 64-bit lane masks in scalar registers)
        LANE_INT_FRESH(idx[0]); LANE_INT_FRESH(idx[1]); LANE_INT_FRESH(idx[2]);
        SchurCol& K = PF ? ((q & 1) ? KB : KA) : KA;
        if (!PF) schur_col_load(K, q, j, jp, jc, Y, L);
        else if (q < 4) { schur_col_load((q & 1) ? KA : KB, q + 1, j, jp, jc, Y, L); SCHED_FENCE(); }
        const double *kx = K.kx, *kpx = K.kpx, *kcx = K.kcx, *kb = K.kb, *ka = K.ka, *kpb = K.kpb, *kca = K.kca;
#pragma unroll
        for (int s = 0; s < NR; s++) {
            const bool xpart = q < 3 && s != 1;          // slot B is a rotational row: no x part
            double ojj = PB[s][0] * kb[0] + PB[s][1] * kb[1] + PB[s][2] * kb[2] + (PA[s][0] * ka[0] + PA[s][1] * ka[1] + PA[s][2] * ka[2]);
            double ojp = PA[s][0] * kpb[0] + PA[s][1] * kpb[1] + PA[s][2] * kpb[2];
            double ojc = PB[s][0] * kca[0] + PB[s][1] * kca[1] + PB[s][2] * kca[2];
            if (xpart) {
                ojj += sx * (XT[s][0] * kx[0] + XT[s][1] * kx[1] + XT[s][2] * kx[2]);
                ojp -= c.sxa * (XT[s][0] * kpx[0] + XT[s][1] * kpx[1] + XT[s][2] * kpx[2]);
                ojc -= c.sxb * (XT[s][0] * kcx[0] + XT[s][1] * kcx[1] + XT[s][2] * kcx[2]);
            }
            if (store && idx[s] >= 0) {
                L[Y.SJJ + 25 * j + 5 * q + idx[s]] = ojj;
                if (c.has_a()) L[Y.SJP + 25 * j + 5 * q + idx[s]] = ojp;
                if (c.has_c()) L[Y.SPJ + 25 * jc + 5 * q + idx[s]] = ojc;
            }
        }
        SCHED_FENCE();      // (as in ck_schur_rows; measured without it: same registers, same time)
    }
    if (store) {
#pragma unroll
        for (int s = 0; s < NR; s++) {
            const double bd = PB[s][0] * d[3] + PB[s][1] * d[4] + PB[s][2] * d[5], ad = PA[s][0] * pd[3] + PA[s][1] * pd[4] + PA[s][2] * pd[5];
            double rr = g[s] - bd - ad;
            if (s != 1) {
                const double xd = XT[s][0] * d[0] + XT[s][1] * d[1] + XT[s][2] * d[2], xa = XT[s][0] * pd[0] + XT[s][1] * pd[1] + XT[s][2] * pd[2];
                rr = g[s] - (c.sxb * xd + bd) - (ad - c.sxa * xa);
            }
            if (idx[s] >= 0) L[Y.R + 5 * j + idx[s]] = rr;
        }
    }
}

// G_k of the owned joint into its LDS slot / B' y against the slot's rows (own[6] child side, par[6] parent side)
HD void gk_store(int j, const Lay& Y, double* L, const double (*XT)[3], const double (*PB)[3], const double (*PA)[3]) {
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int o = Y.GKA + GKSZ * j + gk_row(q), ob = q < 3 ? 3 : 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (q < 3) L[o + i] = XT[q][i];
            L[o + ob + i] = PB[q][i]; L[o + ob + 3 + i] = PA[q][i];
        }
    }
}
HD void gk_t_apply(const LinkC& c, int j, const Lay& Y, const double* L, const double* y, double* own, double* par) {
    double x[3] = {0, 0, 0}, pb[3] = {0, 0, 0}, pa[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const int o = Y.GKA + GKSZ * j + gk_row(r), ob = r < 3 ? 3 : 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (r < 3) x[i] += L[o + i] * y[r];
            pb[i] += L[o + ob + i] * y[r]; pa[i] += L[o + ob + 3 + i] * y[r];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { own[i] = x[i]; own[3 + i] = pb[i]; par[i] = c.has_a() ? -x[i] : 0.0; par[3 + i] = pa[i]; }
}

// ---- body solve: ds = D^-1 (d + cd) with cd = Gk_b(own joint)' dl + Gk_a(child joint)' dl_child (the latter arrives from the child lane)
HD void ck_body_solve(const LinkC& c, const double* d, const double* cd, const double* DINV, double* ds) {
    double tv[6];
#pragma unroll
    for (int k = 0; k < 6; k++) tv[k] = d[k] + cd[k];
#pragma unroll
    for (int k = 0; k < 3; k++) { ds[k] = tv[k] * c.dtm; ds[3 + k] = DINV[3 * k] * tv[3] + DINV[3 * k + 1] * tv[4] + DINV[3 * k + 2] * tv[5]; }
}

// ================================================================== block-tridiagonal solve (chain kernel version)
// Same twisted two-front elimination as cclqr_dev.h S3 (front 0 = lanes 0..5 sweeps from the leaf, front 1 = lanes 8..13 from
// the root; lane (t & 7) = c < 5 owns column c of Z = S_ll^-1 S_lq and of the update of S_qq, lane 5 owns y = S_ll^-1 r_l and
// the update of r_q), re-cut for latency, since one wavefront per SIMD has nothing to hide it with:
//  * blocks are column-major, so every lane's right-hand side, target and result are five consecutive doubles;
//  * every offset is a per-lane cursor advanced by a constant per step (no index arithmetic in the loop);
//  * loads are issued in the order of need (S_ll, right-hand side, target, S_ql), so the pivot chain of the factorisation
//    starts after the first 25 and hides the rest (carrying prefetched blocks across the barrier in registers instead made
//    the register allocator spill all over the kernel);
// Plan of the chain kernel: like tri_plan, but with BALANCED fronts when the chain has an odd number of links (17: 8 + 8 steps
// instead of 7 + 9).  Both fronts then fold into the middle link in the same (last) step; front 1 writes its contribution
// -S_ql Z, -S_ql y to a scratch block instead of the middle link's own blocks (`merge`), and the middle solve adds it.  The
// scratch is the (SJP, SPJ) pair of the chain's second link, which front 1 consumed in its first step.
// The swept chain may be a REDUCED one (every second link of the original, after cr_level below): its links are cs + st i,
// i < cn, and the coupling blocks of neighbours x < x' = x + st sit where the level below left them: S_{x',x} in SJP[x']
// ("lower" block of x'), S_{x,x'} in SPJ[x + 1] ("upper" block of x) -- for st = 1 the plain layout.
struct TriPlanB { TriPlan P; int merge, st; };      // P.cs, P.mid are links; P.cn, P.nA, P.nB, P.steps count links of the swept chain
// fronts = 1: the 8-lane instantiation (mechanisms of up to 4 links, eight instances per wavefront) has room for one front only,
// which sweeps from the leaf all the way to the chain's first link.
HD TriPlanB tri_plan_balanced(int cs, int cn, int st = 1, int fronts = 2) {
    TriPlanB B;
    B.P = tri_plan(0, cn);
    const int rest = cn - 1;
    B.merge = (fronts == 2 && rest > 0 && rest % 2 == 0) ? 1 : 0;
    if (B.merge) { B.P.nA = B.P.nB = rest / 2; B.P.steps = rest / 2; }
    if (fronts == 1) { B.P.nB = 0; B.P.nA = rest; B.P.steps = rest; }
    B.P.cs = cs; B.P.mid = cs + st * B.P.nB; B.st = st;
    return B;
}
HD int tri_scratch_S(const TriPlanB& B, const Lay& Y) { return Y.SJP + 25 * (B.P.cs + B.st); }
HD int tri_scratch_R(const TriPlanB& B, const Lay& Y) { return Y.SPJ + 25 * (B.P.cs + 1); }
struct TriCur {
    int oLL, oQL, oRhs, oTgt, oOut;   // LDS offsets at the current step: S_ll, S_ql, this lane's right-hand side (column c of S_lq, or
                                      // r_l), its target (column c of S_qq, or r_q) and where its solution goes (column c of S_ll, or r_l)
    int dblk, dvec;                   // per-step increments of the block offsets / of this lane's vector offsets
    int n;                            // steps of this lane's front (0: the lane takes no part)
    int imerge, oScr;                 // front 1 of a balanced plan: in step imerge the target is the scratch block (-1: never)
    bool isy;
};
HD TriCur tri_cursor(int t, const TriPlanB& B, const Lay& Y) {
    const TriPlan& P = B.P;
    TriCur K;
    const int front = t >> 3, col = t & 7;
    const bool in = t < 16 && col < 6;
    K.n = in ? (front ? P.nB : P.nA) : 0;
    K.isy = col == 5;
    const int st = B.st;
    const int l = front ? P.cs : P.cs + st * (P.cn - 1), q = front ? l + st : l - st;
    const int cc = col < 5 ? col : 0;
    K.dblk = front ? 25 * st : -25 * st;
    K.dvec = K.isy ? (front ? 5 * st : -5 * st) : K.dblk;
    K.oLL = Y.SJJ + 25 * l;
    K.oQL = front ? Y.SJP + 25 * q : Y.SPJ + 25 * (q + 1);                                  // S_ql: lower block of q / upper block of q
    K.oRhs = K.isy ? Y.R + 5 * l : (front ? Y.SPJ + 25 * (l + 1) : Y.SJP + 25 * l) + 5 * cc;   // S_lq: upper / lower block of l
    K.oTgt = K.isy ? Y.R + 5 * q : Y.SJJ + 25 * q + 5 * cc;
    K.oOut = K.isy ? Y.R + 5 * l : Y.SJJ + 25 * l + 5 * cc;
    K.imerge = (front && B.merge) ? P.nB - 1 : -1;
    K.oScr = K.isy ? tri_scratch_R(B, Y) : tri_scratch_S(B, Y) + 5 * cc;
    return K;
}
// LU (no pivoting; S is SPD-like) of the row-major 5x5 block A in registers, packed as in lu5 (cclqr_dev.h)
HD void lu5_factor(double* A) {
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const double inv = fast_rcp(A[k * 5 + k]);
        A[k * 5 + k] = inv;
#pragma unroll
        for (int i = k + 1; i < 5; i++) {
            const double f = A[i * 5 + k] * inv;
            A[i * 5 + k] = f;
#pragma unroll
            for (int j = k + 1; j < 5; j++) A[i * 5 + j] -= f * A[k * 5 + j];
        }
    }
}
// One elimination step of this lane: tg = updated target column, zy = solution column, to be stored by the caller at *otg,
// *oout AFTER every lane has done its loads (the caller's store sits behind this call in the wavefront's instruction stream).
// Load order = order of need: S_ll (the LU starts as soon as it has arrived), the right-hand side, the target, and S_ql last
// (only the final update reads it), so that most of the loads' latency hides under the pivot chain of the factorisation.
// Returns false (and touches nothing) when the lane has no work in step i.
HD bool tri_step(TriCur& K, int i, const double* L, double* tg, double* zy, int* otg, int* oout) {
    if (i >= K.n) return false;
    double lu[25], sql[25];
#pragma unroll
    for (int cI = 0; cI < 5; cI++)
#pragma unroll
        for (int r = 0; r < 5; r++) lu[r * 5 + cI] = L[K.oLL + 5 * cI + r];
#pragma unroll
    for (int r = 0; r < 5; r++) zy[r] = L[K.oRhs + r];
    const bool mstep = i == K.imerge;          // both fronts fold into the middle link now: this one's share goes to the scratch block
    // (loaded unconditionally and then selected: a load inside the conditional becomes five separately EXEC-masked reads)
    double tl[5];
#pragma unroll
    for (int r = 0; r < 5; r++) tl[r] = L[K.oTgt + r];
#pragma unroll
    for (int r = 0; r < 5; r++) tg[r] = mstep ? 0.0 : tl[r];
#pragma unroll
    for (int e = 0; e < 25; e++) sql[e] = L[K.oQL + e];
    lu5_factor(lu);
    lu5_solve(lu, zy);
#pragma unroll
    for (int r = 0; r < 5; r++) tg[r] -= sql[r] * zy[0] + sql[5 + r] * zy[1] + sql[10 + r] * zy[2] + sql[15 + r] * zy[3] + sql[20 + r] * zy[4];
    *otg = mstep ? K.oScr : K.oTgt; *oout = K.oOut;
    K.oLL += K.dblk; K.oQL += K.dblk; K.oRhs += K.dvec; K.oTgt += K.dvec; K.oOut += K.dvec;
    return true;
}
HD void tri_step_store(double* L, int otg, int oout, const double* tg, const double* zy) {
#pragma unroll
    for (int r = 0; r < 5; r++) { L[otg + r] = tg[r]; L[oout + r] = zy[r]; }
}
// middle link: both sides have been folded in (a balanced plan: front 1's share is added from the scratch block here); one lane
// factorises and solves
HD void ck_tri_mid(int t, const TriPlanB& B, const Lay& Y, double* L) {
    if (t != 0) return;
    const TriPlan& P = B.P;
    double A[25], b[5];
#pragma unroll
    for (int i = 0; i < 5; i++) b[i] = L[Y.R + 5 * P.mid + i] + (B.merge ? L[tri_scratch_R(B, Y) + i] : 0.0);
#pragma unroll
    for (int cI = 0; cI < 5; cI++)
#pragma unroll
        for (int r = 0; r < 5; r++) A[r * 5 + cI] = L[Y.SJJ + 25 * P.mid + 5 * cI + r] + (B.merge ? L[tri_scratch_S(B, Y) + 5 * cI + r] : 0.0);
    lu5_factor(A);
    lu5_solve(A, b);
#pragma unroll
    for (int i = 0; i < 5; i++) L[Y.DL + 5 * P.mid + i] = b[i];
}
// back substitution step j: dl_l = y_l - Z_l dl_nbr; front 0 (lanes 0..4, one row each) l = mid+1+j, nbr = l-1; front 1 (lanes
// 8..12) l = mid-1-j, nbr = l+1.  Measured alternatives, all slower than these 4.7 k cycles per Newton iteration at 17 links: a
// one-lane-per-front sweep with the next link's Z prefetched (6.2 k), and keeping dl in the five lanes' registers with a DPP
// rotation instead of the LDS round trip (5.7 k branch-free, 8.5 k as the compiler first laid it out).
HD void ck_tri_back(int t, int j, const TriPlanB& B, const Lay& Y, double* L) {
    const TriPlan& P = B.P;
    const int front = t >> 3, row = t & 7;
    if (t >= 16 || row >= 5) return;
    if (j >= (front ? P.nB : P.nA)) return;
    const int l = front ? P.mid - B.st * (1 + j) : P.mid + B.st * (1 + j);
    const int nbr = front ? l + B.st : l - B.st;
    double z[5], dn[5];
#pragma unroll
    for (int cI = 0; cI < 5; cI++) { z[cI] = L[Y.SJJ + 25 * l + 5 * cI + row]; dn[cI] = L[Y.DL + 5 * nbr + cI]; }
    const double y = L[Y.R + 5 * l + row];
    L[Y.DL + 5 * l + row] = y - (z[0] * dn[0] + z[1] * dn[1] + z[2] * dn[2] + z[3] * dn[3] + z[4] * dn[4]);
}

// ---- one level of odd-even (cyclic) reduction ahead of the two-front sweep.  The sweep is a chain of dependent 5x5
// factorisations (one per step and front); eliminating every second link FIRST costs about two such steps -- all odd links at once,
// W lanes per link -- and halves the chain that is left.  For the links x_i = cs + st i, i < n, every odd i (l = x_i, p = l - st,
// nx = l + st) gives, with the lower / upper block convention of TriPlanB:
//   Z- = S_ll^-1 S_lp -> in place (lower block of l);  Z+ = S_ll^-1 S_l,nx -> in place (upper block of l);  y = S_ll^-1 r_l -> R[l]
//   S_pp -= S_pl Z-;   S_p,nx = -S_pl Z+  -> upper block of p (over S_pl);    r_p  -= S_pl y          (phase A)
//   S_nx,nx -= S_nx,l Z+;   S_nx,p = -S_nx,l Z- -> lower block of nx (over S_nx,l);   r_nx -= S_nx,l y   (phase B)
// A and B are separate passes because link p of one odd link is link nx of the one below: both update the same S and r.
// The 11 right-hand-side columns of a link (5 of Z-, 5 of Z+, y) are dealt to its W lanes, lane w taking k = w, w + W, ...;
// every lane factorises S_ll itself (as in the sweep).  Afterwards dl_l = y - Z- dl_p - Z+ dl_nx (cr_back), after the sweep.
template <int W>
struct CrLane {
    static const int NS = (11 + W - 1) / W;
    int fl;                      // bit 0: the lane takes part, bit 1: the link has a next link, bit 2 + s: column slot s is in use
    int w;                       // lane within the link: its columns are k = w + W s
    int oLL, oPL, oNL;
    int oRhs[NS], oA[NS], oB[NS];
    double z[NS][5];
    HD bool act() const { return (fl & 1) != 0; }
    HD bool has_n() const { return (fl & 2) != 0; }
    HD bool ok(int s) const { return (fl & (4 << s)) != 0; }
    HD int kind(int s) const { const int k = w + W * s; return (k >= 5 ? 1 : 0) + (k >= 10 ? 1 : 0); }   // 0: column of Z-, 1: of Z+, 2: y
};
#if defined(__HIP_DEVICE_COMPILE__)
#define CR_FLAGS_FRESH(K) asm volatile("" : "+v"((K).fl), "+v"((K).w))
#else
#define CR_FLAGS_FRESH(K) ((void)0)
#endif
template <int W>
HD void cr_setup(CrLane<W>& K, int t, int cs, int n, int st, const Lay& Y, bool skip) {
    const int i = t / W;
    K.w = t - W * i;
    const int nodd = n / 2, idx = 2 * i + 1;
    const bool act = i < nodd && !skip, has_n = idx + 1 < n;
    const int l = act ? cs + st * idx : cs + st, p = l - st, nx = l + st;
    K.oLL = Y.SJJ + 25 * l;
    K.oPL = Y.SPJ + 25 * (p + 1);
    K.oNL = has_n ? Y.SJP + 25 * nx : K.oPL;
    K.fl = (act ? 1 : 0) | (has_n ? 2 : 0);
#pragma unroll
    for (int s = 0; s < CrLane<W>::NS; s++) {
        const int k = K.w + W * s;
        const int T = K.kind(s), cI = k - 5 * T;
        const bool ok = act && k < 11 && (T != 1 || has_n);
        K.fl |= ok ? (4 << s) : 0;
        const int up_l = has_n ? Y.SPJ + 25 * (l + 1) : K.oLL;     // never read when there is no next link; kept inside the image
        K.oRhs[s] = T == 0 ? Y.SJP + 25 * l + 5 * cI : (T == 1 ? up_l + 5 * cI : Y.R + 5 * l);
        K.oA[s] = T == 0 ? Y.SJJ + 25 * p + 5 * cI : (T == 1 ? K.oPL + 5 * cI : Y.R + 5 * p);
        const int nn = has_n ? nx : p;
        K.oB[s] = T == 0 ? K.oNL + 5 * cI : (T == 1 ? Y.SJJ + 25 * nn + 5 * cI : Y.R + 5 * nn);
        if (!ok) { K.oRhs[s] = K.oLL; K.oA[s] = K.oLL; K.oB[s] = K.oLL; }
    }
}
// phase A, loads and arithmetic: the lane's solutions z (kept for phase B) and its p-side columns tA; cr_store_a writes them AFTER
// every lane has loaded (the caller puts the store behind this call in the wavefront's instruction stream).  A fill column
// (S_p,nx, S_nx,p) replaces what its slot held: the old value enters with factor 0.
template <int W>
HD void cr_phase_a(CrLane<W>& K, const double* L, double (*tA)[5]) {
    if (!K.act()) return;
    double lu[25], blk[25];
#pragma unroll
    for (int cI = 0; cI < 5; cI++)
#pragma unroll
        for (int r = 0; r < 5; r++) lu[r * 5 + cI] = L[K.oLL + 5 * cI + r];
#pragma unroll
    for (int s = 0; s < CrLane<W>::NS; s++)
#pragma unroll
        for (int r = 0; r < 5; r++) K.z[s][r] = L[K.oRhs[s] + r];
    lu5_factor(lu);
#pragma unroll
    for (int s = 0; s < CrLane<W>::NS; s++) lu5_solve(lu, K.z[s]);
    SCHED_FENCE();      // the factorisation is dead here: the neighbour block and the targets take its registers (fetching them
                        // ahead of the pivot chain costs ~80 registers more than the kernel has)
#pragma unroll
    for (int e = 0; e < 25; e++) blk[e] = L[K.oPL + e];
#pragma unroll
    for (int s = 0; s < CrLane<W>::NS; s++)
#pragma unroll
        for (int r = 0; r < 5; r++) tA[s][r] = L[K.oA[s] + r];
#pragma unroll
    for (int s = 0; s < CrLane<W>::NS; s++) {
        const double* z = K.z[s];
        const double keep = K.kind(s) != 1 ? 1.0 : 0.0;
#pragma unroll
        for (int r = 0; r < 5; r++) tA[s][r] = keep * tA[s][r] - (blk[r] * z[0] + blk[5 + r] * z[1] + blk[10 + r] * z[2] + blk[15 + r] * z[3] + blk[20 + r] * z[4]);
    }
}
template <int W>
HD void cr_store_a(const CrLane<W>& K, double* L, const double (*tA)[5]) {
#pragma unroll
    for (int s = 0; s < CrLane<W>::NS; s++)
        if (K.ok(s)) {
#pragma unroll
            for (int r = 0; r < 5; r++) { L[K.oRhs[s] + r] = K.z[s][r]; L[K.oA[s] + r] = tA[s][r]; }
        }
}
template <int W>
HD void cr_phase_b(const CrLane<W>& K, const double* L, double (*tB)[5]) {
    if (!(K.act() && K.has_n())) return;
    double blk[25];
#pragma unroll
    for (int s = 0; s < CrLane<W>::NS; s++)
#pragma unroll
        for (int r = 0; r < 5; r++) tB[s][r] = L[K.oB[s] + r];
#pragma unroll
    for (int e = 0; e < 25; e++) blk[e] = L[K.oNL + e];
#pragma unroll
    for (int s = 0; s < CrLane<W>::NS; s++) {
        const double* z = K.z[s];
        const double keep = K.kind(s) != 0 ? 1.0 : 0.0;
#pragma unroll
        for (int r = 0; r < 5; r++) tB[s][r] = keep * tB[s][r] - (blk[r] * z[0] + blk[5 + r] * z[1] + blk[10 + r] * z[2] + blk[15 + r] * z[3] + blk[20 + r] * z[4]);
    }
}
template <int W>
HD void cr_store_b(const CrLane<W>& K, double* L, const double (*tB)[5]) {
#pragma unroll
    for (int s = 0; s < CrLane<W>::NS; s++)
        if (K.ok(s) && K.has_n()) {
#pragma unroll
            for (int r = 0; r < 5; r++) L[K.oB[s] + r] = tB[s][r];
        }
}
// multiplier steps of the eliminated (odd) links, rows dealt to the link's W lanes
template <int W>
HD void cr_back(int t, int cs, int n, int st, const Lay& Y, double* L, bool skip) {
    const int i = t / W, w = t - W * i;
    const int idx = 2 * i + 1;
    if (!(i < n / 2) || skip) return;
    const int l = cs + st * idx, p = l - st, nx = l + st;
    const bool has_n = idx + 1 < n;
    double dp[5], dn[5];
#pragma unroll
    for (int cI = 0; cI < 5; cI++) { dp[cI] = L[Y.DL + 5 * p + cI]; const double dnx = L[Y.DL + 5 * nx + cI]; dn[cI] = has_n ? dnx : 0.0; }   // nx <= one link past the chain: inside the image
    const int oZm = Y.SJP + 25 * l, oZp = has_n ? Y.SPJ + 25 * (l + 1) : oZm;
    double out[(5 + W - 1) / W];
#pragma unroll
    for (int q = 0; q < (5 + W - 1) / W; q++) {
        const int row = w + W * q, rr = row < 5 ? row : 0;
        double acc = L[Y.R + 5 * l + rr];
#pragma unroll
        for (int cI = 0; cI < 5; cI++) { acc -= L[oZm + 5 * cI + rr] * dp[cI]; const double zp = L[oZp + 5 * cI + rr]; acc -= (has_n ? zp : 0.0) * dn[cI]; }
        out[q] = acc;
    }
#pragma unroll
    for (int q = 0; q < (5 + W - 1) / W; q++) {
        const int row = w + W * q;
        if (row < 5) L[Y.DL + 5 * l + row] = out[q];
    }
}
// chains of at least this many links get one reduction level: below, the two passes cost what they save (9 links: 4 sweep steps
// either way)
#define CR_MIN_LINKS 12

// LDS image of the chain kernel.  Gathered across lanes by the block-tridiagonal elimination: the Schur blocks, the
// right-hand side R and the multiplier step DL.  Read by the neighbour lanes: the sparse G_k of every joint (GKA).  Private
// per-lane slots that only relieve the register file: multipliers LAM, D_R^-1 (DINV), the per-step invariants cT|cR (D),
// the constraint force C = G_k' lambda at the accepted point.  The state staging area (trajectory rows go to HBM through it so that the stores coalesce) and the
// control error alias the Schur blocks, which are dead then.  150 doubles per link: 20.4 KB for the 17-body chain, so that
// four workgroups of two instances fit a CU's 160 KB.
HD Lay make_chain_layout(int nb) {
    Lay L;
    L.S = L.ST = L.LT = L.DS = L.XQ = L.NB = L.DTM = L.G = L.GKB = L.GVA = L.GVB = L.UJ = L.CD = L.SS = 0;
    int o = 0;
    L.SJJ = o; o += 25 * nb; L.SJP = o; o += 25 * nb; L.SPJ = o; o += 25 * nb;
    L.R = o; o += 5 * nb; L.DL = o; o += 5 * nb;
    L.GKA = o; o += GKSZ * nb;
    L.LAM = o; o += 5 * nb;
    L.DINV = o; o += 9 * nb;
    L.D = o; o += 6 * nb;
    L.C = o; o += 6 * nb;
    // (cr_back and the clamped operand loads of the sweep read up to one link PAST a chain's last -- DL[nb], R[nb] -- and discard the value:
    // DL and R must therefore not be the image's last arrays; launch_rollout_chain checks DL + 5 (nb + 1) <= total and R + 5 (nb + 1) <= total)
    L.Z = L.SJJ;               // 13 nb staging
    L.DZ = L.SJJ + 13 * nb;    // 12 nb control error
    L.total = o | 1;
    return L;
}

}  // namespace cclqr
