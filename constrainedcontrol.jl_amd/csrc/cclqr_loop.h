// cclqr_loop.h -- phase functions of the rollout kernel for mechanisms with CLOSED kinematic loops (examples/lqr_deltabot.jl:25-33:
// five bodies, seven joints, 33 constraint rows on 30 body coordinates).  Same discretisation, Newton rules and sign conventions
// as the tree kernels (cclqr_dev.h / cclqr_newton.h); what changes is the bookkeeping and the linear solve:
//   * bodies and joints are separate index sets (joint j: parent body ja = M->parent[j] or the origin, child body jb =
//     M->jchild[j]); a body lists its incident joints (M->inc_*), in the caller's joint order;
//   * a loop makes the constraint rows redundant, so the Schur complement S = G_v D^-1 G_k' on the multipliers is singular
//     (rank 28 of 35 for the deltabot, FixedOrientation padded to five rows with two null rows): it is assembled DENSE and solved by
//     Gaussian elimination with complete pivoting that stops at the numerical rank.  Multipliers are then one solution of a
//     consistent singular system (the pivoted basic solution); velocities and poses -- what a rollout returns -- are unique.
// One instance per wavefront; lane t < nb owns body t, lane t < nj owns joint t, lane r < 5 nj owns row r of S.
// Shared with tests/emu/emu_loop.cpp, which runs the same functions lane by lane on the CPU.
#pragma once
#include "cclqr_dev.h"

namespace cclqr {

#define CCLQR_LOOP_MAXB 8      // bodies
#define CCLQR_LOOP_MAXJ 12     // joints: 5 rows each, one row of S per lane of the wavefront
#define LOOP_RANK_TOL 1e-11    // pivots below this fraction of the first pivot are rank deficiency

// LDS image of one instance.  Body-indexed arrays keep the names (and meaning) of the tree kernels' layout so that their body phases
// (ph_body_eval, ph_control_error, ph_gain_partial, ph_accept, ph_update) run unchanged; joint-indexed arrays have nj entries.
HD Lay make_loop_layout(int nb, int nj) {
    Lay L; int o = 0;
    L.Z = o; o += 13 * nb;   L.S = o; o += 6 * nb;   L.ST = o; o += 6 * nb;   L.DS = o; o += 6 * nb;
    L.LAM = o; o += 5 * nj;  L.LT = o; o += 5 * nj;  L.DL = o; o += 5 * nj;
    L.XQ = o; o += 7 * nb;   L.NB = o; o += 9 * nb;  L.DINV = o; o += 9 * nb; L.DTM = o; o += nb;
    L.D = o; o += 6 * nb;    L.G = o; o += 5 * nj;   L.R = o; o += 10 * nj;     // R: the solve's pivot columns, then its pivot rows (as doubles)
    L.GKA = o; o += BLK * nj; L.GKB = o; o += BLK * nj; L.GVA = o; o += BLK * nj; L.GVB = o; o += BLK * nj;
    L.UJ = o; o += nj;
    L.C = o; o += 6 * nb;    L.CD = o; o += 6 * nb;
    L.DZ = o; o += 12 * nb;
    L.SS = 0;            // (the dense system is assembled straight into the lanes' registers since round 4: lpr_assemble)
    L.SJJ = L.SJP = L.SPJ = 0;
    L.total = o | 1;
    return L;
}

// constants of the owned body (t < nb: m, J) and of the owned joint (t < nj: vertices, axis, offset, row selectors)
HD void loop_load_link_consts(LaneRegs& r, const MechDev* M, int t) {
    const int b = t < M->nb ? t : 0, j = t < M->nj ? t : 0;
    r.m = M->m[b];
    for (int i = 0; i < 9; i++) r.J[i] = M->J[b][i];
    for (int i = 0; i < 3; i++) { r.p1[i] = M->p1[j][i]; r.p2[i] = M->p2[j][i]; r.axis[i] = M->axis[j][i]; }
    for (int i = 0; i < 4; i++) r.qoc[i] = M->qoc[j][i];
    for (int i = 0; i < 5; i++)
        for (int k = 0; k < 3; k++) r.sel[i][k] = M->sel[j][i][k];
    r.parent = M->parent[j]; r.childl = M->jchild[j]; r.rotmask = M->rotmask[j]; r.type = M->type[j];
}
HD void loop_load_consts(LaneRegs& r, const MechDev* M, int t) {
    loop_load_link_consts(r, M, t);
    for (int i = 0; i < 3; i++) { r.cT[i] = 0; r.cR[i] = 0; }
    r.pid_int = 0.0; r.pid_last = 0.0;
}

// passive friction of joint t (examples/trackingLQR_triple_cartpole.jl:93-101: -fric * relative joint velocity along the axis), the law of
// ck_friction / ph_control_error on a mechanism whose joints and bodies are separate index sets; a FixedOrientation constraint has none
HD double lp_friction(int t, const Lay& Y, const double* L, const LaneRegs& r, const MechDev* M, double fric) {
    if (t >= M->nj || r.type > 1 || fric == 0.0) return 0.0;
    const int a = r.parent, b = r.childl;
    const double* zb = L + Y.Z + 13 * b;
    double rel;
    if (r.type == 0) {
        rel = r.axis[0] * zb[10] + r.axis[1] * zb[11] + r.axis[2] * zb[12];
        if (a >= 0) { const double* za = L + Y.Z + 13 * a; rel -= r.axis[0] * za[10] + r.axis[1] * za[11] + r.axis[2] * za[12]; }
    } else {
        double dv[3], dva[3], Ra[9];
        for (int i = 0; i < 3; i++) dv[i] = zb[7 + i] - (a >= 0 ? L[Y.Z + 13 * a + 7 + i] : 0.0);
        rotmat(a >= 0 ? L + Y.Z + 13 * a + 3 : QID_, Ra);
        mtv3(Ra, dv, dva);
        rel = r.axis[0] * dva[0] + r.axis[1] * dva[1] + r.axis[2] * dva[2];
    }
    return -fric * rel;
}

// control_pid!(mechanism, pid, k) for joint t (pid.jl:69-88), the law of ph_pid / ck_pid on a mechanism whose joints and bodies are separate
// index sets: minimalCoordinates of the joint from its two bodies' poses (angle about / offset along the axis), wrapped error for revolutes
// (pid.jl:43-57), u = P e + I int(e) + D de/dt added to the joint input.  The integrated / last error stay in the joint's lane.
HD void lp_pid(int t, const Lay& Y, double* L, LaneRegs& r, const MechDev* M, const CtrlDev* C, bool first) {
    if (t >= M->nj || !C->pid_on[t] || r.type > 1) return;
    const int a = r.parent, b = r.childl;
    const double dt = M->dt;
    const double X0[3] = {0, 0, 0};
    const double* za = (a >= 0) ? L + Y.Z + 13 * a : nullptr;
    const double* zb = L + Y.Z + 13 * b;
    const double* qa = za ? za + 3 : QID_;
    double th;
    if (r.type == 0) {
        double qac[4] = {qa[0], -qa[1], -qa[2], -qa[3]}, rel[4], e[4];
        qmul(qac, zb + 3, rel);
        qmul(rel, r.qoc, e);
        th = 2.0 * atan2(r.axis[0] * e[1] + r.axis[1] * e[2] + r.axis[2] * e[3], e[0]);
    } else {
        double Ra[9], Rb[9], rp[3], w[3], gT[3];
        rotmat(qa, Ra); rotmat(zb + 3, Rb);
        mv3(Rb, r.p2, rp);
        for (int i = 0; i < 3; i++) w[i] = zb[i] + rp[i] - (za ? za[i] : X0[i]);
        mtv3(Ra, w, gT);
        th = r.axis[0] * (gT[0] - r.p1[0]) + r.axis[1] * (gT[1] - r.p1[1]) + r.axis[2] * (gT[2] - r.p1[2]);
    }
    const double PI = 3.14159265358979323846;
    double e = C->pid_goal[t] - th;
    if (r.type == 0) { if (e > PI) e -= 2 * PI; else if (e < -PI) e += 2 * PI; }
    if (first) r.pid_last = e;
    r.pid_int += e * dt;
    const double de = (e - r.pid_last) / dt;
    L[Y.UJ + t] += C->pid_P[t] * e + C->pid_I[t] * r.pid_int + C->pid_D[t] * de;
    r.pid_last = e;
}

// F1: joint inputs -> force / torque on body t (SURVEY 8a-bis 'Joint input': a revolute applies +-u axis as a torque, a prismatic
// +-u axis as a force at the joint's vertices; a FixedOrientation constraint takes no input), per-step invariants, solution guess
HD void lp_forces(int t, const Lay& Y, double* L, LaneRegs& r, const MechDev* M) {
    if (t >= M->nb) return;
    const double dt = M->dt;
    const double* z = L + Y.Z + 13 * t;
    double F[3] = {0, 0, 0}, tau[3] = {0, 0, 0}, Rb[9];
    rotmat(z + 3, Rb);
    for (int k = 0; k < M->inc_n[t]; k++) {
        const int j = M->inc_j[t][k];
        const double u = L[Y.UJ + j];
        if (u == 0.0 || M->type[j] > 1) continue;
        const double f[3] = {M->axis[j][0] * u, M->axis[j][1] * u, M->axis[j][2] * u};     // in the parent body's frame
        if (M->inc_side[t][k] == 0) {         // body t is the joint's child
            const int a = M->parent[j];
            double Ra[9], fw[3], fb[3];
            rotmat(a >= 0 ? L + Y.Z + 13 * a + 3 : QID_, Ra);
            mv3(Ra, f, fw); mtv3(Rb, fw, fb);
            if (M->type[j] == 1) { double c[3]; cross3(M->p2[j], fb, c); for (int i = 0; i < 3; i++) { F[i] += fw[i]; tau[i] += c[i]; } }
            else for (int i = 0; i < 3; i++) tau[i] += fb[i];
        } else {                              // body t is the joint's parent
            if (M->type[j] == 1) {
                double fw[3], cr[3];
                mv3(Rb, f, fw); cross3(M->p1[j], f, cr);
                for (int i = 0; i < 3; i++) { F[i] -= fw[i]; tau[i] -= cr[i]; }
            } else for (int i = 0; i < 3; i++) tau[i] -= f[i];
        }
    }
    const double* v1 = z + 7; const double* w1 = z + 10;
    const double sq1 = sqrt(4.0 / (dt * dt) - (w1[0] * w1[0] + w1[1] * w1[1] + w1[2] * w1[2]));
    double Jw1[3], c1[3];
    mv3(r.J, w1, Jw1); cross3(w1, Jw1, c1);
    for (int i = 0; i < 3; i++) {
        r.cT[i] = r.m * (-v1[i] / dt + (i == 2 ? -M->g : 0.0)) - F[i];
        r.cR[i] = -(sq1 * Jw1[i] - c1[i]) - 2.0 * tau[i];
        L[Y.S + 6 * t + i] = v1[i]; L[Y.S + 6 * t + 3 + i] = w1[i];
    }
    L[Y.DTM + t] = dt / r.m;
}

// F2: constraint Jacobians of joint t at the current knot (force mapping G_k)
HD void lp_knot_jac(int t, const Lay& Y, double* L, const LaneRegs& r, const MechDev* M) {
    if (t >= M->nj) return;
    const int a = r.parent, b = r.childl;
    const double X0[3] = {0, 0, 0};
    const double* za = (a >= 0) ? L + Y.Z + 13 * a : nullptr;
    const double* zb = L + Y.Z + 13 * b;
    double g[5];
    joint_eval<true>(r, za ? za : X0, za ? za + 3 : QID_, zb, zb + 3, a >= 0, 1.0, 1.0, nullptr, nullptr, g, L + Y.GKA + BLK * t, L + Y.GKB + BLK * t);
}

// sum over the joints around body t of (its side of G_k)' y_j, y at offset oy (5 per joint)
HD void lp_gk_t_apply(int t, const Lay& Y, const double* L, const MechDev* M, int oy, double* out) {
    for (int c = 0; c < 6; c++) out[c] = 0.0;
    for (int k = 0; k < M->inc_n[t]; k++) {
        const int j = M->inc_j[t][k];
        const double* gk = L + (M->inc_side[t][k] ? Y.GKA : Y.GKB) + BLK * j;
        const double* y = L + oy + 5 * j;
        for (int c = 0; c < 6; c++) out[c] += gk[c] * y[0] + gk[6 + c] * y[1] + gk[12 + c] * y[2] + gk[18 + c] * y[3] + gk[24 + c] * y[4];
    }
}
// F3: C_b = sum G_k' lambda at the knot's multipliers
HD void lp_force_map(int t, const Lay& Y, double* L, const MechDev* M) {
    if (t >= M->nb) return;
    double c[6];
    lp_gk_t_apply(t, Y, L, M, Y.LAM, c);
    for (int i = 0; i < 6; i++) { L[Y.C + 6 * t + i] = c[i]; L[Y.CD + 6 * t + i] = 0.0; }
}

// E2: joint t at the next knot: g and W = G_v D^-1 (JAC) or g only; returns |g_t|^2
template <bool JAC>
HD double lp_joint_eval(int t, const Lay& Y, double* L, const LaneRegs& r, const MechDev* M) {
    if (t >= M->nj) return 0.0;
    const int a = r.parent, b = r.childl;
    const double dt = M->dt;
    const double X0[3] = {0, 0, 0};
    const double* pa = (a >= 0) ? L + Y.XQ + 7 * a : nullptr;
    const double* pb = L + Y.XQ + 7 * b;
    double g[5];
    joint_eval<JAC>(r, pa ? pa : X0, pa ? pa + 3 : QID_, pb, pb + 3, a >= 0, (a >= 0) ? dt * L[Y.DTM + a] : 0.0, dt * L[Y.DTM + b],
                    (a >= 0) ? L + Y.NB + 9 * a : nullptr, L + Y.NB + 9 * b, g, L + Y.GVA + BLK * t, L + Y.GVB + BLK * t);
    double acc = 0.0;
    for (int i = 0; i < 5; i++) { L[Y.G + 5 * t + i] = g[i]; acc += g[i] * g[i]; }
    return acc;
}

// ---- dense solve of the (singular, consistent) system S dl = r: Gauss-Jordan elimination, column by column, pivoting over the rows, that
// SKIPS a column whose entries in the rows still in play are all below the rank tolerance (a redundant direction: its dl stays 0).
// Lane = row for the whole solve, and since round 4 the row LIVES IN THE LANE'S REGISTERS (8 NCB >= 5 nj columns, compile time; the LDS
// row it is loaded from is padded with zeros that far).  Nothing of the solve goes through LDS: with four wavefronts on a CU an LDS
// instruction costs a wavefront ~17 cycles whatever its lane count (profiles/r04/lds_cost_vs_active_lanes_microbench.txt), and the
// LDS-resident elimination with complete pivoting of rounds 2-3 spent 3.1-3.4 k cycles per pivot step on ~100 of them.  What shapes it:
//   * the register file has no run-time index, so the COLUMN of a step must be known at compile time: columns are taken in their natural
//     order (eight steps unrolled per pass of a run-time loop over eight-column blocks; the lane's entry in the step's column is a select
//     over the blocks), and the pivot search runs over the ROWS -- one 32-bit key per lane (lp_row_key), the wavefront's largest key is
//     the pivot row (ties go to the smaller row): six v_max_u32 with a DPP operand;
//   * rank: a column whose largest candidate is below LOOP_RANK_TOL x the largest entry of the assembled system is skipped.  (Complete
//     pivoting -- rounds 2-3, and two register-resident versions of it this round -- needs a column search per row and a run-time column:
//     +120 and +90 instructions per step; measured 2.0 k cycles per step against ~1 k here.)
//   * the pivot row reaches the other lanes by lane reads (v_readlane: the values land in scalar registers and feed the multiply-adds
//     directly); blocks of columns that lie wholly before the current one are skipped (they hold rounding residue, never read again);
//   * Gauss-Jordan, because in SIMD it is free: the retired rows' lanes execute the multiply-adds anyway, so they eliminate the column from
//     their rows too and there is NO back substitution -- at the end a row that was a pivot row holds dl[its column] = rhs / pivot.
// The multipliers of a loop mechanism are not unique; G_k' dl -- velocities, poses -- is (any solution of the consistent system).
template <int NCB>
struct LoopRowR {
    double a[8 * NCB];            // the row (compile-time indices only: registers)
    double rhs;
    double ipiv;                  // 1 / pivot of the row once it has retired
    int col;                      // the column this lane's row was the pivot row of (-1: not yet / never)
};
HD unsigned long long lp_bits(double v) { union { double d; unsigned long long u; } x; x.d = v; return x.u; }
HD double lp_from_bits(unsigned long long u) { union { double d; unsigned long long u; } x; x.u = u; return x.d; }
// a row's candidate as ONE 32-bit key: the high word of the magnitude (exponent and 20 mantissa bits) with its 6 lowest bits replaced by
// 63 - row -- 14 bits of mantissa decide between candidates, which is all a pivot search needs (the pivot VALUE is read exactly, from the row)
HD unsigned lp_row_key(double v, int row) { return ((unsigned)(lp_bits(fabs(v)) >> 32) & ~0x3Fu) | (unsigned)(63 - row); }
HD int lp_key_row(unsigned key) { return 63 - (int)(key & 63); }
// S: row `row` of the dense system [S | r] (lane = row; joint i = row / 5), assembled straight into the lane's registers:
//   S[(i,ri)][(j,rj)] = sum over the bodies joints i and j share of  W_side(i)[ri] . Gk_side(j)[rj]
//   r[(i,ri)] = g_i[ri] - W_b(i)[ri] . d_{jb(i)} - W_a(i)[ri] . d_{ja(i)}
// The loop over the joints j is unrolled to the 8 NCB / 5 joints the instantiation holds (register numbers are compile-time), guarded by the
// mechanism's joint count; columns beyond 5 nj and the rows of lanes beyond 5 nj are zero.  Returns the row's largest magnitude.
template <int NCB>
HD double lpr_assemble(LoopRowR<NCB>& R, int row, const Lay& Y, const double* L, const MechDev* M) {
    const int nj = M->nj, mr = 5 * nj;
    const bool on = row < mr;
    const int i = on ? row / 5 : 0, ri = on ? row - 5 * i : 0;
    const int ia = M->parent[i], ib = M->jchild[i];
    R.col = -1; R.ipiv = 0.0;
    double wa[6], wb[6];
    for (int c = 0; c < 6; c++) { wa[c] = L[Y.GVA + BLK * i + 6 * ri + c]; wb[c] = L[Y.GVB + BLK * i + 6 * ri + c]; }
    double m_ = 0.0;
#pragma unroll
    for (int c = 0; c < 8 * NCB; c++) R.a[c] = 0.0;
#pragma unroll
    for (int j = 0; j < (8 * NCB) / 5; j++) {
        if (j < nj) {                                      // (uniform)
            const int ja = M->parent[j], jb = M->jchild[j];
            const bool bb = on && ib == jb, ba = on && ib == ja, ab = on && ia >= 0 && ia == jb, aa = on && ia >= 0 && ia == ja;
            double sj[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            if (bb || ab) {                 // joint j's child-side block meets joint i's child body (bb) or parent body (ab): one block in flight at a time
                double kb[30];
                for (int e = 0; e < 30; e++) kb[e] = L[Y.GKB + BLK * j + e];
                for (int rj = 0; rj < 5; rj++) {
                    if (bb) sj[rj] += dot6(wb, kb + 6 * rj);
                    if (ab) sj[rj] += dot6(wa, kb + 6 * rj);
                }
            }
            if (ba || aa) {
                double ka[30];
                for (int e = 0; e < 30; e++) ka[e] = L[Y.GKA + BLK * j + e];
                for (int rj = 0; rj < 5; rj++) {
                    if (ba) sj[rj] += dot6(wb, ka + 6 * rj);
                    if (aa) sj[rj] += dot6(wa, ka + 6 * rj);
                }
            }
#pragma unroll
            for (int rj = 0; rj < 5; rj++) { R.a[5 * j + rj] = sj[rj]; m_ = fmax(m_, fabs(sj[rj])); }
        }
    }
    double rr = L[Y.G + 5 * i + ri] - dot6(wb, L + Y.D + 6 * ib);
    if (ia >= 0) rr -= dot6(wa, L + Y.D + 6 * ia);
    R.rhs = on ? rr : 0.0;
    return m_;
}
#if defined(__HIP_DEVICE_COMPILE__)
#define LOOP_OPAQUE(i) asm volatile("" : "+s"(i))
#else
#define LOOP_OPAQUE(i)
#endif
// the row's entry in column 8 kb + U (kb the same in every lane): a select over the blocks.  (Each comparison sees its own opaque copy of
// kb: the optimiser otherwise folds the selects back into ONE load a[8 kb + U] with a run-time index -- and the row moves to scratch memory.)
template <int NCB, int U>
HD double lpr_entry(const LoopRowR<NCB>& R, int kb) {
    double own = R.a[U];
#pragma unroll
    for (int B = 1; B < NCB; B++) { int kq = kb; LOOP_OPAQUE(kq); own = (kq == B) ? R.a[8 * B + U] : own; }
    return own;
}
// the row's candidate for column col: 0 when the row is not in play
template <int NCB>
HD unsigned lpr_key(const LoopRowR<NCB>& R, int row, int mr, double own) { return (row < mr && R.col < 0) ? lp_row_key(own, row) : 0u; }
// bookkeeping of a step (pivot row prow, column col, ip = 1 / pivot); returns the factor this lane's row eliminates with
template <int NCB>
HD double lpr_step(LoopRowR<NCB>& R, int row, int col, int prow, int mr, double own, double ip) {
    if (row == prow) { R.col = col; R.ipiv = ip; }
    return (row < mr && row != prow) ? own * ip : 0.0;          // retired rows too (Gauss-Jordan)
}
// the solution: dl[col] = rhs / pivot for the rows that were pivot rows (the columns that found none keep dl = 0)
template <int NCB>
HD void lpr_solution(const LoopRowR<NCB>& R, int row, int mr, const Lay& Y, double* L) {
    if (row < mr && R.col >= 0) L[Y.DL + R.col] = R.rhs * R.ipiv;
}
// columns the instantiation must hold for nj joints
HD int loop_col_blocks(int nj) { return (5 * nj + 7) / 8; }

// body solve: cd = sum G_k' dl ; ds = D^-1 (d + cd)
HD void lp_body_solve(int t, const Lay& Y, double* L, const MechDev* M) {
    if (t >= M->nb) return;
    double cd[6], tv[6], Di[9];
    lp_gk_t_apply(t, Y, L, M, Y.DL, cd);
    for (int c = 0; c < 6; c++) tv[c] = L[Y.D + 6 * t + c] + cd[c];
    for (int i = 0; i < 9; i++) Di[i] = L[Y.DINV + 9 * t + i];
    const double dtm = L[Y.DTM + t];
    for (int c = 0; c < 3; c++) {
        L[Y.DS + 6 * t + c] = tv[c] * dtm;
        L[Y.DS + 6 * t + 3 + c] = Di[3 * c] * tv[3] + Di[3 * c + 1] * tv[4] + Di[3 * c + 2] * tv[5];
    }
    for (int c = 0; c < 6; c++) L[Y.CD + 6 * t + c] = cd[c];
}
// trial point st = s - alpha ds (body t), lt = lam - alpha dl (joint t); returns the lane's share of ||(ds, dl)||^2
HD double lp_trial(int t, const Lay& Y, double* L, const MechDev* M, double alpha) {
    double acc = 0.0;
    if (t < M->nb)
        for (int i = 0; i < 6; i++) { const double dv = L[Y.DS + 6 * t + i]; L[Y.ST + 6 * t + i] = L[Y.S + 6 * t + i] - alpha * dv; acc += dv * dv; }
    if (t < M->nj)
        for (int i = 0; i < 5; i++) { const double ev = L[Y.DL + 5 * t + i]; L[Y.LT + 5 * t + i] = L[Y.LAM + 5 * t + i] - alpha * ev; acc += ev * ev; }
    return acc;
}
// accept the trial point: s, lambda <- trial ; C -= alpha CD
HD void lp_accept(int t, const Lay& Y, double* L, const MechDev* M, double alpha) {
    if (t < M->nb)
        for (int i = 0; i < 6; i++) { L[Y.S + 6 * t + i] = L[Y.ST + 6 * t + i]; L[Y.C + 6 * t + i] -= alpha * L[Y.CD + 6 * t + i]; L[Y.CD + 6 * t + i] = 0.0; }
    if (t < M->nj)
        for (int i = 0; i < 5; i++) L[Y.LAM + 5 * t + i] = L[Y.LT + 5 * t + i];
}

}  // namespace cclqr
