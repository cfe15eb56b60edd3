// cclqr_lin_loop.h -- linearsystem(mechanism, xd, vd, qd, ωd, Fτd, bodyids, eqcids) (call sites src/control/lqr.jl:63,
// src/control/lqr_tracking.jl:88) for mechanisms with CLOSED kinematic loops (examples/lqr_deltabot.jl:47-53), in the layout of the
// closed-loop rollout kernel (cclqr_loop.h: bodies and joints are separate index sets, a body lists the joints around it).
//
// The model with the multipliers EXOGENOUS,  z+ = A z + Bu u + Bl lambda,  G z+ = 0  (error coordinates per body [x, v, qtilde, w],
// lqr.jl:92-103), is as well defined for a loop as for a tree: every entry is a sum of per-joint and per-body blocks -- the same
// blocks as in cclqr_lin_dev.h (lin_joint_core: geometric stiffness d(G(z)'lambda*)/dz, state dependence of the applied input;
// body_col*: the body's own map) -- only that a body may be the child of several joints and two joints may connect the same two
// bodies, so the rows are ACCUMULATED into zeroed matrices.  What a loop breaks is the elimination of lambda: G Bl is singular (redundant
// constraint rows), which is handled where the projected pair A' = A - Bl X, D = Bu - Bl Y, (G Bl) [X | Y] = G [A | Bu] is formed
// (project_model_kernel, rollout_loop.hip: complete pivoting up to the numerical rank).
//
// Runs after one converged Newton step on the setpoint held in LDS: S = (v+, w+), LAM = lambda*, XQ = next pose, DINV = D_R(w+)^-1,
// GKA/GKB = G at the current knot, UJ = joint inputs.  __host__ __device__: tests/emu/emu_loop.cpp runs the same functions on the CPU.
#pragma once
#include "cclqr_loop.h"
#include "cclqr_lin_dev.h"

namespace cclqr {

// L1: joint t (lane t < nj)
HD void lp_lin_joint(int t, const Lay& Y, int JB, double* L, const LaneRegs& r, const MechDev* M) {
    if (t >= M->nj) return;
    const int a = r.parent, b = r.childl;
    const bool has_a = a >= 0;
    const double X0[3] = {0, 0, 0};
    const double* xb = L + Y.Z + 13 * b;
    lin_joint_core(L + JB + LJB * t, r, has_a, has_a ? L + Y.Z + 13 * a : X0, has_a ? L + Y.Z + 13 * a + 3 : QID_, xb, xb + 3, L + Y.LAM + 5 * t, L[Y.UJ + t]);
}

// M[r0 .. r0+11][c0 .. c0+2] += the body's response to a 3x3 force block FT and torque block FR (either may be null = zero):
// dw+ = Dinv FR ; dqt+ = N dw+ ; dv+ = (dt/m) FT ; dx+ = dt dv+
HD void body_cols_acc(double* Mx, int ld, int r0, int c0, const double* FT, const double* FR, const double* Dinv, const double* N, double dtm, double dt) {
    double dw[9], dq[9];
    if (FR) { mm3(Dinv, FR, dw); mm3(N, dw, dq); } else for (int i = 0; i < 9; i++) { dw[i] = 0; dq[i] = 0; }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const double dv = FT ? dtm * FT[i * 3 + j] : 0.0;
            Mx[(size_t)(r0 + i) * ld + c0 + j] += dt * dv;
            Mx[(size_t)(r0 + 3 + i) * ld + c0 + j] += dv;
            Mx[(size_t)(r0 + 6 + i) * ld + c0 + j] += dq[i * 3 + j];
            Mx[(size_t)(r0 + 9 + i) * ld + c0 + j] += dw[i * 3 + j];
        }
}

// L2: body t (lane t < nb) accumulates its 12 rows of A: its own columns and the columns of every body it shares a joint with.
// O.A must be zero on entry; rows of different bodies are disjoint, so the lanes do not interfere.
HD void lp_lin_rows_A(int t, const Lay& Y, int JB, const double* L, const LaneRegs& r, const MechDev* M, const LinOut& O) {
    if (t >= M->nb) return;
    const double dt = M->dt, dtm = dt / r.m;
    const double* Dinv = L + Y.DINV + 9 * t;
    const double* w2 = L + Y.S + 6 * t + 3;
    const double sq2 = sqrt(4.0 / (dt * dt) - (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2]));
    double N[9];
    make_N(w2, sq2, dt, N);
    const int rb = 12 * t, ld = O.mx;
    double* A = O.A;
    // sums over the joints around the body: as the parent (side 1) it feels d(force, torque)_a / d(x_a, q_a); as the child (side 0) d torque_b / d q_b
    double frx[9], ftq[9], frq[9];
    for (int i = 0; i < 9; i++) { frx[i] = 0.0; ftq[i] = 0.0; frq[i] = 0.0; }
    for (int k = 0; k < M->inc_n[t]; k++) {
        const double* sc = L + JB + LJB * M->inc_j[t][k];
        if (M->inc_side[t][k]) { for (int i = 0; i < 9; i++) { frx[i] += sc[J_PRXA + i]; ftq[i] += sc[J_PTQ + i]; frq[i] += sc[J_PRQA + i]; } }
        else for (int i = 0; i < 9; i++) frq[i] += sc[J_RQB + i];
    }
    // own columns -- x: dx+ = I (+ the torque of the joints it is the parent of);  v: dv+ = I, dx+ = dt I
    body_cols_acc(A, ld, rb, rb, nullptr, frx, Dinv, N, dtm, dt);
    for (int i = 0; i < 3; i++) { A[(size_t)(rb + i) * ld + rb + i] += 1.0; A[(size_t)(rb + i) * ld + rb + 3 + i] += dt; A[(size_t)(rb + 3 + i) * ld + rb + 3 + i] += 1.0; }
    // qtilde: forces / torques of the joints around it, plus the kinematic term d qt+/d qt = R(wq)'
    body_cols_acc(A, ld, rb, rb + 6, ftq, frq, Dinv, N, dtm, dt);
    {
        const double wq[4] = {0.5 * dt * sq2, 0.5 * dt * w2[0], 0.5 * dt * w2[1], 0.5 * dt * w2[2]};
        double Rw[9];
        rotmat(wq, Rw);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) A[(size_t)(rb + 6 + i) * ld + rb + 6 + j] += Rw[j * 3 + i];
    }
    // w: dw+ = Dinv dPsi/dw ; dqt+ = N dw+
    {
        const double* w1 = L + Y.Z + 13 * t + 10;
        const double sq1 = sqrt(4.0 / (dt * dt) - (w1[0] * w1[0] + w1[1] * w1[1] + w1[2] * w1[2]));
        double Jw1[3], S[9], SJ[9], Sj[9], Psi[9];
        mv3(r.J, w1, Jw1);
        S[0] = sq1; S[1] = w1[2]; S[2] = -w1[1]; S[3] = -w1[2]; S[4] = sq1; S[5] = w1[0]; S[6] = w1[1]; S[7] = -w1[0]; S[8] = sq1;   // sq1 I - [w1]x
        mm3(S, r.J, SJ);
        skew3(Jw1, Sj);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Psi[i * 3 + j] = SJ[i * 3 + j] + Sj[i * 3 + j] - Jw1[i] * w1[j] / sq1;
        body_cols_acc(A, ld, rb, rb + 9, nullptr, Psi, Dinv, N, dtm, dt);
    }
    // columns of the bodies on the other side of each joint
    for (int k = 0; k < M->inc_n[t]; k++) {
        const int j = M->inc_j[t][k];
        const double* sc = L + JB + LJB * j;
        if (M->inc_side[t][k] == 0) {       // this body is the child: the parent's orientation moves the constraint force on it
            const int a = M->parent[j];
            if (a >= 0) body_cols_acc(A, ld, rb, 12 * a + 6, sc + J_TQA, sc + J_RQA, Dinv, N, dtm, dt);
        } else {                            // this body is the parent: the child's position and orientation
            const int c = M->jchild[j];
            double nfrx[9];
            for (int i = 0; i < 9; i++) nfrx[i] = -sc[J_PRXA + i];
            body_cols_acc(A, ld, rb, 12 * c, nullptr, nfrx, Dinv, N, dtm, dt);
            body_cols_acc(A, ld, rb, 12 * c + 6, nullptr, sc + J_PRQB, Dinv, N, dtm, dt);
        }
    }
}

// L3: body t accumulates its rows of Bl (five columns per joint around it) and of Bu; joint t writes its rows of G = dg/dz+.
// cj[i] = joint index of input i.  O.Bl, O.Bu, O.G must be zero on entry.
HD void lp_lin_rows_B(int t, const Lay& Y, const double* L, const LaneRegs& r, const MechDev* M, const int* cj, const LinOut& O) {
    const double dt = M->dt;
    if (t < M->nb) {
        const double dtm = dt / r.m;
        const double* Dinv = L + Y.DINV + 9 * t;
        const double* w2 = L + Y.S + 6 * t + 3;
        double N[9];
        make_N(w2, sqrt(4.0 / (dt * dt) - (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2])), dt, N);
        const int rb = 12 * t;
        for (int k = 0; k < M->inc_n[t]; k++) {
            const int j = M->inc_j[t][k];
            const double* Gk = L + (M->inc_side[t][k] ? Y.GKA : Y.GKB) + BLK * j;
            for (int row = 0; row < 5; row++) body_col1(O.Bl, O.ml, rb, 5 * j + row, Gk + 6 * row, Gk + 6 * row + 3, Dinv, N, dtm, dt, true);
        }
        for (int i = 0; i < O.mu; i++) {
            const int j = cj[i];
            if (j < 0 || j >= M->nj || M->type[j] > 1) continue;          // (a FixedOrientation constraint takes no input)
            double ft[3] = {0, 0, 0}, fr[3] = {0, 0, 0};
            bool hit = false;
            const double* ax = M->axis[j];
            if (M->jchild[j] == t) {              // this body is the joint's child
                const int a = M->parent[j];
                double Ra[9], Rb[9], Raa[3], yb[3];
                rotmat(a >= 0 ? L + Y.Z + 13 * a + 3 : QID_, Ra); rotmat(L + Y.Z + 13 * t + 3, Rb);
                mv3(Ra, ax, Raa); mtv3(Rb, Raa, yb);
                if (M->type[j] == 1) { double cr[3]; cross3(M->p2[j], yb, cr); for (int q = 0; q < 3; q++) { ft[q] = Raa[q]; fr[q] = 2.0 * cr[q]; } }
                else for (int q = 0; q < 3; q++) fr[q] = 2.0 * yb[q];
                hit = true;
            } else if (M->parent[j] == t) {       // ... its parent
                double Rb[9], Raa[3];
                rotmat(L + Y.Z + 13 * t + 3, Rb);
                mv3(Rb, ax, Raa);
                if (M->type[j] == 1) { double cr[3]; cross3(M->p1[j], ax, cr); for (int q = 0; q < 3; q++) { ft[q] = -Raa[q]; fr[q] = -2.0 * cr[q]; } }
                else for (int q = 0; q < 3; q++) fr[q] = -2.0 * ax[q];
                hit = true;
            }
            if (hit) body_col1(O.Bu, O.mu, rb, i, ft, fr, Dinv, N, dtm, dt, true);
        }
    }
    if (t < M->nj) {      // G rows of joint t at the next knot (a FixedOrientation's two null rows stay zero)
        const int a = r.parent, b = r.childl;
        const double X0[3] = {0, 0, 0};
        const double* pa = (a >= 0) ? L + Y.XQ + 7 * a : nullptr;
        const double* pb = L + Y.XQ + 7 * b;
        double g[5], Ba[30], Bb[30];
        joint_eval<true>(r, pa ? pa : X0, pa ? pa + 3 : QID_, pb, pb + 3, a >= 0, 1.0, 1.0, nullptr, nullptr, g, Ba, Bb);
        for (int row = 0; row < 5; row++) {
            double* Gr = O.G + (size_t)(5 * t + row) * O.mx;
            for (int q = 0; q < 3; q++) {
                Gr[12 * b + q] += Bb[row * 6 + q];
                Gr[12 * b + 6 + q] += Bb[row * 6 + 3 + q];
                if (a >= 0) { Gr[12 * a + q] += Ba[row * 6 + q]; Gr[12 * a + 6 + q] += Ba[row * 6 + 3 + q]; }
            }
        }
    }
}

}  // namespace cclqr
