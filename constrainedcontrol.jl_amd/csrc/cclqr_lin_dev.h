// cclqr_lin_dev.h -- phase functions of the linearisation kernel: linearsystem(mechanism, xd, vd, qd, ωd, Fτd, bodyids, eqcids)
// (call sites src/control/lqr.jl:63, src/control/lqr_tracking.jl:88; the function itself lives in ConstrainedDynamics).
//
// z+ = A z + Bu u + Bl lambda,  G z+ = 0  about (zd, ud, lambda*), error coordinates per body [x, v, qtilde, w]
// (lqr.jl:92-103), exact Jacobians of the one-step map with lambda exogenous, including the geometric-stiffness
// term d(G(z)' lambda*)/dz and the state dependence of the applied joint input (arXiv:2010.05886).
//
// Runs after one converged Newton step on the setpoint held in LDS: S = (v+, w+), LAM = lambda*, XQ = next pose,
// DINV = D_R(w+)^-1 (N(w+) is recomputed here), GKA/GKB = G at the current knot, UJ = joint inputs.
#pragma once
#include "cclqr_dev.h"

namespace cclqr {

#define LJB 64   // doubles of per-joint scratch: 7 blocks of 9
// block offsets inside a joint's scratch
#define J_TQA 0    // d(F+cT)_b / d qa
#define J_RQA 9    // d(2tau+cR)_b / d qa
#define J_RQB 18   // d(2tau+cR)_b / d qb
#define J_PTQ 27   // d(F+cT)_a / d qa
#define J_PRXA 36  // d(2tau+cR)_a / d xa   ( d/d xb = - this )
#define J_PRQB 45  // d(2tau+cR)_a / d qb
#define J_PRQA 54  // d(2tau+cR)_a / d qa

HD void skew3(const double* a, double* S) {
    S[0] = 0; S[1] = -a[2]; S[2] = a[1]; S[3] = a[2]; S[4] = 0; S[5] = -a[0]; S[6] = -a[1]; S[7] = a[0]; S[8] = 0;
}
HD void Lmat4(const double* q, double* L) {
    double s = q[0], x = q[1], y = q[2], z = q[3];
    L[0] = s; L[1] = -x; L[2] = -y; L[3] = -z; L[4] = x; L[5] = s; L[6] = -z; L[7] = y;
    L[8] = y; L[9] = z; L[10] = s; L[11] = -x; L[12] = z; L[13] = -y; L[14] = x; L[15] = s;
}
HD void Rmat4(const double* q, double* R) {
    double s = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = s; R[1] = -x; R[2] = -y; R[3] = -z; R[4] = x; R[5] = s; R[6] = z; R[7] = -y;
    R[8] = y; R[9] = -z; R[10] = s; R[11] = x; R[12] = z; R[13] = y; R[14] = -x; R[15] = s;
}
HD void mm4(const double* A, bool ta, const double* B, double* C) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (ta ? A[k * 4 + i] : A[i * 4 + k]) * B[k * 4 + j];
            C[i * 4 + j] = s;
        }
}
HD void vblk(const double* M4, double sc, double* o, bool add) {   // o (+)= sc * V M4 V'
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) o[r * 3 + c] = (add ? o[r * 3 + c] : 0.0) + sc * M4[(r + 1) * 4 + 1 + c];
}
HD void axpy9(double sc, const double* A, double* o, bool add) {
    for (int i = 0; i < 9; i++) o[i] = (add ? o[i] : 0.0) + sc * A[i];
}

// L1: one joint: sensitivities of the constraint force G(z)'lambda* and of the applied input u to the poses of both bodies, into the joint's
// scratch o (7 blocks of 9).  r holds the joint's constants; (xa, qa) / (xb, qb) are the parent's / child's pose at the current knot
// (has_a false: the parent is the origin).  Shared by the tree layout (ph_lin_joint) and the closed-loop layout (cclqr_lin_loop.h).
HD void lin_joint_core(double* o, const LaneRegs& r, bool has_a, const double* xa, const double* qa, const double* xb, const double* qb, const double* lam, double u) {
    for (int i = 0; i < 63; i++) o[i] = 0.0;
    double mu3[3] = {0, 0, 0}, nu3[3] = {0, 0, 0};
    for (int row = 0; row < 5; row++) {
        bool rot = (r.rotmask >> row) & 1;
        for (int i = 0; i < 3; i++) { double v = r.sel[row][i] * lam[row]; if (rot) nu3[i] += v; else mu3[i] += v; }
    }
    const double f[3] = {r.axis[0] * u, r.axis[1] * u, r.axis[2] * u};
    double mue[3] = {mu3[0], mu3[1], mu3[2]};
    if (r.type == 1) for (int i = 0; i < 3; i++) mue[i] += f[i];   // prismatic input acts like a translational multiplier on F and tau_b

    double Ra[9], Rb[9], RbtRa[9], rp[3], w[3], Ratw[3];
    rotmat(qa, Ra); rotmat(qb, Rb);
    mtm3(Rb, Ra, RbtRa);
    mv3(Rb, r.p2, rp);
    for (int i = 0; i < 3; i++) w[i] = xb[i] + rp[i] - xa[i];
    mtv3(Ra, w, Ratw);
    double Sme[9], Sm[9], Sp2[9], T1[9], T2[9];
    skew3(mue, Sme); skew3(mu3, Sm); skew3(r.p2, Sp2);
    // child body: F_b + cT_b = Ra mue ;  2tau_b + cR_b = 2 p2 x (Rb'Ra mue)
    double y[3], t3[3], Sy[9];
    mv3(Ra, mue, t3); mtv3(Rb, t3, y);
    skew3(y, Sy);
    mm3(Sp2, Sy, T1);
    axpy9(4.0, T1, o + J_RQB, true);
    if (has_a) {
        double RaS[9];
        mm3(Ra, Sme, RaS);
        axpy9(-2.0, RaS, o + J_TQA, true);
        axpy9(2.0, RaS, o + J_PTQ, true);
        mm3(RbtRa, Sme, T1); mm3(Sp2, T1, T2);
        axpy9(-4.0, T2, o + J_RQA, true);
        // parent body: cR_a = 2 [mu]x Ra' w   (the input torque on the parent, -2 p1 x f, does not depend on the state)
        double Rat[9] = {Ra[0], Ra[3], Ra[6], Ra[1], Ra[4], Ra[7], Ra[2], Ra[5], Ra[8]}, SmRat[9], Sw[9];
        mm3(Sm, Rat, SmRat);
        axpy9(-2.0, SmRat, o + J_PRXA, true);
        mm3(SmRat, Rb, T1); mm3(T1, Sp2, T2);
        axpy9(-4.0, T2, o + J_PRQB, true);
        skew3(Ratw, Sw);
        mm3(Sm, Sw, T1);
        axpy9(4.0, T1, o + J_PRQA, true);
    }
    if (r.type == 0 && u != 0.0) {   // revolute input: 2tau_b = 2 Rb'Ra f
        double Sf[9], yf[3], Syf[9];
        skew3(f, Sf);
        if (has_a) { mm3(RbtRa, Sf, T1); axpy9(-4.0, T1, o + J_RQA, true); }
        mv3(Ra, f, t3); mtv3(Rb, t3, yf);
        skew3(yf, Syf);
        axpy9(4.0, Syf, o + J_RQB, true);
    }
    // rotational rows, multiplier nu3:  f_b = V Lb' E' La n ;  f_a = -V (n qoff qb* qa)
    {
        double n4[4] = {0, nu3[0], nu3[1], nu3[2]};
        double E[16], La[16], Lb[16], Rn[16], M1[16], M2[16], M3[16];
        Rmat4(r.qoc, E); Lmat4(qa, La); Lmat4(qb, Lb); Rmat4(n4, Rn);
        double Et[16];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) Et[i * 4 + j] = E[j * 4 + i];
        mm4(Lb, true, Et, M1); mm4(M1, false, La, M2);
        if (has_a) { mm4(M2, false, Rn, M3); vblk(M3, 1.0, o + J_RQA, true); }
        double y4[4];
        for (int i = 0; i < 4; i++) y4[i] = M2[i * 4] * n4[0] + M2[i * 4 + 1] * n4[1] + M2[i * 4 + 2] * n4[2] + M2[i * 4 + 3] * n4[3];
        Rmat4(y4, M3);
        vblk(M3, -1.0, o + J_RQB, true);
        if (has_a) {
            double qoff[4] = {r.qoc[0], -r.qoc[1], -r.qoc[2], -r.qoc[3]}, qbc[4] = {qb[0], -qb[1], -qb[2], -qb[3]}, t1[4], t2[4], t3q[4];
            qmul(n4, qoff, t1); qmul(qbc, qa, t2); qmul(t1, t2, t3q);
            Lmat4(t3q, M3);
            vblk(M3, -1.0, o + J_PRQA, true);
            double L1[16], R2[16];
            Lmat4(t1, L1); Rmat4(t2, R2);
            mm4(L1, false, R2, M3);
            vblk(M3, 1.0, o + J_PRQB, true);
        }
    }
}

HD void ph_lin_joint(int t, int nb, const Lay& Y, int JB, double* L, const LaneRegs& r) {
    if (t >= nb) return;
    const int a = r.parent;
    const bool has_a = a >= 0;
    const double X0[3] = {0, 0, 0};
    const double* xb = L + Y.Z + 13 * t;
    lin_joint_core(L + JB + LJB * t, r, has_a, has_a ? L + Y.Z + 13 * a : X0, has_a ? L + Y.Z + 13 * a + 3 : QID_, xb, xb + 3, L + Y.LAM + 5 * t, L[Y.UJ + t]);
}

struct LinOut {
    double *A, *Bu, *Bl, *G;   // this knot's matrices (row major), user body order
    int mx, mu, ml;
};

HD void put3(double* A, int ld, int r0, int c0, const double* B) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A[(size_t)(r0 + i) * ld + c0 + j] = B[i * 3 + j];
}

// per-body maps shared by the A/Bu/Bl rows:  dw+ = Dinv * rhs ; dqt+ = N dw+ ; dv+ = (dt/m) * ft ; dx+ = dt dv+
HD void body_cols(double* M, int ld, int r0, int c0, const double* FT, const double* FR, const double* Dinv, const double* N, double dtm,
                  double dt) {
    // FT, FR: 3x3 (may be null = zero).  writes the 12x3 block at (r0, c0)
    double dv[9], dw[9], dq[9], dx[9];
    for (int i = 0; i < 9; i++) { dv[i] = FT ? dtm * FT[i] : 0.0; dx[i] = dt * dv[i]; }
    if (FR) { mm3(Dinv, FR, dw); mm3(N, dw, dq); } else for (int i = 0; i < 9; i++) { dw[i] = 0; dq[i] = 0; }
    put3(M, ld, r0, c0, dx); put3(M, ld, r0 + 3, c0, dv); put3(M, ld, r0 + 6, c0, dq); put3(M, ld, r0 + 9, c0, dw);
}

// L2: body t writes its 12 rows of A (columns of itself, its parent and its child)
HD void ph_lin_rows_A(int t, int nb, const Lay& Y, int JB, const double* L, const LaneRegs& r, const MechDev* M, const LinOut& O) {
    if (t >= nb) return;
    const double dt = M->dt, dtm = dt / r.m;
    const double* Dinv = L + Y.DINV + 9 * t;
    double N[9];
    {
        const double* w2n = L + Y.S + 6 * t + 3;
        make_N(w2n, sqrt(4.0 / (dt * dt) - (w2n[0] * w2n[0] + w2n[1] * w2n[1] + w2n[2] * w2n[2])), dt, N);
    }
    const double* own = L + JB + LJB * t;
    const int a = r.parent, nch = M->nchild[t];
    const int rb = 12 * M->perm[t];
    double* A = O.A;
    const int ld = O.mx;
    // own columns
    {
        const int cb = rb;
        double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        // x columns: dx+ = I (+ via FR[x_b] from the child joint)
        double frx[9];
        for (int i = 0; i < 9; i++) frx[i] = 0.0;
        for (int ci = 0; ci < nch; ci++) { const double* ch = L + JB + LJB * M->child[t][ci]; for (int i = 0; i < 9; i++) frx[i] += ch[J_PRXA + i]; }
        body_cols(A, ld, rb, cb, nullptr, frx, Dinv, N, dtm, dt);
        for (int i = 0; i < 3; i++) A[(size_t)(rb + i) * ld + cb + i] += 1.0;
        // v columns: dv+ = I, dx+ = dt I
        double dtI[9] = {dt, 0, 0, 0, dt, 0, 0, 0, dt};
        put3(A, ld, rb, cb + 3, dtI); put3(A, ld, rb + 3, cb + 3, I3); put3(A, ld, rb + 6, cb + 3, Z9); put3(A, ld, rb + 9, cb + 3, Z9);
        // qtilde columns: FT from the child joint (PTQ), FR from own joint (RQB) and child joint (PRQA); plus the kinematic term Eqq
        double ftq[9], frq[9];
        for (int i = 0; i < 9; i++) { ftq[i] = 0.0; frq[i] = own[J_RQB + i]; }
        for (int ci = 0; ci < nch; ci++) { const double* ch = L + JB + LJB * M->child[t][ci]; for (int i = 0; i < 9; i++) { ftq[i] += ch[J_PTQ + i]; frq[i] += ch[J_PRQA + i]; } }
        body_cols(A, ld, rb, cb + 6, ftq, frq, Dinv, N, dtm, dt);
        const double* w2 = L + Y.S + 6 * t + 3;
        double sq2 = sqrt(4.0 / (dt * dt) - (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2]));
        double wq[4] = {0.5 * dt * sq2, 0.5 * dt * w2[0], 0.5 * dt * w2[1], 0.5 * dt * w2[2]}, Rw[9];
        rotmat(wq, Rw);   // d qt+/d qt = V L(wq)' R(wq) V' = R(wq)'
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) A[(size_t)(rb + 6 + i) * ld + cb + 6 + j] += Rw[j * 3 + i];
        // w columns: dw+ = Dinv dPsi/dw ; dqt+ = N dw+
        const double* w1 = L + Y.Z + 13 * t + 10;
        double sq1 = sqrt(4.0 / (dt * dt) - (w1[0] * w1[0] + w1[1] * w1[1] + w1[2] * w1[2]));
        double Jw1[3], S[9], SJ[9], Psi[9];
        mv3(r.J, w1, Jw1);
        S[0] = sq1; S[1] = w1[2]; S[2] = -w1[1]; S[3] = -w1[2]; S[4] = sq1; S[5] = w1[0]; S[6] = w1[1]; S[7] = -w1[0]; S[8] = sq1;   // sq1 I - [w1]x
        mm3(S, r.J, SJ);
        double Sj[9];
        skew3(Jw1, Sj);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Psi[i * 3 + j] = SJ[i * 3 + j] + Sj[i * 3 + j] - Jw1[i] * w1[j] / sq1;
        body_cols(A, ld, rb, cb + 9, nullptr, Psi, Dinv, N, dtm, dt);
    }
    if (a >= 0) {   // parent columns: only through the own joint, x_a has no effect on body b's forces
        const int ca = 12 * M->perm[a];
        body_cols(A, ld, rb, ca + 6, own + J_TQA, own + J_RQA, Dinv, N, dtm, dt);
    }
    for (int ci = 0; ci < nch; ci++) {   // child columns: through each child joint (this body is its parent)
        const int c = M->child[t][ci];
        const double* ch = L + JB + LJB * c;
        const int cc = 12 * M->perm[c];
        double nfrx[9];
        for (int i = 0; i < 9; i++) nfrx[i] = -ch[J_PRXA + i];
        body_cols(A, ld, rb, cc, nullptr, nfrx, Dinv, N, dtm, dt);
        body_cols(A, ld, rb, cc + 6, nullptr, ch + J_PRQB, Dinv, N, dtm, dt);
    }
}

HD void body_col1(double* M, int ld, int r0, int col, const double* ft, const double* fr, const double* Dinv, const double* N, double dtm,
                  double dt, bool add) {
    double dw[3], dq[3];
    mv3(Dinv, fr, dw); mv3(N, dw, dq);
    for (int i = 0; i < 3; i++) {
        double dv = dtm * ft[i];
        double* p0 = M + (size_t)(r0 + i) * ld + col;
        double* p1 = M + (size_t)(r0 + 3 + i) * ld + col;
        double* p2 = M + (size_t)(r0 + 6 + i) * ld + col;
        double* p3 = M + (size_t)(r0 + 9 + i) * ld + col;
        if (add) { *p0 += dt * dv; *p1 += dv; *p2 += dq[i]; *p3 += dw[i]; } else { *p0 = dt * dv; *p1 = dv; *p2 = dq[i]; *p3 = dw[i]; }
    }
}

// L3: body t writes its rows of Bl (columns of its own joint and of its child joint), Bu and joint t its rows of G
HD void ph_lin_rows_B(int t, int nb, const Lay& Y, const double* L, const LaneRegs& r, const MechDev* M, const int* cj, const LinOut& O) {
    if (t >= nb) return;
    const double dt = M->dt, dtm = dt / r.m;
    const double* Dinv = L + Y.DINV + 9 * t;
    double N[9];
    {
        const double* w2n = L + Y.S + 6 * t + 3;
        make_N(w2n, sqrt(4.0 / (dt * dt) - (w2n[0] * w2n[0] + w2n[1] * w2n[1] + w2n[2] * w2n[2])), dt, N);
    }
    const int rb = 12 * M->perm[t];
    const int nch = M->nchild[t];
    // the joint of link l is the caller's joint jperm(l): constraint rows keep the caller's joint numbering
    for (int side = 0; side <= nch; side++) {   // own joint (this body is the child), then every child joint (this body is the parent)
        const int j = side ? M->child[t][side - 1] : t;
        const double* Gk = L + (side ? Y.GKA : Y.GKB) + BLK * j;
        for (int row = 0; row < 5; row++)
            body_col1(O.Bl, O.ml, rb, 5 * M->jperm[j] + row, Gk + 6 * row, Gk + 6 * row + 3, Dinv, N, dtm, dt, false);
    }
    // Bu: input of controlled link cj[i] acts on its child (that link's body) and on its parent body
    for (int i = 0; i < O.mu; i++) {
        int j = cj[i];
        double ft[3] = {0, 0, 0}, fr[3] = {0, 0, 0};
        bool hit = false;
        if (j == t) {   // this body is the child
            const double* qa = (r.parent >= 0) ? L + Y.Z + 13 * r.parent + 3 : QID_;
            double Ra[9], Rb[9], Raa[3], yb[3];
            rotmat(qa, Ra); rotmat(L + Y.Z + 13 * t + 3, Rb);
            mv3(Ra, r.axis, Raa); mtv3(Rb, Raa, yb);
            if (r.type == 1) { double cr[3]; cross3(r.p2, yb, cr); for (int q = 0; q < 3; q++) { ft[q] = Raa[q]; fr[q] = 2.0 * cr[q]; } }
            else for (int q = 0; q < 3; q++) fr[q] = 2.0 * yb[q];
            hit = true;
        } else if (j >= 0 && j < nb && M->parent[j] == t) {   // this body is the parent of the controlled joint
            const int c = j;
            double Rb[9], Raa[3];
            rotmat(L + Y.Z + 13 * t + 3, Rb);
            const double* ax = M->axis[c];
            mv3(Rb, ax, Raa);
            if (M->type[c] == 1) { double cr[3]; cross3(M->p1[c], ax, cr); for (int q = 0; q < 3; q++) { ft[q] = -Raa[q]; fr[q] = -2.0 * cr[q]; } }
            else for (int q = 0; q < 3; q++) fr[q] = -2.0 * ax[q];
            hit = true;
        }
        if (hit) body_col1(O.Bu, O.mu, rb, i, ft, fr, Dinv, N, dtm, dt, true);
    }
    // G rows of joint t at the next knot
    {
        const int a = r.parent;
        const double X0[3] = {0, 0, 0};
        const double* pa = (a >= 0) ? L + Y.XQ + 7 * a : nullptr;
        const double* pb = L + Y.XQ + 7 * t;
        double g[5], Ba[30], Bb[30];
        joint_eval<true>(r, pa ? pa : X0, pa ? pa + 3 : QID_, pb, pb + 3, a >= 0, 1.0, 1.0, nullptr, nullptr, g, Ba, Bb);
        for (int row = 0; row < 5; row++) {
            double* Gr = O.G + (size_t)(5 * M->jperm[t] + row) * O.mx;
            for (int q = 0; q < 3; q++) {
                Gr[12 * M->perm[t] + q] = Bb[row * 6 + q];
                Gr[12 * M->perm[t] + 6 + q] = Bb[row * 6 + 3 + q];
                if (a >= 0) { Gr[12 * M->perm[a] + q] = Ba[row * 6 + q]; Gr[12 * M->perm[a] + 6 + q] = Ba[row * 6 + 3 + q]; }
            }
        }
    }
}

}  // namespace cclqr
