// cclqr_dev.h -- device-side data model and the per-lane "phase" functions of the rollout kernel.
//
// Execution model (DESIGN.md "rollout kernel"): one mechanism instance is owned by a group of G lanes
// (G = 16/32/64) of ONE wavefront; all per-instance data lives in LDS; a step is a sequence of phases,
// each phase a set of independent tasks spread over the group's lanes, separated by wave barriers.
// Lane t < nb permanently owns link t (= body t and the joint that hangs it off its parent), so the
// link's constants and per-step invariants stay in that lane's registers.
//
// The linear solve is NOT the reference's generic 6x6/5x5 tree LDU: bodies are eliminated first in
// closed form (D_b = blkdiag(m/dt I, D_R 3x3)), which leaves a block-tridiagonal 5x5 system in the
// multipliers along each chain.  Same linear system, same solution up to rounding.
//
// Every function here is __host__ __device__ so tests/emu can run the identical arithmetic serially
// on the CPU (test infrastructure only; the product never executes these on the host).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define CCLQR_MAXL 64          // links per mechanism supported by the device path (chains: one lane per link of a wavefront; branching trees: 32, cclqr_treereg.h)
#define CCLQR_MAXK 4           // child joints per body (general trees)
#define CCLQR_MAXP 48          // sibling pairs (joints that share their parent body)
#define CCLQR_MAXI 8           // joints around one body of a closed-loop mechanism
#define HD __host__ __device__ __forceinline__

namespace cclqr {

// ---- mechanism tables in device memory (internal link order = chain by chain, root to leaf) ----
struct MechDev {
    int nb;
    double dt, g;
    int parent[CCLQR_MAXL];   // parent link, -1 = origin
    int childl[CCLQR_MAXL];   // the single child link (forest of chains), -1 = leaf
    int rotmask[CCLQR_MAXL];  // bit r set: constraint row r is a rotational row
    int type[CCLQR_MAXL];     // 0 revolute, 1 prismatic
    int perm[CCLQR_MAXL];     // user body index of link l
    int jperm[CCLQR_MAXL];    // user joint index of link l's joint
    // links are numbered chain by chain (parent[l] == l-1 inside a chain): bit l of start_mask / end_mask marks the
    // link attached to the origin / the leaf of its chain, so the hot phases need no table lookups
    unsigned long long start_mask, end_mask;       // (64 links)
    int nchains, chain_start[CCLQR_MAXL], chain_len[CCLQR_MAXL];
    double m[CCLQR_MAXL], J[CCLQR_MAXL][9];
    double p1[CCLQR_MAXL][3], p2[CCLQR_MAXL][3], axis[CCLQR_MAXL][3], qoc[CCLQR_MAXL][4]; // qoc = conj(qoffset)
    double sel[CCLQR_MAXL][5][3]; // row r of the joint = sel[r] . (translational | rotational 3-vector)
    // ---- general trees (tree != 0: some body carries several child joints; links are numbered depth-first, first child = l+1).
    // The Schur complement on the multipliers then couples the joints around a body pairwise (siblings), and the solve is the
    // table-driven elimination below instead of the chains' two-front sweep.
    int tree;        // 0: forest of chains; otherwise 8 x (largest neighbour count in the elimination) = lanes one elimination step needs
    int nchild[CCLQR_MAXL], child[CCLQR_MAXL][CCLQR_MAXK];   // child links of body l
    int npairs, pair_i[CCLQR_MAXP], pair_j[CCLQR_MAXP];      // sibling pairs i < j: LDS blocks SS[2p] = S_ij, SS[2p+1] = S_ji
    // elimination program, links in reverse order: when l is eliminated its remaining neighbours are x_0 .. x_{nn-1}
    // (the parent joint and the siblings with a smaller index); all offsets are LDS offsets of 5x5 blocks in the instance layout
    int el_nn[CCLQR_MAXL], el_x[CCLQR_MAXL][CCLQR_MAXK];
    int el_lx[CCLQR_MAXL][CCLQR_MAXK];                 // S_{l,x_g}   (holds Z_{l,x_g} = S_ll^-1 S_{l,x_g} afterwards)
    int el_xl[CCLQR_MAXL][CCLQR_MAXK];                 // S_{x_g,l}
    int el_t[CCLQR_MAXL][CCLQR_MAXK][CCLQR_MAXK];      // [g'][g]: S_{x_g',x_g}
    // ---- closed loops (loop != 0; cclqr_loop.h): nj >= nb joints, bodies and joints in the caller's order (perm, jperm = identity);
    // m, J are indexed by body; parent (parent BODY or -1), type, p1, p2, axis, qoc, sel, rotmask by joint; jchild = child body
    int loop, nj;
    int jchild[CCLQR_MAXL];
    int inc_n[8], inc_j[8][CCLQR_MAXI], inc_side[8][CCLQR_MAXI];   // joints around body b; side 0: b is the child, 1: the parent
};

// ---- controller tables in device memory ----
struct CtrlDev {
    int mu, nK, N, nsp;       // N <= 0: infinite horizon
    int cj[CCLQR_MAXL];       // controlled links (internal index)
    const double* K;          // [nK][mu][12 nb], columns in internal link order
    const double* zd;         // [nsp][nb][13] internal link order
    const double* Fd;         // [nsp][mu]
    double fric[CCLQR_MAXL];  // viscous joint friction per link
    int has_fric;
    double noise_scale;
    // PID on 1-DoF joints (pid.jl): per link, pid_on[l] != 0
    int noise_philox;             // 1: counter-based noise (Philox-4x32-10 + Box-Muller), SURVEY 8d
    unsigned noise_key0;
    int has_pid, pid_on[CCLQR_MAXL];
    double pid_P[CCLQR_MAXL], pid_I[CCLQR_MAXL], pid_D[CCLQR_MAXL], pid_goal[CCLQR_MAXL];
    // per-instance controller tables (n_ctrl > 1): instance with global index n reads K + n K_stride, zd + n zd_stride, Fd + n Fd_stride
    // (strides in doubles; all 0 when one table is shared, the reference's case)
    long long K_stride, zd_stride, Fd_stride;
    int n_ctrl;
};

// ---- LDS layout of one instance (offsets in doubles) ----
// NB holds N(w+) D_R^-1 (so that the joint evaluation emits W = G_v D^-1 directly)
struct Lay {
    int Z, S, ST, LAM, LT, DS, DL, XQ, NB, DINV, DTM, D, G, R, GKA, GKB, GVA, GVB, SJJ, SJP, SPJ, UJ, C, CD, SS, DZ, total;
};
#define BLK 31   // stride of a 5x6 block (30 used; odd stride keeps ds_read_b64 conflict-free across lanes)
HD Lay make_layout(int nb, int nss = 0) {   // nss: sibling blocks (2 per sibling pair), general trees only
    Lay L; int o = 0;
    L.Z = o; o += 13 * nb;   L.S = o; o += 6 * nb;   L.ST = o; o += 6 * nb;
    L.LAM = o; o += 5 * nb;  L.LT = o; o += 5 * nb;  L.DS = o; o += 6 * nb;  L.DL = o; o += 5 * nb;
    L.XQ = o; o += 7 * nb;   L.NB = o; o += 9 * nb;  L.DINV = o; o += 9 * nb; L.DTM = o; o += nb;
    L.D = o; o += 6 * nb;    L.G = o; o += 5 * nb;   L.R = o; o += 5 * nb;
    L.GKA = o; o += BLK * nb; L.GKB = o; o += BLK * nb; L.GVA = o; o += BLK * nb; L.GVB = o; o += BLK * nb;
    L.SJJ = o; o += 25 * nb; L.SJP = o; o += 25 * nb; L.SPJ = o; o += 25 * nb;
    L.UJ = o; o += nb;
    L.C = o; o += 6 * nb;    // G_k' lambda at the accepted point
    L.CD = o; o += 6 * nb;   // G_k' dlambda of the current Newton step
    L.SS = o; o += 25 * nss;
    L.DZ = L.GVA;            // control error aliases the (dead at control time) Gv storage: 12 nb <= 31 nb
    L.total = o | 1;         // odd instance stride
    return L;
}

// ---- per-lane registers that persist across phases ----
struct LaneRegs {
    // link constants (lane t < nb owns link t)
    double m, J[9], p1[3], p2[3], sel[5][3], qoc[4], axis[3];
    int parent, childl, rotmask, type;
    // per-step invariants of the owned body: cT = m(-v/dt + ezg) - F ; cR = -(sq1 I - [w1]x) J w1 - 2 tau
    double cT[3], cR[3];
    // PID state of the owned joint: integratederrors / lasterrors (pid.jl:10-11)
    double pid_int, pid_last;
};

// ------------------------------------------------------------------ small algebra
HD void qmul(const double* a, const double* b, double* o) {
    double s = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    o[0] = s; o[1] = x; o[2] = y; o[3] = z;
}
HD void rotmat(const double* q, double* R) {
    double s = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = s * s + x * x - y * y - z * z; R[1] = 2 * (x * y - s * z); R[2] = 2 * (x * z + s * y);
    R[3] = 2 * (x * y + s * z); R[4] = s * s - x * x + y * y - z * z; R[5] = 2 * (y * z - s * x);
    R[6] = 2 * (x * z - s * y); R[7] = 2 * (y * z + s * x); R[8] = s * s - x * x - y * y + z * z;
}
HD void mv3(const double* R, const double* p, double* o) {   // o = R p
    double a = R[0] * p[0] + R[1] * p[1] + R[2] * p[2], b = R[3] * p[0] + R[4] * p[1] + R[5] * p[2],
           c = R[6] * p[0] + R[7] * p[1] + R[8] * p[2];
    o[0] = a; o[1] = b; o[2] = c;
}
HD void mtv3(const double* R, const double* p, double* o) {  // o = R' p
    double a = R[0] * p[0] + R[3] * p[1] + R[6] * p[2], b = R[1] * p[0] + R[4] * p[1] + R[7] * p[2],
           c = R[2] * p[0] + R[5] * p[1] + R[8] * p[2];
    o[0] = a; o[1] = b; o[2] = c;
}
HD void cross3(const double* a, const double* b, double* o) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
HD void mm3(const double* A, const double* B, double* C) {   // C = A B (3x3 row major)
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
HD void mtm3(const double* A, const double* B, double* C) {  // C = A' B
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
// reciprocal for pivots: v_rcp_f64 seed + two Newton steps (~1 ulp) instead of the ~12-instruction IEEE division sequence
HD double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
#else
    return 1.0 / x;
#endif
}
HD void inv3(const double* A, double* Ai) {
    double c0 = A[4] * A[8] - A[5] * A[7], c1 = A[5] * A[6] - A[3] * A[8], c2 = A[3] * A[7] - A[4] * A[6];
    double id = fast_rcp(A[0] * c0 + A[1] * c1 + A[2] * c2);
    Ai[0] = c0 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c1 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c2 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// ------------------------------------------------------------------ counter-based standard normal (Philox-4x32-10 + Box-Muller)
HD double philox_normal(unsigned key0, unsigned long long instance, int k) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    unsigned c0 = (unsigned)(k - 1), c1 = 0u, c2 = 0u, c3 = 0u, k0 = key0, k1 = (unsigned)instance;
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)M0 * c0, p1 = (unsigned long long)M1 * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    const double u1 = ((double)c0 + 0.5) * (1.0 / 4294967296.0), u2 = ((double)c1 + 0.5) * (1.0 / 4294967296.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

// ------------------------------------------------------------------ link constants -> lane registers
HD void lane_load_consts(LaneRegs& r, const MechDev* M, int l) {
    r.m = M->m[l];
    for (int i = 0; i < 9; i++) r.J[i] = M->J[l][i];
    for (int i = 0; i < 3; i++) { r.p1[i] = M->p1[l][i]; r.p2[i] = M->p2[l][i]; r.axis[i] = M->axis[l][i]; }
    for (int i = 0; i < 4; i++) r.qoc[i] = M->qoc[l][i];
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 3; j++) r.sel[i][j] = M->sel[l][i][j];
    r.parent = M->parent[l]; r.childl = M->childl[l]; r.rotmask = M->rotmask[l]; r.type = M->type[l];
    for (int i = 0; i < 3; i++) { r.cT[i] = 0; r.cR[i] = 0; }
    r.pid_int = 0.0; r.pid_last = 0.0;
}

// ------------------------------------------------------------------ joint: g and d g/d(x, phi) for both bodies
// translational  R(qa)'(xb + R(qb) p2 - xa) - p1 ; rotational vec(qa^-1 qb qoff^-1); rows picked by sel/rotmask.
// Ba/Bb (5x6 row major, row stride 6): [dg/dx * sx_side , (dg/dphi) * Nside]  where N = 3x3 (nullptr = identity).
// JAC = false evaluates g only (rejected line-search trials need no Jacobian).
// Row kinds of the two joint types (Revolute = 3 translational + 2 rotational rows, Prismatic = 2 + 3): rows 0,1 are always
// translational, rows 3,4 always rotational, only row 2 depends on the joint -- so in the unrolled row loops four of the five
// rows fold to one kind at compile time (no selects, and only that kind's Jacobian products are formed).
HD bool row_is_rotational(int rotmask, int row) { return row >= 3 ? true : (row <= 1 ? false : ((rotmask >> 2) & 1) != 0); }

template <bool JAC>
HD void joint_eval(const LaneRegs& r, const double* xa, const double* qa, const double* xb, const double* qb, bool has_a,
                   double sxa, double sxb, const double* Na_, const double* Nb_, double* g, double* Ba, double* Bb) {
    // copy every input into registers first: the outputs go to LDS too, and a store that may alias would otherwise
    // serialise the remaining input loads (one LDS round trip each)
    const double xa_[3] = {xa[0], xa[1], xa[2]}, qa_[4] = {qa[0], qa[1], qa[2], qa[3]};
    const double xb_[3] = {xb[0], xb[1], xb[2]}, qb_[4] = {qb[0], qb[1], qb[2], qb[3]};
    double Na[9], Nb[9];
    const bool hasNa = JAC && Na_ != nullptr, hasNb = JAC && Nb_ != nullptr;
    if (JAC) {
#pragma unroll
        for (int i = 0; i < 9; i++) { Na[i] = hasNa ? Na_[i] : 0.0; Nb[i] = hasNb ? Nb_[i] : 0.0; }
    }
    xa = xa_; qa = qa_; xb = xb_; qb = qb_;
    double Ra[9], Rb[9], rp[3], w[3], RaTw[3], gT[3];
    rotmat(qa, Ra); rotmat(qb, Rb);
    mv3(Rb, r.p2, rp);
    for (int i = 0; i < 3; i++) w[i] = xb[i] + rp[i] - xa[i];
    mtv3(Ra, w, RaTw);
    for (int i = 0; i < 3; i++) gT[i] = RaTw[i] - r.p1[i];
    double qac[4] = {qa[0], -qa[1], -qa[2], -qa[3]}, rel[4], e[4];
    qmul(qac, qb, rel);
    qmul(rel, r.qoc, e);
    if (!JAC) {
#pragma unroll
        for (int row = 0; row < 5; row++) {
            const bool rot = (r.rotmask >> row) & 1;
            double s0 = r.sel[row][0], s1 = r.sel[row][1], s2 = r.sel[row][2];
            g[row] = rot ? (s0 * e[1] + s1 * e[2] + s2 * e[3]) : (s0 * gT[0] + s1 * gT[1] + s2 * gT[2]);
        }
        return;
    }
    // child side 3x3s:  XT_b = Ra' ; PT_b = -2 Ra' Rb [p2]x ; PR_b = V L(rel) R(qoc) V'
    double RaTRb[9], PTb[9], PRb[9];
    mtm3(Ra, Rb, RaTRb);
    {   // M [p]x ;  [p]x = [0 -pz py; pz 0 -px; -py px 0]
        const double* p = r.p2;
        for (int i = 0; i < 3; i++) {
            double a = RaTRb[i * 3], b = RaTRb[i * 3 + 1], c = RaTRb[i * 3 + 2];
            PTb[i * 3 + 0] = -2.0 * (b * p[2] - c * p[1]);
            PTb[i * 3 + 1] = -2.0 * (c * p[0] - a * p[2]);
            PTb[i * 3 + 2] = -2.0 * (a * p[1] - b * p[0]);
        }
    }
    {   // rows 1..3, cols 1..3 of L(rel) R(qoc)
        double s = rel[0], x = rel[1], y = rel[2], z = rel[3];
        double os = r.qoc[0], ox = r.qoc[1], oy = r.qoc[2], oz = r.qoc[3];
        double Lr[3][4] = {{x, s, -z, y}, {y, z, s, -x}, {z, -y, x, s}};
        double Rc[4][3] = {{-ox, -oy, -oz}, {os, oz, -oy}, {-oz, os, ox}, {oy, -ox, os}};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) PRb[i * 3 + j] = Lr[i][0] * Rc[0][j] + Lr[i][1] * Rc[1][j] + Lr[i][2] * Rc[2][j] + Lr[i][3] * Rc[3][j];
    }
    // parent side 3x3s: XT_a = -Ra' ; PT_a = 2 [Ra'w]x ; PR_a = -(e_s I - [e_v]x)
    double PTa[9] = {0, -2 * RaTw[2], 2 * RaTw[1], 2 * RaTw[2], 0, -2 * RaTw[0], -2 * RaTw[1], 2 * RaTw[0], 0};
    double PRa[9] = {-e[0], -e[3], e[2], e[3], -e[0], -e[1], -e[2], e[1], -e[0]};
#pragma unroll
    for (int row = 0; row < 5; row++) {
        const bool rot = row_is_rotational(r.rotmask, row);
        double s0 = r.sel[row][0], s1 = r.sel[row][1], s2 = r.sel[row][2];
        double vT = s0 * gT[0] + s1 * gT[1] + s2 * gT[2];
        double vR = s0 * e[1] + s1 * e[2] + s2 * e[3];
        g[row] = rot ? vR : vT;
        double xb3[3], pb3[3], pa3[3];
        for (int c = 0; c < 3; c++) {
            // sel' Ra'  = (Ra sel)' : component c = sum_i sel[i] Ra[c][i]
            double xt = s0 * Ra[c * 3] + s1 * Ra[c * 3 + 1] + s2 * Ra[c * 3 + 2];
            xb3[c] = rot ? 0.0 : xt;
            double ptb = s0 * PTb[c] + s1 * PTb[3 + c] + s2 * PTb[6 + c];
            double prb = s0 * PRb[c] + s1 * PRb[3 + c] + s2 * PRb[6 + c];
            pb3[c] = rot ? prb : ptb;
            double pta = s0 * PTa[c] + s1 * PTa[3 + c] + s2 * PTa[6 + c];
            double pra = s0 * PRa[c] + s1 * PRa[3 + c] + s2 * PRa[6 + c];
            pa3[c] = rot ? pra : pta;
        }
        for (int c = 0; c < 3; c++) {
            Bb[row * 6 + c] = xb3[c] * sxb;
            Ba[row * 6 + c] = has_a ? -xb3[c] * sxa : 0.0;
            double nb_ = hasNb ? (pb3[0] * Nb[c] + pb3[1] * Nb[3 + c] + pb3[2] * Nb[6 + c]) : pb3[c];
            double na_ = hasNa ? (pa3[0] * Na[c] + pa3[1] * Na[3 + c] + pa3[2] * Na[6 + c]) : pa3[c];
            Bb[row * 6 + 3 + c] = nb_;
            Ba[row * 6 + 3 + c] = has_a ? na_ : 0.0;
        }
    }
}

// ------------------------------------------------------------------ phases (t = lane index inside the group)
static const double QID_[4] = {1.0, 0.0, 0.0, 0.0};

// C1: control error of link t into DZ (order x, v, qtilde, w: lqr.jl:92-95) and passive friction into UJ
HD void ph_control_error(int t, int nb, const Lay& Y, double* L, const LaneRegs& r, const CtrlDev* C, const double* zd) {
    if (t >= nb) return;
    const double* z = L + Y.Z + 13 * t;
    const double* d = zd + 13 * t;
    double qdc[4] = {d[3], -d[4], -d[5], -d[6]}, qe[4];
    qmul(qdc, z + 3, qe);   // qd \ q : raw vector part, no sign fix, no factor 2 (lqr.jl:101-102)
    double* dz = L + Y.DZ + 12 * t;
    for (int i = 0; i < 3; i++) {
        dz[i] = z[i] - d[i]; dz[3 + i] = z[7 + i] - d[7 + i]; dz[6 + i] = qe[1 + i]; dz[9 + i] = z[10 + i] - d[10 + i];
    }
    double u = 0.0;
    if (C->has_fric && C->fric[t] != 0.0) {   // trackingLQR_triple_cartpole.jl:93-101
        int a = r.parent;
        double rel;
        if (r.type == 0) {
            rel = r.axis[0] * z[10] + r.axis[1] * z[11] + r.axis[2] * z[12];
            if (a >= 0) { const double* za = L + Y.Z + 13 * a; rel -= r.axis[0] * za[10] + r.axis[1] * za[11] + r.axis[2] * za[12]; }
        } else {
            double dv[3], dva[3], Ra[9];
            const double* za = (a >= 0) ? L + Y.Z + 13 * a : nullptr;
            for (int i = 0; i < 3; i++) dv[i] = z[7 + i] - (za ? za[7 + i] : 0.0);
            rotmat(za ? za + 3 : QID_, Ra);
            mtv3(Ra, dv, dva);
            rel = r.axis[0] * dva[0] + r.axis[1] * dva[1] + r.axis[2] * dva[2];
        }
        u = -C->fric[t] * rel;
    }
    L[Y.UJ + t] = u;
}

// C1b: control_pid!(mechanism, pid, k) for the joint of link t (pid.jl:69-88): minimalCoordinates (angle about / offset along
// the joint axis), wrapped error for revolutes (pid.jl:43-57), u = P e + I int(e) + D de/dt added to the joint input
HD void ph_pid(int t, int nb, const Lay& Y, double* L, LaneRegs& r, const CtrlDev* C, double dt, bool first) {
    if (t >= nb || !C->pid_on[t]) return;
    const int a = r.parent;
    const double X0[3] = {0, 0, 0};
    const double* za = (a >= 0) ? L + Y.Z + 13 * a : nullptr;
    const double* zb = L + Y.Z + 13 * t;
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

// C2: partial dot product  sum_{t, t+G, ...} K[i][.] * DZ[.]  (caller reduces over the group)
HD double ph_gain_partial(int t, int G, int nb, const Lay& Y, const double* L, const double* Krow) {
    double s = 0.0;
    for (int c = t; c < 12 * nb; c += G) s += Krow[c] * L[Y.DZ + c];
    return s;
}

// F1: joint inputs -> force/torque on the owned body, per-step invariants, solution guess (SURVEY 8a-bis 'Joint input')
template <bool TREE = false>
HD void ph_forces(int t, int nb, const Lay& Y, double* L, LaneRegs& r, const MechDev* M, bool owner = true) {
    if (t >= nb) return;
    const double dt = M->dt;
    const double* z = L + Y.Z + 13 * t;
    double F[3] = {0, 0, 0}, tau[3] = {0, 0, 0};
    double Rb[9];
    rotmat(z + 3, Rb);
    double u = L[Y.UJ + t];
    if (u != 0.0) {   // own joint: this body is the child
        double Ra[9], f[3] = {r.axis[0] * u, r.axis[1] * u, r.axis[2] * u}, fw[3], fb[3];
        rotmat(r.parent >= 0 ? L + Y.Z + 13 * r.parent + 3 : QID_, Ra);
        mv3(Ra, f, fw); mtv3(Rb, fw, fb);
        if (r.type == 1) { double c[3]; cross3(r.p2, fb, c); for (int i = 0; i < 3; i++) { F[i] += fw[i]; tau[i] += c[i]; } }
        else for (int i = 0; i < 3; i++) tau[i] += fb[i];
    }
    const int nch = TREE ? M->nchild[t] : (r.childl >= 0 ? 1 : 0);
    for (int ci = 0; ci < nch; ci++) {     // child joints: this body is the parent
        const int c = TREE ? M->child[t][ci] : r.childl;
        double uc = L[Y.UJ + c];
        if (uc != 0.0) {
            double f[3] = {M->axis[c][0] * uc, M->axis[c][1] * uc, M->axis[c][2] * uc};
            if (M->type[c] == 1) {
                double fw[3], cr[3];
                mv3(Rb, f, fw); cross3(M->p1[c], f, cr);
                for (int i = 0; i < 3; i++) { F[i] -= fw[i]; tau[i] -= cr[i]; }
            } else for (int i = 0; i < 3; i++) tau[i] -= f[i];
        }
    }
    const double* v1 = z + 7; const double* w1 = z + 10;
    double sq1 = sqrt(4.0 / (dt * dt) - (w1[0] * w1[0] + w1[1] * w1[1] + w1[2] * w1[2]));
    double Jw1[3], c1[3];
    mv3(r.J, w1, Jw1); cross3(w1, Jw1, c1);
    for (int i = 0; i < 3; i++) {
        r.cT[i] = r.m * (-v1[i] / dt + (i == 2 ? -M->g : 0.0)) - F[i];
        r.cR[i] = -(sq1 * Jw1[i] - c1[i]) - 2.0 * tau[i];
        if (owner) { L[Y.S + 6 * t + i] = v1[i]; L[Y.S + 6 * t + 3 + i] = w1[i]; }
    }
    if (owner) L[Y.DTM + t] = dt / r.m;
}

// F2: constraint Jacobians at the current knot (force mapping G_k)
HD void ph_knot_jac(int t, int nb, const Lay& Y, double* L, const LaneRegs& r) {
    if (t >= nb) return;
    int a = r.parent;
    const double X0[3] = {0, 0, 0};
    const double* za = (a >= 0) ? L + Y.Z + 13 * a : nullptr;
    const double* zb = L + Y.Z + 13 * t;
    double g[5];
    joint_eval<true>(r, za ? za : X0, za ? za + 3 : QID_, zb, zb + 3, a >= 0, 1.0, 1.0, nullptr, nullptr, g, L + Y.GKA + BLK * t, L + Y.GKB + BLK * t);
}

// N(w) = (dt^2/4)(sq I - [w]x + w w'/sq):  d phi+ = N d w+
HD void make_N(const double* w2, double sq2, double dt, double* N) {
    double k = 0.25 * dt * dt, isq = fast_rcp(sq2);
    N[0] = k * (sq2 + w2[0] * w2[0] * isq); N[1] = k * (w2[2] + w2[0] * w2[1] * isq);  N[2] = k * (-w2[1] + w2[0] * w2[2] * isq);
    N[3] = k * (-w2[2] + w2[1] * w2[0] * isq); N[4] = k * (sq2 + w2[1] * w2[1] * isq); N[5] = k * (w2[0] + w2[1] * w2[2] * isq);
    N[6] = k * (w2[1] + w2[2] * w2[0] * isq);  N[7] = k * (-w2[0] + w2[2] * w2[1] * isq); N[8] = k * (sq2 + w2[2] * w2[2] * isq);
}

// E1: body t at the trial solution s: next pose, residual d = dyn(s) - (C - alpha CD) where C = G_k' lambda, CD = G_k' dlambda
// (the constraint-force map is linear in lambda, so a trial point lambda - alpha dlambda needs no new G_k' product);
// with JAC also D_R^-1 and N D_R^-1.  Returns this lane's partial sum of squares of d.
template <bool JAC>
HD double ph_body_eval(int t, int nb, const Lay& Y, double* L, const LaneRegs& r, double dt, int s_off, double alpha) {
    if (t >= nb) return 0.0;
    double z[7], s[6], cf[6];
#pragma unroll
    for (int i = 0; i < 7; i++) z[i] = L[Y.Z + 13 * t + i];
#pragma unroll
    for (int i = 0; i < 6; i++) { s[i] = L[s_off + 6 * t + i]; cf[i] = L[Y.C + 6 * t + i] - alpha * L[Y.CD + 6 * t + i]; }
    const double* w2 = s + 3;
    double* xq = L + Y.XQ + 7 * t;
    for (int i = 0; i < 3; i++) xq[i] = z[i] + s[i] * dt;
    const double inv_dt = fast_rcp(dt), m_dt = r.m * inv_dt;     // one refined reciprocal instead of four IEEE divisions
    double sq2 = sqrt(4.0 * inv_dt * inv_dt - (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2]));
    double wb[4] = {0.5 * dt * sq2, 0.5 * dt * w2[0], 0.5 * dt * w2[1], 0.5 * dt * w2[2]};
    qmul(z + 3, wb, xq + 3);
    double Jw2[3], c2[3];
    mv3(r.J, w2, Jw2); cross3(w2, Jw2, c2);
    double* d = L + Y.D + 6 * t;
    double acc = 0.0;
    for (int i = 0; i < 3; i++) {
        double dT = m_dt * s[i] + r.cT[i] - cf[i], dR = sq2 * Jw2[i] + c2[i] + r.cR[i] - cf[3 + i];
        d[i] = dT; d[3 + i] = dR;
        acc += dT * dT + dR * dR;
    }
    if (!JAC) return acc;
    // D_R = (sq2 I + [w2]x) J - [J w2]x - (J w2) w2'/sq2
    double S[9] = {sq2, -w2[2], w2[1], w2[2], sq2, -w2[0], -w2[1], w2[0], sq2}, SJ[9], Dr[9], Di[9], N[9];
    mm3(S, r.J, SJ);
    double isq = fast_rcp(sq2);
    double Sj[9] = {0, -Jw2[2], Jw2[1], Jw2[2], 0, -Jw2[0], -Jw2[1], Jw2[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Dr[i * 3 + j] = SJ[i * 3 + j] - Sj[i * 3 + j] - Jw2[i] * w2[j] * isq;
    inv3(Dr, Di);
    for (int i = 0; i < 9; i++) L[Y.DINV + 9 * t + i] = Di[i];
    make_N(w2, sq2, dt, N);
    mm3(N, Di, L + Y.NB + 9 * t);
    return acc;
}

// E2: joint t at the next knot: g and W = G_v D^-1 = [X dt (dt/m), Phi N D_R^-1]  (JAC) or g only; returns |g_t|^2
template <bool JAC>
HD double ph_joint_eval(int t, int nb, const Lay& Y, double* L, const LaneRegs& r, double dt) {
    if (t >= nb) return 0.0;
    int a = r.parent;
    const double X0[3] = {0, 0, 0};
    const double* pa = (a >= 0) ? L + Y.XQ + 7 * a : nullptr;
    const double* pb = L + Y.XQ + 7 * t;
    double g[5];
    joint_eval<JAC>(r, pa ? pa : X0, pa ? pa + 3 : QID_, pb, pb + 3, a >= 0, (a >= 0) ? dt * L[Y.DTM + a] : 0.0, dt * L[Y.DTM + t],
                    (a >= 0) ? L + Y.NB + 9 * a : nullptr, L + Y.NB + 9 * t, g, L + Y.GVA + BLK * t, L + Y.GVB + BLK * t);
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 5; i++) { L[Y.G + 5 * t + i] = g[i]; acc += g[i] * g[i]; }
    return acc;
}


// F3 (once per step, after the knot Jacobians): C_b = Gk_b(own joint)' lambda_b + Gk_a(child joint)' lambda_child
HD void ph_force_map(int t, int G, int nb, const Lay& Y, double* L, unsigned long long end_mask) {
    for (int b = t; b < nb; b += G) {
        const bool has_c = !((end_mask >> b) & 1ull);
        double gb[30], lb[5], ga[30], lc[5];
#pragma unroll
        for (int i = 0; i < 30; i++) gb[i] = L[Y.GKB + BLK * b + i];
#pragma unroll
        for (int i = 0; i < 5; i++) lb[i] = L[Y.LAM + 5 * b + i];
        if (has_c) {
#pragma unroll
            for (int i = 0; i < 30; i++) ga[i] = L[Y.GKA + BLK * (b + 1) + i];
#pragma unroll
            for (int i = 0; i < 5; i++) lc[i] = L[Y.LAM + 5 * (b + 1) + i];
        } else {
#pragma unroll
            for (int i = 0; i < 30; i++) ga[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 5; i++) lc[i] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < 6; c++) {
            L[Y.C + 6 * b + c] = gb[c] * lb[0] + gb[6 + c] * lb[1] + gb[12 + c] * lb[2] + gb[18 + c] * lb[3] + gb[24 + c] * lb[4]
                               + ga[c] * lc[0] + ga[6 + c] * lc[1] + ga[12 + c] * lc[2] + ga[18 + c] * lc[3] + ga[24 + c] * lc[4];
            L[Y.CD + 6 * b + c] = 0.0;
        }
    }
}

HD double dot6(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }

// S2: Schur complement blocks and right-hand side.  Tasks (part, joint j, row) of ONE shape -- two 6-vectors against two
// 5x6 blocks -- so that every pass is branch-uniform:
//   part 0: (W_b[j], Gk_b[j]) + (W_a[j], Gk_a[j]) -> row of S_jj ;  r_j = g_j - W_b d_b - W_a d_a
//   part 1: (W_a[j], Gk_b[p]) -> row of S_jp ,  (W_b[p], Gk_a[j]) -> row of S_pj        (p = j-1, the parent link)
// Each task loads all its operands into registers, computes, then stores (no store between loads).
HD void ph_schur_s(int t, int G, int nb, const Lay& Y, double* L, unsigned long long start_mask) {
    for (int e = t; e < 10 * nb; e += G) {
        const int part = (e >= 5 * nb) ? 1 : 0;
        const int e5 = e - part * 5 * nb;
        const int j = e5 / 5, row = e5 - 5 * j;
        const bool has_p = !((start_mask >> j) & 1ull);
        if (part && !has_p) continue;
        const int p = has_p ? j - 1 : j;
        const int ov1 = (part ? Y.GVA + BLK * j : Y.GVB + BLK * j) + 6 * row;
        const int ov2 = (part ? Y.GVB + BLK * p : Y.GVA + BLK * j) + 6 * row;
        const int om1 = part ? Y.GKB + BLK * p : Y.GKB + BLK * j;
        const int om2 = Y.GKA + BLK * j;
        const int od1 = Y.D + 6 * j, od2 = Y.D + 6 * p;
        double v1[6], v2[6], m1[30], m2[30], d1[6], d2[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { v1[i] = L[ov1 + i]; v2[i] = L[ov2 + i]; d1[i] = L[od1 + i]; d2[i] = L[od2 + i]; }
#pragma unroll
        for (int i = 0; i < 30; i++) { m1[i] = L[om1 + i]; m2[i] = L[om2 + i]; }
        const double g0 = L[Y.G + 5 * j + row];
        const double use2 = (part || has_p) ? 1.0 : 0.0;     // part 0 on a chain root: no parent-side block
        double o1[5], o2[5];
#pragma unroll
        for (int col = 0; col < 5; col++) { o1[col] = dot6(v1, m1 + 6 * col); o2[col] = use2 * dot6(v2, m2 + 6 * col); }
        const double rr = g0 - dot6(v1, d1) - use2 * dot6(v2, d2);
        if (part == 0) {
#pragma unroll
            for (int col = 0; col < 5; col++) L[Y.SJJ + 25 * j + 5 * row + col] = o1[col] + o2[col];
            L[Y.R + 5 * j + row] = rr;
        } else {
#pragma unroll
            for (int col = 0; col < 5; col++) { L[Y.SJP + 25 * j + 5 * row + col] = o1[col]; L[Y.SPJ + 25 * j + 5 * row + col] = o2[col]; }
        }
    }
}

// ---- S3: block-tridiagonal solve along one chain [cs, cs+cn), swept from BOTH ends at once (twisted factorisation).
// Two "fronts" run the same instruction stream on different lanes: front 0 = lanes 0..4 eliminates from the leaf towards
// the middle link, front 1 = lanes 8..12 from the root towards the middle; lane (t & 7) owns one row.  Link numbering
// inside a chain: parent of l is l-1.  Block names: SJP[l] = S_{l,l-1}, SPJ[l] = S_{l-1,l}.
struct TriPlan { int cs, cn, nA, nB, mid, steps; };
HD TriPlan tri_plan(int cs, int cn) {
    TriPlan P;
    P.cs = cs; P.cn = cn;
    int rest = cn - 1;
    P.nB = (rest + 1) / 2 + ((rest % 2 == 0 && rest > 0) ? 1 : 0);   // top-down count; unequal counts keep the two last updates of the middle apart
    if (P.nB > rest) P.nB = rest;
    P.nA = rest - P.nB;                                                // bottom-up count
    P.mid = cs + P.nB;
    P.steps = P.nB > P.nA ? P.nB : P.nA;
    return P;
}
// LU (no pivoting; S is SPD-like) of the 5x5 block at L[off], all in registers with static indices
HD void lu5(const double* L, int off, double* A) {
#pragma unroll
    for (int i = 0; i < 25; i++) A[i] = L[off + i];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        double inv = fast_rcp(A[k * 5 + k]);
        A[k * 5 + k] = inv;
#pragma unroll
        for (int i = k + 1; i < 5; i++) {
            double f = A[i * 5 + k] * inv;
            A[i * 5 + k] = f;
#pragma unroll
            for (int j = k + 1; j < 5; j++) A[i * 5 + j] -= f * A[k * 5 + j];
        }
    }
}
// solve (packed LU in A) for one right-hand side, in place
HD void lu5_solve(const double* A, double* b) {
#pragma unroll
    for (int i = 1; i < 5; i++) {
#pragma unroll
        for (int k = 0; k < i; k++) b[i] -= A[i * 5 + k] * b[k];
    }
#pragma unroll
    for (int i = 4; i >= 0; i--) {
#pragma unroll
        for (int k = i + 1; k < 5; k++) b[i] -= A[i * 5 + k] * b[k];
        b[i] *= A[i * 5 + i];
    }
}
// elimination step i: front 0 folds link l = cs+cn-1-i into q = l-1, front 1 folds l = cs+i into q = l+1.
// A front is 6 lanes (t & 7 = 0..5): every lane factorises S_ll redundantly (same latency as one lane) and then solves ONE
// right-hand side: lanes 0..4 column c of Z = S_ll^-1 S_lq, lane 5 y = S_ll^-1 r_l.  The same lane applies S_ql to its
// solution and updates column c of S_qq (lanes 0..4) or r_q (lane 5):  S_qq -= S_ql Z,  r_q -= S_ql y.
// Z and y are kept (zy[5]) and stored afterwards over S_ll / r_l: the back substitution is then dl_l = y - Z dl_q.
HD bool ph_tri_elim(int t, int i, const TriPlan& P, const Lay& Y, double* L, double* zy, int* l_out) {
    const int front = t >> 3, col = t & 7;
    if (t >= 16 || col >= 6) return false;
    if (i >= (front ? P.nB : P.nA)) return false;
    const int l = front ? P.cs + i : P.cs + P.cn - 1 - i;
    const int q = front ? l + 1 : l - 1;
    // S_{q,l} / S_{l,q}: front 1 reads (SJP, SPJ) of q, front 0 (SPJ, SJP) of l.  Offsets by arithmetic on `front`: a lane-varying
    // select between two layout offsets is turned into a scratch-memory table by the compiler
    const int dsp = Y.SPJ - Y.SJP, ob = 25 * (l + front * (q - l));
    const int oQL = Y.SJP + (1 - front) * dsp + ob;
    const int oLQ = Y.SJP + front * dsp + ob;
    const bool isy = col == 5;
    // ---- loads: own right-hand side (a column of S_lq, or r_l), own target (a column of S_qq, or r_q), S_ql, S_ll
    const int orhs = isy ? Y.R + 5 * l : oLQ + col, srhs = isy ? 1 : 5;
    const int otgt = isy ? Y.R + 5 * q : Y.SJJ + 25 * q + col, stgt = isy ? 1 : 5;
    double sql[25], tg[5], lu[25];
#pragma unroll
    for (int r = 0; r < 5; r++) { zy[r] = L[orhs + srhs * r]; tg[r] = L[otgt + stgt * r]; }
#pragma unroll
    for (int c = 0; c < 25; c++) sql[c] = L[oQL + c];
    lu5(L, Y.SJJ + 25 * l, lu);
    lu5_solve(lu, zy);
#pragma unroll
    for (int r = 0; r < 5; r++) tg[r] -= sql[5 * r] * zy[0] + sql[5 * r + 1] * zy[1] + sql[5 * r + 2] * zy[2] + sql[5 * r + 3] * zy[3] + sql[5 * r + 4] * zy[4];
    // ---- stores
#pragma unroll
    for (int r = 0; r < 5; r++) L[otgt + stgt * r] = tg[r];
    *l_out = l;
    return true;
}
// Z (one column per lane) overwrites S_ll, y overwrites r_l
HD void ph_tri_store(int t, int l, const Lay& Y, double* L, const double* zy) {
    const int col = t & 7;
    const int o = (col == 5) ? Y.R + 5 * l : Y.SJJ + 25 * l + col, st = (col == 5) ? 1 : 5;
#pragma unroll
    for (int r = 0; r < 5; r++) L[o + st * r] = zy[r];
}
// middle link: factorise (both sides have been folded in) and solve; one lane
HD void ph_tri_mid(int t, const TriPlan& P, const Lay& Y, double* L) {
    if (t != 0) return;
    double A[25], b[5];
#pragma unroll
    for (int i = 0; i < 5; i++) b[i] = L[Y.R + 5 * P.mid + i];
    lu5(L, Y.SJJ + 25 * P.mid, A);
    lu5_solve(A, b);
#pragma unroll
    for (int i = 0; i < 5; i++) L[Y.DL + 5 * P.mid + i] = b[i];
}
// back substitution step j: dl_l = y_l - Z_l dl_nbr; front 0 (lanes 0..4) l = mid+1+j, nbr = l-1; front 1 (lanes 8..12) l = mid-1-j, nbr = l+1
HD void ph_tri_back(int t, int j, const TriPlan& P, const Lay& Y, double* L) {
    const int front = t >> 3, row = t & 7;
    if (t >= 16 || row >= 5) return;
    if (j >= (front ? P.nB : P.nA)) return;
    const int l = front ? P.mid - 1 - j : P.mid + 1 + j;
    const int nbr = front ? l + 1 : l - 1;
    double z[5], dn[5];
#pragma unroll
    for (int c = 0; c < 5; c++) { z[c] = L[Y.SJJ + 25 * l + 5 * row + c]; dn[c] = L[Y.DL + 5 * nbr + c]; }
    double y = L[Y.R + 5 * l + row];
    L[Y.DL + 5 * l + row] = y - (z[0] * dn[0] + z[1] * dn[1] + z[2] * dn[2] + z[3] * dn[3] + z[4] * dn[4]);
}

// S4: ds_b = D_b^-1 (d_b + Gk_b(own joint)' dl_b + Gk_a(child joint)' dl_child), one task per body
HD void ph_body_solve(int t, int G, int nb, const Lay& Y, double* L, unsigned long long end_mask) {
    for (int b = t; b < nb; b += G) {
        const bool has_c = !((end_mask >> b) & 1ull);
        double d[6], gb[30], lb[5], ga[30], lc[5], Di[9];
#pragma unroll
        for (int i = 0; i < 6; i++) d[i] = L[Y.D + 6 * b + i];
#pragma unroll
        for (int i = 0; i < 30; i++) gb[i] = L[Y.GKB + BLK * b + i];
#pragma unroll
        for (int i = 0; i < 5; i++) lb[i] = L[Y.DL + 5 * b + i];
#pragma unroll
        for (int i = 0; i < 9; i++) Di[i] = L[Y.DINV + 9 * b + i];
        const double dtm = L[Y.DTM + b];
        if (has_c) {
#pragma unroll
            for (int i = 0; i < 30; i++) ga[i] = L[Y.GKA + BLK * (b + 1) + i];
#pragma unroll
            for (int i = 0; i < 5; i++) lc[i] = L[Y.DL + 5 * (b + 1) + i];
        } else {
#pragma unroll
            for (int i = 0; i < 30; i++) ga[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 5; i++) lc[i] = 0.0;
        }
        double tv[6], cd[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            cd[c] = gb[c] * lb[0] + gb[6 + c] * lb[1] + gb[12 + c] * lb[2] + gb[18 + c] * lb[3] + gb[24 + c] * lb[4]
                  + ga[c] * lc[0] + ga[6 + c] * lc[1] + ga[12 + c] * lc[2] + ga[18 + c] * lc[3] + ga[24 + c] * lc[4];
            tv[c] = d[c] + cd[c];
        }
        double o[6];
#pragma unroll
        for (int c = 0; c < 3; c++) { o[c] = tv[c] * dtm; o[3 + c] = Di[3 * c] * tv[3] + Di[3 * c + 1] * tv[4] + Di[3 * c + 2] * tv[5]; }
#pragma unroll
        for (int c = 0; c < 6; c++) { L[Y.DS + 6 * b + c] = o[c]; L[Y.CD + 6 * b + c] = cd[c]; }
    }
}

// ================================================================== general trees (MechDev::tree)
// F3 for a body with any number of child joints: C_b = Gk_b(own joint)' lambda_b + sum_children Gk_a(child)' lambda_child
HD void ph_force_map_tree(int t, int G, int nb, const Lay& Y, double* L, const MechDev* M) {
    for (int b = t; b < nb; b += G) {
        double acc[6];
        {
            double gb[30], lb[5];
#pragma unroll
            for (int i = 0; i < 30; i++) gb[i] = L[Y.GKB + BLK * b + i];
#pragma unroll
            for (int i = 0; i < 5; i++) lb[i] = L[Y.LAM + 5 * b + i];
#pragma unroll
            for (int c = 0; c < 6; c++) acc[c] = gb[c] * lb[0] + gb[6 + c] * lb[1] + gb[12 + c] * lb[2] + gb[18 + c] * lb[3] + gb[24 + c] * lb[4];
        }
        for (int ci = 0; ci < M->nchild[b]; ci++) {
            const int ch = M->child[b][ci];
            double ga[30], lc[5];
#pragma unroll
            for (int i = 0; i < 30; i++) ga[i] = L[Y.GKA + BLK * ch + i];
#pragma unroll
            for (int i = 0; i < 5; i++) lc[i] = L[Y.LAM + 5 * ch + i];
#pragma unroll
            for (int c = 0; c < 6; c++) acc[c] += ga[c] * lc[0] + ga[6 + c] * lc[1] + ga[12 + c] * lc[2] + ga[18 + c] * lc[3] + ga[24 + c] * lc[4];
        }
#pragma unroll
        for (int c = 0; c < 6; c++) { L[Y.C + 6 * b + c] = acc[c]; L[Y.CD + 6 * b + c] = 0.0; }
    }
}

// S2 for trees.  Same task shape as ph_schur_s (two 6-vectors against two 5x6 blocks):
//   part 0 (joint j, row):  row of S_jj and r_j, parent from the table
//   part 1 (joint j with a parent joint p, row):  rows of S_jp and S_pj
//   part 2 (sibling pair (i, j), row):  row of S_ij = W_a[i] Gk_a[j]'  and of  S_ji = W_a[j] Gk_a[i]'
HD void ph_schur_s_tree(int t, int G, int nb, const Lay& Y, double* L, const MechDev* M) {
    const int ntask = 5 * (2 * nb + M->npairs);
    for (int e = t; e < ntask; e += G) {
        const int part = (e >= 10 * nb) ? 2 : ((e >= 5 * nb) ? 1 : 0);
        const int e5 = e - part * 5 * nb;
        const int idx = e5 / 5, row = e5 - 5 * idx;
        int ov1, ov2, om1, om2, od1 = Y.D, od2 = Y.D, oo1, oo2;
        double use2 = 1.0;
        if (part == 2) {
            const int i = M->pair_i[idx], j = M->pair_j[idx];
            ov1 = Y.GVA + BLK * i + 6 * row; om1 = Y.GKA + BLK * j; oo1 = Y.SS + 25 * (2 * idx) + 5 * row;
            ov2 = Y.GVA + BLK * j + 6 * row; om2 = Y.GKA + BLK * i; oo2 = Y.SS + 25 * (2 * idx + 1) + 5 * row;
        } else {
            const int j = idx, pp = M->parent[j];
            const bool has_p = pp >= 0;
            if (part && !has_p) continue;
            const int p = has_p ? pp : j;
            ov1 = (part ? Y.GVA + BLK * j : Y.GVB + BLK * j) + 6 * row;
            ov2 = (part ? Y.GVB + BLK * p : Y.GVA + BLK * j) + 6 * row;
            om1 = part ? Y.GKB + BLK * p : Y.GKB + BLK * j;
            om2 = Y.GKA + BLK * j;
            od1 = Y.D + 6 * j; od2 = Y.D + 6 * p;
            use2 = (part || has_p) ? 1.0 : 0.0;
            oo1 = (part ? Y.SJP : Y.SJJ) + 25 * j + 5 * row;
            oo2 = Y.SPJ + 25 * j + 5 * row;
        }
        double v1[6], v2[6], m1[30], m2[30], d1[6], d2[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { v1[i] = L[ov1 + i]; v2[i] = L[ov2 + i]; d1[i] = L[od1 + i]; d2[i] = L[od2 + i]; }
#pragma unroll
        for (int i = 0; i < 30; i++) { m1[i] = L[om1 + i]; m2[i] = L[om2 + i]; }
        double o1[5], o2[5];
#pragma unroll
        for (int col = 0; col < 5; col++) { o1[col] = dot6(v1, m1 + 6 * col); o2[col] = use2 * dot6(v2, m2 + 6 * col); }
        if (part == 0) {
            const double g0 = L[Y.G + 5 * idx + row];
            const double rr = g0 - dot6(v1, d1) - use2 * dot6(v2, d2);
#pragma unroll
            for (int col = 0; col < 5; col++) L[oo1 + col] = o1[col] + o2[col];
            L[Y.R + 5 * idx + row] = rr;
        } else {
#pragma unroll
            for (int col = 0; col < 5; col++) { L[oo1 + col] = o1[col]; L[oo2 + col] = o2[col]; }
        }
    }
}

// S3 for trees: eliminate link l (links are taken in reverse order, so its whole subtree is already folded into S_ll and r_l).
// Lane (g = t >> 3, c = t & 7): group g < nn owns neighbour x_g; lanes c < 5 solve column c of Z_{l,x_g} = S_ll^-1 S_{l,x_g}
// and subtract S_{x',l} z from column c of S_{x',x_g} for every neighbour x'; lane (0, 5) does the same with y = S_ll^-1 r_l
// on the right-hand sides.  Z overwrites S_{l,x_g}, y overwrites r_l.  A root with nothing left above it only computes y.
HD void ph_tree_elim(int t, int l, const Lay& Y, double* L, const MechDev* M) {
    const int g = t >> 3, c = t & 7, nn = M->el_nn[l];
    const bool isy = (g == 0 && c == 5);
    if (!(isy || (g < nn && c < 5))) return;
    double lu[25], zy[5];
    const int orhs = isy ? Y.R + 5 * l : M->el_lx[l][g] + c, srhs = isy ? 1 : 5;
#pragma unroll
    for (int r = 0; r < 5; r++) zy[r] = L[orhs + srhs * r];
    lu5(L, Y.SJJ + 25 * l, lu);
    lu5_solve(lu, zy);
    for (int gp = 0; gp < nn; gp++) {
        const int oxl = M->el_xl[l][gp];
        const int otgt = isy ? Y.R + 5 * M->el_x[l][gp] : M->el_t[l][gp][g] + c;
        double sxl[25], tg[5];
#pragma unroll
        for (int i = 0; i < 25; i++) sxl[i] = L[oxl + i];
#pragma unroll
        for (int r = 0; r < 5; r++) tg[r] = L[otgt + srhs * r];
#pragma unroll
        for (int r = 0; r < 5; r++) tg[r] -= sxl[5 * r] * zy[0] + sxl[5 * r + 1] * zy[1] + sxl[5 * r + 2] * zy[2] + sxl[5 * r + 3] * zy[3] + sxl[5 * r + 4] * zy[4];
#pragma unroll
        for (int r = 0; r < 5; r++) L[otgt + srhs * r] = tg[r];
    }
#pragma unroll
    for (int r = 0; r < 5; r++) L[orhs + srhs * r] = zy[r];
}
// back substitution for link l (links in forward order: every neighbour above l is already solved): dl_l = y_l - sum_g Z_{l,x_g} dl_{x_g}
HD void ph_tree_back(int t, int l, const Lay& Y, double* L, const MechDev* M) {
    if (t >= 5) return;
    double acc = L[Y.R + 5 * l + t];
    for (int g = 0; g < M->el_nn[l]; g++) {
        const int oz = M->el_lx[l][g] + 5 * t, od = Y.DL + 5 * M->el_x[l][g];
        acc -= L[oz] * L[od] + L[oz + 1] * L[od + 1] + L[oz + 2] * L[od + 2] + L[oz + 3] * L[od + 3] + L[oz + 4] * L[od + 4];
    }
    L[Y.DL + 5 * l + t] = acc;
}

// S4 for trees: ds_b = D_b^-1 (d_b + Gk_b(own joint)' dl_b + sum_children Gk_a(child)' dl_child)
HD void ph_body_solve_tree(int t, int G, int nb, const Lay& Y, double* L, const MechDev* M) {
    for (int b = t; b < nb; b += G) {
        double d[6], Di[9], cd[6];
#pragma unroll
        for (int i = 0; i < 6; i++) d[i] = L[Y.D + 6 * b + i];
#pragma unroll
        for (int i = 0; i < 9; i++) Di[i] = L[Y.DINV + 9 * b + i];
        const double dtm = L[Y.DTM + b];
        {
            double gb[30], lb[5];
#pragma unroll
            for (int i = 0; i < 30; i++) gb[i] = L[Y.GKB + BLK * b + i];
#pragma unroll
            for (int i = 0; i < 5; i++) lb[i] = L[Y.DL + 5 * b + i];
#pragma unroll
            for (int c = 0; c < 6; c++) cd[c] = gb[c] * lb[0] + gb[6 + c] * lb[1] + gb[12 + c] * lb[2] + gb[18 + c] * lb[3] + gb[24 + c] * lb[4];
        }
        for (int ci = 0; ci < M->nchild[b]; ci++) {
            const int ch = M->child[b][ci];
            double ga[30], lc[5];
#pragma unroll
            for (int i = 0; i < 30; i++) ga[i] = L[Y.GKA + BLK * ch + i];
#pragma unroll
            for (int i = 0; i < 5; i++) lc[i] = L[Y.DL + 5 * ch + i];
#pragma unroll
            for (int c = 0; c < 6; c++) cd[c] += ga[c] * lc[0] + ga[6 + c] * lc[1] + ga[12 + c] * lc[2] + ga[18 + c] * lc[3] + ga[24 + c] * lc[4];
        }
        double tv[6], o[6];
#pragma unroll
        for (int c = 0; c < 6; c++) tv[c] = d[c] + cd[c];
#pragma unroll
        for (int c = 0; c < 3; c++) { o[c] = tv[c] * dtm; o[3 + c] = Di[3 * c] * tv[3] + Di[3 * c + 1] * tv[4] + Di[3 * c + 2] * tv[5]; }
#pragma unroll
        for (int c = 0; c < 6; c++) { L[Y.DS + 6 * b + c] = o[c]; L[Y.CD + 6 * b + c] = cd[c]; }
    }
}

// T1: trial point  st = s - alpha ds ; lt = lam - alpha dl  (one task per link); returns partial ||(ds, dl)||^2
HD double ph_trial(int t, int G, int nb, const Lay& Y, double* L, double alpha, int s_cur, int s_try, int l_cur, int l_try) {
    double acc = 0.0;
    for (int b = t; b < nb; b += G) {
        double sv[6], dv[6], lv[5], ev[5];
#pragma unroll
        for (int i = 0; i < 6; i++) { sv[i] = L[s_cur + 6 * b + i]; dv[i] = L[Y.DS + 6 * b + i]; }
#pragma unroll
        for (int i = 0; i < 5; i++) { lv[i] = L[l_cur + 5 * b + i]; ev[i] = L[Y.DL + 5 * b + i]; }
#pragma unroll
        for (int i = 0; i < 6; i++) { L[s_try + 6 * b + i] = sv[i] - alpha * dv[i]; acc += dv[i] * dv[i]; }
#pragma unroll
        for (int i = 0; i < 5; i++) { L[l_try + 5 * b + i] = lv[i] - alpha * ev[i]; acc += ev[i] * ev[i]; }
    }
    return acc;
}
// ---- line-search levels evaluated side by side.  After the full step has been rejected, the halvings alpha = 2^-j are tried
// NL at a time: lane group lg = t / nb evaluates level j0 + lg for link tl = t - lg nb.  A level's trial point, next pose and
// residual live in the (by then dead) Schur blocks: level slot q uses the 25 nb doubles at SJJ + 25 nb q  (SJJ, SJP, SPJ are
// consecutive):  ST (6 nb) | LT (5 nb) | XQ (7 nb) | D / G (6 nb).  Accepting the FIRST level in order that does not increase
// ||f|| (or level LINE_MAXIT) is the same decision sequence as halving one level at a time.
#define LEVEL_SLOTS 3
// lane groups that evaluate levels side by side: as many whole copies of the mechanism as fit the G lanes of an instance
HD int newton_level_groups(int G, int nb) { int n = G / nb; return n > LEVEL_SLOTS ? LEVEL_SLOTS : (n < 1 ? 1 : n); }
HD Lay level_layout(const Lay& Y, int nb, int slot) {
    Lay V = Y;
    const int base = Y.SJJ + 25 * nb * slot;
    V.ST = base; V.LT = base + 6 * nb; V.XQ = base + 11 * nb; V.D = base + 18 * nb; V.G = base + 18 * nb;
    return V;
}
HD void ph_trial_level(int tl, int nb, const Lay& Y, const Lay& V, double* L, double alpha, int s_cur, int l_cur) {
    if (tl >= nb) return;
    double sv[6], dv[6], lv[5], ev[5];
#pragma unroll
    for (int i = 0; i < 6; i++) { sv[i] = L[s_cur + 6 * tl + i]; dv[i] = L[Y.DS + 6 * tl + i]; }
#pragma unroll
    for (int i = 0; i < 5; i++) { lv[i] = L[l_cur + 5 * tl + i]; ev[i] = L[Y.DL + 5 * tl + i]; }
#pragma unroll
    for (int i = 0; i < 6; i++) L[V.ST + 6 * tl + i] = sv[i] - alpha * dv[i];
#pragma unroll
    for (int i = 0; i < 5; i++) L[V.LT + 5 * tl + i] = lv[i] - alpha * ev[i];
}
// the accepted level's trial point and next pose become the regular trial buffers
HD void ph_level_commit(int t, int nb, const Lay& Y, const Lay& V, double* L, int s_try, int l_try) {
    if (t >= nb) return;
    double a[6], b[5], c[7];
#pragma unroll
    for (int i = 0; i < 6; i++) a[i] = L[V.ST + 6 * t + i];
#pragma unroll
    for (int i = 0; i < 5; i++) b[i] = L[V.LT + 5 * t + i];
#pragma unroll
    for (int i = 0; i < 7; i++) c[i] = L[V.XQ + 7 * t + i];
#pragma unroll
    for (int i = 0; i < 6; i++) L[s_try + 6 * t + i] = a[i];
#pragma unroll
    for (int i = 0; i < 5; i++) L[l_try + 5 * t + i] = b[i];
#pragma unroll
    for (int i = 0; i < 7; i++) L[Y.XQ + 7 * t + i] = c[i];
}

// accept: the trial buffers become the current ones (the caller swaps offsets); C follows lambda:  C -= alpha CD
HD void ph_accept(int t, int G, int nb, const Lay& Y, double* L, double alpha) {
    for (int b = t; b < nb; b += G) {
#pragma unroll
        for (int i = 0; i < 6; i++) { L[Y.C + 6 * b + i] -= alpha * L[Y.CD + 6 * b + i]; L[Y.CD + 6 * b + i] = 0.0; }
    }
}
// copy the solution back into S/LAM when the Newton loop ended on the swapped buffers
HD void ph_copy_solution(int t, int G, int nb, const Lay& Y, double* L, int s_cur, int l_cur) {
    if (s_cur == Y.S) return;
    for (int e = t; e < 6 * nb; e += G) L[Y.S + e] = L[s_cur + e];
    for (int e = t; e < 5 * nb; e += G) L[Y.LAM + e] = L[l_cur + e];
}
// U1: state update from the accepted solution (XQ holds its next pose)
HD void ph_update(int t, int nb, const Lay& Y, double* L) {
    if (t >= nb) return;
    double* z = L + Y.Z + 13 * t;
    for (int i = 0; i < 7; i++) z[i] = L[Y.XQ + 7 * t + i];
    for (int i = 0; i < 6; i++) z[7 + i] = L[Y.S + 6 * t + i];
}

}  // namespace cclqr
