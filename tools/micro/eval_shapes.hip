// eval_shapes.hip -- Stage A microbenchmark of VERDICT r4 item 1 (MEASUREMENT TOOL, not part of libcclqr.so): the parallel part of one Newton
// iteration of the 17-body headline chain -- chain_eval<JAC> = body evaluation + joint evaluation + the Schur complement rows (ck_schur_rows),
// ~45 % of an iteration -- in three decompositions, each at the occupancy the whole rollout kernel would have in that shape:
//   A  today's shape: lane = link, TWO instances per wavefront (32-lane groups), LDS 40 816 B per workgroup -> four workgroups per CU = ONE wavefront
//      per SIMD.  The function timed IS the shipped rollout_chain.hip::chain_eval<32, true> (this file includes that source).
//   B  the re-cut the review asked to be measured: ONE instance per wavefront, THREE lanes per link (51 of 64 lanes; lane = 21 w + t so that the
//      parent link still is one lane below and the neighbour vectors move by the same wave shifts), <= 256 registers, LDS 20 408 B per workgroup ->
//      eight workgroups per CU = TWO wavefronts per SIMD.  The body evaluation and the row-independent part of the joint evaluation are computed by all
//      three lanes of a link (no exchange between them is needed then); the five constraint rows are dealt as {0, 3}, {1, 4}, {2}: every lane computes ONE
//      translational-kind row and ONE rotational-kind row from per-lane selectors -- uniform code; the lanes of row 2 (either kind, by joint type) compute
//      both kinds of that one row and select -- and the Schur rows / right-hand side of exactly those rows.
//   C  the minimal re-cut: lane = link as today but ONE instance per wavefront (17 of 64 lanes), <= 256 registers, two wavefronts per SIMD -- what stall
//      overlap alone buys, with no split of a link.
// Every shape evaluates the same points of the same instances; the norms and the Schur blocks / right-hand sides left in LDS are compared on the host.
// Reported: kernel time (HIP events), instance-evaluations per second, and wavefront cycles per evaluation (s_memtime) -- "cycles per PAIR of instances" is
// cycles per evaluation of one wavefront for A, and the wall time of two co-resident wavefronts' evaluations for B and C (= kernel time x SIMDs / pairs).
//   hipcc -O3 -std=c++17 -ffp-contract=fast --offload-arch=gfx950 -shared -fPIC tools/micro/eval_shapes.hip -o tools/micro/libeval_shapes.so
#include "../../constrainedcontrol.jl_amd/csrc/rollout_chain.hip"
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_tables.h"

namespace cclqr {
// (capi.hip's helper, which the included launchers of rollout_chain.hip reference; never called here)
hipError_t set_max_dynamic_lds_once(const void* fn, size_t lds) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }

struct MicroArgs {
    const MechDev* M;
    const double* z0;      // [n_inst][nb][13] user body order
    int64_t n_inst;
    int reps;
    double* norms;         // [n_inst][reps]
    double* image;         // [n_inst][SJJ SJP SPJ R of the LAST evaluation: 80 nb doubles] or null
    unsigned long long* cycles;   // [workgroups] s_memtime ticks of the timed loop
};

// what the rollout kernel does between the control law and newton! (no input: uj = 0), for the lane's link.  store = this lane writes the link's LDS slots
__device__ __forceinline__ void micro_forces(LinkC& c, LinkS& S, int t, const Lay& Y, double* L, const MechDev* M, double dt, bool store) {
    double zf[13], za[13];
#pragma unroll
    for (int i = 0; i < 7; i++) zf[i] = S.z[i];
#pragma unroll
    for (int i = 0; i < 6; i++) zf[7 + i] = S.s[i];
    from_prev<7>(zf, za);
    if (!c.has_a()) {
#pragma unroll
        for (int i = 0; i < 13; i++) za[i] = (i == 3) ? 1.0 : 0.0;
    }
    double F[3], tau[3], W6[6], cW6[6];
    ck_joint_wrench(c, 0.0, zf + 3, za + 3, F, tau, W6, W6 + 3);
    from_next<6>(W6, cW6);
    if (c.has_c()) {
#pragma unroll
        for (int i = 0; i < 3; i++) { F[i] += cW6[i]; tau[i] += cW6[3 + i]; }
    }
    double cTR[6];
    ck_step_invariants(c, zf, F, tau, dt, M->g, cTR, cTR + 3);
    double gk[5], kXT[3][3], kPB[5][3], kPA[5][3];
    joint_eval_sparse<true>(c, za, za + 3, zf, zf + 3, nullptr, nullptr, gk, kXT, kPB, kPA);
    if (c.live() && store) {
        gk_store(t, Y, L, kXT, kPB, kPA);
#pragma unroll
        for (int i = 0; i < 6; i++) { L[Y.D + 6 * t + i] = cTR[i]; L[Y.C + 6 * t + i] = 0.0; }
    }
}
// a trial direction that differs per link and component (the timed evaluations run at s - alpha ds for a different alpha each)
__device__ __forceinline__ void micro_direction(LinkS& S, int t) {
#pragma unroll
    for (int i = 0; i < 6; i++) { S.ds[i] = 1e-3 * (1 + ((7 * t + 3 * i) % 11)); S.cd[i] = 1e-2 * (1 + ((5 * t + i) % 7)); S.d[i] = 0.0; }
}

// ------------------------------------------------------------------------------------------------ shapes A and C: the shipped chain_eval<G, true>
template <int G, int WPS>
__global__ __launch_bounds__(64, WPS) void micro_eval_linklane(MicroArgs a) {
    extern __shared__ double lds[];
    constexpr int NBP = 17;
    const int lane = threadIdx.x, t = lane % G, grp = lane / G;
    const int64_t inst = (int64_t)blockIdx.x * (64 / G) + grp;
    const MechDev* M = a.M;
    const int nb = M->nb;
    const double dt = M->dt;
    const Lay Y = make_chain_layout(NBP);
    double* L = lds + grp * Y.total;
    LinkC c;
    link_load_consts(c, M, t, nb, dt);
    c.set_valid(inst < a.n_inst);
    const int ut = c.on() ? M->perm[t] : 0;
    LinkS S;
#pragma unroll
    for (int i = 0; i < 7; i++) S.z[i] = c.live() ? a.z0[inst * 13 * nb + ut * 13 + i] : ((i == 3) ? 1.0 : 0.0);
#pragma unroll
    for (int i = 0; i < 6; i++) S.s[i] = c.live() ? a.z0[inst * 13 * nb + ut * 13 + 7 + i] : 0.0;
    for (int e = t; e < Y.total; e += G) L[e] = 0.0;
    __syncthreads();
    micro_forces(c, S, t, Y, L, M, dt, true);
    micro_direction(S, t);
    __syncthreads();
    SubSel Qnone;      // (one lane per link: unused)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < a.reps; r++) {
        const double alpha = 1e-3 * (r + 1);
        const double nrm = chain_eval<G, true>(c, S, t, Y, L, alpha, c.live(), dt, Qnone);
        __syncthreads();
        if (t == 0 && c.valid()) a.norms[inst * a.reps + r] = nrm;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) a.cycles[blockIdx.x] = t1 - t0;
    if (a.image && c.valid()) {
        for (int e = t; e < 80 * NBP; e += G) a.image[inst * 80 * NBP + e] = L[e < 75 * NBP ? Y.SJJ + e : Y.R + (e - 75 * NBP)];
    }
}

// ------------------------------------------------------------------------------------------------ shape B: three lanes per link
constexpr int NL3 = 21;      // lanes of one sub-lane group (>= 17 links)
struct Sel3 {
    double sa[3], sb[3];     // selector of the lane's translational-kind row / rotational-kind row
    int ra, rb;              // their row numbers (rb = -1: the lane has no second row)
    bool x2;                 // the lane holds row 2 (either kind): both kinds of that ONE row are computed and the joint type selects
};
__device__ __forceinline__ void sel3_setup(const LinkC& c, int w, Sel3& Q) {
    // rows (cclqr_chain.h LinkC): revolute e0 e1 e2 | V1 V2, prismatic V1 V2 | e0 e1 e2
    const bool rev = c.rev();
    Q.x2 = w == 2;
    Q.ra = w < 2 ? w : 2;
    Q.rb = w < 2 ? 3 + w : -1;
    const int v = w & 1;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double unit_w = (i == w) ? 1.0 : 0.0, V = c.V12[3 * v + i];
        const double row2 = rev ? ((i == 2) ? 1.0 : 0.0) : ((i == 0) ? 1.0 : 0.0);        // row 2: e2 (revolute) / e0 (prismatic)
        Q.sa[i] = w < 2 ? (rev ? unit_w : V) : row2;
        const double unit_b = (i == 1 + w) ? 1.0 : 0.0;                                      // prismatic rows 3, 4: e1, e2
        Q.sb[i] = w < 2 ? (rev ? V : unit_b) : row2;
    }
}
// the lane's two rows of the joint's (g, sparse Jacobian pair); same formulas as joint_eval_sparse<true> (cclqr_chain.h)
struct Rows3 { double ga, gb, xt[3], pba[3], paa[3], pbb[3], pab[3]; };
__device__ __forceinline__ void joint_eval_rows3(const LinkC& c, const Sel3& Q, const double* xa, const double* qa, const double* xb, const double* qb,
                                                 const double* Na, const double* Nb, Rows3& R) {
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
    // row a as a translational row, row b as a rotational row
    const double a0 = Q.sa[0], a1 = Q.sa[1], a2 = Q.sa[2], b0 = Q.sb[0], b1 = Q.sb[1], b2 = Q.sb[2];
    const double vT = a0 * gT[0] + a1 * gT[1] + a2 * gT[2], vR = b0 * e[1] + b1 * e[2] + b2 * e[3];
    double xt[3], ptb[3], pta[3], prb[3], pra[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        xt[k] = a0 * Ra[k * 3] + a1 * Ra[k * 3 + 1] + a2 * Ra[k * 3 + 2];
        ptb[k] = a0 * PTb[k] + a1 * PTb[3 + k] + a2 * PTb[6 + k];
        pta[k] = a0 * PTa[k] + a1 * PTa[3 + k] + a2 * PTa[6 + k];
        prb[k] = b0 * PRb[k] + b1 * PRb[3 + k] + b2 * PRb[6 + k];
        pra[k] = b0 * PRa[k] + b1 * PRa[3 + k] + b2 * PRa[6 + k];
    }
    // a lane of row 2 keeps ONE row (slot a): the rotational variant when the joint is prismatic; its slot b is empty
    const bool rot = Q.x2 && !c.rev();
    R.ga = rot ? vR : vT;
    R.gb = Q.x2 ? 0.0 : vR;
    double pa3[3], pb3[3], qa3[3], qb3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        R.xt[k] = rot ? 0.0 : xt[k];
        pb3[k] = rot ? prb[k] : ptb[k]; pa3[k] = rot ? pra[k] : pta[k];
        qb3[k] = Q.x2 ? 0.0 : prb[k]; qa3[k] = Q.x2 ? 0.0 : pra[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        R.pba[k] = pb3[0] * Nb[k] + pb3[1] * Nb[3 + k] + pb3[2] * Nb[6 + k];
        R.pbb[k] = qb3[0] * Nb[k] + qb3[1] * Nb[3 + k] + qb3[2] * Nb[6 + k];
        const double na = pa3[0] * Na[k] + pa3[1] * Na[3 + k] + pa3[2] * Na[6 + k], nb2 = qa3[0] * Na[k] + qa3[1] * Na[3 + k] + qa3[2] * Na[6 + k];
        R.paa[k] = c.has_a() ? na : 0.0;
        R.pab[k] = c.has_a() ? nb2 : 0.0;
    }
}
// the lane's rows of S_jj, S_jp, S_jc and r_j (ck_schur_rows for two of the five rows; element (r, q) of a block at 5 q + r)
__device__ __forceinline__ void schur_rows3(const LinkC& c, const Sel3& Q, int j, bool store, const Lay& Y, double* L, const Rows3& R, const double* d, const double* pd) {
    const double sx = c.sxb + c.sxa;
    const int jp = c.has_a() ? j - 1 : j, jc = c.has_c() ? j + 1 : j;
    const bool hb = Q.rb >= 0;
    const int rb = hb ? Q.rb : Q.ra;
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int o = gk_row(q), ob = q < 3 ? 3 : 0;
        double kx[3] = {0, 0, 0}, kpx[3] = {0, 0, 0}, kcx[3] = {0, 0, 0}, kb[3], ka[3], kpb[3], kca[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (q < 3) { kx[i] = L[Y.GKA + GKSZ * j + o + i]; kpx[i] = L[Y.GKA + GKSZ * jp + o + i]; kcx[i] = L[Y.GKA + GKSZ * jc + o + i]; }
            kb[i] = L[Y.GKA + GKSZ * j + o + ob + i]; ka[i] = L[Y.GKA + GKSZ * j + o + ob + 3 + i];
            kpb[i] = L[Y.GKA + GKSZ * jp + o + ob + i]; kca[i] = L[Y.GKA + GKSZ * jc + o + ob + 3 + i];
        }
        // row a (translational kind: x part against the translational columns q < 3)
        double ajj = R.pba[0] * kb[0] + R.pba[1] * kb[1] + R.pba[2] * kb[2] + (R.paa[0] * ka[0] + R.paa[1] * ka[1] + R.paa[2] * ka[2]);
        double ajp = R.paa[0] * kpb[0] + R.paa[1] * kpb[1] + R.paa[2] * kpb[2];
        double ajc = R.pba[0] * kca[0] + R.pba[1] * kca[1] + R.pba[2] * kca[2];
        if (q < 3) {
            ajj += sx * (R.xt[0] * kx[0] + R.xt[1] * kx[1] + R.xt[2] * kx[2]);
            ajp -= c.sxa * (R.xt[0] * kpx[0] + R.xt[1] * kpx[1] + R.xt[2] * kpx[2]);
            ajc -= c.sxb * (R.xt[0] * kcx[0] + R.xt[1] * kcx[1] + R.xt[2] * kcx[2]);
        }
        // row b (rotational kind)
        const double bjj = R.pbb[0] * kb[0] + R.pbb[1] * kb[1] + R.pbb[2] * kb[2] + (R.pab[0] * ka[0] + R.pab[1] * ka[1] + R.pab[2] * ka[2]);
        const double bjp = R.pab[0] * kpb[0] + R.pab[1] * kpb[1] + R.pab[2] * kpb[2];
        const double bjc = R.pbb[0] * kca[0] + R.pbb[1] * kca[1] + R.pbb[2] * kca[2];
        if (store) {
            L[Y.SJJ + 25 * j + 5 * q + Q.ra] = ajj;
            if (hb) L[Y.SJJ + 25 * j + 5 * q + rb] = bjj;
            if (c.has_a()) { L[Y.SJP + 25 * j + 5 * q + Q.ra] = ajp; if (hb) L[Y.SJP + 25 * j + 5 * q + rb] = bjp; }
            if (c.has_c()) { L[Y.SPJ + 25 * jc + 5 * q + Q.ra] = ajc; if (hb) L[Y.SPJ + 25 * jc + 5 * q + rb] = bjc; }
        }
        SCHED_FENCE();
    }
    if (store) {
        const double bd = R.pba[0] * d[3] + R.pba[1] * d[4] + R.pba[2] * d[5], ad = R.paa[0] * pd[3] + R.paa[1] * pd[4] + R.paa[2] * pd[5];
        const double xd = R.xt[0] * d[0] + R.xt[1] * d[1] + R.xt[2] * d[2], xa = R.xt[0] * pd[0] + R.xt[1] * pd[1] + R.xt[2] * pd[2];
        L[Y.R + 5 * j + Q.ra] = R.ga - (c.sxb * xd + bd) - (ad - c.sxa * xa);
        if (hb) L[Y.R + 5 * j + rb] = R.gb - (R.pbb[0] * d[3] + R.pbb[1] * d[4] + R.pbb[2] * d[5]) - (R.pab[0] * pd[3] + R.pab[1] * pd[4] + R.pab[2] * pd[5]);
    }
}
__device__ __forceinline__ double chain_eval3(LinkC& c, const Sel3& Q, LinkS& S, int t, int w, const Lay& Y, double* L, double alpha, bool active, double dt) {
    double part = 0.0;
    double NB[9], xq[7];
    LINK_FLAGS_FRESH(c);
#pragma unroll
    for (int k = 0; k < 7; k++) xq[k] = S.z[k];
    if (active) {
        double cf[6], sv[6], cTR[6], DINV[9];
#pragma unroll
        for (int k = 0; k < 6; k++) { cf[k] = L[Y.C + 6 * t + k] - alpha * S.cd[k]; sv[k] = S.s[k] - alpha * S.ds[k]; cTR[k] = L[Y.D + 6 * t + k]; }
        const double pb = ck_body_eval<true>(c, S.z, sv, cf, cTR, cTR + 3, dt, xq, S.d, DINV, NB);
        part = w == 0 ? pb : 0.0;                 // the body's residual enters the norm once
        if (w == 0) {
#pragma unroll
            for (int k = 0; k < 9; k++) L[Y.DINV + 9 * t + k] = DINV[k];
        }
    }
    double pxq[7], pNB[9], pd[6];
    from_prev<7>(xq, pxq);
    from_prev<9>(NB, pNB);
    if (!c.has_a()) {
#pragma unroll
        for (int i = 0; i < 7; i++) pxq[i] = (i == 3) ? 1.0 : 0.0;
    }
    Rows3 R;
    if (active) {
        joint_eval_rows3(c, Q, pxq, pxq + 3, xq, xq + 3, pNB, NB, R);
        part += R.ga * R.ga + R.gb * R.gb;
    }
    LINK_FLAGS_FRESH(c);
    from_prev<6>(S.d, pd);
    if (active) schur_rows3(c, Q, t, true, Y, L, R, S.d, pd);
    return sqrt(group_sum<64>(part));
}
__global__ __launch_bounds__(64, 2) void micro_eval_3lanes(MicroArgs a) {
    extern __shared__ double lds[];
    constexpr int NBP = 17;
    const int lane = threadIdx.x, w = lane / NL3, t = lane - NL3 * w;
    const int64_t inst = blockIdx.x;
    const MechDev* M = a.M;
    const int nb = M->nb;
    const double dt = M->dt;
    const Lay Y = make_chain_layout(NBP);
    double* L = lds;
    LinkC c;
    link_load_consts(c, M, w < 3 ? t : 64, nb, dt);       // (lane 63: no link)
    c.set_valid(inst < a.n_inst);
    Sel3 Q;
    sel3_setup(c, w < 3 ? w : 0, Q);
    const int ut = c.on() ? M->perm[t] : 0;
    LinkS S;
#pragma unroll
    for (int i = 0; i < 7; i++) S.z[i] = c.live() ? a.z0[inst * 13 * nb + ut * 13 + i] : ((i == 3) ? 1.0 : 0.0);
#pragma unroll
    for (int i = 0; i < 6; i++) S.s[i] = c.live() ? a.z0[inst * 13 * nb + ut * 13 + 7 + i] : 0.0;
    for (int e = lane; e < Y.total; e += 64) L[e] = 0.0;
    __syncthreads();
    micro_forces(c, S, t, Y, L, M, dt, w == 0);
    micro_direction(S, t);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < a.reps; r++) {
        const double alpha = 1e-3 * (r + 1);
        const double nrm = chain_eval3(c, Q, S, t, w, Y, L, alpha, c.live(), dt);
        __syncthreads();
        if (lane == 0 && c.valid()) a.norms[inst * a.reps + r] = nrm;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) a.cycles[blockIdx.x] = t1 - t0;
    if (a.image && c.valid()) {
        for (int e = lane; e < 80 * NBP; e += 64) a.image[inst * 80 * NBP + e] = L[e < 75 * NBP ? Y.SJJ + e : Y.R + (e - 75 * NBP)];
    }
}

}  // namespace cclqr

using namespace cclqr;

// shape: 0 = A (two instances per wavefront, one wavefront per SIMD), 1 = B (three lanes per link, two wavefronts per SIMD), 2 = C (lane = link, one
// instance per wavefront, two wavefronts per SIMD).  Host pointers; returns the kernel's milliseconds (HIP events) or < 0.
extern "C" double micro_eval_run(const cclqr_mech_desc* d, const double* z0, long long n_inst, int reps, int shape, int timed_launches, double* norms, double* image,
                                 unsigned long long* cycles, int* n_workgroups, int* occupancy_wg_per_cu) {
    cclqr_mech m;
    std::string err;
    if (build_mech_tables(d, &m, err) != CCLQR_OK || m.nb != 17 || m.host.tree || m.host.loop) { fprintf(stderr, "micro: %s\n", err.c_str()); return -1.0; }
    MechDev* dM = nullptr;
    double *dz = nullptr, *dn = nullptr, *di = nullptr;
    unsigned long long* dc = nullptr;
    const size_t nz = 13 * 17;
    const int per_wg = shape == 0 ? 2 : 1;
    const unsigned grid = (unsigned)((n_inst + per_wg - 1) / per_wg);
    const Lay Y = make_chain_layout(17);
    const size_t lds = (size_t)per_wg * Y.total * sizeof(double);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "micro: %s: %s\n", #x, hipGetErrorString(e_)); return -2.0; } } while (0)
    CK(hipMalloc((void**)&dM, sizeof(MechDev)));
    CK(hipMemcpy(dM, &m.host, sizeof(MechDev), hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&dz, n_inst * nz * 8)); CK(hipMemcpy(dz, z0, n_inst * nz * 8, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&dn, (size_t)n_inst * reps * 8)); CK(hipMemset(dn, 0, (size_t)n_inst * reps * 8));
    if (image) { CK(hipMalloc((void**)&di, (size_t)n_inst * 80 * 17 * 8)); CK(hipMemset(di, 0, (size_t)n_inst * 80 * 17 * 8)); }
    CK(hipMalloc((void**)&dc, (size_t)grid * 8));
    MicroArgs a{dM, dz, n_inst, reps, dn, di, dc};
    const void* f = shape == 0 ? (const void*)micro_eval_linklane<32, 1> : (shape == 1 ? (const void*)micro_eval_3lanes : (const void*)micro_eval_linklane<64, 2>);
    CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    if (shape == 0) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, micro_eval_linklane<32, 1>, 64, lds));
    else if (shape == 1) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, micro_eval_3lanes, 64, lds));
    else CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, micro_eval_linklane<64, 2>, 64, lds));
    if (occupancy_wg_per_cu) *occupancy_wg_per_cu = occ;
    if (n_workgroups) *n_workgroups = (int)grid;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (shape == 0) hipLaunchKernelGGL((micro_eval_linklane<32, 1>), dim3(grid), dim3(64), lds, nullptr, a);
        else if (shape == 1) hipLaunchKernelGGL(micro_eval_3lanes, dim3(grid), dim3(64), lds, nullptr, a);
        else hipLaunchKernelGGL((micro_eval_linklane<64, 2>), dim3(grid), dim3(64), lds, nullptr, a);
    };
    launch();       // warm-up (code object load, clocks)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < timed_launches; i++) launch();
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(norms, dn, (size_t)n_inst * reps * 8, hipMemcpyDeviceToHost));
    if (image) CK(hipMemcpy(image, di, (size_t)n_inst * 80 * 17 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cycles, dc, (size_t)grid * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dM); (void)hipFree(dz); (void)hipFree(dn); if (di) (void)hipFree(di); (void)hipFree(dc);
    return (double)ms / timed_launches;
#undef CK
}
