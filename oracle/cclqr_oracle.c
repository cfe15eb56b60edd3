/*
 * cclqr_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See cclqr_oracle.h.
 *
 * Conventions:
 *   body state z[13] = x(3) world COM, q(4) unit quaternion body->world scalar first,
 *                      v(3) world, w(3) BODY frame; v,w are the velocities that led to the current knot.
 *   update   x+ = x + v+ dt ;  q+ = q (x) (dt/2) (sqrt(4/dt^2 - w+'w+), w+)
 *   residual d_T = m ((v+ - v)/dt + [0,0,-g]) - F
 *            d_R = (sq+ I + [w+]x) J w+ - (sq I - [w]x) J w - 2 tau
 *            d  -= sum_j G_{j,b}(x_k,q_k)' lambda_j ;  g_j(x+,q+) = 0
 *   Newton on (v+, w+, lambda) with the tree LDU (leaves first), line search, eps = 1e-10.
 */
#include "cclqr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXB 64
#define NEWTON_EPS 1e-10
#define NEWTON_MAXIT 100
#define LINE_MAXIT 10

/* ------------------------------------------------------------------ flop counter */
#ifdef ORC_COUNT_FLOPS
static double g_flops = 0.0;
#define FL(n) (g_flops += (double)(n))
#else
#define FL(n) ((void)0)
#endif
double orc_flops_get(void) {
#ifdef ORC_COUNT_FLOPS
    return g_flops;
#else
    return -1.0;
#endif
}
void orc_flops_reset(void) {
#ifdef ORC_COUNT_FLOPS
    g_flops = 0.0;
#endif
}
/* Newton statistics of the instrumented build: [0..15] line-search halvings summed per iteration index, [16..31] iterations reaching that
 * index, [32..47] histogram of -log10(alpha |ds|) of each solve's LAST iteration (bin b: 1e-(b+1) <= . < 1e-b), [48..58] histogram of halvings in the last iteration */
#ifdef ORC_COUNT_FLOPS
static double g_nstat[64];
#pragma omp threadprivate(g_nstat)
#endif
void orc_newton_stats(double *out, int reset) {
#ifdef ORC_COUNT_FLOPS
    for (int i = 0; i < 64; i++) { out[i] = g_nstat[i]; if (reset) g_nstat[i] = 0.0; }
#else
    (void)out; (void)reset;
#endif
}
/* Residual at the START of every Newton iteration, binned by floor(-log10 ||f||) (bin 0: >= 1 ... bin 31): [0..31] iterations AFTER which
 * ||f|| < eps for the first time in their solve, [32..63] iterations after which it still is not, [64..95] iterations entered with ||f|| < eps
 * already (they only serve the step-size half of the rule).  Calibrates the device's "this factorisation is the last one" predictor. */
#ifdef ORC_COUNT_FLOPS
static double g_nhist[96];
#pragma omp threadprivate(g_nhist)
#endif
void orc_newton_hist(double *out, int reset) {
#ifdef ORC_COUNT_FLOPS
    for (int i = 0; i < 96; i++) { out[i] = g_nhist[i]; if (reset) g_nhist[i] = 0.0; }
#else
    (void)out; (void)reset;
#endif
}
/* MODEL of a device option, not the reference's rule (default 0 = the rule of SURVEY 8a-bis, the only one parity is claimed against):
 * 1 = once ||f|| < eps the Jacobians are frozen (chord iterations: only the residual is re-evaluated), as cclqr_rollout's kernels do
 * for the iterations that only serve the step-size test.  Set before a rollout; read-only while one runs. */
static int g_newton_variant = 0;
void orc_set_newton_variant(int v) { g_newton_variant = v; }

/* ------------------------------------------------------------------ small dense helpers (row major) */
/* C(m x n) = beta*C + alpha * op(A) * op(B);  ta/tb: 0 = as is, 1 = transposed. lda/ldb/ldc = row strides */
static void gemm(int m, int n, int k, double alpha, const double *A, int lda, int ta, const double *B, int ldb, int tb,
                 double beta, double *C, int ldc) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int l = 0; l < k; l++) {
                double a = ta ? A[l * lda + i] : A[i * lda + l];
                double b = tb ? B[j * ldb + l] : B[l * ldb + j];
                s += a * b;
            }
            C[i * ldc + j] = (beta == 0.0 ? 0.0 : beta * C[i * ldc + j]) + alpha * s;
        }
    FL(2.0 * m * n * k);
}

/* LU with partial pivoting, in place; returns 0 ok, -1 singular */
static int lu_factor(int n, double *A, int lda, int *piv) {
    for (int c = 0; c < n; c++) {
        int p = c;
        double best = fabs(A[c * lda + c]);
        for (int r = c + 1; r < n; r++)
            if (fabs(A[r * lda + c]) > best) { best = fabs(A[r * lda + c]); p = r; }
        piv[c] = p;
        if (best == 0.0) return -1;
        if (p != c)
            for (int j = 0; j < n; j++) { double t = A[c * lda + j]; A[c * lda + j] = A[p * lda + j]; A[p * lda + j] = t; }
        double inv = 1.0 / A[c * lda + c];
        for (int r = c + 1; r < n; r++) {
            double l = A[r * lda + c] * inv;
            A[r * lda + c] = l;
            for (int j = c + 1; j < n; j++) A[r * lda + j] -= l * A[c * lda + j];
        }
        FL(1 + (n - c - 1) * (1 + 2.0 * (n - c - 1)));
    }
    return 0;
}
/* solve A X = B for nrhs columns, B (n x nrhs) row major, in place */
static void lu_solve(int n, const double *LU, int lda, const int *piv, double *B, int ldb, int nrhs) {
    for (int c = 0; c < n; c++) {
        int p = piv[c];
        if (p != c)
            for (int j = 0; j < nrhs; j++) { double t = B[c * ldb + j]; B[c * ldb + j] = B[p * ldb + j]; B[p * ldb + j] = t; }
    }
    for (int i = 0; i < n; i++)
        for (int r = 0; r < i; r++) {
            double l = LU[i * lda + r];
            for (int j = 0; j < nrhs; j++) B[i * ldb + j] -= l * B[r * ldb + j];
        }
    for (int i = n - 1; i >= 0; i--) {
        for (int r = i + 1; r < n; r++) {
            double u = LU[i * lda + r];
            for (int j = 0; j < nrhs; j++) B[i * ldb + j] -= u * B[r * ldb + j];
        }
        double inv = 1.0 / LU[i * lda + i];
        for (int j = 0; j < nrhs; j++) B[i * ldb + j] *= inv;
    }
    FL(2.0 * n * n * nrhs);
}
/* inverse of a small n x n (n <= 8) matrix */
static int inv_small(int n, const double *A, double *Ainv) {
    double LU[64];
    int piv[8];
    for (int i = 0; i < n * n; i++) LU[i] = A[i];
    if (lu_factor(n, LU, n, piv)) return -1;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ainv[i * n + j] = (i == j) ? 1.0 : 0.0;
    lu_solve(n, LU, n, piv, Ainv, n, n);
    return 0;
}

/* ------------------------------------------------------------------ quaternion / rotation algebra */
static void qmul(const double *a, const double *b, double *o) {
    double s = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    o[0] = s; o[1] = x; o[2] = y; o[3] = z;
    FL(28);
}
static void qconj(const double *a, double *o) { o[0] = a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = -a[3]; }
/* L(q): q (x) p = L(q) p ;  R(q): p (x) q = R(q) p ; 4x4 row major */
static void Lmat(const double *q, double *L) {
    double s = q[0], x = q[1], y = q[2], z = q[3];
    double t[16] = {s, -x, -y, -z, x, s, -z, y, y, z, s, -x, z, -y, x, s};
    memcpy(L, t, sizeof t);
}
static void Rmat(const double *q, double *R) {
    double s = q[0], x = q[1], y = q[2], z = q[3];
    double t[16] = {s, -x, -y, -z, x, s, z, -y, y, -z, s, x, z, y, -x, s};
    memcpy(R, t, sizeof t);
}
static void cross(const double *a, const double *b, double *o) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
    FL(9);
}
static void skew(const double *a, double *S) {
    S[0] = 0; S[1] = -a[2]; S[2] = a[1];
    S[3] = a[2]; S[4] = 0; S[5] = -a[0];
    S[6] = -a[1]; S[7] = a[0]; S[8] = 0;
}
/* rotation matrix of q (world <- body), row major 3x3 */
static void rotmat(const double *q, double *R) {
    double s = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = s * s + x * x - y * y - z * z; R[1] = 2 * (x * y - s * z); R[2] = 2 * (x * z + s * y);
    R[3] = 2 * (x * y + s * z); R[4] = s * s - x * x + y * y - z * z; R[5] = 2 * (y * z - s * x);
    R[6] = 2 * (x * z - s * y); R[7] = 2 * (y * z + s * x); R[8] = s * s - x * x - y * y + z * z;
    FL(30);
}
static void mat3vec(const double *R, const double *p, double *o) {
    double a = R[0] * p[0] + R[1] * p[1] + R[2] * p[2];
    double b = R[3] * p[0] + R[4] * p[1] + R[5] * p[2];
    double c = R[6] * p[0] + R[7] * p[1] + R[8] * p[2];
    o[0] = a; o[1] = b; o[2] = c;
    FL(15);
}
static void mat3Tvec(const double *R, const double *p, double *o) {
    double a = R[0] * p[0] + R[3] * p[1] + R[6] * p[2];
    double b = R[1] * p[0] + R[4] * p[1] + R[7] * p[2];
    double c = R[2] * p[0] + R[5] * p[1] + R[8] * p[2];
    o[0] = a; o[1] = b; o[2] = c;
    FL(15);
}
/* d(R(q) p)/dq (3x4), valid for the un-normalised form (s^2 - v'v) p + 2 (v'p) v + 2 s (v x p) */
static void drot_dq(const double *q, const double *p, double *D) {
    double s = q[0];
    const double *v = q + 1;
    double vxp[3];
    cross(v, p, vxp);
    double vp = v[0] * p[0] + v[1] * p[1] + v[2] * p[2];
    for (int i = 0; i < 3; i++) D[i * 4 + 0] = 2 * (s * p[i] + vxp[i]);
    double Sp[9];
    skew(p, Sp);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            D[i * 4 + 1 + j] = 2 * ((i == j ? vp : 0.0) + v[i] * p[j] - p[i] * v[j] - s * Sp[i * 3 + j]);
    FL(60);
}
/* d(R(q)' p)/dq (3x4) */
static void drotT_dq(const double *q, const double *p, double *D) {
    double qc[4];
    qconj(q, qc);
    drot_dq(qc, p, D);
    for (int i = 0; i < 3; i++)
        for (int j = 1; j < 4; j++) D[i * 4 + j] = -D[i * 4 + j];
}

/* ------------------------------------------------------------------ mechanism */
typedef struct {
    int nb, ne;
    double dt, g;
    double m[MAXB], J[MAXB][9];
    int parent[MAXB], child[MAXB], type[MAXB];
    double p1[MAXB][3], p2[MAXB][3], ax[MAXB][3], qoff[MAXB][4], V12[MAXB][6];
    int pj[MAXB];    /* parent joint of body b */
    int bfs[MAXB];   /* bodies in breadth-first order from the origin */
} mech_t;

/* two unit rows orthogonal to the (unit) axis a: deterministic choice (any basis gives the same trajectory) */
static void orth_rows(const double *a, double *V12) {
    int e = 0;
    double best = fabs(a[0]);
    for (int i = 1; i < 3; i++)
        if (fabs(a[i]) < best) { best = fabs(a[i]); e = i; }
    double v1[3] = {0, 0, 0};
    v1[e] = 1.0;
    double d = a[e];
    for (int i = 0; i < 3; i++) v1[i] -= d * a[i];
    double n = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
    for (int i = 0; i < 3; i++) v1[i] /= n;
    double v2[3];
    cross(a, v1, v2);
    for (int i = 0; i < 3; i++) { V12[i] = v1[i]; V12[3 + i] = v2[i]; }
}

static int mech_build(const orc_mech_desc *d, mech_t *M) {
    if (d->nb < 1 || d->nb > MAXB || d->ne != d->nb) return -1;
    M->nb = d->nb; M->ne = d->ne; M->dt = d->dt; M->g = d->g;
    for (int b = 0; b < d->nb; b++) {
        M->m[b] = d->mass[b];
        memcpy(M->J[b], d->inertia + 9 * b, 9 * sizeof(double));
        M->pj[b] = -1;
    }
    for (int j = 0; j < d->ne; j++) {
        M->parent[j] = d->parent[j]; M->child[j] = d->child[j]; M->type[j] = d->type[j];
        if (M->child[j] < 0 || M->child[j] >= d->nb || M->parent[j] < -1 || M->parent[j] >= d->nb) return -1;
        if (M->pj[M->child[j]] != -1) return -1; /* closed loop / two parents */
        M->pj[M->child[j]] = j;
        memcpy(M->p1[j], d->p1 + 3 * j, 24); memcpy(M->p2[j], d->p2 + 3 * j, 24);
        memcpy(M->qoff[j], d->qoff + 4 * j, 32);
        double n = sqrt(d->axis[3 * j] * d->axis[3 * j] + d->axis[3 * j + 1] * d->axis[3 * j + 1] + d->axis[3 * j + 2] * d->axis[3 * j + 2]);
        if (n == 0.0) return -1;
        for (int i = 0; i < 3; i++) M->ax[j][i] = d->axis[3 * j + i] / n;
        orth_rows(M->ax[j], M->V12[j]);
    }
    /* BFS from the origin */
    int cnt = 0;
    for (int j = 0; j < M->ne; j++)
        if (M->parent[j] == -1) M->bfs[cnt++] = M->child[j];
    for (int h = 0; h < cnt; h++)
        for (int j = 0; j < M->ne; j++)
            if (M->parent[j] == M->bfs[h]) M->bfs[cnt++] = M->child[j];
    if (cnt != M->nb) return -1; /* not a tree rooted at the origin */
    return 0;
}

static const double QID[4] = {1, 0, 0, 0};
static const double X0[3] = {0, 0, 0};

/* constraint value and raw Jacobians of joint j:  g(5), Xa(5x3), Qa(5x4), Xb(5x3), Qb(5x4)
 * translational  R(qa)'(xb + R(qb) p2 - xa) - p1 ;  rotational vec(qa^-1 qb qoff^-1)
 * Revolute = Translational(3) + Rotational(2 rows _|_ axis); Prismatic = Translational(2 rows _|_ axis) + Rotational(3) */
static void joint_eval(const mech_t *M, int j, const double *xa, const double *qa, const double *xb, const double *qb,
                       double *g, double *Xa, double *Qa, double *Xb, double *Qb, int jac) {
    double Ra[9], Rb[9], w[3], rp[3], gT[3], gR[3];
    rotmat(qa, Ra); rotmat(qb, Rb);
    mat3vec(Rb, M->p2[j], rp);
    for (int i = 0; i < 3; i++) w[i] = xb[i] + rp[i] - xa[i];
    mat3Tvec(Ra, w, gT);
    for (int i = 0; i < 3; i++) gT[i] -= M->p1[j][i];
    double qac[4], qoc[4], t[4], e[4];
    qconj(qa, qac); qconj(M->qoff[j], qoc);
    qmul(qac, qb, t); qmul(t, qoc, e);
    for (int i = 0; i < 3; i++) gR[i] = e[1 + i];
    FL(9);

    double XaT[9], XbT[9], QaT[12], QbT[12], QaR[12], QbR[12];
    if (jac) {
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) { XbT[r * 3 + c] = Ra[c * 3 + r]; XaT[r * 3 + c] = -Ra[c * 3 + r]; }
        drotT_dq(qa, w, QaT);
        double Dp[12];
        drot_dq(qb, M->p2[j], Dp);
        gemm(3, 4, 3, 1.0, Ra, 3, 1, Dp, 4, 0, 0.0, QbT, 4);
        /* e = L(qa*) R(qoff*) qb  ;  e = R(qb qoff*) T qa */
        double La[16], Ro[16], LR[16];
        Lmat(qac, La); Rmat(qoc, Ro);
        gemm(4, 4, 4, 1.0, La, 4, 0, Ro, 4, 0, 0.0, LR, 4);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) QbR[r * 4 + c] = LR[(r + 1) * 4 + c];
        double bo[4], Rb4[16];
        qmul(qb, qoc, bo);
        Rmat(bo, Rb4);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) QaR[r * 4 + c] = Rb4[(r + 1) * 4 + c] * (c == 0 ? 1.0 : -1.0);
    }
    const double *V = M->V12[j];
    int nt = (M->type[j] == ORC_REVOLUTE) ? 3 : 2;
    /* translational rows */
    for (int r = 0; r < nt; r++) {
        if (nt == 3) {
            g[r] = gT[r];
            if (jac) {
                for (int c = 0; c < 3; c++) { Xa[r * 3 + c] = XaT[r * 3 + c]; Xb[r * 3 + c] = XbT[r * 3 + c]; }
                for (int c = 0; c < 4; c++) { Qa[r * 4 + c] = QaT[r * 4 + c]; Qb[r * 4 + c] = QbT[r * 4 + c]; }
            }
        } else {
            g[r] = V[r * 3] * gT[0] + V[r * 3 + 1] * gT[1] + V[r * 3 + 2] * gT[2];
            if (jac) {
                for (int c = 0; c < 3; c++) {
                    Xa[r * 3 + c] = V[r * 3] * XaT[c] + V[r * 3 + 1] * XaT[3 + c] + V[r * 3 + 2] * XaT[6 + c];
                    Xb[r * 3 + c] = V[r * 3] * XbT[c] + V[r * 3 + 1] * XbT[3 + c] + V[r * 3 + 2] * XbT[6 + c];
                }
                for (int c = 0; c < 4; c++) {
                    Qa[r * 4 + c] = V[r * 3] * QaT[c] + V[r * 3 + 1] * QaT[4 + c] + V[r * 3 + 2] * QaT[8 + c];
                    Qb[r * 4 + c] = V[r * 3] * QbT[c] + V[r * 3 + 1] * QbT[4 + c] + V[r * 3 + 2] * QbT[8 + c];
                }
            }
        }
    }
    /* rotational rows */
    int nr = 5 - nt;
    for (int r = 0; r < nr; r++) {
        int o = nt + r;
        if (nr == 3) {
            g[o] = gR[r];
            if (jac) {
                for (int c = 0; c < 3; c++) { Xa[o * 3 + c] = 0; Xb[o * 3 + c] = 0; }
                for (int c = 0; c < 4; c++) { Qa[o * 4 + c] = QaR[r * 4 + c]; Qb[o * 4 + c] = QbR[r * 4 + c]; }
            }
        } else {
            g[o] = V[r * 3] * gR[0] + V[r * 3 + 1] * gR[1] + V[r * 3 + 2] * gR[2];
            if (jac) {
                for (int c = 0; c < 3; c++) { Xa[o * 3 + c] = 0; Xb[o * 3 + c] = 0; }
                for (int c = 0; c < 4; c++) {
                    Qa[o * 4 + c] = V[r * 3] * QaR[c] + V[r * 3 + 1] * QaR[4 + c] + V[r * 3 + 2] * QaR[8 + c];
                    Qb[o * 4 + c] = V[r * 3] * QbR[c] + V[r * 3 + 1] * QbR[4 + c] + V[r * 3 + 2] * QbR[8 + c];
                }
            }
        }
    }
    FL(jac ? 120 : 15);
}

/* (5x4) Q  ->  (5x3) Q * L(q) * V'   (derivative w.r.t. phi in q (x) (1, phi)) */
static void Q_to_phi(const double *Q, const double *q, double *out) {
    double L[16];
    Lmat(q, L);
    for (int r = 0; r < 5; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int l = 0; l < 4; l++) s += Q[r * 4 + l] * L[l * 4 + 1 + c];
            out[r * 3 + c] = s;
        }
    FL(120);
}
/* One joint evaluated from explicit parameters -- the SAME joint_eval / Q_to_phi as every tree joint -- for oracle/loops.py, the
 * dense-KKT stepper that handles closed kinematic loops (examples/lqr_deltabot.jl).  g[5]; Ga, Gb [5][6] = dg/d(x, phi) of the parent /
 * child body (phi: q (x) (1, phi)).  xa, qa = NULL: the parent is the origin (Ga = 0). */
void orc_joint_blocks(int32_t type, const double *p1, const double *p2, const double *axis, const double *qoff, const double *xa,
                      const double *qa, const double *xb, const double *qb, double *g, double *Ga, double *Gb) {
    static __thread mech_t M;
    M.nb = M.ne = 1; M.type[0] = type;
    memcpy(M.p1[0], p1, 24); memcpy(M.p2[0], p2, 24); memcpy(M.qoff[0], qoff, 32);
    double n = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    for (int i = 0; i < 3; i++) M.ax[0][i] = axis[i] / n;
    orth_rows(M.ax[0], M.V12[0]);
    const int has_a = xa != NULL;
    if (!has_a) { xa = X0; qa = QID; }
    double Xa[15], Qa[20], Xb[15], Qb[20], Pa[15], Pb[15];
    joint_eval(&M, 0, xa, qa, xb, qb, g, Xa, Qa, Xb, Qb, 1);
    Q_to_phi(Qa, qa, Pa); Q_to_phi(Qb, qb, Pb);
    for (int r = 0; r < 5; r++)
        for (int c = 0; c < 3; c++) {
            Ga[r * 6 + c] = has_a ? Xa[r * 3 + c] : 0.0; Ga[r * 6 + 3 + c] = has_a ? Pa[r * 3 + c] : 0.0;
            Gb[r * 6 + c] = Xb[r * 3 + c]; Gb[r * 6 + 3 + c] = Pb[r * 3 + c];
        }
}

/* (5x4) Q -> (5x3) Q * (dt/2) L(qk) [-w'/sq ; I]  (derivative w.r.t. w+ of q+ = qk (x) (dt/2)(sq, w+)) */
static void Q_to_omega(const double *Q, const double *qk, const double *w, double sq, double dt, double *out) {
    double L[16], T[12];
    Lmat(qk, L);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 3; c++) T[r * 3 + c] = 0.5 * dt * (L[r * 4 + 1 + c] - L[r * 4] * w[c] / sq);
    gemm(5, 3, 4, 1.0, Q, 4, 0, T, 3, 0, 0.0, out, 3);
    FL(36);
}

/* joint-space input uj -> world force F[b] and body-frame torque tau[b] on both bodies (knot k), SURVEY 8a-bis "Joint input" */
static void apply_input(const mech_t *M, const double *z, const double *uj, double *F, double *tau) {
    memset(F, 0, sizeof(double) * 3 * M->nb);
    memset(tau, 0, sizeof(double) * 3 * M->nb);
    for (int j = 0; j < M->ne; j++) {
        double u = uj[j];
        if (u == 0.0) continue;
        int a = M->parent[j], b = M->child[j];
        const double *qa = (a >= 0) ? z + 13 * a + 3 : QID;
        const double *qb = z + 13 * b + 3;
        double Ra[9], Rb[9], f[3], fw[3], fb[3], t[3];
        rotmat(qa, Ra); rotmat(qb, Rb);
        for (int i = 0; i < 3; i++) f[i] = M->ax[j][i] * u;
        mat3vec(Ra, f, fw);     /* world */
        mat3Tvec(Rb, fw, fb);   /* child frame */
        if (M->type[j] == ORC_PRISMATIC) {
            for (int i = 0; i < 3; i++) F[3 * b + i] += fw[i];
            cross(M->p2[j], fb, t);
            for (int i = 0; i < 3; i++) tau[3 * b + i] += t[i];
            if (a >= 0) {
                for (int i = 0; i < 3; i++) F[3 * a + i] -= fw[i];
                cross(M->p1[j], f, t);
                for (int i = 0; i < 3; i++) tau[3 * a + i] -= t[i];
            }
        } else {
            for (int i = 0; i < 3; i++) tau[3 * b + i] += fb[i];
            if (a >= 0)
                for (int i = 0; i < 3; i++) tau[3 * a + i] -= f[i];
        }
        FL(30);
    }
}

/* per-step workspace */
typedef struct {
    double Gka[MAXB][30], Gkb[MAXB][30]; /* dg/d(x,phi) at the current knot */
    double Gva[MAXB][30], Gvb[MAXB][30]; /* dg/d(v+,w+) at the next knot */
    double F[3 * MAXB], tau[3 * MAXB];
    double d[MAXB][6], g[MAXB][5];
    double Dr[MAXB][9];
} work_t;

static void next_pose(const mech_t *M, const double *zb, const double *sb, double *xn, double *qn, double *sq_out) {
    double dt = M->dt;
    for (int i = 0; i < 3; i++) xn[i] = zb[i] + sb[i] * dt;
    const double *w = sb + 3;
    double sq = sqrt(4.0 / (dt * dt) - (w[0] * w[0] + w[1] * w[1] + w[2] * w[2]));
    double wb[4] = {0.5 * dt * sq, 0.5 * dt * w[0], 0.5 * dt * w[1], 0.5 * dt * w[2]};
    qmul(zb + 3, wb, qn);
    if (sq_out) *sq_out = sq;
    FL(20);
}

static void knot_jacobians(const mech_t *M, const double *z, work_t *W) {
    for (int j = 0; j < M->ne; j++) {
        int a = M->parent[j], b = M->child[j];
        const double *xa = a >= 0 ? z + 13 * a : X0, *qa = a >= 0 ? z + 13 * a + 3 : QID;
        const double *xb = z + 13 * b, *qb = z + 13 * b + 3;
        double g[5], Xa[15], Qa[20], Xb[15], Qb[20], Pa[15], Pb[15];
        joint_eval(M, j, xa, qa, xb, qb, g, Xa, Qa, Xb, Qb, 1);
        Q_to_phi(Qa, qa, Pa); Q_to_phi(Qb, qb, Pb);
        for (int r = 0; r < 5; r++)
            for (int c = 0; c < 3; c++) {
                W->Gka[j][r * 6 + c] = Xa[r * 3 + c]; W->Gka[j][r * 6 + 3 + c] = Pa[r * 3 + c];
                W->Gkb[j][r * 6 + c] = Xb[r * 3 + c]; W->Gkb[j][r * 6 + 3 + c] = Pb[r * 3 + c];
            }
        if (a < 0) memset(W->Gka[j], 0, sizeof W->Gka[j]);
    }
}

/* residuals d (per body) and g (per joint) at solution guess (s, lam); optionally Jacobian blocks. returns ||f||_2 */
static double residual(const mech_t *M, const double *z, const double *s, const double *lam, work_t *W, int jac) {
    double dt = M->dt;
    double xn[MAXB][3], qn[MAXB][4], sqn[MAXB];
    double nrm = 0.0;
    for (int b = 0; b < M->nb; b++) {
        const double *zb = z + 13 * b, *v1 = zb + 7, *w1 = zb + 10, *v2 = s + 6 * b, *w2 = s + 6 * b + 3;
        const double *J = M->J[b];
        next_pose(M, zb, s + 6 * b, xn[b], qn[b], &sqn[b]);
        double sq2 = sqn[b];
        double sq1 = sqrt(4.0 / (dt * dt) - (w1[0] * w1[0] + w1[1] * w1[1] + w1[2] * w1[2]));
        double Jw1[3], Jw2[3], c1[3], c2[3];
        mat3vec(J, w1, Jw1); mat3vec(J, w2, Jw2);
        cross(w1, Jw1, c1); cross(w2, Jw2, c2);
        double ezg[3] = {0, 0, -M->g};
        for (int i = 0; i < 3; i++) {
            W->d[b][i] = M->m[b] * ((v2[i] - v1[i]) / dt + ezg[i]) - W->F[3 * b + i];
            W->d[b][3 + i] = sq2 * Jw2[i] + c2[i] - (sq1 * Jw1[i] - c1[i]) - 2.0 * W->tau[3 * b + i];
        }
        FL(40);
        if (jac) {
            /* D_R = (sq2 I + [w2]x) J - [J w2]x - (J w2) w2'/sq2 */
            double S[9], SJ[9];
            skew(w2, S);
            for (int i = 0; i < 9; i++) S[i] += (i % 4 == 0) ? sq2 : 0.0;
            gemm(3, 3, 3, 1.0, S, 3, 0, J, 3, 0, 0.0, SJ, 3);
            double Sj[9];
            skew(Jw2, Sj);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) W->Dr[b][r * 3 + c] = SJ[r * 3 + c] - Sj[r * 3 + c] - Jw2[r] * w2[c] / sq2;
            FL(30);
        }
    }
    for (int j = 0; j < M->ne; j++) {
        int a = M->parent[j], b = M->child[j];
        const double *lj = lam + 5 * j;
        /* d -= G' lambda */
        for (int c = 0; c < 6; c++) {
            double sb = 0, sa = 0;
            for (int r = 0; r < 5; r++) { sb += W->Gkb[j][r * 6 + c] * lj[r]; sa += W->Gka[j][r * 6 + c] * lj[r]; }
            W->d[b][c] -= sb;
            if (a >= 0) W->d[a][c] -= sa;
        }
        FL(a >= 0 ? 120 : 60);
        const double *xa = a >= 0 ? xn[a] : X0, *qa = a >= 0 ? qn[a] : QID;
        double Xa[15], Qa[20], Xb[15], Qb[20];
        joint_eval(M, j, xa, qa, xn[b], qn[b], W->g[j], Xa, Qa, Xb, Qb, jac);
        if (jac) {
            double Oa[15], Ob[15];
            Q_to_omega(Qb, z + 13 * b + 3, s + 6 * b + 3, sqn[b], dt, Ob);
            if (a >= 0) Q_to_omega(Qa, z + 13 * a + 3, s + 6 * a + 3, sqn[a], dt, Oa);
            for (int r = 0; r < 5; r++)
                for (int c = 0; c < 3; c++) {
                    W->Gvb[j][r * 6 + c] = Xb[r * 3 + c] * dt; W->Gvb[j][r * 6 + 3 + c] = Ob[r * 3 + c];
                    if (a >= 0) { W->Gva[j][r * 6 + c] = Xa[r * 3 + c] * dt; W->Gva[j][r * 6 + 3 + c] = Oa[r * 3 + c]; }
                }
            FL(30);
        }
    }
    for (int b = 0; b < M->nb; b++)
        for (int i = 0; i < 6; i++) nrm += W->d[b][i] * W->d[b][i];
    for (int j = 0; j < M->ne; j++)
        for (int i = 0; i < 5; i++) nrm += W->g[j][i] * W->g[j][i];
    FL(2 * 11 * M->nb + 1);
    return sqrt(nrm);
}

/* Solve [D -Gk'; Gv 0] [ds; dl] = [d; g] by block LDU along the body/joint tree, leaves first
 * (arXiv:2002.11245; SURVEY 8a-bis "Elimination order"). */
static int tree_ldu_solve(const mech_t *M, work_t *W, double *ds, double *dl) {
    static __thread double Db[MAXB][36], Dj[MAXB][25], Dbi[MAXB][36], Dji[MAXB][25], fb[MAXB][6], fj[MAXB][5];
    for (int b = 0; b < M->nb; b++) {
        memset(Db[b], 0, sizeof Db[b]);
        for (int i = 0; i < 3; i++) Db[b][i * 6 + i] = M->m[b] / M->dt;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Db[b][(3 + r) * 6 + 3 + c] = W->Dr[b][r * 3 + c];
        memcpy(fb[b], W->d[b], sizeof fb[b]);
    }
    for (int j = 0; j < M->ne; j++) { memset(Dj[j], 0, sizeof Dj[j]); memcpy(fj[j], W->g[j], sizeof fj[j]); }
    for (int h = M->nb - 1; h >= 0; h--) {
        int b = M->bfs[h], j = M->pj[b], a = M->parent[j];
        double T[30], t6[6], t5[5];
        /* eliminate body b into its parent joint j:  A_jb = Gv_b, A_bj = -Gk_b' */
        if (inv_small(6, Db[b], Dbi[b])) return -1;
        gemm(5, 6, 6, 1.0, W->Gvb[j], 6, 0, Dbi[b], 6, 0, 0.0, T, 6);          /* Gv_b D^-1 */
        gemm(5, 5, 6, 1.0, T, 6, 0, W->Gkb[j], 6, 1, 1.0, Dj[j], 5);             /* D_j += Gv_b D^-1 Gk_b' */
        gemm(5, 1, 6, -1.0, T, 6, 0, fb[b], 1, 0, 1.0, fj[j], 1);                /* f_j -= Gv_b D^-1 f_b */
        /* eliminate joint j into its parent body a:  A_aj = -Gk_a', A_ja = Gv_a */
        if (inv_small(5, Dj[j], Dji[j])) return -1;
        if (a >= 0) {
            gemm(6, 5, 5, 1.0, W->Gka[j], 6, 1, Dji[j], 5, 0, 0.0, T, 5);        /* Gk_a' D_j^-1 (6x5) */
            gemm(6, 6, 5, 1.0, T, 5, 0, W->Gva[j], 6, 0, 1.0, Db[a], 6);         /* D_a += Gk_a' D_j^-1 Gv_a */
            gemm(6, 1, 5, 1.0, T, 5, 0, fj[j], 1, 0, 1.0, fb[a], 1);             /* f_a += Gk_a' D_j^-1 f_j */
        }
        (void)t6; (void)t5;
    }
    for (int h = 0; h < M->nb; h++) {
        int b = M->bfs[h], j = M->pj[b], a = M->parent[j];
        double r5[5], r6[6];
        memcpy(r5, fj[j], sizeof r5);
        if (a >= 0) gemm(5, 1, 6, -1.0, W->Gva[j], 6, 0, ds + 6 * a, 1, 0, 1.0, r5, 1);
        gemm(5, 1, 5, 1.0, Dji[j], 5, 0, r5, 1, 0, 0.0, dl + 5 * j, 1);
        memcpy(r6, fb[b], sizeof r6);
        gemm(6, 1, 5, 1.0, W->Gkb[j], 6, 1, dl + 5 * j, 1, 0, 1.0, r6, 1);
        gemm(6, 1, 6, 1.0, Dbi[b], 6, 0, r6, 1, 0, 0.0, ds + 6 * b, 1);
    }
    return 0;
}

/* newton!: returns iterations used (>0) or -(iterations) when not converged */
static int newton(const mech_t *M, const double *z, double *s, double *lam, work_t *W) {
    int nb = M->nb, ne = M->ne;
    double ds[6 * MAXB], dl[5 * MAXB], st[6 * MAXB], lt[5 * MAXB];
    double normf0 = residual(M, z, s, lam, W, 0);
    for (int it = 1; it <= NEWTON_MAXIT; it++) {
        const int chord = g_newton_variant == 1 && it > 1 && normf0 < NEWTON_EPS;      /* model of the device's frozen-Jacobian iterations */
        residual(M, z, s, lam, W, chord ? 0 : 1);
        if (tree_ldu_solve(M, W, ds, dl)) return -it;
        double alpha = 1.0, normf1 = 0.0;
        int halvings = 0;
        for (int ls = 0; ls <= LINE_MAXIT; ls++) {
            for (int i = 0; i < 6 * nb; i++) st[i] = s[i] - alpha * ds[i];
            for (int i = 0; i < 5 * ne; i++) lt[i] = lam[i] - alpha * dl[i];
            FL(2 * 11 * nb);
            normf1 = residual(M, z, st, lt, W, 0);
            if (normf1 > normf0 && ls < LINE_MAXIT) { alpha *= 0.5; halvings++; } else break;
        }
        double nd = 0.0;
        for (int i = 0; i < 6 * nb; i++) nd += ds[i] * ds[i];
        for (int i = 0; i < 5 * ne; i++) nd += dl[i] * dl[i];
        nd = alpha * sqrt(nd);
        memcpy(s, st, sizeof(double) * 6 * nb);
        memcpy(lam, lt, sizeof(double) * 5 * ne);
#ifdef ORC_COUNT_FLOPS
        { int ii = it < 16 ? it - 1 : 15; g_nstat[ii] += halvings; g_nstat[16 + ii] += 1.0;
          if (normf1 < NEWTON_EPS && nd < NEWTON_EPS) { int b = nd > 0 ? (int)floor(-log10(nd)) : 15; if (b < 0) b = 0; if (b > 15) b = 15; g_nstat[32 + b] += 1.0; g_nstat[48 + halvings] += 1.0; }
          int hb = normf0 > 0 ? (int)floor(-log10(normf0)) : 31; if (hb < 0) hb = 0; if (hb > 31) hb = 31;
          g_nhist[(normf0 < NEWTON_EPS ? 64 : (normf1 < NEWTON_EPS ? 0 : 32)) + hb] += 1.0; }
#else
        (void)halvings;
#endif
        if (normf1 < NEWTON_EPS && nd < NEWTON_EPS) return it;
        if (g_newton_variant == 2 && normf1 < NEWTON_EPS) return it;      /* MODEL of a residual-only stopping rule (the device's newton_mode = 1 at 1e-10) */
        normf0 = normf1;
    }
    return -NEWTON_MAXIT;
}

static int step_core(const mech_t *M, double *z, double *lam, const double *uj, work_t *W, double *s_out) {
    double s[6 * MAXB];
    apply_input(M, z, uj, W->F, W->tau);
    knot_jacobians(M, z, W);
    for (int b = 0; b < M->nb; b++)
        for (int i = 0; i < 6; i++) s[6 * b + i] = z[13 * b + 7 + i];
    int it = newton(M, z, s, lam, W);
    if (s_out) memcpy(s_out, s, sizeof(double) * 6 * M->nb);
    for (int b = 0; b < M->nb; b++) {
        double xn[3], qn[4];
        next_pose(M, z + 13 * b, s + 6 * b, xn, qn, 0);
        memcpy(z + 13 * b, xn, 24); memcpy(z + 13 * b + 3, qn, 32);
        memcpy(z + 13 * b + 7, s + 6 * b, 48);
    }
    return it;
}

int orc_step(const orc_mech_desc *d, double *z, double *lam, const double *uj) {
    mech_t M;
    if (mech_build(d, &M)) return -1000;
    work_t *W = (work_t *)malloc(sizeof(work_t));
    int it = step_core(&M, z, lam, uj, W, 0);
    free(W);
    return it;
}

void orc_constraints(const orc_mech_desc *d, const double *z, double *g) {
    mech_t M;
    if (mech_build(d, &M)) return;
    for (int j = 0; j < M.ne; j++) {
        int a = M.parent[j], b = M.child[j];
        joint_eval(&M, j, a >= 0 ? z + 13 * a : X0, a >= 0 ? z + 13 * a + 3 : QID, z + 13 * b, z + 13 * b + 3, g + 5 * j, 0, 0, 0, 0, 0);
    }
}

/* per-body explicit solve with lambda exogenous: v+ directly, w+ by Newton on d_R */
static void step_fixed_lambda_core(const mech_t *M, const double *z, const double *lam, const double *uj, work_t *W, double *zn) {
    double dt = M->dt;
    apply_input(M, z, uj, W->F, W->tau);
    knot_jacobians(M, z, W);
    double c[MAXB][6];
    memset(c, 0, sizeof c);
    for (int j = 0; j < M->ne; j++) {
        int a = M->parent[j], b = M->child[j];
        for (int k = 0; k < 6; k++)
            for (int r = 0; r < 5; r++) {
                c[b][k] += W->Gkb[j][r * 6 + k] * lam[5 * j + r];
                if (a >= 0) c[a][k] += W->Gka[j][r * 6 + k] * lam[5 * j + r];
            }
    }
    for (int b = 0; b < M->nb; b++) {
        const double *zb = z + 13 * b, *w1 = zb + 10, *J = M->J[b];
        double s[6];
        for (int i = 0; i < 3; i++) s[i] = zb[7 + i] - dt * (i == 2 ? -M->g : 0.0) + dt / M->m[b] * (W->F[3 * b + i] + c[b][i]);
        double sq1 = sqrt(4.0 / (dt * dt) - (w1[0] * w1[0] + w1[1] * w1[1] + w1[2] * w1[2]));
        double Jw1[3], c1[3], rhs[3];
        mat3vec(J, w1, Jw1); cross(w1, Jw1, c1);
        for (int i = 0; i < 3; i++) rhs[i] = sq1 * Jw1[i] - c1[i] + 2.0 * W->tau[3 * b + i] + c[b][3 + i];
        double w2[3] = {w1[0], w1[1], w1[2]};
        for (int it = 0; it < 50; it++) {
            double sq2 = sqrt(4.0 / (dt * dt) - (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2]));
            double Jw2[3], c2[3], f[3], S[9], SJ[9], Sj[9], D[9], Di[9], dw[3];
            mat3vec(J, w2, Jw2); cross(w2, Jw2, c2);
            for (int i = 0; i < 3; i++) f[i] = sq2 * Jw2[i] + c2[i] - rhs[i];
            skew(w2, S);
            for (int i = 0; i < 9; i++) S[i] += (i % 4 == 0) ? sq2 : 0.0;
            gemm(3, 3, 3, 1.0, S, 3, 0, J, 3, 0, 0.0, SJ, 3);
            skew(Jw2, Sj);
            for (int r = 0; r < 3; r++)
                for (int cc = 0; cc < 3; cc++) D[r * 3 + cc] = SJ[r * 3 + cc] - Sj[r * 3 + cc] - Jw2[r] * w2[cc] / sq2;
            inv_small(3, D, Di);
            mat3vec(Di, f, dw);
            double n = 0;
            for (int i = 0; i < 3; i++) { w2[i] -= dw[i]; n += dw[i] * dw[i]; }
            if (sqrt(n) < 1e-15) break;
        }
        for (int i = 0; i < 3; i++) s[3 + i] = w2[i];
        double xn[3], qn[4];
        next_pose(M, zb, s, xn, qn, 0);
        memcpy(zn + 13 * b, xn, 24); memcpy(zn + 13 * b + 3, qn, 32); memcpy(zn + 13 * b + 7, s, 48);
    }
}

void orc_step_fixed_lambda(const orc_mech_desc *d, const double *z, const double *lam, const double *uj, double *znext) {
    mech_t M;
    if (mech_build(d, &M)) return;
    work_t *W = (work_t *)malloc(sizeof(work_t));
    step_fixed_lambda_core(&M, z, lam, uj, W, znext);
    free(W);
}

/* minimalCoordinates(mechanism, eqc)[1]: Revolute -> angle of qa^-1 qb qoff^-1 about the axis, Prismatic -> offset along the axis */
static double joint_coordinate(const mech_t *M, int j, const double *z) {
    int a = M->parent[j], b = M->child[j];
    const double *xa = a >= 0 ? z + 13 * a : X0, *qa = a >= 0 ? z + 13 * a + 3 : QID;
    const double *xb = z + 13 * b, *qb = z + 13 * b + 3;
    if (M->type[j] == ORC_REVOLUTE) {
        double qac[4], qoc[4], t[4], e[4];
        qconj(qa, qac); qconj(M->qoff[j], qoc);
        qmul(qac, qb, t); qmul(t, qoc, e);
        return 2.0 * atan2(M->ax[j][0] * e[1] + M->ax[j][1] * e[2] + M->ax[j][2] * e[3], e[0]);
    }
    double Ra[9], Rb[9], rp[3], w[3], gT[3];
    rotmat(qa, Ra); rotmat(qb, Rb);
    mat3vec(Rb, M->p2[j], rp);
    for (int i = 0; i < 3; i++) w[i] = xb[i] + rp[i] - xa[i];
    mat3Tvec(Ra, w, gT);
    return M->ax[j][0] * (gT[0] - M->p1[j][0]) + M->ax[j][1] * (gT[1] - M->p1[j][1]) + M->ax[j][2] * (gT[2] - M->p1[j][2]);
}
void orc_minimal_coordinates(const orc_mech_desc *d, const double *z, double *theta) {
    mech_t M;
    if (mech_build(d, &M)) return;
    for (int j = 0; j < M.ne; j++) theta[j] = joint_coordinate(&M, j, z);
}
/* control_pid!(mechanism, pid, k), pid.jl:69-88; integ/last = integratederrors / lasterrors of this instance */
static void pid_core(const mech_t *M, const orc_ctrl_desc *c, const double *z, int k, double *integ, double *last, double *uj) {
    const double PI = 3.14159265358979323846;
    for (int i = 0; i < c->npid; i++) {
        int j = c->pid_joint[i];
        double e = c->pid_goal[i] - joint_coordinate(M, j, z);            /* stateError_pid, pid.jl:43-57 */
        if (M->type[j] == ORC_REVOLUTE) { if (e > PI) e -= 2 * PI; else if (e < -PI) e += 2 * PI; }
        if (k == 1) last[i] = e;                                            /* pid.jl:73 */
        integ[i] += e * M->dt;                                              /* pid.jl:76 */
        double de = (e - last[i]) / M->dt;                                  /* pid.jl:77 */
        uj[j] += c->pid_P[i] * e + c->pid_I[i] * integ[i] + c->pid_D[i] * de; /* pid.jl:79 */
        last[i] = e;                                                        /* pid.jl:81 */
    }
}

/* Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), the counter-based generator SURVEY 8d names */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
    for (int i = 0; i < 4; i++) out[i] = c[i];
}

/* standard-normal sample (instance n, step k) of the reproducible noise stream: Box-Muller on the first two words */
double orc_philox_normal(uint64_t seed, uint64_t instance, int k) {
    uint32_t ctr[4] = {(uint32_t)(k - 1), 0, 0, 0}, key[2] = {(uint32_t)(seed & 0xffffffffu) ^ (uint32_t)(seed >> 32), (uint32_t)instance}, x[4];
    orc_philox4x32(ctr, key, x);
    double u1 = ((double)x[0] + 0.5) / 4294967296.0, u2 = ((double)x[1] + 0.5) / 4294967296.0;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

/* ------------------------------------------------------------------ feedback law */
/* lqr.jl:89-139 / lqr_tracking.jl:46-71 / trackingLQR_triple_cartpole.jl:76-115 */
static void control_core(const mech_t *M, const orc_ctrl_desc *c, int64_t inst, const double *z, int k, double noise_sample, double *uj) {
    int nb = M->nb, mx = 12 * nb;
    for (int j = 0; j < M->ne; j++) uj[j] = 0.0;
    int inf = (c->N <= 0);
    if (!inf && !(k < c->N)) return; /* lqr.jl:106: at k >= N no force is written */
    int ksp = (c->nsp > 1) ? (k - 1 < c->nsp ? k - 1 : c->nsp - 1) : 0;
    const size_t tab = c->n_ctrl > 1 ? (size_t)inst : 0; /* per-instance controller tables */
    const double *zd = c->zd + (tab * c->nsp + (size_t)ksp) * 13 * nb;
    const double *Ktab = c->K ? c->K + tab * (size_t)c->nK * c->mu * mx : NULL;
    const double *Fdtab = c->Fd ? c->Fd + tab * (size_t)c->nsp * c->mu : NULL;
    double dz[12 * MAXB];
    for (int b = 0; b < nb; b++) {
        const double *zb = z + 13 * b, *zdb = zd + 13 * b;
        double qdc[4], qe[4];
        qconj(zdb + 3, qdc);
        qmul(qdc, zb + 3, qe); /* qd \ q ; raw vector part, no sign fix, no factor 2 (lqr.jl:101-102) */
        for (int i = 0; i < 3; i++) {
            dz[12 * b + i] = zb[i] - zdb[i];
            dz[12 * b + 3 + i] = zb[7 + i] - zdb[7 + i];
            dz[12 * b + 6 + i] = qe[1 + i];
            dz[12 * b + 9 + i] = zb[10 + i] - zdb[10 + i];
        }
        FL(9);
    }
    /* passive viscous joint friction on the relative joint velocity */
    if (c->fric) {
        for (int j = 0; j < M->ne; j++) {
            if (c->fric[j] == 0.0) continue;
            int a = M->parent[j], b = M->child[j];
            double rel;
            if (M->type[j] == ORC_REVOLUTE) {
                rel = M->ax[j][0] * z[13 * b + 10] + M->ax[j][1] * z[13 * b + 11] + M->ax[j][2] * z[13 * b + 12];
                if (a >= 0) rel -= M->ax[j][0] * z[13 * a + 10] + M->ax[j][1] * z[13 * a + 11] + M->ax[j][2] * z[13 * a + 12];
            } else {
                double dv[3], dva[3];
                for (int i = 0; i < 3; i++) dv[i] = z[13 * b + 7 + i] - (a >= 0 ? z[13 * a + 7 + i] : 0.0);
                double Ra[9];
                rotmat(a >= 0 ? z + 13 * a + 3 : QID, Ra);
                mat3Tvec(Ra, dv, dva);
                rel = M->ax[j][0] * dva[0] + M->ax[j][1] * dva[1] + M->ax[j][2] * dva[2];
            }
            uj[j] += -c->fric[j] * rel;
        }
    }
    int kk = inf ? 0 : (k - 1 < c->nK ? k - 1 : c->nK - 1);
    for (int i = 0; i < c->mu; i++) {
        double u = Fdtab ? Fdtab[(size_t)ksp * c->mu + i] : 0.0;
        if (Ktab) {
            const double *Kr = Ktab + ((size_t)kk * c->mu + i) * mx;
            double s = 0;
            for (int t = 0; t < mx; t++) s += Kr[t] * dz[t];
            u -= s;
            FL(2 * mx);
        }
        u += c->noise_scale * noise_sample;
        uj[c->ctrl_joint[i]] += u;
    }
}

void orc_control(const orc_mech_desc *d, const orc_ctrl_desc *c, const double *z, int k, double noise_sample, double *uj) {
    mech_t M;
    if (mech_build(d, &M)) return;
    control_core(&M, c, 0, z, k, noise_sample, uj);
}

int orc_rollout(const orc_mech_desc *d, const orc_ctrl_desc *c, int64_t n_inst, int32_t steps, const double *z0, double *traj,
                double *zT, int32_t *status, int32_t nthreads) {
    mech_t M;
    if (mech_build(d, &M)) return -1;
    int nz = 13 * M.nb;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        work_t *W = (work_t *)malloc(sizeof(work_t));
#pragma omp for schedule(dynamic, 1)
        for (int64_t n = 0; n < n_inst; n++) {
            double z[13 * MAXB], lam[5 * MAXB], uj[MAXB], pid_integ[MAXB], pid_last[MAXB];
            memset(pid_integ, 0, sizeof pid_integ); memset(pid_last, 0, sizeof pid_last);
            memcpy(z, z0 + n * nz, sizeof(double) * nz);
            memset(lam, 0, sizeof lam);
            int worst = 0, bad = 0;
            for (int k = 1; k <= steps; k++) {
                if (traj) memcpy(traj + ((size_t)n * steps + (k - 1)) * nz, z, sizeof(double) * nz);
                double ns = 0.0;
                if (c->noise_scale != 0.0) {
                    if (c->noise) ns = c->noise[(size_t)n * steps + (k - 1)];
                    else if (c->noise_philox) ns = orc_philox_normal(c->noise_seed, (uint64_t)n, k);
                }
                control_core(&M, c, n, z, k, ns, uj);
                if (c->npid > 0) pid_core(&M, c, z, k, pid_integ, pid_last, uj);
                int it = step_core(&M, z, lam, uj, W, 0);
                if (it < 0) { bad = 1; it = -it; }
                if (it > worst) worst = it;
            }
            memcpy(zT + n * nz, z, sizeof(double) * nz);
            if (status) status[n] = bad ? -worst : worst;
        }
        free(W);
    }
    return 0;
}

/* ------------------------------------------------------------------ linearisation (a8) */
/* Exact Jacobians of the one-step map z+ = f(z, u, lambda) (lambda exogenous) at (zd, ud, lambda*), plus G = dg/dz+.
 * Error coordinates per body: [x, v, qtilde, w] with q = qd (x) (sqrt(1-|qt|^2), qt)   (lqr.jl:92-103). */
static void add3(double *A, int lda, int r0, int c0, const double *B, double scale) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) A[(size_t)(r0 + r) * lda + c0 + c] += scale * B[r * 3 + c];
}
static void mm3(const double *A, const double *B, double *C) { gemm(3, 3, 3, 1.0, A, 3, 0, B, 3, 0, 0.0, C, 3); }
static void mm3T(const double *A, const double *B, double *C) { gemm(3, 3, 3, 1.0, A, 3, 1, B, 3, 0, 0.0, C, 3); }
/* 3x3 block V * M4 * V' of a 4x4 matrix (rows/cols 1..3) */
static void vblock(const double *M4, double *o) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) o[r * 3 + c] = M4[(r + 1) * 4 + 1 + c];
}

static int linearize_core(const mech_t *M, const double *zd, int mu, const int *cj, const double *Fd, double *A, double *Bu, double *Bl,
                          double *G, work_t *W) {
    int nb = M->nb, ne = M->ne, mx = 12 * nb, ml = 5 * ne;
    double dt = M->dt;
    double z[13 * MAXB], lam[5 * MAXB], uj[MAXB], s[6 * MAXB];
    memcpy(z, zd, sizeof(double) * 13 * nb);
    memset(lam, 0, sizeof lam);
    for (int j = 0; j < ne; j++) uj[j] = 0;
    for (int i = 0; i < mu; i++) uj[cj[i]] += Fd ? Fd[i] : 0.0;
    double zn[13 * MAXB];
    memcpy(zn, z, sizeof(double) * 13 * nb);
    int it = step_core(M, zn, lam, uj, W, s); /* W->Gk* now hold the knot-k Jacobians, lam = lambda* */
    if (it < 0) return -1;

    /* FT[b], FR[b]: d(F + c_T)/dz and d(2 tau + c_R)/dz (3 x mx each);  FTu/FRu: w.r.t. u (3 x mu) */
    double *FT = (double *)calloc((size_t)nb * 3 * mx * 2, sizeof(double));
    double *FR = FT + (size_t)nb * 3 * mx;
    double *FTu = (double *)calloc((size_t)nb * 3 * (mu > 0 ? mu : 1) * 2, sizeof(double));
    double *FRu = FTu + (size_t)nb * 3 * (mu > 0 ? mu : 1);
#define FTB(b) (FT + (size_t)(b)*3 * mx)
#define FRB(b) (FR + (size_t)(b)*3 * mx)
    for (int j = 0; j < ne; j++) {
        int a = M->parent[j], b = M->child[j];
        const double *qa = a >= 0 ? z + 13 * a + 3 : QID, *xa = a >= 0 ? z + 13 * a : X0;
        const double *qb = z + 13 * b + 3, *xb = z + 13 * b;
        const double *lj = lam + 5 * j, *V = M->V12[j];
        int nt = (M->type[j] == ORC_REVOLUTE) ? 3 : 2;
        double mu3[3], nu3[3];
        if (nt == 3) {
            for (int i = 0; i < 3; i++) mu3[i] = lj[i];
            for (int i = 0; i < 3; i++) nu3[i] = V[i] * lj[3] + V[3 + i] * lj[4];
        } else {
            for (int i = 0; i < 3; i++) mu3[i] = V[i] * lj[0] + V[3 + i] * lj[1];
            for (int i = 0; i < 3; i++) nu3[i] = lj[2 + i];
        }
        double Ra[9], Rb[9], RbtRa[9], w[3], rp[3], Ratw[3];
        rotmat(qa, Ra); rotmat(qb, Rb);
        mm3T(Rb, Ra, RbtRa);
        mat3vec(Rb, M->p2[j], rp);
        for (int i = 0; i < 3; i++) w[i] = xb[i] + rp[i] - xa[i];
        mat3Tvec(Ra, w, Ratw);
        int cxa = 12 * a, cqa = 12 * a + 6, cxb = 12 * b, cqb = 12 * b + 6;
        double Sm[9], Sp2[9], T1[9], T2[9], T3[9];
        /* ---- translational constraint force, multiplier mu3 in the parent frame (geometric stiffness) ---- */
        {
            skew(mu3, Sm); skew(M->p2[j], Sp2);
            double RaSm[9];
            mm3(Ra, Sm, RaSm);
            double y[3], Sy[9];
            double Ramu[3];
            mat3vec(Ra, mu3, Ramu); mat3Tvec(Rb, Ramu, y);
            skew(y, Sy);
            if (a >= 0) {
                add3(FTB(a), mx, 0, cqa, RaSm, 2.0);  /* c_T,a = -Ra mu */
                add3(FTB(b), mx, 0, cqa, RaSm, -2.0); /* c_T,b =  Ra mu */
                mm3(RbtRa, Sm, T1); mm3(Sp2, T1, T2);
                add3(FRB(b), mx, 0, cqa, T2, -4.0);   /* c_R,b = 2 p2 x (Rb'Ra mu) */
            }
            mm3(Sp2, Sy, T1);
            add3(FRB(b), mx, 0, cqb, T1, 4.0);
            if (a >= 0) {
                /* c_R,a = 2 [mu]x Ra' w */
                double SmRat[9], Rat[9];
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) Rat[r * 3 + c] = Ra[c * 3 + r];
                mm3(Sm, Rat, SmRat);
                add3(FRB(a), mx, 0, cxa, SmRat, -2.0);
                add3(FRB(a), mx, 0, cxb, SmRat, 2.0);
                mm3(SmRat, Rb, T1); mm3(T1, Sp2, T2);
                add3(FRB(a), mx, 0, cqb, T2, -4.0);
                double Sw[9];
                skew(Ratw, Sw);
                mm3(Sm, Sw, T3);
                add3(FRB(a), mx, 0, cqa, T3, 4.0);
            }
        }
        /* ---- rotational constraint force, multiplier nu3 ---- */
        {
            double n4[4] = {0, nu3[0], nu3[1], nu3[2]};
            double qoc[4];
            qconj(M->qoff[j], qoc);
            double E[16], La[16], Lb[16], Rn[16], M1[16], M2[16], M3[16], blk[9];
            Rmat(qoc, E); Lmat(qa, La); Lmat(qb, Lb); Rmat(n4, Rn);
            /* f_b = V Lb' E' La n ;  df_b/dphi_a = V Lb' E' La R(n) V' ; df_b/dphi_b = -V R(Lb' E' La n) V' */
            gemm(4, 4, 4, 1.0, Lb, 4, 1, E, 4, 1, 0.0, M1, 4);
            gemm(4, 4, 4, 1.0, M1, 4, 0, La, 4, 0, 0.0, M2, 4); /* Lb' E' La */
            if (a >= 0) {
                gemm(4, 4, 4, 1.0, M2, 4, 0, Rn, 4, 0, 0.0, M3, 4);
                vblock(M3, blk);
                add3(FRB(b), mx, 0, cqa, blk, 1.0);
            }
            double y4[4];
            gemm(4, 1, 4, 1.0, M2, 4, 0, n4, 1, 0, 0.0, y4, 1);
            Rmat(y4, M3);
            vblock(M3, blk);
            add3(FRB(b), mx, 0, cqb, blk, -1.0);
            if (a >= 0) {
                /* f_a = -V (n qoff qb* qa) ; df_a/dphi_a = -V L(n qoff qb* qa) V' ; df_a/dphi_b = V L(n qoff) R(qb* qa) V' */
                double qbc[4], t1[4], t2[4], t3[4], t4[4];
                qconj(qb, qbc);
                qmul(n4, M->qoff[j], t1); qmul(qbc, qa, t2); qmul(t1, t2, t3);
                Lmat(t3, M3);
                vblock(M3, blk);
                add3(FRB(a), mx, 0, cqa, blk, -1.0);
                double L1[16], R2[16];
                Lmat(t1, L1); Rmat(t2, R2);
                gemm(4, 4, 4, 1.0, L1, 4, 0, R2, 4, 0, 0.0, M3, 4);
                vblock(M3, blk);
                add3(FRB(a), mx, 0, cqb, blk, 1.0);
                (void)t4;
            }
        }
        /* ---- joint input: derivative w.r.t. state (only when u != 0) and w.r.t. u ---- */
        int iu = -1;
        for (int i = 0; i < mu; i++)
            if (cj[i] == j) iu = i;
        double u = uj[j];
        double ahat[3] = {M->ax[j][0], M->ax[j][1], M->ax[j][2]};
        double f[3] = {ahat[0] * u, ahat[1] * u, ahat[2] * u};
        double Sf[9], RaSf[9], yb[3], Syb[9], Raf[3];
        skew(f, Sf); mm3(Ra, Sf, RaSf);
        mat3vec(Ra, f, Raf); mat3Tvec(Rb, Raf, yb);
        skew(yb, Syb);
        if (M->type[j] == ORC_PRISMATIC) {
            if (u != 0.0) {
                if (a >= 0) {
                    add3(FTB(b), mx, 0, cqa, RaSf, -2.0);
                    add3(FTB(a), mx, 0, cqa, RaSf, 2.0);
                    mm3(RbtRa, Sf, T1); mm3(Sp2, T1, T2);
                    add3(FRB(b), mx, 0, cqa, T2, -4.0);
                }
                mm3(Sp2, Syb, T1);
                add3(FRB(b), mx, 0, cqb, T1, 4.0);
            }
            if (iu >= 0) {
                double Raa[3], yba[3], t[3];
                mat3vec(Ra, ahat, Raa); mat3Tvec(Rb, Raa, yba);
                cross(M->p2[j], yba, t);
                for (int i = 0; i < 3; i++) {
                    FTu[((size_t)b * 3 + i) * mu + iu] += Raa[i];
                    FRu[((size_t)b * 3 + i) * mu + iu] += 2.0 * t[i];
                }
                if (a >= 0) {
                    cross(M->p1[j], ahat, t);
                    for (int i = 0; i < 3; i++) {
                        FTu[((size_t)a * 3 + i) * mu + iu] -= Raa[i];
                        FRu[((size_t)a * 3 + i) * mu + iu] -= 2.0 * t[i];
                    }
                }
            }
        } else {
            if (u != 0.0) {
                if (a >= 0) {
                    mm3(RbtRa, Sf, T1);
                    add3(FRB(b), mx, 0, cqa, T1, -4.0);
                }
                add3(FRB(b), mx, 0, cqb, Syb, 4.0);
            }
            if (iu >= 0) {
                double Raa[3], yba[3];
                mat3vec(Ra, ahat, Raa); mat3Tvec(Rb, Raa, yba);
                for (int i = 0; i < 3; i++) FRu[((size_t)b * 3 + i) * mu + iu] += 2.0 * yba[i];
                if (a >= 0)
                    for (int i = 0; i < 3; i++) FRu[((size_t)a * 3 + i) * mu + iu] -= 2.0 * ahat[i];
            }
        }
    }

    memset(A, 0, sizeof(double) * mx * mx);
    if (mu > 0) memset(Bu, 0, sizeof(double) * mx * mu);
    memset(Bl, 0, sizeof(double) * mx * ml);
    memset(G, 0, sizeof(double) * ml * mx);
    for (int b = 0; b < nb; b++) {
        const double *w1 = z + 13 * b + 10, *w2 = s + 6 * b + 3, *J = M->J[b];
        double sq1 = sqrt(4.0 / (dt * dt) - (w1[0] * w1[0] + w1[1] * w1[1] + w1[2] * w1[2]));
        double sq2 = sqrt(4.0 / (dt * dt) - (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2]));
        double Jw1[3], Jw2[3], S[9], SJ[9], Sj[9], Dr[9], Dri[9], Psi[9];
        mat3vec(J, w1, Jw1); mat3vec(J, w2, Jw2);
        skew(w2, S);
        for (int i = 0; i < 9; i++) S[i] += (i % 4 == 0) ? sq2 : 0.0;
        mm3(S, J, SJ); skew(Jw2, Sj);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Dr[r * 3 + c] = SJ[r * 3 + c] - Sj[r * 3 + c] - Jw2[r] * w2[c] / sq2;
        if (inv_small(3, Dr, Dri)) return -1;
        /* dPsi/dw = (sq1 I - [w1]x) J + [J w1]x - (J w1) w1'/sq1 */
        skew(w1, S);
        for (int i = 0; i < 9; i++) S[i] = ((i % 4 == 0) ? sq1 : 0.0) - S[i];
        mm3(S, J, SJ); skew(Jw1, Sj);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Psi[r * 3 + c] = SJ[r * 3 + c] + Sj[r * 3 + c] - Jw1[r] * w1[c] / sq1;
        /* quaternion increment wq = (dt/2)(sq2, w2):  dqt+/dqt = V L(wq)' R(wq) V' ;  dqt+/dw+ = (dt/2) V L(wq)' [-w2'/sq2; I] */
        double wq[4] = {0.5 * dt * sq2, 0.5 * dt * w2[0], 0.5 * dt * w2[1], 0.5 * dt * w2[2]};
        double Lw[16], Rw[16], LtR[16], Eqq[9], Eqw[9], Tm[12];
        Lmat(wq, Lw); Rmat(wq, Rw);
        gemm(4, 4, 4, 1.0, Lw, 4, 1, Rw, 4, 0, 0.0, LtR, 4);
        vblock(LtR, Eqq);
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 3; c++) Tm[r * 3 + c] = (r == 0) ? -w2[c] / sq2 : ((r - 1 == c) ? 1.0 : 0.0);
        double LtT[12];
        gemm(4, 3, 4, 0.5 * dt, Lw, 4, 1, Tm, 3, 0, 0.0, LtT, 3);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Eqw[r * 3 + c] = LtT[(r + 1) * 3 + c];

        int r0 = 12 * b;
        /* ---- A rows of body b ---- */
        double *dv = (double *)calloc((size_t)3 * mx * 2, sizeof(double)), *dw = dv + 3 * mx;
        for (int i = 0; i < 3; i++) dv[(size_t)i * mx + r0 + 3 + i] = 1.0;
        for (int i = 0; i < 3; i++)
            for (int c = 0; c < mx; c++) dv[(size_t)i * mx + c] += dt / M->m[b] * FTB(b)[(size_t)i * mx + c];
        double *rhs = (double *)calloc((size_t)3 * mx, sizeof(double));
        memcpy(rhs, FRB(b), sizeof(double) * 3 * mx);
        add3(rhs, mx, 0, r0 + 9, Psi, 1.0);
        gemm(3, mx, 3, 1.0, Dri, 3, 0, rhs, mx, 0, 0.0, dw, mx);
        for (int i = 0; i < 3; i++)
            for (int c = 0; c < mx; c++) {
                A[(size_t)(r0 + i) * mx + c] = ((c == r0 + i) ? 1.0 : 0.0) + dt * dv[(size_t)i * mx + c];
                A[(size_t)(r0 + 3 + i) * mx + c] = dv[(size_t)i * mx + c];
                A[(size_t)(r0 + 9 + i) * mx + c] = dw[(size_t)i * mx + c];
            }
        gemm(3, mx, 3, 1.0, Eqw, 3, 0, dw, mx, 0, 0.0, A + (size_t)(r0 + 6) * mx, mx);
        add3(A, mx, r0 + 6, r0 + 6, Eqq, 1.0);
        free(rhs); free(dv);
        /* ---- Bu rows ---- */
        if (mu > 0) {
            double dwu[3 * MAXB], dqu[3 * MAXB];
            gemm(3, mu, 3, 1.0, Dri, 3, 0, FRu + (size_t)b * 3 * mu, mu, 0, 0.0, dwu, mu);
            gemm(3, mu, 3, 1.0, Eqw, 3, 0, dwu, mu, 0, 0.0, dqu, mu);
            for (int i = 0; i < 3; i++)
                for (int c = 0; c < mu; c++) {
                    double dvu = dt / M->m[b] * FTu[((size_t)b * 3 + i) * mu + c];
                    Bu[(size_t)(r0 + i) * mu + c] = dt * dvu;
                    Bu[(size_t)(r0 + 3 + i) * mu + c] = dvu;
                    Bu[(size_t)(r0 + 6 + i) * mu + c] = dqu[i * mu + c];
                    Bu[(size_t)(r0 + 9 + i) * mu + c] = dwu[i * mu + c];
                }
        }
        /* ---- Bl rows: columns of the joints touching b ---- */
        for (int j = 0; j < ne; j++) {
            const double *Gk = (M->child[j] == b) ? W->Gkb[j] : ((M->parent[j] == b) ? W->Gka[j] : 0);
            if (!Gk) continue;
            for (int r = 0; r < 5; r++) {
                double gx[3] = {Gk[r * 6], Gk[r * 6 + 1], Gk[r * 6 + 2]}, gp[3] = {Gk[r * 6 + 3], Gk[r * 6 + 4], Gk[r * 6 + 5]};
                double dwl[3], dql[3];
                mat3vec(Dri, gp, dwl); mat3vec(Eqw, dwl, dql);
                int col = 5 * j + r;
                for (int i = 0; i < 3; i++) {
                    double dvl = dt / M->m[b] * gx[i];
                    Bl[(size_t)(r0 + i) * ml + col] = dt * dvl;
                    Bl[(size_t)(r0 + 3 + i) * ml + col] = dvl;
                    Bl[(size_t)(r0 + 6 + i) * ml + col] = dql[i];
                    Bl[(size_t)(r0 + 9 + i) * ml + col] = dwl[i];
                }
            }
        }
    }
    /* ---- G = dg/dz+ at the next knot ---- */
    for (int j = 0; j < ne; j++) {
        int a = M->parent[j], b = M->child[j];
        const double *xa = a >= 0 ? zn + 13 * a : X0, *qa = a >= 0 ? zn + 13 * a + 3 : QID;
        double g[5], Xa[15], Qa[20], Xb[15], Qb[20], Pa[15], Pb[15];
        joint_eval(M, j, xa, qa, zn + 13 * b, zn + 13 * b + 3, g, Xa, Qa, Xb, Qb, 1);
        Q_to_phi(Qa, qa, Pa); Q_to_phi(Qb, zn + 13 * b + 3, Pb);
        for (int r = 0; r < 5; r++)
            for (int c = 0; c < 3; c++) {
                G[(size_t)(5 * j + r) * mx + 12 * b + c] = Xb[r * 3 + c];
                G[(size_t)(5 * j + r) * mx + 12 * b + 6 + c] = Pb[r * 3 + c];
                if (a >= 0) {
                    G[(size_t)(5 * j + r) * mx + 12 * a + c] = Xa[r * 3 + c];
                    G[(size_t)(5 * j + r) * mx + 12 * a + 6 + c] = Pa[r * 3 + c];
                }
            }
    }
    free(FT); free(FTu);
    return 0;
}

int orc_linearize(const orc_mech_desc *d, const double *zd, int32_t mu, const int32_t *ctrl_joint, const double *Fd, double *A, double *Bu,
                  double *Bl, double *G) {
    mech_t M;
    if (mech_build(d, &M)) return -1;
    work_t *W = (work_t *)malloc(sizeof(work_t));
    int rc = linearize_core(&M, zd, mu, ctrl_joint, Fd, A, Bu, Bl, G, W);
    free(W);
    return rc;
}

/* ------------------------------------------------------------------ Riccati recursion (a5, a10) */
typedef struct {
    int mx, mu, ml, m;
    double *GBl, *GBu, *Y, *D, *DtP, *Mm, *bb, *Abar, *T1, *Pn, *KRK;
    int *piv;
} ric_ws;
static void ric_alloc(ric_ws *w, int mx, int mu, int ml) {
    w->mx = mx; w->mu = mu; w->ml = ml; w->m = mu + ml;
    int m = w->m;
    w->GBl = (double *)malloc(sizeof(double) * (ml * ml + 1)); w->GBu = (double *)malloc(sizeof(double) * (ml * mu + 1));
    w->Y = (double *)malloc(sizeof(double) * (mx * ml + 1)); w->D = (double *)malloc(sizeof(double) * (mx * mu + 1));
    w->DtP = (double *)malloc(sizeof(double) * (mu * mx + 1)); w->Mm = (double *)malloc(sizeof(double) * (m * m + 1));
    w->bb = (double *)malloc(sizeof(double) * (m * mx + 1)); w->Abar = (double *)malloc(sizeof(double) * mx * mx);
    w->T1 = (double *)malloc(sizeof(double) * mx * mx); w->Pn = (double *)malloc(sizeof(double) * mx * mx);
    w->KRK = (double *)malloc(sizeof(double) * (mu * mx + 1)); w->piv = (int *)malloc(sizeof(int) * (m + ml + 1));
}
static void ric_free(ric_ws *w) {
    free(w->GBl); free(w->GBu); free(w->Y); free(w->D); free(w->DtP); free(w->Mm); free(w->bb); free(w->Abar); free(w->T1);
    free(w->Pn); free(w->KRK); free(w->piv);
}
/* one backward step, lqr.jl:151-170.  P (in) -> w->Pn (out); Kk rows written to Ku (mu x mx). returns ||P - Pn||_F or <0 */
static double ric_step(ric_ws *w, const double *A, const double *Bu, const double *Bl, const double *G, const double *Q, const double *R,
                       const double *P, double *Ku) {
    int mx = w->mx, mu = w->mu, ml = w->ml, m = w->m;
    /* D = Bu - Bl/(G*Bl)*G*Bu                                            lqr.jl:151 */
    memcpy(w->D, Bu, sizeof(double) * mx * mu);
    if (ml > 0) {
        gemm(ml, ml, mx, 1.0, G, mx, 0, Bl, ml, 0, 0.0, w->GBl, ml);
        gemm(ml, mu, mx, 1.0, G, mx, 0, Bu, mu, 0, 0.0, w->GBu, mu);
        /* Y = Bl (G Bl)^-1  via  (G Bl)' Y' = Bl' */
        double *GBlT = (double *)malloc(sizeof(double) * ml * ml), *Yt = (double *)malloc(sizeof(double) * ml * mx);
        for (int r = 0; r < ml; r++)
            for (int c = 0; c < ml; c++) GBlT[r * ml + c] = w->GBl[c * ml + r];
        for (int r = 0; r < ml; r++)
            for (int c = 0; c < mx; c++) Yt[r * mx + c] = Bl[c * ml + r];
        if (lu_factor(ml, GBlT, ml, w->piv)) { free(GBlT); free(Yt); return -1.0; }
        lu_solve(ml, GBlT, ml, w->piv, Yt, mx, mx);
        for (int r = 0; r < mx; r++)
            for (int c = 0; c < ml; c++) w->Y[r * ml + c] = Yt[c * mx + r];
        free(GBlT); free(Yt);
        gemm(mx, mu, ml, -1.0, w->Y, ml, 0, w->GBu, mu, 0, 1.0, w->D, mu);
    }
    /* M = [R + D'P Bu, D'P Bl; G Bu, G Bl] ; b = [D'P; G] A              lqr.jl:152-158 */
    gemm(mu, mx, mx, 1.0, w->D, mu, 1, P, mx, 0, 0.0, w->DtP, mx);
    for (int r = 0; r < mu; r++)
        for (int c = 0; c < mu; c++) w->Mm[r * m + c] = R[r * mu + c];
    gemm(mu, mu, mx, 1.0, w->DtP, mx, 0, Bu, mu, 0, 1.0, w->Mm, m);
    if (ml > 0) {
        gemm(mu, ml, mx, 1.0, w->DtP, mx, 0, Bl, ml, 0, 0.0, w->Mm + mu, m);
        for (int r = 0; r < ml; r++) {
            for (int c = 0; c < mu; c++) w->Mm[(mu + r) * m + c] = w->GBu[r * mu + c];
            for (int c = 0; c < ml; c++) w->Mm[(mu + r) * m + mu + c] = w->GBl[r * ml + c];
        }
    }
    gemm(mu, mx, mx, 1.0, w->DtP, mx, 0, A, mx, 0, 0.0, w->bb, mx);
    if (ml > 0) gemm(ml, mx, mx, 1.0, G, mx, 0, A, mx, 0, 0.0, w->bb + (size_t)mu * mx, mx);
    /* Kk = M \ b                                                          lqr.jl:160 */
    if (lu_factor(m, w->Mm, m, w->piv)) return -1.0;
    lu_solve(m, w->Mm, m, w->piv, w->bb, mx, mx);
    memcpy(Ku, w->bb, sizeof(double) * mu * mx);
    /* Abar = A - Bu Kuk - Bl Klk ; Pkp1 = Q + Kuk' R Kuk + Abar' Pk Abar  lqr.jl:169-170 */
    memcpy(w->Abar, A, sizeof(double) * mx * mx);
    gemm(mx, mx, mu, -1.0, Bu, mu, 0, w->bb, mx, 0, 1.0, w->Abar, mx);
    if (ml > 0) gemm(mx, mx, ml, -1.0, Bl, ml, 0, w->bb + (size_t)mu * mx, mx, 0, 1.0, w->Abar, mx);
    memcpy(w->Pn, Q, sizeof(double) * mx * mx);
    gemm(mu, mx, mu, 1.0, R, mu, 0, w->bb, mx, 0, 0.0, w->KRK, mx);
    gemm(mx, mx, mu, 1.0, w->bb, mx, 1, w->KRK, mx, 0, 1.0, w->Pn, mx);
    gemm(mx, mx, mx, 1.0, w->Abar, mx, 1, P, mx, 0, 0.0, w->T1, mx);
    gemm(mx, mx, mx, 1.0, w->T1, mx, 0, w->Abar, mx, 0, 1.0, w->Pn, mx);
    double n = 0;
    for (int i = 0; i < mx * mx; i++) { double df = P[i] - w->Pn[i]; n += df * df; }
    FL(3.0 * mx * mx);
    return sqrt(n);
}

int orc_riccati(int32_t mx, int32_t mu, int32_t ml, const double *A, const double *Bu, const double *Bl, const double *G, const double *Q,
                const double *R, int32_t N, double tol, double *K, int32_t *kbreak) {
    ric_ws w;
    ric_alloc(&w, mx, mu, ml);
    double *P = (double *)malloc(sizeof(double) * mx * mx);
    memcpy(P, Q, sizeof(double) * mx * mx); /* Pk = Q, lqr.jl:147 */
    int k = 0, rc = 0;
    for (k = N - 1; k >= 1; k--) {
        double nrm = ric_step(&w, A, Bu, Bl, G, Q, R, P, K + (size_t)(k - 1) * mu * mx);
        if (nrm < 0) { rc = -2; break; }
        if (nrm < tol) break;                           /* lqr.jl:172-174: break BEFORE Pk is updated */
        memcpy(P, w.Pn, sizeof(double) * mx * mx);      /* lqr.jl:176 */
    }
    if (k < 1 && N - 1 >= 1) k = 1; /* Julia: after a completed loop the outer k holds its last value, 1 */
    if (N - 1 < 1) k = 0;
    for (int k2 = k - 1; k2 >= 1; k2--)                 /* lqr.jl:179-181 back-fill */
        memcpy(K + (size_t)(k2 - 1) * mu * mx, K + (size_t)k2 * mu * mx, sizeof(double) * mu * mx);
    if (kbreak) *kbreak = k;
    free(P);
    ric_free(&w);
    return rc;
}

int orc_riccati_tracking(const orc_mech_desc *d, int32_t mu, const int32_t *cj, const double *zd, const double *Fd, const double *Q,
                         const double *R, int32_t N, double tol, double *K, int32_t *kbreak) {
    mech_t M;
    if (mech_build(d, &M)) return -1;
    int mx = 12 * M.nb, ml = 5 * M.ne;
    ric_ws w;
    ric_alloc(&w, mx, mu, ml);
    work_t *W = (work_t *)malloc(sizeof(work_t));
    double *P = (double *)malloc(sizeof(double) * mx * mx), *A = (double *)malloc(sizeof(double) * mx * mx);
    double *Bu = (double *)malloc(sizeof(double) * mx * (mu + 1)), *Bl = (double *)malloc(sizeof(double) * mx * ml),
           *G = (double *)malloc(sizeof(double) * ml * mx);
    memcpy(P, Q, sizeof(double) * mx * mx);
    int k = 0, rc = 0;
    for (k = N - 1; k >= 1; k--) {
        /* lqr_tracking.jl:88: re-linearise at knot k */
        if (linearize_core(&M, zd + (size_t)(k - 1) * 13 * M.nb, mu, cj, Fd + (size_t)(k - 1) * mu, A, Bu, Bl, G, W)) { rc = -3; break; }
        double nrm = ric_step(&w, A, Bu, Bl, G, Q, R, P, K + (size_t)(k - 1) * mu * mx);
        if (nrm < 0) { rc = -2; break; }
        if (nrm < tol) break;
        memcpy(P, w.Pn, sizeof(double) * mx * mx);
    }
    if (k < 1 && N - 1 >= 1) k = 1;
    if (N - 1 < 1) k = 0;
    for (int k2 = k - 1; k2 >= 1; k2--) memcpy(K + (size_t)(k2 - 1) * mu * mx, K + (size_t)k2 * mu * mx, sizeof(double) * mu * mx);
    if (kbreak) *kbreak = k;
    free(P); free(A); free(Bu); free(Bl); free(G); free(W);
    ric_free(&w);
    return rc;
}
