/*
 * cclqr_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C fp64 restatement of the batched LQR-rollout hot path of
 * janbruedigam/ConstrainedControl.jl v0.3.0:
 *   - constrained Riccati recursion        src/control/lqr.jl:141-184
 *   - time-varying twin                    src/control/lqr_tracking.jl:73-122
 *   - feedback law                         src/control/lqr.jl:89-139,
 *                                          src/control/lqr_tracking.jl:46-71,
 *                                          examples/trackingLQR_triple_cartpole.jl:76-115
 *   - integrator / Newton / LDU / linearsystem: these live in the un-vendored
 *     dependency ConstrainedDynamics.jl (^0.9.1, Project.toml:7,12) which is NOT
 *     present in /root/reference.  They are restated from the published
 *     algorithm (arXiv:2002.11245, arXiv:2010.05886) following SURVEY.md 8a-bis
 *     and anchored on the reference's call sites (lqr.jl:63,98-103,109).
 *
 * PARITY STATUS: the Riccati recursion follows the reference source line by
 * line and is pinned against scipy's DARE (the algorithm of src/util/util.jl:1-19).
 * Everything that lives in ConstrainedDynamics is **parity unpinned**: the
 * reference's tests hold no numeric fixture (every test ends in `@test true`)
 * and no Julia toolchain exists here or on the GPU box.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in oracle/ -- as the checker, never as the thing shipped.
 */
#ifndef CCLQR_ORACLE_H
#define CCLQR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_REVOLUTE 0
#define ORC_PRISMATIC 1

/* Mechanism description: a tree of nb bodies hanging off the origin through ne == nb
 * 1-DoF joints (5 constraint rows each).  Same field order as cclqr_mech_desc. */
typedef struct {
    int32_t nb, ne;
    double dt, g;          /* step, gravity along z (g = -9.81 pulls towards -z) */
    const double *mass;    /* [nb] */
    const double *inertia; /* [nb][9] body-frame inertia, row major */
    const int32_t *parent; /* [ne] body index, -1 = origin */
    const int32_t *child;  /* [ne] body index */
    const int32_t *type;   /* [ne] ORC_REVOLUTE / ORC_PRISMATIC */
    const double *p1;      /* [ne][3] joint vertex in the parent frame */
    const double *p2;      /* [ne][3] joint vertex in the child frame */
    const double *axis;    /* [ne][3] joint axis in the parent frame */
    const double *qoff;    /* [ne][4] orientation offset (scalar first) */
} orc_mech_desc;

/* Controller description (LQR / TrackingLQR / open loop), see lqr.jl:3-15, lqr_tracking.jl:3-15 */
typedef struct {
    int32_t mu;                /* number of controlled joints */
    const int32_t *ctrl_joint; /* [mu] joint indices */
    int32_t nK;                /* gain matrices stored: N-1, or 1 for the Inf-horizon controller */
    int32_t N;                 /* horizon in steps (gate k<N, lqr.jl:106); N<=0: infinite horizon (lqr.jl:116) */
    const double *K;           /* [nK][mu][12*nb], NULL => no feedback */
    int32_t nsp;               /* setpoints stored: 1 (LQR) or N (TrackingLQR) */
    const double *zd;          /* [nsp][nb][13] = x(3) q(4) v(3) w(3) */
    const double *Fd;          /* [nsp][mu] feed-forward */
    const double *fric;        /* [ne] viscous joint friction (trackingLQR_triple_cartpole.jl:98-101) or NULL */
    double noise_scale;        /* multiplies noise[] on every controlled joint (same file :98) */
    /* PID{T,N} on 1-DoF joints in minimal coordinates, src/control/pid.jl:3-88 (npid = 0: none) */
    int32_t npid;
    const int32_t *pid_joint;  /* [npid] joint indices */
    const double *pid_P, *pid_I, *pid_D, *pid_goal; /* [npid] */
    int32_t noise_philox;      /* != 0 and noise == NULL: Philox-4x32-10 + Box-Muller stream per instance */
    uint64_t noise_seed;
    int32_t n_ctrl;            /* > 1: instance n uses table n of K [n_ctrl][nK][mu][12 nb], zd [n_ctrl][nsp][nb][13], Fd [n_ctrl][nsp][mu] */
    /* (fields above = cclqr_ctrl_desc; the injected noise array is oracle-only and therefore last) */
    const double *noise;       /* [n_inst][steps] injected standard-normal samples or NULL */
} orc_ctrl_desc;

/* one integrator step for one instance; uj[ne] = joint-space input per joint.
 * z[nb][13] in/out, lam[5*ne] in/out (warm start). returns Newton iterations (<0: not converged) */
int orc_step(const orc_mech_desc *m, double *z, double *lam, const double *uj);

/* step with lambda treated as an exogenous input and no constraint solve (for validating the linearisation) */
void orc_step_fixed_lambda(const orc_mech_desc *m, const double *z, const double *lam, const double *uj, double *znext);

/* one joint from explicit parameters (g[5], Ga/Gb [5][6] = dg/d(x, phi) of parent / child; xa = qa = NULL: parent is the origin):
 * building block of oracle/loops.py, the dense-KKT stepper for closed loops (examples/lqr_deltabot.jl:25-33) */
void orc_joint_blocks(int32_t type, const double *p1, const double *p2, const double *axis, const double *qoff, const double *xa,
                      const double *qa, const double *xb, const double *qb, double *g, double *Ga, double *Gb);

/* constraint values g (5*ne) at state z */
void orc_constraints(const orc_mech_desc *m, const double *z, double *g);

/* minimalCoordinates(mechanism, eqc)[1] of every joint (angle about / offset along the joint axis), pid.jl:45,55 */
void orc_minimal_coordinates(const orc_mech_desc *m, const double *z, double *theta);

/* feedback law: writes uj[ne] for step k (1-based) */
void orc_control(const orc_mech_desc *m, const orc_ctrl_desc *c, const double *z, int k, double noise_sample, double *uj);

/* batched rollout. z0 [n_inst][nb][13]; traj (or NULL) [n_inst][steps][nb][13] holds the state *before* each step;
 * zT [n_inst][nb][13]; status[n_inst] = max Newton iterations, negative if any step failed to converge. */
int orc_rollout(const orc_mech_desc *m, const orc_ctrl_desc *c, int64_t n_inst, int32_t steps,
                const double *z0, double *traj, double *zT, int32_t *status, int32_t nthreads);

/* linearsystem(...) call sites lqr.jl:63, lqr_tracking.jl:88.  Row-major outputs:
 * A [mx][mx], Bu [mx][mu], Bl [mx][ml], G [ml][mx]  with mx = 12 nb, ml = 5 ne */
int orc_linearize(const orc_mech_desc *m, const double *zd, int32_t mu, const int32_t *ctrl_joint, const double *Fd,
                  double *A, double *Bu, double *Bl, double *G);

/* dlqr(A,Bu,Bl,G,Q,R,N), lqr.jl:141-184.  K [N-1][mu][mx]; *kbreak = value of k after the loop (1-based, 0 if none) */
int orc_riccati(int32_t mx, int32_t mu, int32_t ml, const double *A, const double *Bu, const double *Bl, const double *G,
                const double *Q, const double *R, int32_t N, double tol, double *K, int32_t *kbreak);

/* dlqr(mechanism, xd,...,N), lqr_tracking.jl:73-122.  zd [N][nb][13], Fd [N][mu], K [N-1][mu][mx] */
int orc_riccati_tracking(const orc_mech_desc *m, int32_t mu, const int32_t *ctrl_joint, const double *zd, const double *Fd,
                         const double *Q, const double *R, int32_t N, double tol, double *K, int32_t *kbreak);

/* Philox-4x32-10 block function and the standard-normal sample (instance, step k) derived from it (noise_philox above) */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_philox_normal(uint64_t seed, uint64_t instance, int k);

/* flop counter (instrumented build only: -DORC_COUNT_FLOPS), see SURVEY 8d */
double orc_flops_get(void);
void orc_flops_reset(void);
/* Newton / line-search statistics of the calling thread (instrumented build only; layout documented at the definition) */
void orc_newton_stats(double *out64, int reset);
/* residual at the start of every Newton iteration, binned (instrumented build only; layout at the definition) */
void orc_newton_hist(double *out96, int reset);
/* 0 (default) = the reference's rule, ||f|| < eps AND ||step taken|| < eps; 1 = MODEL of the device's frozen-Jacobian iterations once ||f|| < eps (not a
 * parity mode); 2 = MODEL of a residual-only rule, ||f|| < eps alone (what tests/test_reference_fixtures.py compares the reference's Newton iteration counts
 * with, next to rule 0, to say which of the two the un-vendored dependency follows) */
void orc_set_newton_variant(int v);

#ifdef __cplusplus
}
#endif
#endif
