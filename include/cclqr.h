/*
 * cclqr.h -- C ABI of libcclqr.so: the MI355X (gfx950) implementation of ConstrainedControl.jl's batched
 * LQR-rollout hot path.  Plain pointers and sizes only; every entry point returns an int status
 * (0 = ok, < 0 = error, see CCLQR_E*); no exceptions cross the boundary.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference repo,
 * janbruedigam/ConstrainedControl.jl v0.3.0).  The reference has no batch API and no FFI of its own:
 * INTEGRATION.md shows the `ccall` stubs a maintainer adds on the Julia side.
 *
 * Layouts (all fp64, all dense arrays row-major unless stated):
 *   body state   z[13]   = x(3) world, q(4) scalar-first unit quaternion body->world, v(3) world, w(3) body frame
 *   batch state  z0[n_inst][nb][13]            (an instance = 13*nb contiguous doubles -> coalesced wave loads)
 *   trajectory   traj[n_inst][steps][nb][13]   = Storage x/q/v/w of every body at every step (lqr_tracking.jl:32-35)
 *   gains        K[nK][mu][12*nb]              = lqr.K[k][i] rows (lqr.jl:4); error order per body x,v,q~,w (lqr.jl:92-95)
 *   linear model A[mx][mx], Bu[mx][mu], Bl[mx][ml], G[ml][mx], mx = 12*nb, ml = 5*ne
 */
#ifndef CCLQR_H
#define CCLQR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCLQR_OK 0
#define CCLQR_EINVAL -1       /* size / topology mismatch  (the reference's @assert, lqr.jl:59-60) */
#define CCLQR_ESINGULAR -2    /* G*Bl or M singular        (LAPACK exception from lqr.jl:151,160) */
#define CCLQR_ENOCONV -3      /* soft: Newton / Riccati did not converge (lqr.jl:41 `@info`) */
#define CCLQR_EHIP -4         /* HIP runtime error; cclqr_last_error() has the text */
#define CCLQR_EUNSUPPORTED -5 /* valid for the reference, outside this build's scope (> 4 child joints on a body, nb > 64, loop mechanisms beyond 8 bodies / 12 joints) */

/* ABI version = cclqr_version().  A shim built against another header must refuse to run: the structs below are passed by pointer and
 * read in full.  200: cclqr_ctrl_desc.n_ctrl, cclqr_rollout_opts {noise_ws_dev, noise_ws_len, newton_mode}, cclqr_riccati_opts.keep_last,
 * the thread-local setters (cclqr_set_instance_offset, cclqr_set_pid_state, cclqr_riccati_path) removed.
 * 201 (additive over 200: same struct sizes and offsets): cclqr_rollout_opts.reserved became `flags` (CCLQR_ROLLOUT_NO_ALLOC), new entry points
 * cclqr_ctrl_set_feedforward and cclqr_abi_layout.  A shim checks cclqr_version() == the version it was written against AND, through
 * cclqr_abi_layout(), the sizeof / offsetof of every struct it mirrors. */
#define CCLQR_ABI_VERSION 201

#define CCLQR_REVOLUTE 0      /* EqualityConstraint(Revolute(a, b, axis; p1, p2, qoffset)),  examples/lqr_cartpole.jl:26 */
#define CCLQR_PRISMATIC 1     /* EqualityConstraint(Prismatic(a, b, axis; p1, p2, qoffset)), examples/lqr_cartpole.jl:25 */
#define CCLQR_FIXED_ORIENTATION 2 /* EqualityConstraint(FixedOrientation(a, b; qoffset)), examples/lqr_deltabot.jl:25 (closed-loop mechanisms) */

/* Mechanism(origin, bodies, eqconstraints; g, Δt) -- examples/lqr_cartpole.jl:32.
 * A tree of nb bodies, each hung off its parent (or the origin, -1) by one 1-DoF joint (ne == nb); a body may carry up to 4 child joints.
 * Forests of chains and trees with branching bodies: up to 64 bodies (one lane of a wavefront per link).
 * Closed kinematic loops (examples/lqr_deltabot.jl:25-33: ne > nb, or a body that is the child of two joints, or a FixedOrientation
 * constraint; up to 8 bodies and 12 joints): cclqr_rollout* run them (LQR / TrackingLQR law with joint friction and noise, PID; one instance per wavefront, multipliers lam
 * [n_inst][5*ne]); cclqr_linearize returns their A, Bu, Bλ, G (ml = 5*ne rows, a FixedOrientation contributing two null rows), but G*Bλ
 * is singular for a loop, so LQR / TrackingLQR construction goes through cclqr_linearize_projected + cclqr_riccati / cclqr_riccati_tv with
 * ml = 0 (cclqr_riccati_tracking, which divides by G*Bλ at every knot, returns CCLQR_EUNSUPPORTED for them). */
typedef struct {
    int32_t nb, ne;
    double dt, g;          /* mechanism.Δt (lqr.jl:65), gravity along z */
    const double *mass;    /* [nb] */
    const double *inertia; /* [nb][9] body frame */
    const int32_t *parent; /* [ne] body index or -1 */
    const int32_t *child;  /* [ne] body index */
    const int32_t *type;   /* [ne] CCLQR_REVOLUTE / CCLQR_PRISMATIC */
    const double *p1;      /* [ne][3] vertex in the parent frame */
    const double *p2;      /* [ne][3] vertex in the child frame */
    const double *axis;    /* [ne][3] axis in the parent frame */
    const double *qoff;    /* [ne][4] orientation offset */
} cclqr_mech_desc;

/* LQR{T,N,NK} (lqr.jl:3-15) / TrackingLQR{T,N,NK} (lqr_tracking.jl:3-15) as flat tables. */
typedef struct {
    int32_t mu;                /* length(eqcids) */
    const int32_t *ctrl_joint; /* [mu] joint indices (eqcids, 0-based joint index) */
    int32_t nK;                /* gain matrices: N-1, or 1 for LQR{T,Inf} (lqr.jl:42) */
    int32_t N;                 /* horizon in steps; feedback is gated by k < N (lqr.jl:106); N <= 0: infinite horizon (lqr.jl:116) */
    const double *K;           /* [nK][mu][12*nb] or NULL (open loop) */
    int32_t nsp;               /* 1: xd,vd,qd,ωd of LQR ; N: per-step setpoints of TrackingLQR (lqr_tracking.jl:6-9) */
    const double *zd;          /* [nsp][nb][13] */
    const double *Fd;          /* [nsp][mu]  Fτd (lqr.jl:12, lqr_tracking.jl:12) */
    const double *fric;        /* [ne] viscous joint friction of examples/trackingLQR_triple_cartpole.jl:98-101, or NULL */
    double noise_scale;        /* cart noise amplitude, same file :98 (`randn()*2`); 0 = none */
    /* PID{T,N} (src/control/pid.jl:3-40) on 1-DoF joints in minimal coordinates; control_pid! (pid.jl:69-88) runs every step.
     * The integrated / last errors live for one launch unless cclqr_rollout_opts.pid_state_dev provides a buffer. npid = 0: none. */
    int32_t npid;
    const int32_t *pid_joint;  /* [npid] joint indices (eqcids) */
    const double *pid_P, *pid_I, *pid_D, *pid_goal; /* [npid]  P, I, D, goals (pid.jl:4-9) */
    /* `randn()` of trackingLQR_triple_cartpole.jl:98,125 as a reproducible counter-based stream: when noise_philox != 0
     * and no noise array is passed, sample (instance n, step k) = Box-Muller of Philox-4x32-10(counter = (k-1, 0, 0, 0),
     * key = (noise_seed low 32 bits ^ high 32 bits, n)), times noise_scale. */
    int32_t noise_philox;
    uint64_t noise_seed;
    /* Batched controllers ("Riccati run per instance on distinct setpoints"): n_ctrl > 1 gives every instance its
     * own table -- K [n_ctrl][nK][mu][12*nb], zd [n_ctrl][nsp][nb][13], Fd [n_ctrl][nsp][mu] -- and instance n of a launch reads table
     * first_instance + n (cclqr_rollout_opts), which must be < n_ctrl.  0 or 1: one table for all instances = the reference's one
     * LQR object per simulate! call (lqr.jl:106-111). */
    int32_t n_ctrl;
} cclqr_ctrl_desc;

typedef struct cclqr_mech cclqr_mech; /* opaque: device-resident mechanism tables */
typedef struct cclqr_ctrl cclqr_ctrl; /* opaque: device-resident controller tables */

const char *cclqr_last_error(void);
int cclqr_version(void);     /* CCLQR_ABI_VERSION of the library that was loaded */
/* sizeof and offsetof of the four structs that cross this boundary, as THIS library was compiled, so that a foreign-language mirror (the
 * Julia structs of julia/CCLQR.jl, the ctypes Structures of _capi.py -- the reference has no FFI of its own: INTEGRATION.md) can verify its layout
 * at load time instead of trusting a version number.  out[0..n) receives, in this order (CCLQR_ABI_LAYOUT_LEN values):
 *   cclqr_mech_desc:    sizeof, offsetof nb, ne, dt, g, mass, inertia, parent, child, type, p1, p2, axis, qoff                       (14)
 *   cclqr_ctrl_desc:    sizeof, offsetof mu, ctrl_joint, nK, N, K, nsp, zd, Fd, fric, noise_scale, npid, pid_joint, pid_P, pid_I, pid_D,
 *                       pid_goal, noise_philox, noise_seed, n_ctrl                                                                   (20)
 *   cclqr_riccati_opts: sizeof, offsetof path, bf16_terms, keep_last, reserved                                                       (5)
 *   cclqr_rollout_opts: sizeof, offsetof first_instance, pid_state_dev, pid_state_len, noise_ws_dev, noise_ws_len, newton_mode, flags,
 *                       newton_eps_alone                                                                                             (9)
 * Returns the number of values the library has (48), writes min(n, 48) of them; out may be NULL with n = 0. */
#define CCLQR_ABI_LAYOUT_LEN 48
int cclqr_abi_layout(int32_t *out, int32_t n);
int cclqr_device_count(int32_t *n);
int cclqr_set_device(int32_t dev);

/* Mechanism(...) constructor (examples/lqr_cartpole.jl:32): validates the topology, orders links chain by chain, uploads the tables. */
int cclqr_mech_create(const cclqr_mech_desc *desc, cclqr_mech **out);
int cclqr_mech_destroy(cclqr_mech *m);

/* LQR(...) (lqr.jl:17-48) / TrackingLQR(...) (lqr_tracking.jl:17-44) result as a device object consumed by cclqr_rollout*. */
int cclqr_ctrl_create(const cclqr_mech *m, const cclqr_ctrl_desc *desc, cclqr_ctrl **out);
int cclqr_ctrl_destroy(cclqr_ctrl *c);

/* LQR(mechanism, bodyids, eqcids, Q, R, horizon; xd, vd, qd, ωd, Fτd) (lqr.jl:50-66) for n_ctrl setpoints at once, entirely on the device
 * (SURVEY 8d configs[3]: "Riccati run per instance on distinct setpoints"): linearsystem at every zd[i] (lqr.jl:63), dlqr (lqr.jl:141-184)
 * and the per-instance controller tables of cclqr_ctrl_create (instance n of a rollout reads table first_instance + n) without the gains
 * -- n_ctrl (N-1) mu 12 nb doubles: 1 GB for 1024 seven-joint arms at N = 200 -- ever visiting the host.  Q [mx][mx], R [mu][mu] already Δt-scaled
 * (lqr.jl:18-19), N = horizon in steps, zd [n_ctrl][nb][13], Fd [n_ctrl][mu] or NULL, kbreak [n_ctrl] or NULL.  Tree mechanisms.
 * infinite_horizon != 0: LQR{T,Inf} (horizon = Inf, lqr.jl:25-27 and 40-43) -- N is Ntemp = ceil(10/Δt), the recursion runs its N-1 steps
 * (or breaks, lqr.jl:172), only Ku[1] is kept (n_ctrl mu 12 nb doubles: 39 MB for 8192 seven-joint arms) and control_lqr! is never gated
 * (lqr.jl:116-139); a problem whose recursion has not converged is reported through kbreak[i] == 1 (lqr.jl:41 `@info`). */
int cclqr_ctrl_create_lqr_batch(const cclqr_mech *m, int32_t n_ctrl, const double *zd, int32_t mu, const int32_t *ctrl_joint, const double *Fd,
                                const double *Q, const double *R, int32_t N, int32_t infinite_horizon, double tol, int32_t *kbreak,
                                cclqr_ctrl **out);

/* linearsystem(mechanism, xd, vd, qd, ωd, Fτd, bodyids, eqcids) -- call sites lqr.jl:63, lqr_tracking.jl:88.
 * Batched over nk knots (nk = 1 for LQR, N-1 for TrackingLQR).  Host pointers.
 * zd [nk][nb][13], Fd [nk][mu]; outputs A [nk][mx][mx], Bu [nk][mx][mu], Bl [nk][mx][ml], G [nk][ml][mx], ml = 5*ne. */
int cclqr_linearize(const cclqr_mech *m, int32_t nk, const double *zd, int32_t mu, const int32_t *ctrl_joint, const double *Fd,
                    double *A, double *Bu, double *Bl, double *G);

/* The same linear model with the multipliers eliminated -- A' = A - Bλ (G Bλ)^-1 G A and D = Bu - Bλ (G Bλ)^-1 G Bu, the projected pair the
 * recursion of lqr.jl:151-170 works with.  Defined for every topology cclqr_mech_create takes, and the only linearisation of closed-loop
 * mechanisms (examples/lqr_deltabot.jl:47-53), where G Bλ is singular (redundant constraint rows) but A', D are still unique; feed them to
 * cclqr_riccati with ml = 0.
 *   h <= 0: ANALYTIC -- the exact Jacobians of the one-step map on the device (linearsystem of lqr.jl:63 with the multipliers exogenous, for
 *           loops in their own bookkeeping), then G Bλ eliminated with complete pivoting up to its numerical rank.  The elimination keeps
 *           [G Bλ | G A | G Bu] of a knot in one compute unit's 160 KB of LDS; a mechanism it does not fit (from about 15 bodies: the 17-body
 *           chain needs 198 KB) is differenced with h = 1e-6 instead, so the call is defined for every mechanism either way;
 *   h > 0:  central differences (step h in every error coordinate x, v, q~, ω of lqr.jl:92-103 and every input) of the DEVICE's own constrained
 *           one-step map, one launch of nk (1 + 24 nb + 2 mu) single-step rollouts: an independent cross-check of the analytic form.
 * Ap [nk][mx][mx], D [nk][mx][mu]. Host pointers. */
int cclqr_linearize_projected(const cclqr_mech *m, int32_t nk, const double *zd, int32_t mu, const int32_t *ctrl_joint, const double *Fd,
                              double h, double *Ap, double *D);

/* dlqr(A, Bu, Bλ, G, Q, R, N) -- lqr.jl:141-184, batched over nprob independent problems (nprob = 1 in the reference).
 * Q [mx][mx] and R [mu][mu] are the already Δt-scaled block-diagonal weights (lqr.jl:18-19).
 * K [nprob][N-1][mu][mx] ([nprob][mu][mx] with cclqr_riccati_opts.keep_last); kbreak [nprob] = value of the loop index k after the loop
 * (lqr.jl:172-181). Host pointers. */
int cclqr_riccati(int32_t nprob, int32_t mx, int32_t mu, int32_t ml, const double *A, const double *Bu, const double *Bl, const double *G,
                  const double *Q, const double *R, int32_t N, double tol, double *K, int32_t *kbreak);

/* Options of cclqr_riccati_ex / cclqr_riccati_tracking_ex (NULL or all zero = defaults).
 *   path        launch shape: 0 = choose by problem size and count, 1 = one LDS-resident workgroup per problem whenever the problem fits
 *               (mx up to ~96), 2 = every backward step tiled over the whole device.  Same results either way.
 *   bf16_terms  0 = fp64 MFMA, the PARITY mode (gains equal the reference recursion's to 1e-7).  1..3 = measured-error mode of
 *               BASELINE configs[3] ("dense Riccati on MFMA bf16 -> fp32 accumulate"): the two mx^3 products of a backward step
 *               (lqr.jl:170) run on v_mfma_f32_16x16x16_bf16 with fp32 accumulation, every fp64 operand split into that many bf16
 *               terms (1 = plain bf16, 3 = bf16x3).  Available on the tiled path for every shape and on the LDS-resident batched
 *               kernel for the Sawyer (mx 84, mu 7) and cartpole (mx 24, mu 1) shapes; other shapes take the tiled path.  Its gain
 *               error and speed are reported, not promised: it does NOT reproduce the fp64 gains and the 1e-5 break
 *               test of lqr.jl:172 never fires.
 *   keep_last   != 0: K is [nprob][mu][mx] and receives only the gain of the last executed backward step = Ku[1] after the back-fill of
 *               lqr.jl:179-181 = the single gain LQR{T,Inf} keeps (lqr.jl:40-43); the (N-1)-fold table is never materialised. */
typedef struct {
    int32_t path;
    int32_t bf16_terms;
    int32_t keep_last;
    int32_t reserved;
} cclqr_riccati_opts;

int cclqr_riccati_ex(int32_t nprob, int32_t mx, int32_t mu, int32_t ml, const double *A, const double *Bu, const double *Bl, const double *G,
                     const double *Q, const double *R, int32_t N, double tol, double *K, int32_t *kbreak, const cclqr_riccati_opts *opts);
int cclqr_riccati_tracking_ex(const cclqr_mech *m, int32_t mu, const int32_t *ctrl_joint, const double *zd, const double *Fd, const double *Q,
                              const double *R, int32_t N, double tol, double *K, int32_t *kbreak, const cclqr_riccati_opts *opts);

/* Device workspaces of the host-pointer entry points (cclqr_linearize, cclqr_riccati*, cclqr_rollout) are cached per calling thread, on
 * the device that thread was on when they were allocated, and reused by the next call; after cclqr_set_device to another GPU the next
 * call releases them there and starts a cache on the new device.  This returns them to the driver (they are re-allocated on demand);
 * a thread that exits without calling it leaves its blocks allocated until the process ends.  Every entry point that takes a
 * cclqr_mech returns CCLQR_EINVAL when the calling thread is not on the device the handle was created on. */
int cclqr_release_workspaces(void);

/* The recursion of lqr_tracking.jl:73-122 on per-knot linear models the caller brings: A [N-1][mx][mx], Bu [N-1][mx][mu], Bl [N-1][mx][ml],
 * G [N-1][ml][mx] (knot k = 1 .. N-1 at index k-1, as cclqr_riccati_tracking linearises them itself); ml = 0 with the projected pairs of
 * cclqr_linearize_projected gives the TrackingLQR of a closed-loop mechanism.  K [N-1][mu][mx], kbreak [1]. Host pointers. */
int cclqr_riccati_tv(int32_t mx, int32_t mu, int32_t ml, const double *A, const double *Bu, const double *Bl, const double *G,
                     const double *Q, const double *R, int32_t N, double tol, double *K, int32_t *kbreak);

/* dlqr(mechanism, xd, vd, qd, ωd, Fτd, eqcids, Q, R, N) -- lqr_tracking.jl:73-122: re-linearises at every knot (:88).
 * zd [N][nb][13], Fd [N][mu], K [N-1][mu][mx]. Host pointers. */
int cclqr_riccati_tracking(const cclqr_mech *m, int32_t mu, const int32_t *ctrl_joint, const double *zd, const double *Fd, const double *Q,
                           const double *R, int32_t N, double tol, double *K, int32_t *kbreak);

/* simulate!(mechanism, steps, controller; record) for n_inst independent instances -- the rollout loop of every
 * example's last line (e.g. examples/lqr_cartpole.jl:44) with control_lqr! (lqr.jl:89-139) / control_trackinglqr!
 * (lqr_tracking.jl:46-71) fused in.  Step indices run k0 .. k0+steps-1 (1-based like the reference).
 * HOST pointers; traj may be NULL (record=false); noise [n_inst][steps] standard-normal samples or NULL;
 * status[n_inst] = max Newton iterations used, negative if a step failed: -CCLQR_NEWTON_MAXIT when a solve hit the iteration cap, a smaller magnitude when a
 * step ended on a non-finite residual (the instance is LOST: frozen at its last pose, at rest, for the rest of the launch). */
int cclqr_rollout(const cclqr_mech *m, const cclqr_ctrl *c, int64_t n_inst, int32_t steps, int32_t k0, const double *z0,
                  const double *noise, double *traj, double *zT, int32_t *status);

/* Same with DEVICE pointers and an explicit hipStream_t (passed as void*): nothing is copied, the launch is
 * asynchronous.  lam [n_inst][5*ne] carries the multipliers (Newton warm start) between calls; it is read when
 * k0 > 1 and always written; pass NULL to start from zero and discard. */
int cclqr_rollout_dev(const cclqr_mech *m, const cclqr_ctrl *c, int64_t n_inst, int32_t steps, int32_t k0, const double *z0_dev,
                      double *lam_dev, const double *noise_dev, int64_t noise_stride, double *traj_dev, double *zT_dev,
                      int32_t *status_dev, void *stream);

/* Per-launch options of cclqr_rollout_ex (NULL or all zero = defaults).
 *   first_instance  global index of instance 0 of this launch: the Philox noise stream (noise_philox) and the per-instance
 *                   controller table (n_ctrl > 1) of an instance are keyed by its GLOBAL index, so rank r of a sharded batch
 *                   passes its shard's first index and reproduces the unsharded batch;
 *   pid_state_dev   DEVICE buffer [n_inst][nb][2] ([n_inst][joints][2] for a closed-loop mechanism, which has more joints than bodies;
 *                   integrated error, last error per joint, opaque order) that carries the PID
 *                   integrators of pid.jl:10-11 between launches: read when k0 > 1, always written; only used by controllers with
 *                   npid > 0.  NULL: they live for one launch;
 *   pid_state_len   doubles in that buffer (checked against n_inst * nb * 2, closed loops: n_inst * joints * 2);
 *   noise_ws_dev    DEVICE workspace of n_inst * steps doubles for the launch's Philox samples (noise_philox controllers without an
 *                   injected noise array).  NULL: the controller handle's own workspace is used -- ONE buffer per handle, so two
 *                   launches that share a controller on different streams or host threads must each bring their own here, and
 *                   the handle's buffer cannot grow while `stream` is being captured into a hipGraph (CCLQR_EINVAL: size it first
 *                   with cclqr_ctrl_reserve_noise).  Launches of at most CCLQR_PHILOX_INKERNEL_STEPS steps (the step-per-launch / hipGraph
 *                   form of BASELINE configs[4]) on forests of chains generate their samples INSIDE the rollout kernel and touch no
 *                   workspace at all;
 *   noise_ws_len    doubles in that workspace (checked against n_inst * steps);
 *   newton_mode     0 = the reference's stopping rule, ||f|| < eps AND ||step taken|| < eps (the PARITY mode).
 *                   1 = measured-error mode: a Newton solve ALSO stops as soon as ||f|| < newton_eps_alone, whatever the step size.
 *                   Saves the iterations the exact rule spends halving steps on round-off noise; the state deviation from mode 0
 *                   is reported, not promised.  Forests of chains and branching trees under the plain LQR / TrackingLQR
 *                   law (CCLQR_EUNSUPPORTED with the friction / noise / PID laws there), closed-loop mechanisms under every law;
 *   flags           CCLQR_ROLLOUT_NO_ALLOC: the call must neither allocate nor synchronise -- what a caller who is capturing ANY stream of the
 *                   device into a hipGraph needs (a hipMalloc / hipDeviceSynchronize during a global-mode capture invalidates it, whichever
 *                   stream it is issued on).  A launch that would have to grow the handle's Philox workspace is then refused with
 *                   CCLQR_EINVAL (the message names cclqr_ctrl_reserve_noise) before anything is touched, whether or not `stream` itself is
 *                   capturing; without the flag the library still refuses on a capturing `stream` (it can see that one) and grows otherwise;
 *                   CCLQR_ROLLOUT_PACK_WAVEFRONTS: forests of chains and branching trees: 64 / lanes-per-instance instances in every wavefront whatever the batch
 *                   size.  Without it a batch too small to give every SIMD a wavefront that way is spread over more wavefronts (the spare lane
 *                   groups take part in their neighbours' line searches: up to 20 % less time per step at a few hundred instances; a persistent
 *                   launch, steps >= 8, spreads over the whole device, a shorter one over a quarter of it) -- bitwise the same results.  Set it
 *                   when MANY launches share the device at once (more than four concurrent step-per-launch chains, several processes on one
 *                   GPU): spread launches then queue behind one another;
 *                   CCLQR_ROLLOUT_CARRY_STATUS: `status` is read as well as written -- it carries an instance's status ACROSS launches, for callers who
 *                   step a batch one launch (or a few steps) at a time (k0 continuation: the `controlfunction` loops, MPC, a hipGraph of step
 *                   launches).  Zero it before the first launch.  An instance that comes in lost (-CCLQR_NEWTON_MAXIT < status < 0: an earlier step ended on
 *                   a non-finite residual) is not stepped: it stays frozen at its last pose, at rest, as it would inside ONE launch over the whole
 *                   horizon, and keeps its status; for the others this launch's result is merged in (largest Newton count so far, negative once any
 *                   step failed).  A lost instance that had ALSO failed to converge earlier reports -(CCLQR_NEWTON_MAXIT - 1), so that "lost" stays
 *                   readable from the number.  Without the flag every launch reports on its own steps only and forgets what came before;
 *   newton_eps_alone  threshold of mode 1 (<= 0: 1e-10, the rule's own eps: stop on the residual alone). */
#define CCLQR_NEWTON_MAXIT 100        /* newton!'s iteration cap */
#define CCLQR_ROLLOUT_NO_ALLOC 1
#define CCLQR_ROLLOUT_PACK_WAVEFRONTS 2
#define CCLQR_ROLLOUT_CARRY_STATUS 4
#define CCLQR_PHILOX_INKERNEL_STEPS 8
typedef struct {
    int64_t first_instance;
    double *pid_state_dev;
    int64_t pid_state_len;
    double *noise_ws_dev;
    int64_t noise_ws_len;
    int32_t newton_mode;
    int32_t flags;
    double newton_eps_alone;
} cclqr_rollout_opts;

/* cclqr_rollout_dev with explicit options (opts may be NULL = cclqr_rollout_dev). */
int cclqr_rollout_ex(const cclqr_mech *m, const cclqr_ctrl *c, int64_t n_inst, int32_t steps, int32_t k0, const double *z0_dev,
                     double *lam_dev, const double *noise_dev, int64_t noise_stride, double *traj_dev, double *zT_dev,
                     int32_t *status_dev, const cclqr_rollout_opts *opts, void *stream);

/* cclqr_rollout (HOST pointers) with options: first_instance and newton_mode apply, the device-buffer fields must be NULL. */
int cclqr_rollout_host_ex(const cclqr_mech *m, const cclqr_ctrl *c, int64_t n_inst, int32_t steps, int32_t k0, const double *z0,
                          const double *noise, double *traj, double *zT, int32_t *status, const cclqr_rollout_opts *opts);

/* Sizes the controller handle's Philox noise workspace for launches of up to n_inst * steps samples (the `randn()` stream of
 * examples/trackingLQR_triple_cartpole.jl:98,125 as generated by noise_philox).  Synchronises the device when it has to re-allocate.
 * Call it before capturing step-per-launch rollouts of a noise_philox controller into a hipGraph (BASELINE configs[4]), from a thread that is on
 * the controller's device (CCLQR_EINVAL otherwise), and never while ANY stream of that device is being captured in the global capture mode (the
 * re-allocation is a device synchronisation plus an allocation, which such a capture does not survive; a launch on the capturing stream itself
 * that would have to grow the workspace is refused with CCLQR_EINVAL before anything is touched, and so is ANY launch that carries
 * CCLQR_ROLLOUT_NO_ALLOC in cclqr_rollout_opts.flags: cclqr_rollout_ex). */
int cclqr_ctrl_reserve_noise(cclqr_ctrl *c, int64_t n_inst, int32_t steps);

/* The `controlfunction` hook of the reference's controllers (src/control/lqr.jl:14, :56; lqr_tracking.jl:19; pid.jl:16 -- a Julia closure
 * `controlfunction(mechanism, controller, k)` that ends in setForce!(mechanism, eqc, u), e.g. examples/trackingLQR_triple_cartpole.jl:93-117)
 * with the closure on the HOST and the batch on the device: the caller steps the rollout one launch per step (k0 continuation of
 * cclqr_rollout_dev), reads the states, lets its closure compute every instance's joint inputs and hands them over here; the next launch applies
 * them as feed-forward inputs.  Fd: [n_ctrl][nsp][mu] doubles exactly as given to cclqr_ctrl_create (len is checked against that), a HOST
 * pointer (on_device = 0: copied synchronously on the null stream -- the caller must have synchronised every NON-BLOCKING stream on which a launch
 * that reads this controller may still be running, e.g. by reading that launch's states back, as the closure loop does) or a DEVICE pointer
 * (on_device = 1: copied on `stream`, ordered with the launches on it).
 * The controller must have been created with a feed-forward table (Fd != NULL). */
int cclqr_ctrl_set_feedforward(cclqr_ctrl *c, const double *Fd, int64_t len, int32_t on_device, void *stream);

/* kernel launch geometry chosen for a mechanism (for roofline bookkeeping in bench.py; no counterpart in the reference's simulate!,
 * examples/lqr_cartpole.jl:44): lanes per instance and LDS bytes per workgroup; links the chain kernel's LDS image is laid out for
 * (names the instantiation rollout_chain_kernel<lanes, links, law> for forests of chains and rollout_treereg_kernel<lanes, links, law, relax>
 * for branching trees; 0 for closed-loop mechanisms, whose one kernel takes its layout at run time) */
int cclqr_rollout_geometry(const cclqr_mech *m, int32_t *lanes_per_instance, int32_t *lds_bytes_per_workgroup);
int cclqr_rollout_layout_links(const cclqr_mech *m, int32_t *links);
/* lanes that work for ONE link and links per sub-lane group of the chain kernel's instantiation (rollout_chain_kernel<lanes, links, law, relax, lanes_per_link,
 * links_per_group>): a forest of chains that leaves lanes of its lane group idle (1-2 links in 8 lanes: 3 lanes per link; 3-4 in 8, 5-8 in 16, 9-16 in 32: 2) deals
 * the constraint rows of an evaluation to them at unchanged occupancy; 1 and the lane count for longer chains, branching trees and closed loops.  Bookkeeping
 * for bench.py like the two above (no counterpart in the reference's simulate!, examples/lqr_cartpole.jl:44). */
int cclqr_rollout_lanes_per_link(const cclqr_mech *m, int32_t *lanes_per_link, int32_t *links_per_group);
/* instances a launch of n_inst instances x steps steps with these cclqr_rollout_opts.flags puts into one wavefront (fewer than 64 / lanes-per-instance
 * when the batch is spread, see CCLQR_ROLLOUT_PACK_WAVEFRONTS; closed loops: always 1).  Bookkeeping. */
int cclqr_rollout_instances_per_wavefront(const cclqr_mech *m, int64_t n_inst, int32_t steps, int32_t flags, int32_t *instances);

#ifdef __cplusplus
}
#endif
#endif
