// rollout_loop.hip -- simulate! with the LQR feedback law for mechanisms with closed kinematic loops (examples/lqr_deltabot.jl:25-53),
// one instance per wavefront, persistent over the horizon.  Phases and the singular dense solve: cclqr_loop.h.
//
// Replaces: ConstrainedDynamics.simulate!/newton! on a Mechanism whose constraint graph has cycles, driven by control_lqr!
// (src/control/lqr.jl:89-139).  Not a throughput kernel: loop mechanisms are small (the reference's only one has five bodies) and
// the solve is a 5 nj-step pivoted elimination; the chain / tree kernels keep every mechanism without loops.
#include "cclqr_loop.h"
#include "cclqr_lin_loop.h"
#include "cclqr_internal.h"
#include "cclqr_newton.h"

namespace cclqr {

// residual (+ Jacobians) at the point s_off with multipliers lambda - alpha dlambda already folded into C - alpha CD
template <bool JAC>
__device__ __forceinline__ double loop_eval(int t, const Lay& Y, double* L, const LaneRegs& r, const MechDev* M, int s_off, double alpha PROF_ARG) {
    double part = ph_body_eval<JAC>(t, M->nb, Y, L, r, M->dt, s_off, alpha);
    __syncthreads();
    STAMP(PF_EVAL_BODY);
    part += lp_joint_eval<JAC>(t, Y, L, r, M);
    __syncthreads();
    STAMP(PF_EVAL_JOINT);
    PCOUNT(PF_EVALS);
    return sqrt(group_sum<64>(part));
}

// maximum of one 32-bit key per lane over the wavefront, through the DPP crossbar (no LDS round trips): running maximum along
// each 16-lane row, row 0 -> 1 and 2 -> 3, rows {0,1} -> {2,3}; lane 63 holds the result, read back as a wavefront-uniform value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned key_dpp_max(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_key(unsigned v) {
    v = key_dpp_max<0x111, 0xf>(v);      // row_shr:1
    v = key_dpp_max<0x112, 0xf>(v);      // row_shr:2
    v = key_dpp_max<0x114, 0xf>(v);      // row_shr:4
    v = key_dpp_max<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of each row holds the row's maximum
    v = key_dpp_max<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = key_dpp_max<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wavefront's maximum
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// the wavefront's largest non-negative double
__device__ __forceinline__ double wave_max_d(double v) {
    for (int o = 1; o < 64; o <<= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double lane_read_d(double v, int lane) {       // lane: the same in every lane
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// one step of the elimination: column 8 kb + U.  Returns whether the column found a pivot (the same in every lane).
template <int NCB, int U>
__device__ __forceinline__ bool loop_solve_step(LoopRowR<NCB>& R, int t, int kb, int mr, double tol) {
    const int col = 8 * kb + U;
    const double own = lpr_entry<NCB, U>(R, kb);
    const int prow = lp_key_row(wave_max_key(lpr_key(R, t, mr, own)));            // uniform: every lane holds the same pivot row
    const double piv = lane_read_d(own, prow);          // (no row in play: row 63's entry, 0 -- lanes beyond 5 nj hold zero rows)
    const bool found = fabs(piv) > tol;
    if (found) {
        const double ip = 1.0 / piv;
        const double f = lpr_step(R, t, col, prow, mr, own, ip);
        R.rhs -= f * lane_read_d(R.rhs, prow);
#pragma unroll
        for (int B = 0; B < NCB; B++)
            if (B >= kb) {                                     // (uniform) blocks before the current one hold residue nobody reads again
                double p[8];                                   // eight lane reads ahead of the multiply-adds that use them (a scalar register
#pragma unroll                                                 // written by a lane read needs two idle cycles before the vector ALU may read it)
                for (int u = 0; u < 8; u++) p[u] = lane_read_d(R.a[8 * B + u], prow);
#pragma unroll
                for (int u = 0; u < 8; u++) R.a[8 * B + u] -= f * p[u];
            }
    }
    return found;
}
// S dl = r by Gauss-Jordan elimination over the columns that have a pivot, rows in registers (cclqr_loop.h); dl of the free (redundant) directions is 0
template <int NCB>
__device__ __forceinline__ void loop_solve(int t, const Lay& Y, double* L, const MechDev* M PROF_ARG) {
    const int mr = 5 * M->nj;
    LoopRowR<NCB> R;
    const double tol = LOOP_RANK_TOL * wave_max_d(lpr_assemble(R, t, Y, L, M));
    if (t < mr) L[Y.DL + t] = 0.0;
    STAMP(PF_SCHUR_S);
#pragma unroll 1
    for (int kb = 0; kb < NCB; kb++) {            // (columns beyond 5 nj are zero padding: no pivot, skipped)
        loop_solve_step<NCB, 0>(R, t, kb, mr, tol); loop_solve_step<NCB, 1>(R, t, kb, mr, tol);
        loop_solve_step<NCB, 2>(R, t, kb, mr, tol); loop_solve_step<NCB, 3>(R, t, kb, mr, tol);
        loop_solve_step<NCB, 4>(R, t, kb, mr, tol); loop_solve_step<NCB, 5>(R, t, kb, mr, tol);
        loop_solve_step<NCB, 6>(R, t, kb, mr, tol); loop_solve_step<NCB, 7>(R, t, kb, mr, tol);
    }
    STAMP(PF_TRI_FWD);
    lpr_solution(R, t, mr, Y, L);
    __syncthreads();
    STAMP(PF_TRI_BWD);
}

// newton! on the instance held in LDS (S / LAM = guess in, solution out), the rules of the tree kernels: returns the
// iterations used; *failed = a non-finite residual (the instance has left the integrator's domain) or no convergence in NEWTON_MAXIT
// eps_alone > 0: the measured-error mode of cclqr_rollout_opts.newton_mode = 1 (a solve also stops on ||f|| < eps_alone, whatever the step)
template <int NCB>
__device__ __forceinline__ int loop_newton(int t, const Lay& Y, double* L, LaneRegs& r, const MechDev* M, bool* failed, double eps_alone PROF_ARG) {
    bool done = false, fail = false;
    int its = 0;
    double normf0 = loop_eval<true>(t, Y, L, r, M, Y.S, 0.0 PROF_PASS);
    for (int iter = 1; iter <= NEWTON_MAXIT && !done; iter++) {
        PCOUNT(PF_NEWTON_ITERS);
        loop_solve<NCB>(t, Y, L, M PROF_PASS);
        {   // the link constants (38 doubles per lane) are re-read from the mechanism tables behind the solve, through a pointer the optimiser
            // cannot see through: the solve's row (up to 64 doubles per lane) then takes their registers instead of coming on top of them
            const MechDev* Mq = M;
            asm volatile("" : "+s"(Mq));
            loop_load_link_consts(r, Mq, t);
        }
        lp_body_solve(t, Y, L, M);
        __syncthreads();
        STAMP(PF_BODY_SOLVE);
        double alpha = 1.0, normf1 = 0.0;
        const double nd = sqrt(group_sum<64>(lp_trial(t, Y, L, M, alpha)));
        __syncthreads();
        STAMP(PF_TRIAL);
        for (int ls = 0; ls <= LINE_MAXIT; ls++) {     // halve while ||f|| grows; level LINE_MAXIT is taken as it is
            normf1 = loop_eval<false>(t, Y, L, r, M, Y.ST, alpha PROF_PASS);
            if (!(normf1 > normf0) || ls == LINE_MAXIT) break;
            alpha *= 0.5;
            lp_trial(t, Y, L, M, alpha);
            __syncthreads();
        }
        lp_accept(t, Y, L, M, alpha);
        __syncthreads();
        STAMP(PF_ACCEPT);
        its = iter;
        if (normf1 < NEWTON_EPS && alpha * nd < NEWTON_EPS) done = true;
        if (normf1 < eps_alone) done = true;
        if (!(normf1 < 1e300)) { done = true; fail = true; }
        if (!done) normf0 = loop_eval<true>(t, Y, L, r, M, Y.S, 0.0 PROF_PASS);     // Jacobians at the accepted point
    }
    *failed = fail || !done;
    return its;
}

// relax: 0 = the reference's stopping rule, 1 = newton_mode 1 (a.eps_alone).  NCB: the dense system's rows are held in registers 8 NCB columns wide
// Two wavefronts per SIMD (<= 256 registers): with the dense system assembled in registers the deltabot's LDS image is 12.2 KB (24.8 KB with the
// system staged in LDS), so LDS admits thirteen workgroups per CU where the 307 registers the kernel would like admit four; at 256 the compiler
// parks ~50 doubles of the evaluation / assembly phases in scratch (364 bytes per lane, none inside the elimination) and the rollout is faster:
// 9.8 M instance-steps/s at 8192 instances with one wavefront per SIMD, 11.6 M with two and the staged system (six per CU), 14.6 M with two and
// the register assembly (eight per CU); three per SIMD (168 registers, 992 bytes of scratch) falls back to 12.4 M.
template <int NCB>
__global__ __launch_bounds__(64, 2) void rollout_loop_kernel(RolloutArgs a, int relax) {
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    const int64_t inst = blockIdx.x;
    const MechDev* M = a.M;
    const CtrlDev* C = a.C;
    const int nb = M->nb, nj = M->nj, nz = 13 * nb;
    const Lay Y = make_loop_layout(nb, nj);
    double* L = lds;
    LaneRegs r;
    loop_load_consts(r, M, t);
    for (int e = t; e < Y.total; e += 64) L[e] = 0.0;
    __syncthreads();
    for (int e = t; e < nz; e += 64) L[Y.Z + e] = a.z0[inst * nz + e];
    if (a.lam && a.k0 > 1)
        for (int e = t; e < 5 * nj; e += 64) L[Y.LAM + e] = a.lam[inst * 5 * nj + e];
    if (a.pid_state && a.k0 > 1 && t < nj) { r.pid_int = a.pid_state[(inst * nj + t) * 2]; r.pid_last = a.pid_state[(inst * nj + t) * 2 + 1]; }
    __syncthreads();

#ifdef CCLQR_PROFILE
    Prof prof;
    prof.start();
#endif
    int worst = 0;
    bool bad = false, dead = false;
    if (a.carry && a.status) {                    // CCLQR_ROLLOUT_CARRY_STATUS (rollout_chain.hip)
        const int carried = a.status[inst];
        worst = carried < 0 ? -carried : carried;
        bad = carried < 0;
        dead = carried < 0 && carried > -NEWTON_MAXIT;
    }
    const long long ginst = a.inst0 + inst;
    for (int kk = 0; kk < a.steps; kk++) {
        const int k = a.k0 + kk;
        if (a.traj)
            for (int e = t; e < nz; e += 64) a.traj[((size_t)inst * a.steps + kk) * nz + e] = L[Y.Z + e];
        STAMP(PF_IO);
        // ---------------- feedback law (lqr.jl:89-139)
        const bool gate = (C->N <= 0) || (k < C->N);
        const int ksp = (C->nsp > 1) ? ((k - 1 < C->nsp) ? k - 1 : C->nsp - 1) : 0;
        const int kidx = (C->N <= 0) ? 0 : ((k - 1 < C->nK) ? k - 1 : C->nK - 1);
        if (t < nj) L[Y.UJ + t] = 0.0;
        if (gate) ph_control_error(t, nb, Y, L, r, C, C->zd + ginst * C->zd_stride + (size_t)ksp * nz);
        __syncthreads();
        // joint friction in the loop's own bookkeeping (ph_control_error's friction term assumes link t = body t = joint t and is overwritten here)
        if (t < nj) L[Y.UJ + t] = (gate && C->has_fric) ? lp_friction(t, Y, L, r, M, C->fric[t]) : 0.0;
        if (C->has_pid) lp_pid(t, Y, L, r, M, C, k == 1);            // control_pid! runs every step (pid.jl:69-88), whatever the LQR horizon
        __syncthreads();
        if (gate) {
            for (int i = 0; i < C->mu; i++) {
                double part = 0.0;
                if (C->K) part = ph_gain_partial(t, 64, nb, Y, L, C->K + ginst * C->K_stride + ((size_t)kidx * C->mu + i) * 12 * nb);
                const double s = group_sum<64>(part);
                if (t == 0) {
                    double u = (C->Fd ? C->Fd[ginst * C->Fd_stride + (size_t)ksp * C->mu + i] : 0.0) - s;
                    if (C->noise_scale != 0.0 && a.noise) u += C->noise_scale * a.noise[(size_t)inst * a.noise_stride + (k - 1)];     // injected, or this launch's Philox samples
                    L[Y.UJ + C->cj[i]] += u;
                }
                __syncthreads();
            }
        }
        STAMP(PF_CONTROL);
        // ---------------- per-step invariants
        lp_forces(t, Y, L, r, M);
        lp_knot_jac(t, Y, L, r, M);
        __syncthreads();
        lp_force_map(t, Y, L, M);
        __syncthreads();
        STAMP(PF_FORCES);
        PCOUNT(PF_STEPS);
        // ---------------- newton! (tolerances and line search: SURVEY 8a-bis)
        bool failed = false;
        const int its = dead ? 0 : loop_newton<NCB>(t, Y, L, r, M, &failed, relax ? a.eps_alone : 0.0 PROF_PASS);
        const bool done = true;
        if (!dead) {
            const bool conv = done && !failed;
            if (!conv) bad = true;
            if (its > worst) worst = its;
            if (!conv && its < NEWTON_MAXIT) {       // non-finite residual: freeze the instance at its last pose, at rest
                dead = true;
                for (int e = t; e < nb; e += 64)
                    for (int i = 0; i < 6; i++) L[Y.Z + 13 * e + 7 + i] = 0.0;
            } else {
                ph_update(t, nb, Y, L);       // XQ = next pose of the accepted solution (the last evaluation was at that point)
            }
        }
        __syncthreads();
    }
    for (int e = t; e < nz; e += 64) a.zT[inst * nz + e] = L[Y.Z + e];
    if (a.lam) for (int e = t; e < 5 * nj; e += 64) a.lam[inst * 5 * nj + e] = L[Y.LAM + e];
    if (a.status && t == 0) a.status[inst] = bad ? -((a.carry && dead && worst >= NEWTON_MAXIT) ? NEWTON_MAXIT - 1 : worst) : worst;
    if (a.pid_state && t < nj) { a.pid_state[(inst * nj + t) * 2] = r.pid_int; a.pid_state[(inst * nj + t) * 2 + 1] = r.pid_last; }
#ifdef CCLQR_PROFILE
    prof.stamp(PF_IO);
    prof.flush();
#endif
}

#ifdef CCLQR_PROFILE
extern "C" int cclqr_prof_read_loop(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * PF_N);
    if (e == hipSuccess && reset) { unsigned long long z[PF_N] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return e == hipSuccess ? PF_N : -1;
}
#endif

// linearsystem(mechanism, xd, vd, qd, ωd, Fτd, bodyids, eqcids) (lqr.jl:63, lqr_tracking.jl:88) for a closed-loop mechanism: one knot per
// wavefront -- one converged Newton step at the setpoint, then A, Bu, Bl, G with the multipliers exogenous (cclqr_lin_loop.h).  a.cj = joint
// indices of the inputs; zd in the caller's body order (the closed-loop tables keep it).
template <int NCB>
__global__ __launch_bounds__(64) void linearize_loop_kernel(LinArgs a) {
    extern __shared__ double lds[];
    const int t = threadIdx.x, knot = blockIdx.x;
    const MechDev* M = a.M;
    const int nb = M->nb, nj = M->nj, nz = 13 * nb, mx = 12 * nb, ml = 5 * nj, mu = a.mu;
    const Lay Y = make_loop_layout(nb, nj);
    const int JB = Y.total;
    double* L = lds;
    LinOut O;
    O.A = a.A + (size_t)knot * mx * mx; O.Bu = a.Bu + (size_t)knot * mx * mu; O.Bl = a.Bl + (size_t)knot * mx * ml;
    O.G = a.G + (size_t)knot * ml * mx; O.mx = mx; O.mu = mu; O.ml = ml;
    for (int e = t; e < mx * mx; e += 64) O.A[e] = 0.0;
    for (int e = t; e < mx * mu; e += 64) O.Bu[e] = 0.0;
    for (int e = t; e < mx * ml; e += 64) O.Bl[e] = 0.0;
    for (int e = t; e < ml * mx; e += 64) O.G[e] = 0.0;
    LaneRegs r;
    loop_load_consts(r, M, t);
    for (int e = t; e < Y.total + LJB * nj; e += 64) L[e] = 0.0;
    __syncthreads();
    for (int e = t; e < nz; e += 64) L[Y.Z + e] = a.zd[(size_t)knot * nz + e];
    __syncthreads();
    if (t == 0)
        for (int i = 0; i < mu; i++) L[Y.UJ + a.cj[i]] += a.Fd ? a.Fd[(size_t)knot * mu + i] : 0.0;
    __syncthreads();
    lp_forces(t, Y, L, r, M);
    lp_knot_jac(t, Y, L, r, M);
    __syncthreads();
    lp_force_map(t, Y, L, M);
    __syncthreads();
#ifdef CCLQR_PROFILE
    Prof prof;
    prof.start();
#endif
    bool failed = false;
    const int its = loop_newton<NCB>(t, Y, L, r, M, &failed, 0.0 PROF_PASS);
    loop_eval<true>(t, Y, L, r, M, Y.S, 0.0 PROF_PASS);      // D_R^-1, N D_R^-1 and the next pose at the converged solution
    __syncthreads();
    lp_lin_joint(t, Y, JB, L, r, M);
    __syncthreads();
    __threadfence_block();
    lp_lin_rows_A(t, Y, JB, L, r, M, O);
    lp_lin_rows_B(t, Y, L, r, M, a.cj, O);
    if (t == 0 && a.status) a.status[knot] = failed ? -its : its;
}

// The projected pair of the recursion (lqr.jl:151: D = Bu - Bl/(G Bl) G Bu, and A' = A - Bl/(G Bl) G A likewise) from a linear model
// whose G Bl may be SINGULAR (closed loops: redundant constraint rows): (G Bl) [X | Y] = G [A | Bu] is consistent, so it is solved by
// Gauss-Jordan elimination with complete pivoting that stops at the numerical rank -- free components 0, a basic solution; Bl X is the same
// for every solution because the null space of G Bl is that of the force map G_k'.  One workgroup per knot, the system in LDS.
// res[knot] = largest |entry| left in the rows that found no pivot, relative to the largest entry of the right-hand side (consistency).
#define PROJ_THREADS 256
__global__ __launch_bounds__(PROJ_THREADS) void project_model_kernel(int mx, int mu, int ml, const double* A, const double* Bu, const double* Bl, const double* G,
                                                                     double* Ap, double* D, double* res, int* rank_out) {
    extern __shared__ double ps[];
    __shared__ double red_v[PROJ_THREADS];
    __shared__ int red_i[PROJ_THREADS];
    __shared__ int s_pi, s_pj, s_rank;
    __shared__ double s_first;
    const int tid = threadIdx.x, knot = blockIdx.x, na = mx + mu, ld = ml + na;
    A += (size_t)knot * mx * mx; Bu += (size_t)knot * mx * mu; Bl += (size_t)knot * mx * ml; G += (size_t)knot * ml * mx;
    Ap += (size_t)knot * mx * mx; D += (size_t)knot * mx * mu;
    double* S = ps;                         // [ml][ld]: G Bl | G A | G Bu
    double* f = S + (size_t)ml * ld;        // [ml] elimination factors
    int* rowp = (int*)(f + ml);             // [ml] pivot column of a row (-1: none)
    int* colu = rowp + ml;                  // [ml] 1: the column has been a pivot column
    for (int e = tid; e < ml * ld; e += PROJ_THREADS) {
        const int i = e / ld, j = e - i * ld;
        double acc = 0.0;
        for (int k = 0; k < mx; k++) {
            const double b = j < ml ? Bl[(size_t)k * ml + j] : (j < ml + mx ? A[(size_t)k * mx + (j - ml)] : Bu[(size_t)k * mu + (j - ml - mx)]);
            acc += G[(size_t)i * mx + k] * b;
        }
        S[e] = acc;
    }
    for (int e = tid; e < ml; e += PROJ_THREADS) { rowp[e] = -1; colu[e] = 0; }
    if (tid == 0) s_rank = 0;
    __syncthreads();
    double rhs_max = 0.0;
    for (int e = tid; e < ml * na; e += PROJ_THREADS) { const int i = e / na, j = e - i * na; rhs_max = fmax(rhs_max, fabs(S[i * ld + ml + j])); }
    red_v[tid] = rhs_max;
    __syncthreads();
    for (int o = PROJ_THREADS / 2; o > 0; o >>= 1) { if (tid < o) red_v[tid] = fmax(red_v[tid], red_v[tid + o]); __syncthreads(); }
    const double rhs_scale = red_v[0];
    __syncthreads();
    for (int k = 0; k < ml; k++) {
        double best = -1.0; int bi = 0;
        for (int e = tid; e < ml * ml; e += PROJ_THREADS) {
            const int i = e / ml, j = e - i * ml;
            if (rowp[i] < 0 && !colu[j]) { const double v = fabs(S[i * ld + j]); if (v > best) { best = v; bi = e; } }
        }
        red_v[tid] = best; red_i[tid] = bi;
        __syncthreads();
        for (int o = PROJ_THREADS / 2; o > 0; o >>= 1) {
            if (tid < o && (red_v[tid + o] > red_v[tid] || (red_v[tid + o] == red_v[tid] && red_i[tid + o] < red_i[tid]))) { red_v[tid] = red_v[tid + o]; red_i[tid] = red_i[tid + o]; }
            __syncthreads();
        }
        if (tid == 0) {
            if (k == 0) s_first = red_v[0];
            if (red_v[0] > LOOP_RANK_TOL * s_first && red_v[0] > 0.0) { s_pi = red_i[0] / ml; s_pj = red_i[0] - (red_i[0] / ml) * ml; s_rank = k + 1; }
            else s_pi = -1;
        }
        __syncthreads();
        if (s_pi < 0) break;
        const int pi = s_pi, pj = s_pj;
        const double pinv = 1.0 / S[pi * ld + pj];
        for (int i = tid; i < ml; i += PROJ_THREADS) f[i] = (i == pi) ? 0.0 : S[i * ld + pj] * pinv;
        if (tid == 0) { rowp[pi] = pj; colu[pj] = 1; }
        __syncthreads();
        for (int e = tid; e < ml * ld; e += PROJ_THREADS) {
            const int i = e / ld, j = e - i * ld;
            if (i != pi) S[e] = (j == pj) ? 0.0 : S[e] - f[i] * S[pi * ld + j];
        }
        __syncthreads();
    }
    // consistency: what is left of the right-hand side in the rows without a pivot
    double left = 0.0;
    for (int e = tid; e < ml * na; e += PROJ_THREADS) { const int i = e / na, j = e - i * na; if (rowp[i] < 0) left = fmax(left, fabs(S[i * ld + ml + j])); }
    red_v[tid] = left;
    __syncthreads();
    for (int o = PROJ_THREADS / 2; o > 0; o >>= 1) { if (tid < o) red_v[tid] = fmax(red_v[tid], red_v[tid + o]); __syncthreads(); }
    if (tid == 0) { if (res) res[knot] = rhs_scale > 0.0 ? red_v[0] / rhs_scale : 0.0; if (rank_out) rank_out[knot] = s_rank; }
    // X[c][:] = (row with pivot column c)[rhs] / pivot, 0 for free columns;  [A' | D] = [A | Bu] - Bl X
    for (int e = tid; e < mx * na; e += PROJ_THREADS) {
        const int i = e / na, j = e - i * na;
        double acc = j < mx ? A[(size_t)i * mx + j] : Bu[(size_t)i * mu + (j - mx)];
        for (int rI = 0; rI < ml; rI++) {
            const int c = rowp[rI];
            if (c >= 0) acc -= Bl[(size_t)i * ml + c] * (S[rI * ld + ml + j] / S[rI * ld + c]);
        }
        if (j < mx) Ap[(size_t)i * mx + j] = acc; else D[(size_t)i * mu + (j - mx)] = acc;
    }
}

size_t linearize_loop_lds_bytes(int nb, int nj) { return (size_t)(make_loop_layout(nb, nj).total + LJB * nj) * sizeof(double); }
template <int NCB>
static hipError_t launch_linearize_loop_t(const LinArgs& a, size_t lds, hipStream_t stream) {
    hipError_t e = set_max_dynamic_lds_once((const void*)linearize_loop_kernel<NCB>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(linearize_loop_kernel<NCB>, dim3(a.nk), dim3(64), lds, stream, a);
    return hipGetLastError();
}
hipError_t launch_linearize_loop(const LinArgs& a, int nb, int nj, hipStream_t stream) {
    if (a.nk <= 0) return hipSuccess;
    const size_t lds = linearize_loop_lds_bytes(nb, nj);
    switch (loop_col_blocks(nj)) {
        case 1: return launch_linearize_loop_t<1>(a, lds, stream);
        case 2: return launch_linearize_loop_t<2>(a, lds, stream);
        case 3: return launch_linearize_loop_t<3>(a, lds, stream);
        case 4: return launch_linearize_loop_t<4>(a, lds, stream);
        case 5: return launch_linearize_loop_t<5>(a, lds, stream);
        case 6: return launch_linearize_loop_t<6>(a, lds, stream);
        case 7: return launch_linearize_loop_t<7>(a, lds, stream);
        case 8: return launch_linearize_loop_t<8>(a, lds, stream);
        default: return hipErrorInvalidValue;
    }
}
// dynamic LDS of project_model_kernel; project_model_fits adds the kernel's static LDS (red_v, red_i, four scalars) before comparing with a CU's 160 KB
size_t project_model_lds_bytes(int mx, int mu, int ml) { return ((size_t)ml * (ml + mx + mu) + ml) * sizeof(double) + 2 * (size_t)ml * sizeof(int) + 16; }
bool project_model_fits(int mx, int mu, int ml) {
    const size_t stat = PROJ_THREADS * (sizeof(double) + sizeof(int)) + 3 * sizeof(int) + sizeof(double) + 64;      // (+ alignment slack)
    return project_model_lds_bytes(mx, mu, ml) + stat <= 160 * 1024;
}
hipError_t launch_project_model(int nk, int mx, int mu, int ml, const double* A, const double* Bu, const double* Bl, const double* G, double* Ap, double* D,
                                double* res, int* rank, hipStream_t stream) {
    if (nk <= 0) return hipSuccess;
    const size_t lds = project_model_lds_bytes(mx, mu, ml);
    hipError_t e = set_max_dynamic_lds_once((const void*)project_model_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(project_model_kernel, dim3(nk), dim3(PROJ_THREADS), lds, stream, mx, mu, ml, A, Bu, Bl, G, Ap, D, res, rank);
    return hipGetLastError();
}

size_t loop_lds_bytes(int nb, int nj) { return (size_t)make_loop_layout(nb, nj).total * sizeof(double); }

template <int NCB>
static hipError_t launch_rollout_loop_t(const RolloutArgs& a, size_t lds, int newton_mode, hipStream_t stream) {
    hipError_t e = set_max_dynamic_lds_once((const void*)rollout_loop_kernel<NCB>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rollout_loop_kernel<NCB>, dim3((unsigned)a.n_inst), dim3(64), lds, stream, a, newton_mode != 0 ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_rollout_loop(const RolloutArgs& a, int nb, int nj, int newton_mode, hipStream_t stream) {
    if (a.n_inst <= 0) return hipSuccess;
    const size_t lds = loop_lds_bytes(nb, nj);
    switch (loop_col_blocks(nj)) {       // 5 nj <= 60 columns (CCLQR_LOOP_MAXJ) in blocks of eight
        case 1: return launch_rollout_loop_t<1>(a, lds, newton_mode, stream);
        case 2: return launch_rollout_loop_t<2>(a, lds, newton_mode, stream);
        case 3: return launch_rollout_loop_t<3>(a, lds, newton_mode, stream);
        case 4: return launch_rollout_loop_t<4>(a, lds, newton_mode, stream);
        case 5: return launch_rollout_loop_t<5>(a, lds, newton_mode, stream);
        case 6: return launch_rollout_loop_t<6>(a, lds, newton_mode, stream);
        case 7: return launch_rollout_loop_t<7>(a, lds, newton_mode, stream);
        case 8: return launch_rollout_loop_t<8>(a, lds, newton_mode, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace cclqr
