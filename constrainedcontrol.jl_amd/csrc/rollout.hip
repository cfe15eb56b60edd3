// rollout.hip -- the LQR-controlled rollout kernel (simulate! + control_lqr!/control_trackinglqr! fused),
// persistent over the whole horizon: state, multipliers and all Jacobian blocks of an instance stay in LDS;
// HBM sees one state load, one gain row per step, one trajectory row per step (if recorded) and the final state.
//
// Replaces: ConstrainedDynamics.simulate!/newton! as driven by the reference (examples/lqr_cartpole.jl:44) with
//           control_lqr! (src/control/lqr.jl:89-139) / control_trackinglqr! (src/control/lqr_tracking.jl:46-71).
#include "cclqr_dev.h"
#include "cclqr_internal.h"
#include "cclqr_newton.h"

namespace cclqr {

// EXTRA: 0 = plain LQR / TrackingLQR feedback, 1 = + joint friction and injected / pre-generated noise, 2 = + PID (as in rollout_chain.hip)
template <int G, bool TREE, int EXTRA>
__global__ __launch_bounds__(64) void rollout_kernel(RolloutArgs a) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x, t = lane % G, grp = lane / G;
    const int64_t inst = (int64_t)blockIdx.x * (64 / G) + grp;
    // the instance exists / a step of it did not converge / it is frozen: bits of ONE vector register, re-tested where needed
    // (a 64-bit lane mask each, kept in scalar registers for the whole launch, otherwise -- see rollout_chain.hip)
    int fl = inst < a.n_inst ? 1 : 0;
#define valid ((fl & 1) != 0)
#define dead ((fl & 2) != 0)
#define bad ((fl & 4) != 0)
#define FRESH_FLAGS asm volatile("" : "+v"(fl))
    const MechDev* M = a.M;
    const CtrlDev* C = a.C;
    const int nb = M->nb;
    const double dt = M->dt;
    const Lay Y = make_layout(nb, TREE ? 2 * M->npairs : 0);
    double* L = lds + grp * Y.total;
    const int nz = 13 * nb;

    LaneRegs r;
    const int NL = newton_level_groups(G, nb), lg = t / nb, tl = t - lg * nb;   // lane groups of the level-parallel line search
    lane_load_consts(r, M, lg < NL ? tl : 0);
    if (EXTRA >= 2 && a.pid_state && a.k0 > 1 && valid && t < nb) { r.pid_int = a.pid_state[(inst * nb + t) * 2]; r.pid_last = a.pid_state[(inst * nb + t) * 2 + 1]; }
#ifdef CCLQR_PROFILE
    Prof prof;
    prof.start();
#endif

    if (valid) {
        for (int e = t; e < nz; e += G) { int l = e / 13, c = e - 13 * l; L[Y.Z + e] = a.z0[inst * nz + M->perm[l] * 13 + c]; }
        for (int e = t; e < 5 * nb; e += G) L[Y.LAM + e] = (a.lam && a.k0 > 1) ? a.lam[inst * 5 * nb + e] : 0.0;
    } else {
        for (int e = t; e < Y.total; e += G) L[e] = 0.0;
        for (int e = t; e < nb; e += G) L[Y.Z + 13 * e + 3] = 1.0;
    }
    __syncthreads();

    int worst = 0;
    // Launch arguments that are only needed once per step or at the end are read from the kernel-argument segment where they are
    // used, through a pointer the optimiser cannot see through, instead of sitting in (spilled) scalar registers for the whole launch.
    typedef const __attribute__((address_space(4))) RolloutArgs* KernArgs;
    KernArgs ap = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const int k0 = a.k0;
    int nsteps = a.steps;
    for (int kk = 0; kk < nsteps; kk++) {
        const int k = k0 + kk;
        asm volatile("" : "+s"(ap));
        FRESH_PHASE(M, r); FRESH_FLAGS;
        double* const traj_out = ap->traj;
        if (traj_out && valid)
            for (int e = fresh_lane(t); e < nz; e += G) { int l = e / 13, c = e - 13 * l; traj_out[((size_t)inst * ap->steps + kk) * nz + M->perm[l] * 13 + c] = L[FRESH_Y.Z + e]; }

        STAMP(PF_IO);
        // ---------------- feedback law (lqr.jl:89-139 / lqr_tracking.jl:46-71)
        const bool gate = (C->N <= 0) || (k < C->N);
        const int ksp = (C->nsp > 1) ? ((k - 1 < C->nsp) ? k - 1 : C->nsp - 1) : 0;
        const int kidx = (C->N <= 0) ? 0 : ((k - 1 < C->nK) ? k - 1 : C->nK - 1);
        const long long ginst = ap->inst0 + inst;     // global instance index: selects the controller table when there is one per instance
        if (gate && valid) ph_control_error(FRESH_T, nb, FRESH_Y, L, r, C, C->zd + ginst * C->zd_stride + (size_t)ksp * nz);
        else if (t < nb) L[FRESH_Y.UJ + t] = 0.0;
        __syncthreads();
        if (gate) {
            for (int i = 0; i < C->mu; i++) {
                double part = 0.0;
                if (C->K && valid) part = ph_gain_partial(FRESH_T, G, nb, FRESH_Y, L, C->K + ginst * C->K_stride + ((size_t)kidx * C->mu + i) * 12 * nb);
                double s = group_sum<G>(part);
                if (FRESH_T == 0 && valid) {
                    double u = (C->Fd ? C->Fd[ginst * C->Fd_stride + (size_t)ksp * C->mu + i] : 0.0) - s;
                    // noise: injected by the caller, or generated for this launch by philox_fill_kernel (capi.hip)
                    if (EXTRA >= 1 && C->noise_scale != 0.0 && ap->noise) u += C->noise_scale * ap->noise[(size_t)inst * ap->noise_stride + (k - 1)];
                    L[FRESH_Y.UJ + C->cj[i]] += u;
                }
                __syncthreads();
            }
        }
        if (EXTRA >= 2 && C->has_pid) {
            if (valid) ph_pid(FRESH_T, nb, FRESH_Y, L, r, C, dt, k == 1);
            __syncthreads();
        }
        STAMP(PF_CONTROL);
        FRESH_PHASE(M, r); FRESH_FLAGS;
        // ---------------- per-step invariants
        if (lg < NL) ph_forces<TREE>(fresh_lane(tl), nb, FRESH_Y, L, r, M, lg == 0);
        ph_knot_jac(FRESH_T, nb, FRESH_Y, L, r);
        __syncthreads();
        if (TREE) ph_force_map_tree(FRESH_T, G, nb, FRESH_Y, L, M);
        else ph_force_map(FRESH_T, G, nb, FRESH_Y, L, M->end_mask);
        __syncthreads();
        STAMP(PF_FORCES);
        PCOUNT(PF_STEPS);

        // ---------------- newton! (tolerances and line search: SURVEY 8a-bis)
        FRESH_PHASE(M, r); FRESH_FLAGS;
        bool done = false;
        int its = newton_solve<G, TREE>(t, nb, FRESH_Y, L, r, M, dt, valid && !dead, &done PROF_PASS);
        if (valid && !dead) {
            if (!done) fl |= 4;
            if (its > worst) worst = its;
            if (!done && its < NEWTON_MAXIT) fl |= 2;   // stopped early on a non-finite residual: the instance is frozen from then on
            else ph_update(FRESH_T, nb, FRESH_Y, L);
        }
        __syncthreads();
        asm volatile("" : "+s"(ap));
        nsteps = ap->steps;      // read again rather than kept in a scalar register through the step
    }
    asm volatile("" : "+s"(ap));
    FRESH_PHASE(M, r); FRESH_FLAGS;
    if (valid) {
        double* const zT = ap->zT;
        double* const lam = ap->lam;
        int* const status = ap->status;
        const MechDev* const MT = ap->M;
        const int nbT = MT->nb, nzT = 13 * nbT;      // read again here rather than kept in scalar registers through the launch
        const Lay YT = make_layout(nbT, 0);
        for (int e = t; e < nzT; e += G) { int l = e / 13, c = e - 13 * l; zT[inst * nzT + MT->perm[l] * 13 + c] = L[YT.Z + e]; }
        if (lam) for (int e = t; e < 5 * nbT; e += G) lam[inst * 5 * nbT + e] = L[YT.LAM + e];
        if (status && t == 0) status[inst] = bad ? -worst : worst;
        if (EXTRA >= 2 && ap->pid_state && t < nbT) { double* const ps = ap->pid_state; ps[(inst * nbT + t) * 2] = r.pid_int; ps[(inst * nbT + t) * 2 + 1] = r.pid_last; }
    }
#ifdef CCLQR_PROFILE
    prof.stamp(PF_IO);
    prof.flush();
#endif
}

#undef valid
#undef dead
#undef bad
#undef FRESH_FLAGS

#ifdef CCLQR_PROFILE
extern "C" int cclqr_prof_read(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * PF_N);
    if (e == hipSuccess && reset) { unsigned long long z[PF_N] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return e == hipSuccess ? PF_N : -1;
}
#endif

// lanes per instance: one lane per link, and a tree needs 8 lanes per neighbour group of its elimination (up to CCLQR_MAXK groups)
int rollout_lanes_per_instance(int nb, int tree) {
    if (!tree) return chain_lanes_per_instance(nb);
    const int g = nb <= 4 ? 16 : (nb <= 8 ? 32 : 64);
    return tree > g ? (tree <= 16 ? 16 : (tree <= 32 ? 32 : 64)) : g;   // tree = lanes an elimination step needs (0 for chains)
}

size_t rollout_lds_bytes(int nb, int tree, int npairs) {
    if (!tree) return chain_lds_bytes(nb);
    int G = rollout_lanes_per_instance(nb, tree);
    return (size_t)(64 / G) * make_layout(nb, tree ? 2 * npairs : 0).total * sizeof(double);
}

template <int G, bool TREE>
static hipError_t launch_one(const RolloutArgs& a, int extra, unsigned grid, size_t lds, hipStream_t stream) {
    const void* f = extra == 0 ? (const void*)rollout_kernel<G, TREE, 0> : (extra == 1 ? (const void*)rollout_kernel<G, TREE, 1> : (const void*)rollout_kernel<G, TREE, 2>);
    hipError_t e = set_max_dynamic_lds_once(f, lds);
    if (e != hipSuccess) return e;
    if (extra == 0) hipLaunchKernelGGL((rollout_kernel<G, TREE, 0>), dim3(grid), dim3(64), lds, stream, a);
    else if (extra == 1) hipLaunchKernelGGL((rollout_kernel<G, TREE, 1>), dim3(grid), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL((rollout_kernel<G, TREE, 2>), dim3(grid), dim3(64), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_rollout(const RolloutArgs& a, int nb, int tree, int npairs, int extra, int newton_mode, hipStream_t stream) {
    if (!tree) return launch_rollout_chain(a, nb, extra, newton_mode, stream);   // forests of chains: the register-resident kernel (rollout_chain.hip)
    const int G = rollout_lanes_per_instance(nb, tree);
    const int per_wg = 64 / G;
    const size_t lds = rollout_lds_bytes(nb, tree, npairs);
    const unsigned grid = (unsigned)((a.n_inst + per_wg - 1) / per_wg);
    if (grid == 0) return hipSuccess;
    return G == 16 ? launch_one<16, true>(a, extra, grid, lds, stream) : (G == 32 ? launch_one<32, true>(a, extra, grid, lds, stream) : launch_one<64, true>(a, extra, grid, lds, stream));
}

}  // namespace cclqr
