// rollout.hip -- launch geometry and dispatch of the LQR-controlled rollout (simulate! + control_lqr!/control_trackinglqr! fused).
// The kernels live in rollout_chain.hip (forests of chains: every BASELINE config), rollout_treereg.hip (branching trees) and rollout_loop.hip
// (closed loops).  The LDS-resident tree kernel of rounds 1-3 that used to be here was replaced by rollout_treereg.hip in round 4; its phase
// functions (cclqr_dev.h ph_*, cclqr_newton.h newton_solve) remain in use by the linearisation kernels (linearize.hip).
//
// Replaces: ConstrainedDynamics.simulate!/newton! as driven by the reference (examples/lqr_cartpole.jl:44) with
//           control_lqr! (src/control/lqr.jl:89-139) / control_trackinglqr! (src/control/lqr_tracking.jl:46-71).
#include "cclqr_dev.h"
#include "cclqr_internal.h"

namespace cclqr {

// lanes per instance / dynamic LDS per workgroup of the chain kernel.  (The `tree` / `npairs` arguments belong to the interface of rounds 1-3,
// cclqr_internal.h; branching trees are launched through launch_rollout_treereg, cclqr_treereg_tables.h, which has its own geometry.)
int rollout_lanes_per_instance(int nb, int tree) { (void)tree; return chain_lanes_per_instance(nb); }
size_t rollout_lds_bytes(int nb, int tree, int npairs) { (void)tree; (void)npairs; return chain_lds_bytes(nb); }

hipError_t launch_rollout(const RolloutArgs& a, int nb, int tree, int npairs, int extra, int newton_mode, hipStream_t stream) {
    (void)npairs;
    if (tree) return hipErrorInvalidValue;      // capi.hip sends branching trees to launch_rollout_treereg
    return launch_rollout_chain(a, nb, extra, newton_mode, stream);
}

}  // namespace cclqr
