// cclqr_internal.h -- host-side handles and kernel launch arguments shared by the translation units of libcclqr.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "cclqr_dev.h"

#define CCLQR_ESINGULAR_ (-2)

namespace cclqr {

struct RolloutArgs {
    const MechDev* M;
    const CtrlDev* C;
    int64_t n_inst;
    int steps, k0;
    const double* z0;     // [n_inst][nb][13] user body order
    double* lam;          // [n_inst][5 nb] internal order, or null
    const double* noise;  // [n_inst][noise_stride] or null
    int64_t noise_stride;
    double* pid_state;    // [n_inst][nb][2] integrated / last PID errors carried between launches (opaque link order), or nullptr
    int64_t inst0;        // global index of instance 0 of this launch (Philox stream = global instance index, so shards reproduce the whole batch)
    double* traj;         // [n_inst][steps][nb][13] or null
    double* zT;           // [n_inst][nb][13]
    int* status;          // [n_inst] or null
    double eps_alone;     // measured-error Newton mode (RELAX kernels only): a solve also stops when ||f|| falls below this
    int carry;            // CCLQR_ROLLOUT_CARRY_STATUS: `status` comes in with the instance's status of the launches before (0: none): an instance that was lost stays
                          // frozen and keeps its status, the others merge this launch's Newton count / failure into it
    int ipw;              // chain and tree kernels: instances per wavefront, 1 .. 64 / lanes per instance (lane groups beyond it hold no instance).  The caller of
                          // launch_rollout_chain / launch_rollout_treereg passes 0 (the launch chooses: chain_instances_per_wavefront) or nonzero = pack every wavefront full
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel) instead of once per launch: remembers the largest size set so
// far and only calls the runtime when a launch needs more (capi.hip)
hipError_t set_max_dynamic_lds_once(const void* fn, size_t lds);

hipError_t launch_philox_fill(double* out, unsigned key0, long long inst0, long long n_inst, int k0, int steps, hipStream_t stream);
// forests of chains (rollout_chain.hip)
int chain_lanes_per_instance(int nb);
int chain_layout_links(int nb);
int chain_lanes_per_link(int nb);
size_t chain_lds_bytes(int nb);
int chain_instances_per_wavefront(int nb, int64_t n_inst, int steps, bool packed);
int spread_instances_per_wavefront(int full, int64_t n_inst, int steps, bool packed);      // the rule itself (rollout_chain.hip); full = 64 / lanes per instance
hipError_t launch_rollout_chain(const RolloutArgs& a, int nb, int extra, int newton_mode, hipStream_t stream);
// closed-loop mechanisms (rollout_loop.hip)
size_t loop_lds_bytes(int nb, int nj);
hipError_t launch_rollout_loop(const RolloutArgs& a, int nb, int nj, int newton_mode, hipStream_t stream);

struct LinArgs {
    const MechDev* M;
    int nk, mu;
    int cj[CCLQR_MAXL];      // controlled links (internal)
    const double* zd;        // [nk][nb][13] internal link order
    const double* Fd;        // [nk][mu]
    double *A, *Bu, *Bl, *G; // [nk][...] internal link order
    int* status;             // [nk]
};
size_t linearize_lds_bytes(int nb, int tree, int npairs);
// closed-loop mechanisms (rollout_loop.hip): the model with the multipliers exogenous, and the rank-revealing projection onto (A', D)
size_t linearize_loop_lds_bytes(int nb, int nj);
hipError_t launch_linearize_loop(const LinArgs& a, int nb, int nj, hipStream_t stream);
size_t project_model_lds_bytes(int mx, int mu, int ml);
bool project_model_fits(int mx, int mu, int ml);      // dynamic + static LDS of project_model_kernel within one CU's 160 KB
hipError_t launch_project_model(int nk, int mx, int mu, int ml, const double* A, const double* Bu, const double* Bl, const double* G, double* Ap, double* D,
                                double* res, int* rank, hipStream_t stream);
hipError_t launch_linearize(const LinArgs& a, int nb, int tree, int npairs, hipStream_t stream);

struct RicArgs {
    int nprob, mx, mu, ml, N;
    int time_varying;        // 1: A,Bu,Bl,G are [N-1][...] per problem (knot k uses index k-1)
    double tol;
    const double *A, *Bu, *Bl, *G, *Q, *R;
    double* K;               // [nprob][N-1][mu][mx] ([nprob][mu][mx] with keep_last)
    int* kbreak;             // [nprob]
    int* status;             // [nprob]
    double* work;            // ric_total_work_doubles(args) doubles
    int* stop;               // [nprob] scratch flags of the tiled path
    int path;                // 0: chosen by problem size and count, 1 resident, 2 tiled
    int bf16_terms;          // 0: fp64 MFMA; 1..3: split-bf16 products with fp32 accumulation (tiled path)
    int keep_last;           // 1: K is [nprob][mu][mx], the gain of the last executed backward step (= Ku[1] after the back-fill)
    long long kpad = 0;      // doubles left free between the tables of consecutive problems in K (0: contiguous)
};
size_t ric_total_work_doubles(const RicArgs& a);
hipError_t launch_riccati(const RicArgs& a, hipStream_t stream);

}  // namespace cclqr

// opaque handles of include/cclqr.h
struct cclqr_mech {
    cclqr::MechDev host;       // internal link order
    cclqr::MechDev* dev;
    int nb;
    int nj;                    // joints: nb for trees, >= nb for closed-loop mechanisms
    int link_of_body[CCLQR_MAXL];   // user body  -> internal link
    int link_of_joint[CCLQR_MAXL];  // user joint -> internal link (= link of its child body)
    int device;
};
// zero doubles behind a controller's gain table: the control phase fetches ceil(12 NBP / G) G entries of a row whatever the mechanism's own
// 12 nb, i.e. up to 12 x 64 - 12 past the end of the LAST row when a short chain runs on a long image.  With one table per instance (n_ctrl > 1)
// every instance's table carries its own zero pad of that overrun (capi.hip gain_row_overrun, CtrlDev::K_stride includes it), so that the entries
// past an instance's last row are zeros and never a NEIGHBOUR's gains -- which may be Inf / NaN when that neighbour's recursion diverged, and
// 0 * NaN would poison a healthy instance's input
#define CCLQR_K_PAD 768
struct cclqr_ctrl {
    cclqr::CtrlDev host;
    cclqr::CtrlDev* dev;
    double *K_dev, *zd_dev, *Fd_dev;
    size_t Fd_len;       // doubles in Fd_dev
    int nb;
    int device;           // the device the tables (and the noise workspace) live on = the mechanism's
    // workspace of the counter-based noise of one launch (noise_philox), grown on demand and kept with the handle
    double* noise_ws;
    size_t noise_ws_cap;
};
