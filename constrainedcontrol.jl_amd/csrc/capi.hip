// capi.hip -- the extern "C" boundary of libcclqr.so (include/cclqr.h).  Host-side work here is limited to
// validation, index permutations (user body order <-> breadth-first link order) and device memory plumbing;
// every piece of hot-path arithmetic runs in the HIP kernels.
#include "../../include/cclqr.h"
#include "cclqr_internal.h"
#include "cclqr_tables.h"
#include "cclqr_treereg_tables.h"
#include "cclqr_newton.h"
static_assert(CCLQR_NEWTON_MAXIT == NEWTON_MAXIT, "include/cclqr.h and cclqr_newton.h disagree on the Newton iteration cap");
#include "cclqr_wscache.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

using namespace cclqr;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                         \
    do {                                                                                                  \
        hipError_t e_ = (x);                                                                              \
        if (e_ != hipSuccess) return fail(CCLQR_EHIP, std::string(#x) + ": " + hipGetErrorString(e_));   \
    } while (0)

extern "C" const char* cclqr_last_error(void) { return g_err.c_str(); }

#include <mutex>
namespace cclqr {
hipError_t set_max_dynamic_lds_once(const void* fn, size_t lds) {
    struct Entry { int dev; const void* fn; size_t lds; };
    static std::mutex mu;
    static std::vector<Entry> seen;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& s : seen)
        if (s.dev == dev && s.fn == fn) {
            if (s.lds >= lds) return hipSuccess;
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) s.lds = lds;
            return e;
        }
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) seen.push_back({dev, fn, lds});
    return e;
}
}  // namespace cclqr

// Device workspaces of the host-pointer entry points (linearize / riccati / rollout staging) are kept per thread and reused by the
// next call instead of a hipMalloc + hipFree (both synchronise the device) per call; cclqr_release_workspaces() returns them.
static thread_local WsCache g_ws;
static hipError_t ws_get(void** p, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const int rc = ws_get_on(g_ws, dev, p, bytes, [](void** q, size_t n) { return (int)hipMalloc(q, n); }, [](void* q) { return (int)hipFree(q); },
                             [](int d) { (void)hipSetDevice(d); });
    return rc < 0 ? hipErrorInvalidDevice : (hipError_t)rc;
}
struct WsScope { ~WsScope() { g_ws.used = 0; } };     // every block is free again when the entry point returns
// blocks of a thread that exits without calling this stay allocated until the process ends (thread_local destructors must not call
// into a HIP runtime that may already be shutting down)
extern "C" int cclqr_release_workspaces(void) {
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    if (have && g_ws.device >= 0 && g_ws.device != cur) (void)hipSetDevice(g_ws.device);
    for (auto& b : g_ws.blocks) if (b.first) (void)hipFree(b.first);
    if (have && g_ws.device >= 0 && g_ws.device != cur) (void)hipSetDevice(cur);
    g_ws.blocks.clear(); g_ws.used = 0; g_ws.device = -1;
    return CCLQR_OK;
}
// every entry point that takes a handle runs on the device the handle's tables live on
static int check_device(const cclqr_mech* m) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(CCLQR_EHIP, "hipGetDevice failed");
    if (m && m->device != dev) return fail(CCLQR_EINVAL, "the mechanism was created on device " + std::to_string(m->device) + ", the calling thread is on device " + std::to_string(dev));
    return CCLQR_OK;
}
extern "C" int cclqr_version(void) { return CCLQR_ABI_VERSION; }
// sizeof / offsetof of the structs of include/cclqr.h as compiled here (the order is the header's comment): a foreign-language mirror checks itself against it
extern "C" int cclqr_abi_layout(int32_t* out, int32_t n) {
#define OFF(T, f) (int32_t) offsetof(T, f)
    const int32_t v[CCLQR_ABI_LAYOUT_LEN] = {
        (int32_t)sizeof(cclqr_mech_desc), OFF(cclqr_mech_desc, nb), OFF(cclqr_mech_desc, ne), OFF(cclqr_mech_desc, dt), OFF(cclqr_mech_desc, g), OFF(cclqr_mech_desc, mass),
        OFF(cclqr_mech_desc, inertia), OFF(cclqr_mech_desc, parent), OFF(cclqr_mech_desc, child), OFF(cclqr_mech_desc, type), OFF(cclqr_mech_desc, p1), OFF(cclqr_mech_desc, p2),
        OFF(cclqr_mech_desc, axis), OFF(cclqr_mech_desc, qoff),
        (int32_t)sizeof(cclqr_ctrl_desc), OFF(cclqr_ctrl_desc, mu), OFF(cclqr_ctrl_desc, ctrl_joint), OFF(cclqr_ctrl_desc, nK), OFF(cclqr_ctrl_desc, N), OFF(cclqr_ctrl_desc, K),
        OFF(cclqr_ctrl_desc, nsp), OFF(cclqr_ctrl_desc, zd), OFF(cclqr_ctrl_desc, Fd), OFF(cclqr_ctrl_desc, fric), OFF(cclqr_ctrl_desc, noise_scale), OFF(cclqr_ctrl_desc, npid),
        OFF(cclqr_ctrl_desc, pid_joint), OFF(cclqr_ctrl_desc, pid_P), OFF(cclqr_ctrl_desc, pid_I), OFF(cclqr_ctrl_desc, pid_D), OFF(cclqr_ctrl_desc, pid_goal),
        OFF(cclqr_ctrl_desc, noise_philox), OFF(cclqr_ctrl_desc, noise_seed), OFF(cclqr_ctrl_desc, n_ctrl),
        (int32_t)sizeof(cclqr_riccati_opts), OFF(cclqr_riccati_opts, path), OFF(cclqr_riccati_opts, bf16_terms), OFF(cclqr_riccati_opts, keep_last), OFF(cclqr_riccati_opts, reserved),
        (int32_t)sizeof(cclqr_rollout_opts), OFF(cclqr_rollout_opts, first_instance), OFF(cclqr_rollout_opts, pid_state_dev), OFF(cclqr_rollout_opts, pid_state_len),
        OFF(cclqr_rollout_opts, noise_ws_dev), OFF(cclqr_rollout_opts, noise_ws_len), OFF(cclqr_rollout_opts, newton_mode), OFF(cclqr_rollout_opts, flags),
        OFF(cclqr_rollout_opts, newton_eps_alone)};
#undef OFF
    if (n < 0 || (n > 0 && !out)) return fail(CCLQR_EINVAL, "bad argument");
    for (int i = 0; i < n && i < CCLQR_ABI_LAYOUT_LEN; i++) out[i] = v[i];
    return CCLQR_ABI_LAYOUT_LEN;
}
extern "C" int cclqr_device_count(int32_t* n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(CCLQR_EHIP, hipGetErrorString(e)); }
    *n = c;
    return CCLQR_OK;
}
extern "C" int cclqr_set_device(int32_t dev) { HIPCHK(hipSetDevice(dev)); return CCLQR_OK; }

extern "C" int cclqr_mech_create(const cclqr_mech_desc* d, cclqr_mech** out) {
    if (!d || !out) return fail(CCLQR_EINVAL, "null argument");
    cclqr_mech* m = new cclqr_mech();
    std::string err;
    int rc = build_mech_tables(d, m, err);
    if (rc != CCLQR_OK) { delete m; return fail(rc, err); }
    // branching trees: the tables of the register-resident tree kernel (sibling lists, elimination schedule) ride behind the MechDev in the
    // same allocation (cclqr_treereg.h treereg_of)
    std::vector<char> image(m->host.tree ? treereg_offset() + sizeof(TreeRegDev) : sizeof(MechDev), 0);
    memcpy(image.data(), &m->host, sizeof(MechDev));
    if (m->host.tree && !m->host.loop) {
        TreeRegDev* R = new (image.data() + treereg_offset()) TreeRegDev;
        if (!build_treereg_tables(m->host, *R, err)) { delete m; return fail(CCLQR_EUNSUPPORTED, err); }
    }
    hipError_t e = hipGetDevice(&m->device);
    if (e == hipSuccess) e = hipMalloc((void**)&m->dev, image.size());
    if (e == hipSuccess) e = hipMemcpy(m->dev, image.data(), image.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete m; return fail(CCLQR_EHIP, std::string("mechanism upload: ") + hipGetErrorString(e)); }
    (void)spread_instances_per_wavefront(2, 1, 8, false);      // (reads the device's compute-unit count once, here: never inside a caller's hipGraph capture)
    *out = m;
    return CCLQR_OK;
}

extern "C" int cclqr_mech_destroy(cclqr_mech* m) {
    if (!m) return CCLQR_OK;
    if (m->dev) (void)hipFree(m->dev);
    delete m;
    return CCLQR_OK;
}

// doubles the rollout kernels' control phase fetches past the end of a gain row: they read ceil(12 NBP / G) G entries of a row (G lanes per instance,
// the LDS image laid out for NBP >= nb links) and let the surplus meet a zero factor.  Closed-loop kernel: exact row length.
static size_t gain_row_overrun(const cclqr_mech* m) {
    if (m->host.loop) return 0;
    const int G = m->host.tree ? treereg_lanes(m->nb, m->host.tree) : chain_lanes_per_instance(m->nb);
    const int nbp = m->host.tree ? treereg_layout_links(m->nb, m->host.tree) : chain_layout_links(m->nb);
    const long long over = (long long)((12 * nbp + G - 1) / G) * G - 12LL * m->nb;
    return over > 0 ? (size_t)((over + 1) & ~1LL) : 0;
}

extern "C" int cclqr_ctrl_create(const cclqr_mech* m, const cclqr_ctrl_desc* d, cclqr_ctrl** out) {
    if (!m || !d || !out) return fail(CCLQR_EINVAL, "null argument");
    { int rc = check_device(m); if (rc != CCLQR_OK) return rc; }
    CtrlHostTables T;
    std::string err;
    int rc = build_ctrl_tables(m, d, T, err);
    if (rc != CCLQR_OK) return fail(rc, err);
    cclqr_ctrl* c = new cclqr_ctrl();
    memset(c, 0, sizeof(*c));
    c->nb = m->nb;
    c->device = m->device;
    hipError_t e = hipMalloc((void**)&c->zd_dev, T.zd.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(c->zd_dev, T.zd.data(), T.zd.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && !T.K.empty()) {
        // CCLQR_K_PAD zero doubles behind the last gain row: the rollout kernels fetch a row in whole strides of their lane group and let the
        // entries past its end meet a zero factor (rollout_chain.hip, control phase) -- past the LAST row that read must stay inside the table
        // (one table per instance: each with its own pad, see CCLQR_K_PAD)
        const size_t ntab = T.H.n_ctrl > 1 ? (size_t)T.H.n_ctrl : 1, per = T.K.size() / ntab, padi = ntab > 1 ? gain_row_overrun(m) : 0;
        const size_t total = ntab * (per + padi) + CCLQR_K_PAD;
        e = hipMalloc((void**)&c->K_dev, total * sizeof(double));
        if (e == hipSuccess) e = hipMemset(c->K_dev, 0, total * sizeof(double));
        if (e == hipSuccess) e = padi ? hipMemcpy2D(c->K_dev, (per + padi) * sizeof(double), T.K.data(), per * sizeof(double), per * sizeof(double), ntab, hipMemcpyHostToDevice)
                                      : hipMemcpy(c->K_dev, T.K.data(), T.K.size() * sizeof(double), hipMemcpyHostToDevice);
        if (ntab > 1) T.H.K_stride = (long long)(per + padi);
    }
    if (e == hipSuccess && !T.Fd.empty()) {
        e = hipMalloc((void**)&c->Fd_dev, T.Fd.size() * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(c->Fd_dev, T.Fd.data(), T.Fd.size() * sizeof(double), hipMemcpyHostToDevice);
        c->Fd_len = T.Fd.size();
    }
    c->host = T.H;
    c->host.K = c->K_dev; c->host.zd = c->zd_dev; c->host.Fd = c->Fd_dev;
    if (!c->Fd_dev) c->host.Fd_stride = 0;
    if (!c->K_dev) c->host.K_stride = 0;
    if (e == hipSuccess) e = hipMalloc((void**)&c->dev, sizeof(CtrlDev));
    if (e == hipSuccess) e = hipMemcpy(c->dev, &c->host, sizeof(CtrlDev), hipMemcpyHostToDevice);
    if (e != hipSuccess) { cclqr_ctrl_destroy(c); return fail(CCLQR_EHIP, std::string("controller upload: ") + hipGetErrorString(e)); }
    *out = c;
    return CCLQR_OK;
}

// controlfunction hook (lqr.jl:14, :56): the host's closure has computed the joint inputs of the next step
extern "C" int cclqr_ctrl_set_feedforward(cclqr_ctrl* c, const double* Fd, int64_t len, int32_t on_device, void* stream) {
    if (!c || !Fd) return fail(CCLQR_EINVAL, "null argument");
    if (!c->Fd_dev) return fail(CCLQR_EINVAL, "the controller was created without a feed-forward table");
    if (len < 0 || (size_t)len != c->Fd_len) return fail(CCLQR_EINVAL, "feed-forward table length differs from the one the controller was created with");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != c->device) return fail(CCLQR_EINVAL, "the calling thread is not on the controller's device");
    hipError_t e = on_device ? hipMemcpyAsync(c->Fd_dev, Fd, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream)
                             : hipMemcpy(c->Fd_dev, Fd, (size_t)len * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(CCLQR_EHIP, std::string("feed-forward upload: ") + hipGetErrorString(e));
    return CCLQR_OK;
}

// in-place permutation of the 12-column body blocks of every gain row from the caller's body order to the kernels' link order
// (one workgroup per row, the row staged in LDS); what build_ctrl_tables does on the host for caller-supplied gains
__global__ void k_rows_to_link_order_kernel(double* K, long long nrows, long long rows_per_table, long long table_stride, int nb, const MechDev* M) {
    extern __shared__ double row[];
    const long long r = blockIdx.x;
    if (r >= nrows) return;
    double* p = K + (r / rows_per_table) * table_stride + (r % rows_per_table) * 12 * nb;
    for (int e = threadIdx.x; e < 12 * nb; e += blockDim.x) row[e] = p[e];
    __syncthreads();
    for (int e = threadIdx.x; e < 12 * nb; e += blockDim.x) { const int l = e / 12; p[e] = row[12 * M->perm[l] + (e - 12 * l)]; }
}

extern "C" int cclqr_ctrl_create_lqr_batch(const cclqr_mech* m, int32_t n_ctrl, const double* zd, int32_t mu, const int32_t* ctrl_joint,
                                           const double* Fd, const double* Q, const double* R, int32_t N, int32_t infinite_horizon, double tol,
                                           int32_t* kbreak, cclqr_ctrl** out) {
    const bool inf = infinite_horizon != 0;
    if (!m || !zd || !Q || !out || (mu > 0 && (!ctrl_joint || !R))) return fail(CCLQR_EINVAL, "null argument");
    { int rc = check_device(m); if (rc != CCLQR_OK) return rc; }
    if (m->host.loop) return fail(CCLQR_EUNSUPPORTED, "batched LQR construction is for tree mechanisms (closed loops: cclqr_linearize_projected)");
    if (n_ctrl < 1 || N < 2 || mu < 1 || mu > m->nb) return fail(CCLQR_EINVAL, "bad sizes");
    const int nb = m->nb;
    const size_t nz = 13 * (size_t)nb, mx = 12 * (size_t)nb, ml = 5 * (size_t)nb, np = (size_t)n_ctrl;
    if (linearize_lds_bytes(nb, m->host.tree, m->host.npairs) > 160 * 1024) return fail(CCLQR_EUNSUPPORTED, "instance does not fit LDS");
    LinArgs la;
    memset(&la, 0, sizeof(la));
    la.M = m->dev; la.nk = n_ctrl; la.mu = mu;
    cclqr_ctrl* c = new cclqr_ctrl();
    memset(c, 0, sizeof(*c));
    c->nb = nb;
    c->device = m->device;
    CtrlDev& H = c->host;
    // LQR{T,Inf} (lqr.jl:25-27, 40-43): the recursion runs its N = Ntemp steps, only Ku[1] is kept and the feedback is never gated
    const size_t nKtab = inf ? 1 : (size_t)(N - 1);
    H.mu = mu; H.nK = (int)nKtab; H.N = inf ? 0 : N; H.nsp = 1; H.n_ctrl = n_ctrl;
    for (int i = 0; i < mu; i++) {
        if (ctrl_joint[i] < 0 || ctrl_joint[i] >= nb) { delete c; return fail(CCLQR_EINVAL, "controlled joint out of range"); }
        la.cj[i] = m->link_of_joint[ctrl_joint[i]];
        H.cj[i] = la.cj[i];
    }
    const size_t padi = n_ctrl > 1 ? gain_row_overrun(m) : 0;       // every instance's table ends in its own zero pad (CCLQR_K_PAD)
    const long long tab_stride = (long long)nKtab * mu * (long long)mx + (long long)padi;
    H.K_stride = n_ctrl > 1 ? tab_stride : 0;
    H.zd_stride = n_ctrl > 1 ? (long long)nz : 0;
    H.Fd_stride = (n_ctrl > 1 && Fd) ? mu : 0;
    // setpoints in link order for the rollout's control law
    std::vector<double> zl(np * nz);
    for (size_t s = 0; s < np; s++)
        for (int l = 0; l < nb; l++) memcpy(&zl[(s * nb + l) * 13], zd + (s * nb + m->host.perm[l]) * 13, 13 * sizeof(double));
    const size_t nK = np * (size_t)tab_stride;
    double *dzd = nullptr, *dA = nullptr, *dBu = nullptr, *dBl = nullptr, *dG = nullptr, *dQ = nullptr, *dR = nullptr, *dwork = nullptr;
    int *dlst = nullptr, *dkb = nullptr, *dst = nullptr, *dstop = nullptr;
    std::vector<int> lst(np), kb(np), st(np);
    WsScope scope;
    hipError_t e = hipMalloc((void**)&c->zd_dev, np * nz * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(c->zd_dev, zl.data(), np * nz * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&c->K_dev, (nK + CCLQR_K_PAD) * sizeof(double));      // (padding: see cclqr_ctrl_create)
    if (e == hipSuccess && Fd) {
        e = hipMalloc((void**)&c->Fd_dev, np * mu * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(c->Fd_dev, Fd, np * mu * sizeof(double), hipMemcpyHostToDevice);
        c->Fd_len = (size_t)np * mu;
    }
    if (e == hipSuccess) e = ws_get((void**)&dzd, np * nz * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dA, np * mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBu, np * mx * mu * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBl, np * mx * ml * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dG, np * ml * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dlst, np * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(dzd, zd, np * nz * sizeof(double), hipMemcpyHostToDevice);
    // linearsystem at every setpoint (lqr.jl:63), one launch; the matrices stay on the device
    la.zd = dzd; la.Fd = c->Fd_dev; la.A = dA; la.Bu = dBu; la.Bl = dBl; la.G = dG; la.status = dlst;
    if (e == hipSuccess) e = launch_linearize(la, nb, m->host.tree, m->host.npairs, nullptr);
    // dlqr for every setpoint (lqr.jl:141-184), gains written straight into the controller's table
    RicArgs ra;
    ra.nprob = n_ctrl; ra.mx = (int)mx; ra.mu = mu; ra.ml = (int)ml; ra.N = N; ra.time_varying = 0; ra.tol = tol; ra.path = 0; ra.bf16_terms = 0; ra.keep_last = inf ? 1 : 0; ra.kpad = (long long)padi;
    const size_t wd = ric_total_work_doubles(ra);
    if (e == hipSuccess) e = ws_get((void**)&dQ, mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dR, (size_t)(mu * mu + 1) * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dwork, wd * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dstop, np * sizeof(int));
    if (e == hipSuccess) e = ws_get((void**)&dkb, np * sizeof(int));
    if (e == hipSuccess) e = ws_get((void**)&dst, np * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(dQ, Q, mx * mx * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dR, R, (size_t)mu * mu * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(c->K_dev, 0, (nK + CCLQR_K_PAD) * sizeof(double));
    ra.stop = dstop; ra.A = dA; ra.Bu = dBu; ra.Bl = dBl; ra.G = dG; ra.Q = dQ; ra.R = dR; ra.K = c->K_dev; ra.kbreak = dkb; ra.status = dst; ra.work = dwork;
    if (e == hipSuccess) e = launch_riccati(ra, nullptr);
    if (e == hipSuccess) {
        const long long nrows = (long long)np * (long long)nKtab * mu;
        hipLaunchKernelGGL(k_rows_to_link_order_kernel, dim3((unsigned)nrows), dim3(128), mx * sizeof(double), nullptr, c->K_dev, nrows, (long long)nKtab * mu, tab_stride, nb, m->dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(lst.data(), dlst, np * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(kb.data(), dkb, np * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(st.data(), dst, np * sizeof(int), hipMemcpyDeviceToHost);
    H.K = c->K_dev; H.zd = c->zd_dev; H.Fd = c->Fd_dev;
    if (e == hipSuccess) e = hipMalloc((void**)&c->dev, sizeof(CtrlDev));
    if (e == hipSuccess) e = hipMemcpy(c->dev, &c->host, sizeof(CtrlDev), hipMemcpyHostToDevice);
    if (e != hipSuccess) { cclqr_ctrl_destroy(c); return fail(CCLQR_EHIP, std::string("batched LQR construction: ") + hipGetErrorString(e)); }
    for (size_t p = 0; p < np; p++) {
        if (kbreak) kbreak[p] = kb[p];
        if (lst[p] <= 0) { cclqr_ctrl_destroy(c); return fail(CCLQR_ENOCONV, "Newton did not converge at setpoint " + std::to_string(p)); }
        if (st[p] != 0) { cclqr_ctrl_destroy(c); return fail(CCLQR_ESINGULAR, "G*Bl or M is singular at setpoint " + std::to_string(p)); }
    }
    *out = c;
    return CCLQR_OK;
}

extern "C" int cclqr_ctrl_destroy(cclqr_ctrl* c) {
    if (!c) return CCLQR_OK;
    if (c->K_dev) (void)hipFree(c->K_dev);
    if (c->zd_dev) (void)hipFree(c->zd_dev);
    if (c->Fd_dev) (void)hipFree(c->Fd_dev);
    if (c->noise_ws) (void)hipFree(c->noise_ws);
    if (c->dev) (void)hipFree(c->dev);
    delete c;
    return CCLQR_OK;
}

extern "C" int cclqr_rollout_layout_links(const cclqr_mech* m, int32_t* links) {
    if (!m || !links) return fail(CCLQR_EINVAL, "null argument");
    // links the rollout kernel's LDS image is laid out for (it names the instantiation): chains rollout_chain_kernel, branching trees
    // rollout_treereg_kernel; 0 for closed-loop mechanisms (one kernel, runtime layout)
    *links = m->host.loop ? 0 : (m->host.tree ? treereg_layout_links(m->nb, m->host.tree) : chain_layout_links(m->nb));
    return CCLQR_OK;
}

extern "C" int cclqr_rollout_lanes_per_link(const cclqr_mech* m, int32_t* lanes_per_link, int32_t* links_per_group) {
    if (!m) return fail(CCLQR_EINVAL, "null argument");
    int kl = 1, nl = 64;
    if (m->host.loop) nl = 64;
    else if (m->host.tree) nl = treereg_lanes(m->nb, m->host.tree);
    else {
        kl = chain_lanes_per_link(m->nb);
        nl = kl == 1 ? chain_lanes_per_instance(m->nb) : (m->nb <= 2 ? 2 : chain_layout_links(m->nb));
    }
    if (lanes_per_link) *lanes_per_link = kl;
    if (links_per_group) *links_per_group = nl;
    return CCLQR_OK;
}

extern "C" int cclqr_rollout_instances_per_wavefront(const cclqr_mech* m, int64_t n_inst, int32_t steps, int32_t flags, int32_t* instances) {
    if (!m || !instances) return fail(CCLQR_EINVAL, "null argument");
    if (m->host.loop) *instances = 1;
    else if (m->host.tree) *instances = spread_instances_per_wavefront(64 / treereg_lanes(m->nb, m->host.tree), n_inst, steps, (flags & CCLQR_ROLLOUT_PACK_WAVEFRONTS) != 0);
    else *instances = chain_instances_per_wavefront(m->nb, n_inst, steps, (flags & CCLQR_ROLLOUT_PACK_WAVEFRONTS) != 0);
    return CCLQR_OK;
}

extern "C" int cclqr_rollout_geometry(const cclqr_mech* m, int32_t* lanes, int32_t* lds_bytes) {
    if (!m) return fail(CCLQR_EINVAL, "null argument");
    if (m->host.loop) { if (lanes) *lanes = 64; if (lds_bytes) *lds_bytes = (int32_t)loop_lds_bytes(m->nb, m->nj); return CCLQR_OK; }
    if (m->host.tree) {       // branching trees: rollout_treereg.hip
        if (lanes) *lanes = treereg_lanes(m->nb, m->host.tree);
        if (lds_bytes) *lds_bytes = (int32_t)treereg_lds_bytes(m->nb, m->host.tree, m->host.npairs);
        return CCLQR_OK;
    }
    if (lanes) *lanes = chain_lanes_per_instance(m->nb);
    if (lds_bytes) *lds_bytes = (int32_t)chain_lds_bytes(m->nb);
    return CCLQR_OK;
}

extern "C" int cclqr_rollout_ex(const cclqr_mech* m, const cclqr_ctrl* c, int64_t n_inst, int32_t steps, int32_t k0, const double* z0,
                                double* lam, const double* noise, int64_t noise_stride, double* traj, double* zT, int32_t* status,
                                const cclqr_rollout_opts* opts, void* stream) {
    if (m && c && n_inst == 0) return CCLQR_OK;   // empty batch
    if (!m || !c || !z0 || !zT) return fail(CCLQR_EINVAL, "null argument");
    if (n_inst < 0 || steps < 0 || k0 < 1) return fail(CCLQR_EINVAL, "bad sizes");
    if (c->nb != m->nb) return fail(CCLQR_EINVAL, "controller was built for another mechanism");
    { int rc = check_device(m); if (rc != CCLQR_OK) return rc; }
    if (!m->host.loop && (m->host.tree ? treereg_lds_bytes(m->nb, m->host.tree, m->host.npairs) : chain_lds_bytes(m->nb)) > 160 * 1024)
        return fail(CCLQR_EUNSUPPORTED, "instance does not fit LDS");
    const int64_t first = opts ? opts->first_instance : 0;
    if (first < 0) return fail(CCLQR_EINVAL, "negative first_instance");
    if (c->host.n_ctrl > 1 && first + n_inst > c->host.n_ctrl) return fail(CCLQR_EINVAL, "more instances than per-instance controller tables");
    const CtrlDev& H = c->host;
    double* pid_state = (opts && H.has_pid) ? opts->pid_state_dev : nullptr;      // never forwarded to a controller without a PID law
    const int pid_slots = m->host.loop ? m->nj : m->nb;           // one (integrated, last) pair per joint; a tree has as many joints as bodies
    if (pid_state && opts->pid_state_len != n_inst * (int64_t)pid_slots * 2) return fail(CCLQR_EINVAL, "pid_state_len must be n_inst * nb * 2 (closed loops: n_inst * joints * 2)");
    // counter-based noise: generated for this launch into a workspace, read by the rollout like an injected array.  The workspace is the
    // caller's (opts->noise_ws_dev: required for launches that share one controller on different streams or threads) or the handle's,
    // which only ever grows OUTSIDE stream capture: hipMalloc / hipFree are illegal while a stream is being captured, so a captured
    // launch needs the workspace sized beforehand (cclqr_ctrl_reserve_noise) or passed in.
    const bool use_noise = H.noise_scale != 0.0 && H.mu > 0;
    if (opts && (opts->flags & ~(CCLQR_ROLLOUT_NO_ALLOC | CCLQR_ROLLOUT_PACK_WAVEFRONTS | CCLQR_ROLLOUT_CARRY_STATUS))) return fail(CCLQR_EINVAL, "unknown bit in cclqr_rollout_opts.flags");
    if (opts && (opts->flags & CCLQR_ROLLOUT_CARRY_STATUS) && !status) return fail(CCLQR_EINVAL, "CCLQR_ROLLOUT_CARRY_STATUS needs the status array (it is read and written)");
    const bool no_alloc = opts && (opts->flags & CCLQR_ROLLOUT_NO_ALLOC);
    // launches of a few steps on forests of chains (the step-per-launch form a hipGraph replays, BASELINE configs[4]) generate their samples inside the
    // rollout kernel (rollout_chain_kernel<.., 3>): one kernel per step instead of two, and no workspace that could have to grow
    bool philox_in_kernel = false;
    if (use_noise && !noise && H.noise_philox && steps > 0) {
        if (!m->host.loop && !m->host.tree && !H.has_pid && steps <= CCLQR_PHILOX_INKERNEL_STEPS && !(opts && opts->noise_ws_dev)) philox_in_kernel = true;
    }
    if (use_noise && !noise && H.noise_philox && steps > 0 && !philox_in_kernel) {
        const size_t need = (size_t)n_inst * steps;
        double* ws = nullptr;
        if (opts && opts->noise_ws_dev) {
            if (opts->noise_ws_len < (int64_t)need) return fail(CCLQR_EINVAL, "noise_ws_len must be at least n_inst * steps");
            ws = opts->noise_ws_dev;
        } else {
            cclqr_ctrl* cm = const_cast<cclqr_ctrl*>(c);
            if (cm->noise_ws_cap < need) {
                // growing = a device synchronisation + an allocation.  A caller who has said CCLQR_ROLLOUT_NO_ALLOC (anybody with a capture open on
                // this device, on whichever stream) is refused outright; without the flag the library can only see a capture of `stream` itself
                if (no_alloc)
                    return fail(CCLQR_EINVAL, "CCLQR_ROLLOUT_NO_ALLOC: the Philox noise workspace of this controller holds " + std::to_string(cm->noise_ws_cap) + " samples, the launch needs " +
                                              std::to_string(need) + ": call cclqr_ctrl_reserve_noise(ctrl, n_inst, steps) beforehand or pass cclqr_rollout_opts.noise_ws_dev");
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                if (stream) HIPCHK(hipStreamIsCapturing((hipStream_t)stream, &cap));
                if (cap != hipStreamCaptureStatusNone)
                    return fail(CCLQR_EINVAL, "the Philox noise workspace cannot grow during stream capture: call cclqr_ctrl_reserve_noise(ctrl, n_inst, steps) "
                                              "before the capture or pass cclqr_rollout_opts.noise_ws_dev");
                int rc = cclqr_ctrl_reserve_noise(cm, n_inst, steps);
                if (rc != CCLQR_OK) return rc;
            }
            ws = cm->noise_ws;
        }
        HIPCHK(launch_philox_fill(ws, H.noise_key0, first, n_inst, k0, steps, (hipStream_t)stream));
        noise = ws - (k0 - 1);      // indexed by the absolute step k-1
        noise_stride = steps;
    }
    const int extra = H.has_pid ? 2 : (philox_in_kernel ? 3 : ((H.has_fric || (use_noise && noise)) ? 1 : 0));
    RolloutArgs a;
    a.M = m->dev; a.C = c->dev; a.n_inst = n_inst; a.steps = steps; a.k0 = k0; a.z0 = z0; a.lam = lam; a.noise = use_noise ? noise : nullptr;
    a.noise_stride = noise_stride; a.traj = traj; a.zT = zT; a.status = status; a.inst0 = first; a.pid_state = pid_state;
    a.ipw = (opts && (opts->flags & CCLQR_ROLLOUT_PACK_WAVEFRONTS)) ? 1 : 0;
    a.carry = (opts && (opts->flags & CCLQR_ROLLOUT_CARRY_STATUS)) ? 1 : 0;
    const int newton_mode = opts ? opts->newton_mode : 0;
    a.eps_alone = (opts && opts->newton_eps_alone > 0.0) ? opts->newton_eps_alone : 1e-10;
    if (newton_mode != 0 && newton_mode != 1) return fail(CCLQR_EINVAL, "newton_mode must be 0 (exact rule) or 1 (residual-only stop)");
    if (m->host.loop) {      // closed loops: one kernel, every law at run time (LQR / TrackingLQR, friction, noise, PID); newton_mode 1 under any of them
        HIPCHK(launch_rollout_loop(a, m->nb, m->nj, newton_mode, (hipStream_t)stream));
        return CCLQR_OK;
    }
    if (newton_mode != 0 && extra != 0)
        return fail(CCLQR_EUNSUPPORTED, "newton_mode 1 exists under the plain LQR / TrackingLQR law only on chains and branching trees (closed-loop mechanisms: every law)");
    if (m->host.tree) HIPCHK(launch_rollout_treereg(a, m->nb, m->host.tree, m->host.npairs, extra, newton_mode, (hipStream_t)stream));
    else HIPCHK(launch_rollout_chain(a, m->nb, extra, newton_mode, (hipStream_t)stream));
    return CCLQR_OK;
}

extern "C" int cclqr_rollout_dev(const cclqr_mech* m, const cclqr_ctrl* c, int64_t n_inst, int32_t steps, int32_t k0, const double* z0,
                                 double* lam, const double* noise, int64_t noise_stride, double* traj, double* zT, int32_t* status,
                                 void* stream) {
    return cclqr_rollout_ex(m, c, n_inst, steps, k0, z0, lam, noise, noise_stride, traj, zT, status, nullptr, stream);
}

extern "C" int cclqr_ctrl_reserve_noise(cclqr_ctrl* c, int64_t n_inst, int32_t steps) {
    if (!c || n_inst < 0 || steps < 0) return fail(CCLQR_EINVAL, "bad argument");
    const size_t need = (size_t)n_inst * steps;
    if (c->noise_ws_cap >= need) return CCLQR_OK;
    {   // the block lives on the handle's device: growing it from a thread that is on another one would drain and allocate there
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return fail(CCLQR_EHIP, "hipGetDevice failed");
        if (dev != c->device) return fail(CCLQR_EINVAL, "the controller was created on device " + std::to_string(c->device) + ", the calling thread is on device " + std::to_string(dev));
    }
    // a launch that still reads the old block may be in flight on some stream: the device is drained before the block is replaced
    HIPCHK(hipDeviceSynchronize());
    if (c->noise_ws) HIPCHK(hipFree(c->noise_ws));
    c->noise_ws = nullptr; c->noise_ws_cap = 0;
    HIPCHK(hipMalloc((void**)&c->noise_ws, (need ? need : 1) * sizeof(double)));
    c->noise_ws_cap = need;
    return CCLQR_OK;
}

extern "C" int cclqr_rollout_host_ex(const cclqr_mech* m, const cclqr_ctrl* c, int64_t n_inst, int32_t steps, int32_t k0, const double* z0,
                                     const double* noise, double* traj, double* zT, int32_t* status, const cclqr_rollout_opts* opts) {
    if (m && c && n_inst == 0) return CCLQR_OK;   // empty batch
    if (!m || !c || !z0 || !zT) return fail(CCLQR_EINVAL, "null argument");
    if (opts && (opts->pid_state_dev || opts->noise_ws_dev)) return fail(CCLQR_EINVAL, "the host-pointer rollout takes no device buffers in its options");
    { int rc = check_device(m); if (rc != CCLQR_OK) return rc; }
    const size_t nz = (size_t)13 * m->nb;
    double *dz0 = nullptr, *dzT = nullptr, *dtraj = nullptr, *dnoise = nullptr;
    int32_t* dst = nullptr;
    int rc = CCLQR_OK;
    WsScope scope;
    hipError_t e = ws_get((void**)&dz0, n_inst * nz * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dzT, n_inst * nz * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dst, n_inst * sizeof(int32_t));
    if (e == hipSuccess && traj) e = ws_get((void**)&dtraj, n_inst * steps * nz * sizeof(double));
    if (e == hipSuccess && noise) e = ws_get((void**)&dnoise, (size_t)n_inst * steps * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(dz0, z0, n_inst * nz * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && noise) e = hipMemcpy(dnoise, noise, (size_t)n_inst * steps * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        // noise is indexed by the absolute step k-1: shift the base so that k0 maps to column 0 of the caller's array
        const double* nbase = dnoise ? dnoise - (k0 - 1) : nullptr;
        rc = cclqr_rollout_ex(m, c, n_inst, steps, k0, dz0, nullptr, nbase, steps, dtraj, dzT, dst, opts, nullptr);
        if (rc == CCLQR_OK) e = hipDeviceSynchronize();
    }
    if (rc == CCLQR_OK && e == hipSuccess) e = hipMemcpy(zT, dzT, n_inst * nz * sizeof(double), hipMemcpyDeviceToHost);
    if (rc == CCLQR_OK && e == hipSuccess && traj) e = hipMemcpy(traj, dtraj, n_inst * steps * nz * sizeof(double), hipMemcpyDeviceToHost);
    if (rc == CCLQR_OK && e == hipSuccess && status) e = hipMemcpy(status, dst, n_inst * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (rc != CCLQR_OK) return rc;
    if (e != hipSuccess) return fail(CCLQR_EHIP, std::string("rollout: ") + hipGetErrorString(e));
    return CCLQR_OK;
}

extern "C" int cclqr_rollout(const cclqr_mech* m, const cclqr_ctrl* c, int64_t n_inst, int32_t steps, int32_t k0, const double* z0,
                             const double* noise, double* traj, double* zT, int32_t* status) {
    return cclqr_rollout_host_ex(m, c, n_inst, steps, k0, z0, noise, traj, zT, status, nullptr);
}

extern "C" int cclqr_linearize(const cclqr_mech* m, int32_t nk, const double* zd, int32_t mu, const int32_t* ctrl_joint, const double* Fd,
                               double* A, double* Bu, double* Bl, double* G) {
    if (!m || !zd || !A || !Bl || !G || (mu > 0 && (!ctrl_joint || !Bu))) return fail(CCLQR_EINVAL, "null argument");
    { int rc = check_device(m); if (rc != CCLQR_OK) return rc; }
    // a closed-loop mechanism has its own tables (bodies and joints in the caller's order, ml = 5 rows per joint incl. the two null rows of a
    // FixedOrientation); its G*Bl is singular, so the recursion takes the projected pair of cclqr_linearize_projected, not these four
    const bool loop = m->host.loop != 0;
    const int nb = m->nb, nj = loop ? m->nj : nb;
    if (nk < 0 || mu < 0 || mu > nj) return fail(CCLQR_EINVAL, "Missmatched length for constraints");
    if (nk == 0) return CCLQR_OK;
    const size_t nz = 13 * (size_t)nb, mx = 12 * (size_t)nb, ml = 5 * (size_t)nj;
    if (!loop && linearize_lds_bytes(nb, m->host.tree, m->host.npairs) > 160 * 1024) return fail(CCLQR_EUNSUPPORTED, "instance does not fit LDS");
    LinArgs a;
    memset(&a, 0, sizeof(a));
    a.M = m->dev; a.nk = nk; a.mu = mu;
    for (int i = 0; i < mu; i++) {
        if (ctrl_joint[i] < 0 || ctrl_joint[i] >= nj) return fail(CCLQR_EINVAL, "controlled joint out of range");
        a.cj[i] = loop ? ctrl_joint[i] : m->link_of_joint[ctrl_joint[i]];
    }
    double *dzd = nullptr, *dFd = nullptr, *dA = nullptr, *dBu = nullptr, *dBl = nullptr, *dG = nullptr;
    int* dst = nullptr;
    std::vector<int> st(nk);
    WsScope scope;
    hipError_t e = ws_get((void**)&dzd, nk * nz * sizeof(double));
    if (e == hipSuccess && Fd && mu > 0) e = ws_get((void**)&dFd, (size_t)nk * mu * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dA, nk * mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBu, (nk * mx * (size_t)(mu > 0 ? mu : 1)) * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBl, nk * mx * ml * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dG, nk * ml * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dst, nk * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(dzd, zd, nk * nz * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && dFd) e = hipMemcpy(dFd, Fd, (size_t)nk * mu * sizeof(double), hipMemcpyHostToDevice);
    a.zd = dzd; a.Fd = dFd; a.A = dA; a.Bu = dBu; a.Bl = dBl; a.G = dG; a.status = dst;
    if (e == hipSuccess) e = loop ? launch_linearize_loop(a, nb, nj, nullptr) : launch_linearize(a, nb, m->host.tree, m->host.npairs, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(A, dA, nk * mx * mx * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && mu > 0) e = hipMemcpy(Bu, dBu, nk * mx * mu * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(Bl, dBl, nk * mx * ml * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(G, dG, nk * ml * mx * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(st.data(), dst, nk * sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(CCLQR_EHIP, std::string("linearize: ") + hipGetErrorString(e));
    for (int k = 0; k < nk; k++)
        if (st[k] <= 0) return fail(CCLQR_ENOCONV, "Newton did not converge at the setpoint of knot " + std::to_string(k));
    return CCLQR_OK;
}

// ---- projected linear model by central differences of the DEVICE step map (any topology; the only linearisation of closed loops)
static inline void h_qmul(const double* a, const double* b, double* o) {
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
// difference quotients of the perturbed single-step results, on the device: thread = (knot, column, body).  Error coordinates of a
// state about the nominal next state, per body x, v, q~ = vec(q0^-1 q), w  (lqr.jl:92-103)
__global__ void fd_quotient_kernel(const double* zT, int nk, int per, int nb, int mu, double h, double* Ap, double* D) {
    const int mx = 12 * nb, ncol = mx + mu;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)nk * ncol * nb) return;
    const int b = (int)(id % nb), col = (int)((id / nb) % ncol), k = (int)(id / ((long long)nb * ncol));
    const size_t nz = 13 * (size_t)nb;
    const double* z0 = zT + ((size_t)k * per) * nz + 13 * b;
    const double* zp = zT + ((size_t)k * per + 1 + 2 * col) * nz + 13 * b;
    const double* zm = zp + nz;
    const double qc[4] = {z0[3], -z0[4], -z0[5], -z0[6]};
    double qp[4], qm[4], e[12];
    qmul(qc, zp + 3, qp);
    qmul(qc, zm + 3, qm);
    for (int i = 0; i < 3; i++) {
        e[i] = zp[i] - zm[i]; e[3 + i] = zp[7 + i] - zm[7 + i]; e[6 + i] = qp[1 + i] - qm[1 + i]; e[9 + i] = zp[10 + i] - zm[10 + i];
    }
    for (int i = 0; i < 12; i++) {
        const double v = e[i] / (2.0 * h);
        const int r = 12 * b + i;
        if (col < mx) Ap[((size_t)k * mx + r) * mx + col] = v;
        else D[((size_t)k * mx + r) * mu + (col - mx)] = v;
    }
}
// The projected pair ANALYTICALLY (h <= 0): the exact Jacobians of the one-step map with the multipliers exogenous -- linearize_kernel for
// trees, linearize_loop_kernel (cclqr_lin_loop.h) for closed loops, nothing leaves the device -- and then the multipliers eliminated by
// project_model_kernel, whose complete pivoting stops at the numerical rank of G Bl (singular for a loop: redundant constraint rows).
static int linearize_projected_analytic(const cclqr_mech* m, int32_t nk, const double* zd, int32_t mu, const int32_t* ctrl_joint, const double* Fd,
                                        double* Ap, double* D) {
    const int nb = m->nb, nj = m->host.loop ? m->nj : nb, mx = 12 * nb, ml = 5 * nj;
    const size_t nz = 13 * (size_t)nb;
    if (!m->host.loop && linearize_lds_bytes(nb, m->host.tree, m->host.npairs) > 160 * 1024) return fail(CCLQR_EUNSUPPORTED, "instance does not fit LDS");
    if (!project_model_fits(mx, mu, ml)) return fail(CCLQR_EUNSUPPORTED, "the projection of this model does not fit LDS (use h > 0)");
    LinArgs a;
    memset(&a, 0, sizeof(a));
    a.M = m->dev; a.nk = nk; a.mu = mu;
    for (int i = 0; i < mu; i++) {
        if (ctrl_joint[i] < 0 || ctrl_joint[i] >= nj) return fail(CCLQR_EINVAL, "controlled joint out of range");
        a.cj[i] = m->host.loop ? ctrl_joint[i] : m->link_of_joint[ctrl_joint[i]];      // closed-loop tables keep the caller's joint order
    }
    const size_t mu1 = (size_t)(mu > 0 ? mu : 1);
    double *dzd = nullptr, *dFd = nullptr, *dA = nullptr, *dBu = nullptr, *dBl = nullptr, *dG = nullptr, *dAp = nullptr, *dD = nullptr, *dres = nullptr;
    int *dst = nullptr, *drank = nullptr;
    std::vector<int> st(nk), rank(nk);
    std::vector<double> res(nk);
    WsScope scope;
    hipError_t e = ws_get((void**)&dzd, nk * nz * sizeof(double));
    if (e == hipSuccess && Fd && mu > 0) e = ws_get((void**)&dFd, (size_t)nk * mu * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dA, (size_t)nk * mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBu, (size_t)nk * mx * mu1 * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBl, (size_t)nk * mx * ml * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dG, (size_t)nk * ml * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dAp, (size_t)nk * mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dD, (size_t)nk * mx * mu1 * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dres, (size_t)nk * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dst, (size_t)nk * sizeof(int));
    if (e == hipSuccess) e = ws_get((void**)&drank, (size_t)nk * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(dzd, zd, nk * nz * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && dFd) e = hipMemcpy(dFd, Fd, (size_t)nk * mu * sizeof(double), hipMemcpyHostToDevice);
    a.zd = dzd; a.Fd = dFd; a.A = dA; a.Bu = dBu; a.Bl = dBl; a.G = dG; a.status = dst;
    if (e == hipSuccess) e = m->host.loop ? launch_linearize_loop(a, nb, nj, nullptr) : launch_linearize(a, nb, m->host.tree, m->host.npairs, nullptr);
    if (e == hipSuccess) e = launch_project_model(nk, mx, mu, ml, dA, dBu, dBl, dG, dAp, dD, dres, drank, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(Ap, dAp, (size_t)nk * mx * mx * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && mu > 0) e = hipMemcpy(D, dD, (size_t)nk * mx * mu * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(st.data(), dst, nk * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(res.data(), dres, nk * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(CCLQR_EHIP, std::string("linearize_projected: ") + hipGetErrorString(e));
    for (int k = 0; k < nk; k++) {
        if (st[k] <= 0) return fail(CCLQR_ENOCONV, "Newton did not converge at the setpoint of knot " + std::to_string(k));
        if (!(res[k] < 1e-6)) return fail(CCLQR_ESINGULAR, "the constraint rows of the linear model are inconsistent at knot " + std::to_string(k) + " (G Bl rank-deficient beyond redundancy)");
    }
    return CCLQR_OK;
}

extern "C" int cclqr_linearize_projected(const cclqr_mech* m, int32_t nk, const double* zd, int32_t mu, const int32_t* ctrl_joint, const double* Fd,
                                         double h, double* Ap, double* D) {
    if (!m || !zd || !Ap || (mu > 0 && (!ctrl_joint || !D))) return fail(CCLQR_EINVAL, "null argument");
    { int rc = check_device(m); if (rc != CCLQR_OK) return rc; }
    if (nk < 0 || mu < 0 || mu > m->nj) return fail(CCLQR_EINVAL, "Missmatched length for constraints");
    if (nk == 0) return CCLQR_OK;
    const int nb = m->nb, mx = 12 * nb;
    if (!(h > 0.0)) {
        // the analytic projection keeps [G Bl | G A | G Bu] of a knot in one CU's LDS (project_model_kernel): 175 KB for a 16-body tree, 198 KB for
        // the 17-body headline chain.  What does not fit is differenced instead (the h > 0 form with its documented step), so that the default
        // call is defined for every mechanism cclqr_mech_create takes
        const int nj = m->host.loop ? m->nj : nb;
        const bool fits = project_model_fits(mx, mu, 5 * nj) && (m->host.loop || linearize_lds_bytes(nb, m->host.tree, m->host.npairs) <= 160 * 1024);
        if (fits) return linearize_projected_analytic(m, nk, zd, mu, ctrl_joint, Fd, Ap, D);
        h = 1e-6;
    }
    const size_t nz = 13 * (size_t)nb;
    const int per = 1 + 2 * mx + 2 * mu;            // nominal, +-h in every state error coordinate, +-h in every input
    const size_t n = (size_t)nk * per;
    std::vector<double> z0(n * nz), fd(n * (size_t)(mu > 0 ? mu : 1), 0.0), zdum(n * nz, 0.0);
    std::vector<int32_t> st(n);
    for (size_t i = 0; i < n; i++)
        for (int b = 0; b < nb; b++) zdum[i * nz + 13 * b + 3] = 1.0;
    for (int k = 0; k < nk; k++) {
        const double* zk = zd + (size_t)k * nz;
        for (int q = 0; q < per; q++) {
            double* z = &z0[((size_t)k * per + q) * nz];
            memcpy(z, zk, nz * sizeof(double));
            for (int i = 0; i < mu; i++) fd[((size_t)k * per + q) * mu + i] = Fd ? Fd[(size_t)k * mu + i] : 0.0;
            if (q >= 1 && q <= 2 * mx) {
                const int col = (q - 1) >> 1, b = col / 12, e = col % 12;
                const double s = ((q - 1) & 1) ? -h : h;
                double* p = z + 13 * b;
                if (e < 3) p[e] += s;
                else if (e < 6) p[7 + e - 3] += s;
                else if (e < 9) {                       // q = qd (sqrt(1 - s^2), s e_i): vec(qd^-1 q) = s e_i exactly
                    double dq[4] = {sqrt(1.0 - s * s), 0.0, 0.0, 0.0}, qn[4];
                    dq[1 + e - 6] = s;
                    h_qmul(zk + 13 * b + 3, dq, qn);
                    for (int i = 0; i < 4; i++) p[3 + i] = qn[i];
                } else p[10 + e - 9] += s;
            } else if (q > 2 * mx) {
                const int i = (q - 1 - 2 * mx) >> 1;
                fd[((size_t)k * per + q) * mu + i] += ((q - 1) & 1) ? -h : h;
            }
        }
    }
    cclqr_ctrl_desc cd;
    memset(&cd, 0, sizeof(cd));
    cd.mu = mu; cd.ctrl_joint = ctrl_joint; cd.nK = 0; cd.N = 0; cd.K = nullptr; cd.nsp = 1; cd.zd = zdum.data(); cd.Fd = mu > 0 ? fd.data() : nullptr;
    cd.n_ctrl = (int32_t)n;
    cclqr_ctrl* c = nullptr;
    int rc = cclqr_ctrl_create(m, &cd, &c);
    if (rc != CCLQR_OK) return rc;
    double *dz0 = nullptr, *dzT = nullptr, *dAp = nullptr, *dD = nullptr;
    int32_t* dst = nullptr;
    WsScope scope;
    hipError_t e = ws_get((void**)&dz0, n * nz * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dzT, n * nz * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dst, n * sizeof(int32_t));
    if (e == hipSuccess) e = ws_get((void**)&dAp, (size_t)nk * mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dD, (size_t)nk * mx * (size_t)(mu > 0 ? mu : 1) * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(dz0, z0.data(), n * nz * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = cclqr_rollout_ex(m, c, (int64_t)n, 1, 1, dz0, nullptr, nullptr, 0, nullptr, dzT, dst, nullptr, nullptr);
    }
    if (rc == CCLQR_OK && e == hipSuccess) {
        const long long work = (long long)nk * (mx + mu) * nb;
        hipLaunchKernelGGL(fd_quotient_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, nullptr, dzT, nk, per, nb, mu, h, dAp, dD);
        e = hipGetLastError();
    }
    if (rc == CCLQR_OK && e == hipSuccess) e = hipDeviceSynchronize();
    if (rc == CCLQR_OK && e == hipSuccess) e = hipMemcpy(st.data(), dst, n * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (rc == CCLQR_OK && e == hipSuccess) e = hipMemcpy(Ap, dAp, (size_t)nk * mx * mx * sizeof(double), hipMemcpyDeviceToHost);
    if (rc == CCLQR_OK && e == hipSuccess && mu > 0) e = hipMemcpy(D, dD, (size_t)nk * mx * mu * sizeof(double), hipMemcpyDeviceToHost);
    cclqr_ctrl_destroy(c);
    if (rc != CCLQR_OK) return rc;
    if (e != hipSuccess) return fail(CCLQR_EHIP, std::string("linearize_projected: ") + hipGetErrorString(e));
    for (size_t i = 0; i < n; i++)
        if (st[i] <= 0) return fail(CCLQR_ENOCONV, "Newton did not converge at a perturbed setpoint of knot " + std::to_string(i / per));
    return CCLQR_OK;
}

// shared tail of the two dlqr entry points: run the recursion on device-resident (A,Bu,Bl,G), download K and kbreak
static int run_riccati(int nprob, int mx, int mu, int ml, int N, int time_varying, double tol, const double* dA, const double* dBu,
                       const double* dBl, const double* dG, const double* Q, const double* R, double* K, int32_t* kbreak,
                       const cclqr_riccati_opts* opts) {
    const int keep_last = (opts && opts->keep_last) ? 1 : 0;
    const size_t nK = (size_t)nprob * (N > 1 ? (keep_last ? 1 : N - 1) : 0) * mu * mx;
    double *dQ = nullptr, *dR = nullptr, *dK = nullptr, *dwork = nullptr;
    int *dkb = nullptr, *dst = nullptr, *dstop = nullptr;
    std::vector<int> st(nprob), kb(nprob);
    RicArgs a;
    a.nprob = nprob; a.mx = mx; a.mu = mu; a.ml = ml; a.N = N; a.time_varying = time_varying; a.tol = tol;
    a.path = opts ? opts->path : 0;
    a.bf16_terms = opts ? opts->bf16_terms : 0;
    a.keep_last = keep_last;
    if (a.path < 0 || a.path > 2 || a.bf16_terms < 0 || a.bf16_terms > 3) return fail(CCLQR_EINVAL, "riccati options: path in 0..2, bf16_terms in 0..3");
    const size_t wd = ric_total_work_doubles(a);
    hipError_t e = ws_get((void**)&dQ, (size_t)mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dR, (size_t)(mu * mu + 1) * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dK, (nK + 1) * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dwork, wd * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dstop, nprob * sizeof(int));
    if (e == hipSuccess) e = ws_get((void**)&dkb, nprob * sizeof(int));
    if (e == hipSuccess) e = ws_get((void**)&dst, nprob * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(dQ, Q, (size_t)mx * mx * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && mu > 0) e = hipMemcpy(dR, R, (size_t)mu * mu * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dK, 0, (nK + 1) * sizeof(double));
    a.stop = dstop;
    a.A = dA; a.Bu = dBu; a.Bl = dBl; a.G = dG; a.Q = dQ; a.R = dR; a.K = dK; a.kbreak = dkb; a.status = dst; a.work = dwork;
    if (e == hipSuccess) e = launch_riccati(a, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && nK) e = hipMemcpy(K, dK, nK * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(kb.data(), dkb, nprob * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(st.data(), dst, nprob * sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(CCLQR_EHIP, std::string("riccati: ") + hipGetErrorString(e));
    for (int p = 0; p < nprob; p++) {
        if (kbreak) kbreak[p] = kb[p];
        if (st[p] != 0) return fail(CCLQR_ESINGULAR, "G*Bl or M is singular in problem " + std::to_string(p));
    }
    return CCLQR_OK;
}

extern "C" int cclqr_riccati(int32_t nprob, int32_t mx, int32_t mu, int32_t ml, const double* A, const double* Bu, const double* Bl,
                             const double* G, const double* Q, const double* R, int32_t N, double tol, double* K, int32_t* kbreak) {
    return cclqr_riccati_ex(nprob, mx, mu, ml, A, Bu, Bl, G, Q, R, N, tol, K, kbreak, nullptr);
}

extern "C" int cclqr_riccati_ex(int32_t nprob, int32_t mx, int32_t mu, int32_t ml, const double* A, const double* Bu, const double* Bl,
                                const double* G, const double* Q, const double* R, int32_t N, double tol, double* K, int32_t* kbreak,
                                const cclqr_riccati_opts* opts) {
    if (!A || !Q || (mu > 0 && (!Bu || !R)) || (ml > 0 && (!Bl || !G)) || !K) return fail(CCLQR_EINVAL, "null argument");
    if (nprob < 1 || mx < 1 || mu < 0 || ml < 0 || N < 1) return fail(CCLQR_EINVAL, "bad sizes");
    double *dA = nullptr, *dBu = nullptr, *dBl = nullptr, *dG = nullptr;
    const size_t np = nprob;
    WsScope scope;
    hipError_t e = ws_get((void**)&dA, np * mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBu, (np * mx * mu + 1) * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBl, (np * mx * ml + 1) * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dG, (np * ml * mx + 1) * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(dA, A, np * mx * mx * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && mu > 0) e = hipMemcpy(dBu, Bu, np * mx * mu * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && ml > 0) e = hipMemcpy(dBl, Bl, np * mx * ml * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && ml > 0) e = hipMemcpy(dG, G, np * ml * mx * sizeof(double), hipMemcpyHostToDevice);
    int rc = CCLQR_OK;
    if (e == hipSuccess) rc = run_riccati(nprob, mx, mu, ml, N, 0, tol, dA, dBu, dBl, dG, Q, R, K, kbreak, opts);
    if (e != hipSuccess) return fail(CCLQR_EHIP, std::string("riccati upload: ") + hipGetErrorString(e));
    return rc;
}

extern "C" int cclqr_riccati_tv(int32_t mx, int32_t mu, int32_t ml, const double* A, const double* Bu, const double* Bl, const double* G,
                                const double* Q, const double* R, int32_t N, double tol, double* K, int32_t* kbreak) {
    if (!A || !Q || (mu > 0 && (!Bu || !R)) || (ml > 0 && (!Bl || !G)) || !K) return fail(CCLQR_EINVAL, "null argument");
    if (mx < 1 || mu < 0 || ml < 0 || N < 2) return fail(CCLQR_EINVAL, "bad sizes");
    double *dA = nullptr, *dBu = nullptr, *dBl = nullptr, *dG = nullptr;
    const size_t nk = (size_t)N - 1;
    WsScope scope;
    hipError_t e = ws_get((void**)&dA, nk * mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBu, (nk * mx * mu + 1) * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBl, (nk * mx * ml + 1) * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dG, (nk * ml * mx + 1) * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(dA, A, nk * mx * mx * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && mu > 0) e = hipMemcpy(dBu, Bu, nk * mx * mu * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && ml > 0) e = hipMemcpy(dBl, Bl, nk * mx * ml * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && ml > 0) e = hipMemcpy(dG, G, nk * ml * mx * sizeof(double), hipMemcpyHostToDevice);
    int rc = CCLQR_OK;
    if (e == hipSuccess) rc = run_riccati(1, mx, mu, ml, N, 1, tol, dA, dBu, dBl, dG, Q, R, K, kbreak, nullptr);
    if (e != hipSuccess) return fail(CCLQR_EHIP, std::string("riccati upload: ") + hipGetErrorString(e));
    return rc;
}

extern "C" int cclqr_riccati_tracking(const cclqr_mech* m, int32_t mu, const int32_t* ctrl_joint, const double* zd, const double* Fd,
                                      const double* Q, const double* R, int32_t N, double tol, double* K, int32_t* kbreak) {
    return cclqr_riccati_tracking_ex(m, mu, ctrl_joint, zd, Fd, Q, R, N, tol, K, kbreak, nullptr);
}

extern "C" int cclqr_riccati_tracking_ex(const cclqr_mech* m, int32_t mu, const int32_t* ctrl_joint, const double* zd, const double* Fd,
                                         const double* Q, const double* R, int32_t N, double tol, double* K, int32_t* kbreak,
                                         const cclqr_riccati_opts* opts) {
    if (!m || !zd || !Q || !K || (mu > 0 && (!ctrl_joint || !R))) return fail(CCLQR_EINVAL, "null argument");
    if (m->host.loop) return fail(CCLQR_EUNSUPPORTED, "TrackingLQR of a closed-loop mechanism is outside this build's scope");
    if (N < 2 || mu < 0 || mu > m->nb) return fail(CCLQR_EINVAL, "bad sizes");
    const int nb = m->nb, nk = N - 1;
    const size_t nz = 13 * (size_t)nb, mx = 12 * (size_t)nb, ml = 5 * (size_t)nb;
    LinArgs a;
    memset(&a, 0, sizeof(a));
    a.M = m->dev; a.nk = nk; a.mu = mu;
    for (int i = 0; i < mu; i++) {
        if (ctrl_joint[i] < 0 || ctrl_joint[i] >= nb) return fail(CCLQR_EINVAL, "controlled joint out of range");
        a.cj[i] = m->link_of_joint[ctrl_joint[i]];
    }
    double *dzd = nullptr, *dFd = nullptr, *dA = nullptr, *dBu = nullptr, *dBl = nullptr, *dG = nullptr;
    int* dst = nullptr;
    std::vector<int> st(nk);
    // knots 1..N-1 (lqr_tracking.jl:87-88): linearise all of them in one launch, keep the matrices on the device
    WsScope scope;
    hipError_t e = ws_get((void**)&dzd, nk * nz * sizeof(double));
    if (e == hipSuccess && Fd && mu > 0) e = ws_get((void**)&dFd, (size_t)nk * mu * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dA, nk * mx * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBu, (nk * mx * (size_t)(mu > 0 ? mu : 1)) * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dBl, nk * mx * ml * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dG, nk * ml * mx * sizeof(double));
    if (e == hipSuccess) e = ws_get((void**)&dst, nk * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(dzd, zd, nk * nz * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && dFd) e = hipMemcpy(dFd, Fd, (size_t)nk * mu * sizeof(double), hipMemcpyHostToDevice);
    a.zd = dzd; a.Fd = dFd; a.A = dA; a.Bu = dBu; a.Bl = dBl; a.G = dG; a.status = dst;
    if (e == hipSuccess) e = launch_linearize(a, nb, m->host.tree, m->host.npairs, nullptr);
    if (e == hipSuccess) e = hipMemcpy(st.data(), dst, nk * sizeof(int), hipMemcpyDeviceToHost);
    int rc = CCLQR_OK;
    if (e == hipSuccess) {
        for (int k = 0; k < nk && rc == CCLQR_OK; k++)
            if (st[k] <= 0) rc = fail(CCLQR_ENOCONV, "Newton did not converge at the setpoint of knot " + std::to_string(k));
        if (rc == CCLQR_OK) rc = run_riccati(1, (int)mx, mu, (int)ml, N, 1, tol, dA, dBu, dBl, dG, Q, R, K, kbreak, opts);
    }
    if (e != hipSuccess) return fail(CCLQR_EHIP, std::string("riccati_tracking: ") + hipGetErrorString(e));
    return rc;
}
