"""ctypes binding of libcclqr.so (include/cclqr.h).  There is no fallback: if the HIP library is missing or a
call fails, this raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcclqr.so")
if os.environ.get("CCLQR_LIB_VARIANT", "base") != "base":      # experiment builds of tools/gpu_*_ab.sh (make variant / ricvariant): never set by the product
    LIB_PATH = os.path.join(_HERE, "libcclqr_%s.so" % os.environ["CCLQR_LIB_VARIANT"])
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

OK, EINVAL, ESINGULAR, ENOCONV, EHIP, EUNSUPPORTED = 0, -1, -2, -3, -4, -5
_ERRNAME = {EINVAL: "CCLQR_EINVAL", ESINGULAR: "CCLQR_ESINGULAR", ENOCONV: "CCLQR_ENOCONV", EHIP: "CCLQR_EHIP", EUNSUPPORTED: "CCLQR_EUNSUPPORTED"}

EXPORTS = ["cclqr_last_error", "cclqr_version", "cclqr_device_count", "cclqr_set_device", "cclqr_mech_create", "cclqr_mech_destroy",
           "cclqr_ctrl_create", "cclqr_ctrl_create_lqr_batch", "cclqr_ctrl_destroy", "cclqr_linearize", "cclqr_linearize_projected", "cclqr_riccati", "cclqr_riccati_tv", "cclqr_riccati_tracking", "cclqr_rollout",
           "cclqr_rollout_dev", "cclqr_rollout_ex", "cclqr_rollout_host_ex", "cclqr_ctrl_reserve_noise", "cclqr_riccati_ex", "cclqr_riccati_tracking_ex",
           "cclqr_release_workspaces", "cclqr_rollout_geometry", "cclqr_rollout_layout_links", "cclqr_ctrl_set_feedforward", "cclqr_abi_layout", "cclqr_rollout_lanes_per_link", "cclqr_rollout_instances_per_wavefront"]
ABI_VERSION = 201     # include/cclqr.h CCLQR_ABI_VERSION: the structs below mirror that header (verified field by field against cclqr_abi_layout at load time)
ROLLOUT_NO_ALLOC = 1  # cclqr_rollout_opts.flags: the call may neither allocate nor synchronise (a hipGraph capture is open on the device)
ROLLOUT_CARRY_STATUS = 4  # ... `status` is read and written: an instance lost in an earlier launch stays frozen, the others merge this launch's result into it
ROLLOUT_PACK_WAVEFRONTS = 2  # ... every wavefront of a chain launch full, whatever the batch size (many launches sharing the device at once)
PHILOX_INKERNEL_STEPS = 8
NEWTON_MAXIT = 100      # newtonIter of ConstrainedDynamics' newton!: |status| of an instance that hit the cap


class CclqrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (_ERRNAME.get(code, "?"), code, msg))
        self.code = code


class MechDesc(C.Structure):
    _fields_ = [("nb", C.c_int32), ("ne", C.c_int32), ("dt", C.c_double), ("g", C.c_double),
                ("mass", _dp), ("inertia", _dp), ("parent", _ip), ("child", _ip), ("type", _ip),
                ("p1", _dp), ("p2", _dp), ("axis", _dp), ("qoff", _dp)]


class CtrlDesc(C.Structure):
    _fields_ = [("mu", C.c_int32), ("ctrl_joint", _ip), ("nK", C.c_int32), ("N", C.c_int32), ("K", _dp),
                ("nsp", C.c_int32), ("zd", _dp), ("Fd", _dp), ("fric", _dp), ("noise_scale", C.c_double),
                ("npid", C.c_int32), ("pid_joint", _ip), ("pid_P", _dp), ("pid_I", _dp), ("pid_D", _dp), ("pid_goal", _dp),
                ("noise_philox", C.c_int32), ("noise_seed", C.c_uint64), ("n_ctrl", C.c_int32)]


class RiccatiOpts(C.Structure):
    _fields_ = [("path", C.c_int32), ("bf16_terms", C.c_int32), ("keep_last", C.c_int32), ("reserved", C.c_int32)]


class RolloutOpts(C.Structure):
    _fields_ = [("first_instance", C.c_int64), ("pid_state_dev", C.c_void_p), ("pid_state_len", C.c_int64),
                ("noise_ws_dev", C.c_void_p), ("noise_ws_len", C.c_int64), ("newton_mode", C.c_int32), ("flags", C.c_int32),
                ("newton_eps_alone", C.c_double)]


_lib = None


def mirrored_layout():
    """sizeof / offsetof of the four ctypes mirrors in the order cclqr_abi_layout reports the library's own (include/cclqr.h)"""
    out = []
    for S in (MechDesc, CtrlDesc, RiccatiOpts, RolloutOpts):
        out.append(C.sizeof(S))
        out += [getattr(S, name).offset for name, _ in S._fields_]
    return out


def _preload_shared_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7) and links it by the name
    `libamdhip64.so`; libcclqr.so needs `libamdhip64.so.7`.  If libcclqr.so is loaded first the system copy comes in, torch later
    adds its bundled copy, and the second runtime reports `no ROCm-capable device`.  Loading torch's copy first (when torch is
    installed; torch itself is NOT imported) makes both resolve to the same runtime whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """load libcclqr.so; raises if it has not been built (`python -c 'import __graft_entry__ as g; g.build()'`)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libcclqr.so is missing at %s: the HIP extension must be built (no CPU fallback exists)" % LIB_PATH)
        _preload_shared_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.cclqr_last_error.restype = C.c_char_p
        L.cclqr_version.restype = C.c_int
        if L.cclqr_version() != ABI_VERSION:      # before any other symbol is touched: an older library lacks the newer exports
            raise ImportError("libcclqr.so has ABI version %d, this binding was written for %d: rebuild the library" % (L.cclqr_version(), ABI_VERSION))
        for name in EXPORTS[1:]:
            try:
                getattr(L, name).restype = C.c_int
            except AttributeError:
                raise ImportError("libcclqr.so (ABI version %d) does not export %s, which include/cclqr.h declares: rebuild the library" % (L.cclqr_version(), name))
        want = mirrored_layout()
        got = (C.c_int32 * len(want))()
        n = L.cclqr_abi_layout(got, C.c_int32(len(want)))
        if n != len(want) or list(got) != want:
            raise ImportError("struct layout mismatch between libcclqr.so and this binding (cclqr_abi_layout: %s, ctypes mirrors: %s)" % (list(got), want))
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise CclqrError(rc, lib().cclqr_last_error().decode())


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def device_count():
    n = C.c_int32(0)
    check(lib().cclqr_device_count(C.byref(n)))
    return n.value


def set_device(dev):
    check(lib().cclqr_set_device(C.c_int32(dev)))


class MechHandle:
    """cclqr_mech*: device-resident mechanism tables"""

    def __init__(self, tables):
        t = tables
        self.tables = t
        self._arrs = [f64(t.mass), f64(t.inertia), i32(t.parent), i32(t.child), i32(t.type), f64(t.p1), f64(t.p2), f64(t.axis), f64(t.qoff)]
        a = self._arrs
        self.desc = MechDesc(t.nb, t.ne, t.dt, t.g, _d(a[0]), _d(a[1]), _i(a[2]), _i(a[3]), _i(a[4]), _d(a[5]), _d(a[6]), _d(a[7]), _d(a[8]))
        self.ptr = C.c_void_p()
        check(lib().cclqr_mech_create(C.byref(self.desc), C.byref(self.ptr)))

    def geometry(self):
        lanes, ldsb = C.c_int32(0), C.c_int32(0)
        check(lib().cclqr_rollout_geometry(self.ptr, C.byref(lanes), C.byref(ldsb)))
        return lanes.value, ldsb.value

    def layout_links(self):
        """links the rollout kernel's LDS image is laid out for (names the instantiation of the chain / tree kernel); 0 for closed-loop mechanisms"""
        n = C.c_int32(0)
        check(lib().cclqr_rollout_layout_links(self.ptr, C.byref(n)))
        return n.value

    def lanes_per_link(self):
        """(lanes that work for one link, links per sub-lane group) of the chain kernel's instantiation; (1, lanes per instance) elsewhere"""
        kl, nl = C.c_int32(0), C.c_int32(0)
        check(lib().cclqr_rollout_lanes_per_link(self.ptr, C.byref(kl), C.byref(nl)))
        return kl.value, nl.value

    def instances_per_wavefront(self, n_inst, steps, flags=0):
        """instances one wavefront of such a launch holds (chains: a small batch is spread over more wavefronts unless ROLLOUT_PACK_WAVEFRONTS)"""
        v = C.c_int32(0)
        check(lib().cclqr_rollout_instances_per_wavefront(self.ptr, C.c_int64(int(n_inst)), C.c_int32(int(steps)), C.c_int32(int(flags)), C.byref(v)))
        return v.value

    def close(self):
        if self.ptr:
            lib().cclqr_mech_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CtrlHandle:
    """cclqr_ctrl*: device-resident controller tables"""

    def __init__(self, mech, ctrl_joint, K=None, N=0, zd=None, Fd=None, fric=None, noise_scale=0.0, pid=None, noise_seed=None, n_ctrl=0):
        """n_ctrl > 1: one controller table per instance -- K [n_ctrl][nK][mu][12 nb], zd [n_ctrl][nsp][nb][13], Fd [n_ctrl][nsp][mu]"""
        nb = mech.tables.nb
        cj = i32(ctrl_joint).reshape(-1)
        mu = len(cj)
        nc = n_ctrl if n_ctrl > 1 else 1
        if zd is None:
            zd = np.zeros((nc, nb, 13))
            zd[:, :, 3] = 1.0
        zd = f64(zd).reshape(-1, nb, 13)
        nsp = zd.shape[0] // nc
        Fd = f64(np.zeros((nc * nsp, mu)) if Fd is None else Fd).reshape(nc * nsp, mu)
        K = None if K is None else f64(K).reshape(-1, mu, 12 * nb)
        fric = None if fric is None else f64(fric).reshape(mech.tables.ne)          # one entry per joint (ne == nb on a tree)
        pj = pP = pI = pD = pg = None
        if pid is not None:
            pj, pP, pI, pD, pg = i32(pid["joint"]).reshape(-1), f64(pid["P"]).reshape(-1), f64(pid["I"]).reshape(-1), f64(pid["D"]).reshape(-1), f64(pid["goal"]).reshape(-1)
        self._arrs = [cj, K, zd, Fd, fric, pj, pP, pI, pD, pg]
        self.mu, self.N, self.nsp = mu, int(N), nsp
        self.desc = CtrlDesc(mu, _i(cj), 0 if K is None else K.shape[0] // nc, int(N), _d(K), nsp, _d(zd), _d(Fd), _d(fric), float(noise_scale),
                             0 if pj is None else len(pj), _i(pj), _d(pP), _d(pI), _d(pD), _d(pg),
                             0 if noise_seed is None else 1, 0 if noise_seed is None else int(noise_seed), int(n_ctrl))
        self.ptr = C.c_void_p()
        check(lib().cclqr_ctrl_create(mech.ptr, C.byref(self.desc), C.byref(self.ptr)))

    def reserve_noise(self, n_inst, steps):
        """size the handle's Philox workspace (needed before a noise_philox launch is captured into a hipGraph)"""
        check(lib().cclqr_ctrl_reserve_noise(self.ptr, C.c_int64(int(n_inst)), C.c_int32(int(steps))))

    def set_feedforward(self, Fd=None, dev_ptr=None, length=None, stream=0):
        """the controlfunction hook (lqr.jl:14, :56): replace the feed-forward table [n_ctrl][nsp][mu] -- host array Fd, or device address dev_ptr
        (+ length in doubles) copied on `stream`"""
        if dev_ptr is not None:
            check(lib().cclqr_ctrl_set_feedforward(self.ptr, C.c_void_p(int(dev_ptr)), C.c_int64(int(length)), C.c_int32(1), C.c_void_p(int(stream)) if stream else None))
            return
        Fd = f64(Fd).reshape(-1)
        check(lib().cclqr_ctrl_set_feedforward(self.ptr, _d(Fd), C.c_int64(Fd.size), C.c_int32(0), None))

    def close(self):
        if self.ptr:
            lib().cclqr_ctrl_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchLqrHandle:
    """cclqr_ctrl* built by cclqr_ctrl_create_lqr_batch: one LQR per setpoint (linearsystem + dlqr + controller tables), gains device-resident"""

    def __init__(self, mech, zd, ctrl_joint, Q, R, N, Fd=None, tol=1e-5, infinite_horizon=False):
        """infinite_horizon: LQR{T,Inf} -- N = Ntemp = ceil(10/Δt) (lqr.jl:26), only Ku[1] per setpoint is kept"""
        nb = mech.tables.nb
        zd = f64(zd).reshape(-1, nb, 13)
        n = zd.shape[0]
        cj = i32(ctrl_joint).reshape(-1)
        mu = len(cj)
        Fd = None if Fd is None else f64(Fd).reshape(n, mu)
        Q, R = f64(Q).reshape(12 * nb, 12 * nb), f64(R).reshape(mu, mu)
        self.kbreak = np.zeros(n, dtype=np.int32)
        self.mu, self.N, self.nsp, self.n_ctrl = mu, (0 if infinite_horizon else int(N)), 1, n
        self._arrs = [zd, cj, Fd, Q, R]
        self.ptr = C.c_void_p()
        check(lib().cclqr_ctrl_create_lqr_batch(mech.ptr, C.c_int32(n), _d(zd), C.c_int32(mu), _i(cj), _d(Fd), _d(Q), _d(R), C.c_int32(int(N)),
                                                C.c_int32(1 if infinite_horizon else 0), C.c_double(float(tol)), _i(self.kbreak), C.byref(self.ptr)))

    def close(self):
        if self.ptr:
            lib().cclqr_ctrl_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rollout(mech, ctrl, z0, steps, k0=1, noise=None, record=False, first_instance=0, newton_mode=0, newton_eps_alone=0.0, flags=0):
    """host-pointer rollout: returns (zT, traj or None, status)"""
    nb = mech.tables.nb
    z0 = f64(z0).reshape(-1, nb, 13)
    n = z0.shape[0]
    traj = np.zeros((n, steps, nb, 13)) if record else None
    zT = np.zeros_like(z0)
    status = np.zeros(n, dtype=np.int32)
    noise = None if noise is None else f64(noise).reshape(n, steps)
    o = RolloutOpts(int(first_instance), None, 0, None, 0, int(newton_mode), int(flags), float(newton_eps_alone))
    check(lib().cclqr_rollout_host_ex(mech.ptr, ctrl.ptr, C.c_int64(n), C.c_int32(steps), C.c_int32(k0), _d(z0), _d(noise), _d(traj), _d(zT),
                                      _i(status), C.byref(o)))
    return zT, traj, status


def rollout_dev(mech, ctrl, n_inst, steps, k0, z0_ptr, lam_ptr, noise_ptr, noise_stride, traj_ptr, zT_ptr, status_ptr, stream=0,
                first_instance=None, pid_state=None, noise_ws=None, noise_ws_len=0, newton_mode=0, newton_eps_alone=0.0, flags=0):
    """device-pointer rollout (integers are raw device addresses, e.g. torch.Tensor.data_ptr()); asynchronous.
    Options (cclqr_rollout_opts): first_instance, pid_state = device address of [n_inst][joints][2] doubles, noise_ws / noise_ws_len = caller's
    Philox workspace, newton_mode, flags (ROLLOUT_NO_ALLOC: what a caller with a hipGraph capture open passes; ROLLOUT_PACK_WAVEFRONTS); none given: cclqr_rollout_dev
    (= NULL options)"""
    vp = lambda p: C.c_void_p(int(p)) if p else None
    if first_instance is None and pid_state is None and noise_ws is None and not newton_mode and not flags:
        check(lib().cclqr_rollout_dev(mech.ptr, ctrl.ptr, C.c_int64(n_inst), C.c_int32(steps), C.c_int32(k0), vp(z0_ptr), vp(lam_ptr),
                                      vp(noise_ptr), C.c_int64(noise_stride), vp(traj_ptr), vp(zT_ptr), vp(status_ptr), vp(stream)))
        return
    o = RolloutOpts(int(first_instance or 0), int(pid_state) if pid_state else None, n_inst * mech.tables.ne * 2 if pid_state else 0,       # one pair per joint (a tree has nb joints, a loop mechanism more)
                    int(noise_ws) if noise_ws else None, int(noise_ws_len) if noise_ws else 0, int(newton_mode), int(flags), float(newton_eps_alone))
    check(lib().cclqr_rollout_ex(mech.ptr, ctrl.ptr, C.c_int64(n_inst), C.c_int32(steps), C.c_int32(k0), vp(z0_ptr), vp(lam_ptr),
                                 vp(noise_ptr), C.c_int64(noise_stride), vp(traj_ptr), vp(zT_ptr), vp(status_ptr), C.byref(o), vp(stream)))


def linearize(mech, zd, ctrl_joint, Fd=None):
    """batched linearsystem: zd [nk][nb][13] -> A [nk][mx][mx], Bu, Bl, G"""
    t = mech.tables
    zd = f64(zd).reshape(-1, t.nb, 13)
    nk = zd.shape[0]
    cj = i32(ctrl_joint).reshape(-1)
    mu, mx, ml = len(cj), 12 * t.nb, 5 * t.ne
    Fd = f64(np.zeros((nk, mu)) if Fd is None else Fd).reshape(nk, mu)
    A, Bu, Bl, G = np.zeros((nk, mx, mx)), np.zeros((nk, mx, mu)), np.zeros((nk, mx, ml)), np.zeros((nk, ml, mx))
    check(lib().cclqr_linearize(mech.ptr, C.c_int32(nk), _d(zd), C.c_int32(mu), _i(cj), _d(Fd), _d(A), _d(Bu), _d(Bl), _d(G)))
    return A, Bu, Bl, G


def linearize_projected(mech, zd, ctrl_joint, Fd=None, h=0.0):
    """projected linear model (A', D) of the constrained step map: zd [nk][nb][13] -> A' [nk][mx][mx], D [nk][mx][mu] (cclqr_linearize_projected;
    any topology, the only linearisation of closed-loop mechanisms).  h <= 0: analytic on the device; h > 0: central differences of the device's step"""
    t = mech.tables
    zd = f64(zd).reshape(-1, t.nb, 13)
    nk = zd.shape[0]
    cj = i32(ctrl_joint).reshape(-1)
    mu, mx = len(cj), 12 * t.nb
    Fd = f64(np.zeros((nk, mu)) if Fd is None else Fd).reshape(nk, mu)
    Ap, D = np.zeros((nk, mx, mx)), np.zeros((nk, mx, mu))
    check(lib().cclqr_linearize_projected(mech.ptr, C.c_int32(nk), _d(zd), C.c_int32(mu), _i(cj), _d(Fd), C.c_double(float(h)), _d(Ap), _d(D)))
    return Ap, D


def riccati(A, Bu, Bl, G, Q, R, N, tol=1e-5, path=0, bf16_terms=0, keep_last=False):
    """batched dlqr: A [nprob][mx][mx] (or [mx][mx]) -> K [nprob][N-1][mu][mx], kbreak [nprob].
    path: 0 auto / 1 resident / 2 tiled; bf16_terms: 0 = fp64 MFMA (parity), 1..3 = split-bf16 measured-error mode (cclqr_riccati_opts);
    keep_last: K [nprob][1][mu][mx] = Ku[1] only (LQR{T,Inf})"""
    A = f64(A)
    single = A.ndim == 2
    mx = A.shape[-1]
    A = A.reshape(-1, mx, mx)
    nprob = A.shape[0]
    Bu = f64(Bu).reshape(nprob, mx, -1)
    mu = Bu.shape[2]
    Bl = f64(Bl).reshape(nprob, mx, -1)
    ml = Bl.shape[2]
    G = f64(G).reshape(nprob, ml, mx)
    Q, R = f64(Q).reshape(mx, mx), f64(R).reshape(mu, mu)
    K = np.zeros((nprob, (1 if keep_last else N - 1) if N > 1 else 0, mu, mx))
    kb = np.zeros(nprob, dtype=np.int32)
    o = RiccatiOpts(int(path), int(bf16_terms), 1 if keep_last else 0, 0)
    check(lib().cclqr_riccati_ex(C.c_int32(nprob), C.c_int32(mx), C.c_int32(mu), C.c_int32(ml), _d(A), _d(Bu), _d(Bl), _d(G), _d(Q), _d(R),
                                 C.c_int32(N), C.c_double(tol), _d(K), _i(kb), C.byref(o)))
    return (K[0], int(kb[0])) if single else (K, kb)


def riccati_tv(A, Bu, Bl, G, Q, R, N, tol=1e-5):
    """time-varying dlqr on caller-supplied per-knot models: A [N-1][mx][mx], Bu [N-1][mx][mu], Bl [N-1][mx][ml], G [N-1][ml][mx] -> K [N-1][mu][mx], kbreak"""
    A = f64(A)
    nk, mx = A.shape[0], A.shape[1]
    assert nk == N - 1
    Bu = f64(Bu).reshape(nk, mx, -1)
    mu = Bu.shape[2]
    Bl = f64(Bl).reshape(nk, mx, -1)
    ml = Bl.shape[2]
    G = f64(G).reshape(nk, ml, mx)
    K = np.zeros((nk, mu, mx))
    kb = np.zeros(1, dtype=np.int32)
    check(lib().cclqr_riccati_tv(C.c_int32(mx), C.c_int32(mu), C.c_int32(ml), _d(A), _d(Bu), _d(Bl) if ml else None, _d(G) if ml else None,
                                 _d(f64(Q)), _d(f64(R)), C.c_int32(int(N)), C.c_double(float(tol)), _d(K), _i(kb)))
    return K, int(kb[0])


def riccati_tracking(mech, ctrl_joint, zd, Fd, Q, R, N, tol=1e-5, path=0, bf16_terms=0):
    t = mech.tables
    cj = i32(ctrl_joint).reshape(-1)
    mu, mx = len(cj), 12 * t.nb
    zd = f64(zd).reshape(N, t.nb, 13)
    Fd = f64(Fd).reshape(N, mu)
    K = np.zeros((N - 1, mu, mx))
    kb = C.c_int32(0)
    o = RiccatiOpts(int(path), int(bf16_terms), 0, 0)
    check(lib().cclqr_riccati_tracking_ex(mech.ptr, C.c_int32(mu), _i(cj), _d(zd), _d(Fd), _d(f64(Q).reshape(mx, mx)), _d(f64(R).reshape(mu, mu)),
                                          C.c_int32(N), C.c_double(tol), _d(K), C.byref(kb), C.byref(o)))
    return K, kb.value


def rate_or_refusal(n_units, seconds, status):
    """throughput string of a run, or a refusal when any instance came back with status <= 0 (a rate of failed rollouts means nothing)"""
    import numpy as np
    bad = int((np.asarray(status) <= 0).sum())
    if bad:
        return "NO RATE: %d of %d instances failed (status <= 0)" % (bad, np.asarray(status).size)
    return "%.3g inst-steps/s" % (n_units / seconds)
