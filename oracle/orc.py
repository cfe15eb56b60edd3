"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT the product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
See oracle/cclqr_oracle.h for what is restated from where and for the parity status
("parity unpinned" for everything that lives in ConstrainedDynamics.jl).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class MechDesc(C.Structure):
    _fields_ = [("nb", C.c_int32), ("ne", C.c_int32), ("dt", C.c_double), ("g", C.c_double),
                ("mass", _dp), ("inertia", _dp), ("parent", _ip), ("child", _ip), ("type", _ip),
                ("p1", _dp), ("p2", _dp), ("axis", _dp), ("qoff", _dp)]


class CtrlDesc(C.Structure):
    _fields_ = [("mu", C.c_int32), ("ctrl_joint", _ip), ("nK", C.c_int32), ("N", C.c_int32), ("K", _dp),
                ("nsp", C.c_int32), ("zd", _dp), ("Fd", _dp), ("fric", _dp), ("noise_scale", C.c_double),
                ("npid", C.c_int32), ("pid_joint", _ip), ("pid_P", _dp), ("pid_I", _dp), ("pid_D", _dp), ("pid_goal", _dp),
                ("noise_philox", C.c_int32), ("noise_seed", C.c_uint64), ("n_ctrl", C.c_int32), ("noise", _dp)]


def build(force=False):
    """compile oracle/liborc.so (+ the flop-counting twin) with gcc"""
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "cclqr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


_libs = {}


def use_library(path):
    """time another build of the SAME source as the plain library (bench.py's -O3 -march=native baseline build); None restores
    oracle/liborc.so.  The checker used by the tests is always the portable build."""
    _libs.pop("plain", None)
    _libs["_override"] = path
    if path is None:
        _libs.pop("_override", None)


def lib(flops=False):
    key = "flops" if flops else "plain"
    if key not in _libs:
        build()
        so = os.path.join(_HERE, "liborc_flops.so" if flops else "liborc.so")
        if not flops and _libs.get("_override"):
            so = _libs["_override"]
        L = C.CDLL(so)
        L.orc_step.restype = C.c_int
        L.orc_rollout.restype = C.c_int
        L.orc_linearize.restype = C.c_int
        L.orc_riccati.restype = C.c_int
        L.orc_riccati_tracking.restype = C.c_int
        L.orc_flops_get.restype = C.c_double
        _libs[key] = L
    return _libs[key]


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return None if a is None else a.ctypes.data_as(_ip)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


class _Keep:
    """holds the numpy arrays a ctypes struct points into"""

    def __init__(self, desc, arrays):
        self.desc, self.arrays = desc, arrays


def mech_desc(t):
    """t: any object with nb, ne, dt, g, mass, inertia, parent, child, type, p1, p2, axis, qoff"""
    arrs = dict(mass=_f64(t.mass), inertia=_f64(t.inertia), parent=_i32(t.parent), child=_i32(t.child), type=_i32(t.type),
                p1=_f64(t.p1), p2=_f64(t.p2), axis=_f64(t.axis), qoff=_f64(t.qoff))
    d = MechDesc(int(t.nb), int(t.ne), float(t.dt), float(t.g), _d(arrs["mass"]), _d(arrs["inertia"]), _i(arrs["parent"]),
                 _i(arrs["child"]), _i(arrs["type"]), _d(arrs["p1"]), _d(arrs["p2"]), _d(arrs["axis"]), _d(arrs["qoff"]))
    return _Keep(d, arrs)


def ctrl_desc(nb, ctrl_joint, K=None, N=0, zd=None, Fd=None, fric=None, noise_scale=0.0, noise=None, pid=None, noise_seed=None, n_ctrl=0):
    """pid = dict(joint=[...], P=[...], I=[...], D=[...], goal=[...]) adds a PID law (pid.jl) on those joints;
    n_ctrl > 1: K [n_ctrl][nK][mu][12 nb], zd [n_ctrl][nsp][nb][13], Fd [n_ctrl][nsp][mu], one table per instance"""
    cj = _i32(ctrl_joint)
    mu = len(cj)
    K = _f64(K)
    nc = n_ctrl if n_ctrl > 1 else 1
    zd = _f64(zd if zd is not None else _identity_state(nb)).reshape(-1, nb, 13)
    Fd = _f64(Fd if Fd is not None else np.zeros((zd.shape[0], mu))).reshape(zd.shape[0], mu)
    nK = 0 if K is None else K.reshape(-1, mu, 12 * nb).shape[0] // nc
    arrs = dict(cj=cj, K=K, zd=zd, Fd=Fd, fric=_f64(fric), noise=_f64(noise))
    npid = 0
    if pid is not None:
        arrs.update(pj=_i32(pid["joint"]), pP=_f64(pid["P"]), pI=_f64(pid["I"]), pD=_f64(pid["D"]), pg=_f64(pid["goal"]))
        npid = len(arrs["pj"])
    d = CtrlDesc(mu, _i(cj), nK, int(N), _d(K), zd.shape[0] // nc, _d(zd), _d(Fd), _d(arrs["fric"]), float(noise_scale),
                 npid, _i(arrs.get("pj")), _d(arrs.get("pP")), _d(arrs.get("pI")), _d(arrs.get("pD")), _d(arrs.get("pg")),
                 0 if noise_seed is None else 1, 0 if noise_seed is None else int(noise_seed), int(n_ctrl), _d(arrs["noise"]))
    return _Keep(d, arrs)


def _identity_state(nb):
    z = np.zeros((nb, 13))
    z[:, 3] = 1.0
    return z


def step(t, z, lam, uj, flops=False):
    m = mech_desc(t)
    z = _f64(z).copy()
    lam = _f64(lam).copy()
    uj = _f64(uj)
    it = lib(flops).orc_step(C.byref(m.desc), _d(z), _d(lam), _d(uj))
    return z, lam, it


def step_fixed_lambda(t, z, lam, uj):
    m = mech_desc(t)
    z = _f64(z)
    out = np.zeros_like(z)
    lib().orc_step_fixed_lambda(C.byref(m.desc), _d(z), _d(_f64(lam)), _d(_f64(uj)), _d(out))
    return out


def constraints(t, z):
    m = mech_desc(t)
    g = np.zeros(5 * t.ne)
    lib().orc_constraints(C.byref(m.desc), _d(_f64(z)), _d(g))
    return g


def minimal_coordinates(t, z):
    m = mech_desc(t)
    th = np.zeros(t.ne)
    lib().orc_minimal_coordinates(C.byref(m.desc), _d(_f64(z)), _d(th))
    return th


def control(t, ctrl, z, k, noise_sample=0.0):
    m = mech_desc(t)
    uj = np.zeros(t.ne)
    lib().orc_control(C.byref(m.desc), C.byref(ctrl.desc), _d(_f64(z)), C.c_int(int(k)), C.c_double(noise_sample), _d(uj))
    return uj


def rollout(t, ctrl, z0, steps, record=False, nthreads=0, flops=False):
    m = mech_desc(t)
    z0 = _f64(z0).reshape(-1, t.nb, 13)
    n = z0.shape[0]
    traj = np.zeros((n, steps, t.nb, 13)) if record else None
    zT = np.zeros_like(z0)
    status = np.zeros(n, dtype=np.int32)
    rc = lib(flops).orc_rollout(C.byref(m.desc), C.byref(ctrl.desc), C.c_int64(n), C.c_int32(steps), _d(z0), _d(traj), _d(zT),
                                _i(status), C.c_int32(nthreads))
    if rc != 0:
        raise RuntimeError("orc_rollout failed: %d" % rc)
    return zT, traj, status


def linearize(t, zd, ctrl_joint, Fd=None):
    m = mech_desc(t)
    cj = _i32(ctrl_joint)
    mu, mx, ml = len(cj), 12 * t.nb, 5 * t.ne
    A, Bu, Bl, G = np.zeros((mx, mx)), np.zeros((mx, mu)), np.zeros((mx, ml)), np.zeros((ml, mx))
    Fd = _f64(Fd if Fd is not None else np.zeros(mu))
    rc = lib().orc_linearize(C.byref(m.desc), _d(_f64(zd)), C.c_int32(mu), _i(cj), _d(Fd), _d(A), _d(Bu), _d(Bl), _d(G))
    if rc != 0:
        raise RuntimeError("orc_linearize failed: %d" % rc)
    return A, Bu, Bl, G


def riccati(A, Bu, Bl, G, Q, R, N, tol=1e-5, flops=False):
    A, Bu, Bl, G, Q, R = (_f64(x) for x in (A, Bu, Bl, G, Q, R))
    mx, mu, ml = A.shape[0], Bu.shape[1], Bl.shape[1]
    K = np.zeros((max(N - 1, 0), mu, mx))
    kb = C.c_int32(0)
    rc = lib(flops).orc_riccati(mx, mu, ml, _d(A), _d(Bu), _d(Bl), _d(G), _d(Q), _d(R), C.c_int32(N), C.c_double(tol), _d(K), C.byref(kb))
    if rc != 0:
        raise RuntimeError("orc_riccati failed: %d" % rc)
    return K, kb.value


def riccati_tracking(t, ctrl_joint, zd, Fd, Q, R, N, tol=1e-5):
    m = mech_desc(t)
    cj = _i32(ctrl_joint)
    mu, mx = len(cj), 12 * t.nb
    K = np.zeros((N - 1, mu, mx))
    kb = C.c_int32(0)
    rc = lib().orc_riccati_tracking(C.byref(m.desc), C.c_int32(mu), _i(cj), _d(_f64(zd)), _d(_f64(Fd)), _d(_f64(Q)), _d(_f64(R)),
                                    C.c_int32(N), C.c_double(tol), _d(K), C.byref(kb))
    if rc != 0:
        raise RuntimeError("orc_riccati_tracking failed: %d" % rc)
    return K, kb.value


def philox4x32(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, out)
    return list(out)


def philox_normal(seed, instance, k):
    L = lib()
    L.orc_philox_normal.restype = C.c_double
    return L.orc_philox_normal(C.c_uint64(seed), C.c_uint64(instance), C.c_int(k))


def newton_stats(reset=True):
    """instrumented build only: (halvings per iteration index [16], solves reaching that index [16], histogram of -log10(alpha|d|) of the
    last iteration [16], histogram of halvings in the last iteration [11]) accumulated by the calling thread"""
    out = (C.c_double * 64)()
    lib(True).orc_newton_stats(out, C.c_int(1 if reset else 0))
    v = np.array(list(out))
    return v[0:16], v[16:32], v[32:48], v[48:59]


def newton_hist(reset=True):
    """instrumented build only: residual at the start of every Newton iteration, binned by floor(-log10 ||f||): (iterations after which
    ||f|| < eps for the first time, iterations after which it still is not, iterations entered with ||f|| < eps already)"""
    out = (C.c_double * 96)()
    lib(True).orc_newton_hist(out, C.c_int(1 if reset else 0))
    v = np.array(list(out))
    return v[0:32], v[32:64], v[64:96]


def set_newton_variant(v, flops=False):
    """0 = the reference's rule (default); 1 = MODEL of the device's frozen-Jacobian iterations (not a parity mode); 2 = MODEL of a residual-only stopping rule"""
    lib(flops).orc_set_newton_variant(C.c_int(v))


def flops_reset():
    lib(True).orc_flops_reset()


def flops_get():
    return lib(True).orc_flops_get()


# ----------------------------------------------------------------------------------------------
# numpy restatement of dlqr(A,Bu,Bλ,G,Q,R,N), src/control/lqr.jl:141-184, statement by statement.
# Used to cross-check the C restatement; Julia's `\` and `/` on square matrices are LU solves.
def dlqr_np(A, Bu, Bl, G, Q, R, N, tol=1e-5):
    mx, mu, ml = A.shape[1], Bu.shape[1], Bl.shape[1]           # :142-144
    Ku = [[np.zeros((1, Q.shape[0])) for _ in range(mu)] for _ in range(N - 1)]  # :145
    Pk = Q                                                       # :147
    k = 0                                                        # :149
    for k in range(N - 1, 0, -1):                                # :150
        if ml > 0:
            D = Bu - np.linalg.solve((G @ Bl).T, Bl.T).T @ G @ Bu  # :151  Bλ/(G*Bλ)*G*Bu
        else:
            D = Bu
        M11 = R + D.T @ Pk @ Bu                                  # :152
        M12 = D.T @ Pk @ Bl                                      # :153
        M21 = G @ Bu                                             # :154
        M22 = G @ Bl                                             # :155
        M = np.block([[M11, M12], [M21, M22]])                   # :157
        b = np.vstack([D.T @ Pk, G]) @ A                         # :158
        Kk = np.linalg.solve(M, b)                               # :160
        for i in range(mu):                                      # :162-164
            Ku[k - 1][i] = Kk[i:i + 1, :]
        Kuk = Kk[:mu, :]                                         # :166
        Klk = Kk[mu:mu + ml, :]                                  # :167
        Abar = A - Bu @ Kuk - Bl @ Klk                           # :169
        Pkp1 = Q + Kuk.T @ R @ Kuk + Abar.T @ Pk @ Abar          # :170
        if np.linalg.norm(Pk - Pkp1) < tol:                      # :172
            break
        Pk = Pkp1                                                # :176
    for k2 in range(k - 1, 0, -1):                               # :179-181
        Ku[k2 - 1] = Ku[k2]
    K = np.array([[Ku[i][j][0] for j in range(mu)] for i in range(N - 1)]).reshape(N - 1, mu, mx)
    return K, k
