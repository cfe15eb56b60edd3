"""Host-side mirror of the reference's controllers.  Same constructor arguments, same assertions, same field
meaning; the arithmetic (linearsystem, dlqr, simulate!) runs in HIP through the C-ABI (include/cclqr.h).

    LQR(mechanism, bodyids, eqcids, Q, R, horizon; xd, vd, qd, ωd, Fτd)            src/control/lqr.jl:49-66
    LQR(A, Bu, Bλ, G, Q, R, horizon, eqcids, xd, vd, qd, ωd, Fτd, Δt)              src/control/lqr.jl:17-47
    TrackingLQR(mechanism, storage, Fτ, eqcids, Q, R)                                src/control/lqr_tracking.jl:17-43
    simulate!(mechanism, tend | storage, controller; record)                         e.g. examples/lqr_cartpole.jl:44
    Storage{T}(steps, Nb)                                                            examples/trackingLQR_triple_cartpole.jl:50-51

What is new relative to the reference (which has no batch API): simulate(..., z0=batch) rolls out n_inst
independent instances at once.  A custom `controlfunction(batch, controller, k)` (lqr.jl:14, :56) is a host closure: simulate then steps the
device one launch per step, shows the closure the batch's states (BatchState) and applies the joint inputs it sets with setForce
(cclqr_ctrl_set_feedforward) -- the built-in laws (LQR / TrackingLQR / friction + noise / PID) stay fused in one persistent launch.
"""
import math

import numpy as np

from . import _capi
from .mechanism import Mechanism, minimal_to_maximal, one_quaternion


def _blockdiag(blocks):
    n = sum(b.shape[0] for b in blocks)
    M = np.zeros((n, n))
    o = 0
    for b in blocks:
        k = b.shape[0]
        M[o:o + k, o:o + k] = b
        o += k
    return M


def _pack_setpoint(nb, xd, vd, qd, ωd):
    z = np.zeros((nb, 13))
    for i in range(nb):
        z[i, 0:3] = np.asarray(xd[i], dtype=np.float64).reshape(3)
        z[i, 3:7] = np.asarray(qd[i], dtype=np.float64).reshape(4)
        z[i, 7:10] = np.asarray(vd[i], dtype=np.float64).reshape(3)
        z[i, 10:13] = np.asarray(ωd[i], dtype=np.float64).reshape(3)
    return z


def _device_mech(mechanism):
    h = getattr(mechanism, "_cclqr_handle", None)
    if h is None or not h.ptr:
        h = _capi.MechHandle(mechanism.tables())
        mechanism._cclqr_handle = h
    return h


class Controller:
    """abstract type owned by ConstrainedDynamics (lqr.jl:3 `<: Controller`)"""


class LQR(Controller):
    """LQR{T,N,NK}: K[k][i] (1 x NK rows), xd, vd, qd, ωd, eqcids, Fτd   (lqr.jl:3-15)"""

    def __init__(self, mechanism, bodyids, eqcids, Q, R, horizon, xd=None, vd=None, qd=None, ωd=None, Fτd=None, controlfunction=None, **kw):
        if "wd" in kw:
            ωd = kw.pop("wd")
        if "Ftd" in kw:
            Fτd = kw.pop("Ftd")
        if "xθd" in kw:          # minimal-coordinate constructor keywords (lqr.jl:70-72)
            xd = kw.pop("xθd")
        if "vωd" in kw:
            vd = kw.pop("vωd")
        if kw:
            raise TypeError("unexpected keyword(s): %s" % list(kw))
        self.controlfunction = controlfunction          # lqr.jl:14: a host closure (batch, lqr, k); None = control_lqr! on the device
        if not isinstance(mechanism, Mechanism):
            raise TypeError("LQR(mechanism, bodyids, eqcids, Q, R, horizon; ...)")
        nb = len(mechanism.bodies)
        if len(Q) and np.ndim(Q[0]) == 0:
            # LQR(mechanism, controlledids, controlids, Q::Vector{T}, R::Vector{T}, horizon; xθd, vωd, Fτd)      lqr.jl:68-86
            controlledids, controlids = bodyids, eqcids
            xθd = np.zeros(len(controlledids)) if xd is None else np.asarray(xd, dtype=np.float64)
            vωd = np.zeros(len(controlledids)) if vd is None else np.asarray(vd, dtype=np.float64)
            Fτ = np.zeros(len(controlids)) if Fτd is None else np.asarray(Fτd, dtype=np.float64).reshape(-1)
            assert len(controlledids) == len(Q) == len(xθd) == len(vωd) == nb, "Missmatched length for bodies"          # lqr.jl:76
            assert len(controlids) == len(R) == len(Fτ), "Missmatched length for constraints"                          # lqr.jl:77
            xd, vd, qd, ωd = minimal_to_maximal(mechanism, controlledids, xθd, vωd)                                    # lqr.jl:80
            Q = [np.eye(12) * float(q) for q in Q]                                                                     # lqr.jl:82
            R = [np.eye(1) * float(r) for r in R]                                                                      # lqr.jl:83
            Fτd = [[f] for f in Fτ]                                                                                    # lqr.jl:85
            bodyids = [b.id for b in mechanism.bodies]
        z3 = [np.zeros(3) for _ in range(nb)]
        xd = z3 if xd is None else xd                       # lqr.jl:51-55 defaults
        vd = z3 if vd is None else vd
        qd = [one_quaternion() for _ in range(nb)] if qd is None else qd
        ωd = z3 if ωd is None else ωd
        Fτd = [np.zeros(1) for _ in range(len(eqcids))] if Fτd is None else Fτd
        # lqr.jl:59-60
        assert len(bodyids) == len(Q) == len(xd) == len(vd) == len(qd) == len(ωd) == nb, "Missmatched length for bodies"
        assert len(eqcids) == len(R) == len(Fτd), "Missmatched length for constraints"
        order = [int(b) - 1 for b in bodyids]
        if sorted(order) != list(range(nb)):
            raise ValueError("bodyids must name every body exactly once")
        # state blocks follow the order of `bodyids`; the device works in mechanism body order
        inv = np.argsort(order)
        Q = [np.asarray(Q[i], dtype=np.float64) for i in inv]
        xd, vd, qd, ωd = ([a[i] for i in inv] for a in (xd, vd, qd, ωd))
        self.mechanism = mechanism
        self.eqcids = [int(e) for e in eqcids]
        self.ctrl_joints = [mechanism.joint_index(e) for e in self.eqcids]
        self.xd, self.vd, self.qd, self.ωd = xd, vd, qd, ωd
        self.Fτd = [np.asarray(f, dtype=np.float64).reshape(1) for f in Fτd]
        self.zd = _pack_setpoint(nb, xd, vd, qd, ωd)[None]
        self.Fd = np.array([f[0] for f in self.Fτd]).reshape(1, -1)
        dev = _device_mech(mechanism)
        self.projected = bool(getattr(mechanism, "has_loops", False))
        if self.projected:
            # closed loops (examples/lqr_deltabot.jl:47-53): G Bλ of lqr.jl:151 is singular (redundant constraint rows), but the pair the
            # recursion works with -- A' = A - Bλ (G Bλ)^-1 G A, D = Bu - Bλ (G Bλ)^-1 G Bu -- is unique: it is the Jacobian of the
            # constrained one-step map, formed analytically on the device (cclqr_linearize_projected), and the recursion runs on it with no multipliers
            mx = 12 * nb
            Ap, D = _capi.linearize_projected(dev, self.zd, self.ctrl_joints, self.Fd)
            self.A, self.Bu, self.Bλ, self.G = Ap[0], D[0], np.zeros((mx, 0)), np.zeros((0, mx))
        else:
            # linearize (lqr.jl:63)
            A, Bu, Bl, G = _capi.linearize(dev, self.zd, self.ctrl_joints, self.Fd)
            self.A, self.Bu, self.Bλ, self.G = A[0], Bu[0], Bl[0], G[0]
        self._finish(self.A, self.Bu, self.Bλ, self.G, Q, [np.asarray(r, dtype=np.float64) for r in R], horizon, mechanism.Δt)

    def _finish(self, A, Bu, Bl, G, Q, R, horizon, Δt):
        """LQR(A, Bu, Bλ, G, Q, R, horizon, eqcids, xd, ..., Δt)   lqr.jl:17-47"""
        self.Q = _blockdiag(Q) * Δt                         # lqr.jl:18
        self.R = _blockdiag(R) * Δt                         # lqr.jl:19
        N = horizon / Δt                                    # lqr.jl:21
        if N < math.inf:
            N = int(math.ceil(horizon / Δt))                # lqr.jl:23
            Ntemp = N
        else:
            Ntemp = int(math.ceil(10 / Δt))                 # lqr.jl:26: 10 s as maximal horizon for Inf
        if G.shape[0] == 0 and N == math.inf and not getattr(self, "projected", False):
            # lqr.jl:33 calls dlqr(A,Bu,Q,R,N) which binds N=Inf to the Δt method of util.jl:50 — a defect, not reproduced
            raise ValueError("unconstrained infinite-horizon LQR is not reachable in the reference (lqr.jl:33)")
        K, kbreak = _capi.riccati(A, Bu, Bl, G, self.Q, self.R, Ntemp)   # lqr.jl:36 / :39
        self.kbreak = kbreak
        self.converged = True
        if N == math.inf:
            self.converged = bool(Ntemp < 3 or np.array_equal(K[0], K[1]))
            if not self.converged:
                print("[ Info: Riccati recursion did not converge.")      # lqr.jl:41
            K = K[:1]                                                     # lqr.jl:42
        self.K = K
        self.N = 0 if N == math.inf else N                  # device convention: N <= 0 means LQR{T,Inf}
        self.horizon_steps = N
        self.NK = K.shape[2]

    @classmethod
    def from_matrices(cls, A, Bu, Bλ, G, Q, R, horizon, eqcids, xd, vd, qd, ωd, Fτd, Δt, mechanism=None):
        """the inner constructor lqr.jl:17-47 for callers that bring their own linear model"""
        self = cls.__new__(cls)
        nb = len(xd)
        self.controlfunction = None
        self.mechanism = mechanism
        self.eqcids = [int(e) for e in eqcids]
        self.ctrl_joints = [mechanism.joint_index(e) for e in self.eqcids] if mechanism is not None else list(range(len(eqcids)))
        self.xd, self.vd, self.qd, self.ωd = xd, vd, qd, ωd
        self.Fτd = [np.asarray(f, dtype=np.float64).reshape(1) for f in Fτd]
        self.zd = _pack_setpoint(nb, xd, vd, qd, ωd)[None]
        self.Fd = np.array([f[0] for f in self.Fτd]).reshape(1, -1)
        self.A, self.Bu, self.Bλ, self.G = (np.asarray(M, dtype=np.float64) for M in (A, Bu, Bλ, G))
        self._finish(self.A, self.Bu, self.Bλ, self.G, [np.asarray(q, dtype=np.float64) for q in Q], [np.asarray(r, dtype=np.float64) for r in R],
                     horizon, Δt)
        return self

    def _ctrl_handle(self, dev, fric=None, noise_scale=0.0, noise_seed=None):
        return _capi.CtrlHandle(dev, self.ctrl_joints, K=self.K, N=self.N, zd=self.zd, Fd=self.Fd, fric=fric, noise_scale=noise_scale, noise_seed=noise_seed)


class Storage:
    """Storage{T}(steps, Nb): x[i][k], q[i][k], v[i][k], ω[i][k]  (lqr_tracking.jl:32-35).  Batched: leading instance axis.
    `z` is the raw [n_inst][steps][nb][13] array; x/q/v/ω index as storage.x[i][k] for instance 0 (or .instance(n))."""

    def __init__(self, steps, nb, n_inst=1, z=None, status=None, zT=None):
        self.steps, self.nb, self.n_inst = int(steps), int(nb), int(n_inst)
        self.z = np.zeros((n_inst, steps, nb, 13)) if z is None else z
        if z is None:
            self.z[..., 3] = 1.0
        self.status = status
        self.zT = zT

    def instance(self, n):
        return Storage(self.steps, self.nb, 1, self.z[n:n + 1], None if self.status is None else self.status[n:n + 1],
                       None if self.zT is None else self.zT[n:n + 1])

    def _field(self, lo, hi):
        return [self.z[0, :, i, lo:hi] for i in range(self.nb)]

    x = property(lambda self: self._field(0, 3))
    q = property(lambda self: self._field(3, 7))
    v = property(lambda self: self._field(7, 10))
    ω = property(lambda self: self._field(10, 13))
    w = ω


class TrackingLQR(Controller):
    """TrackingLQR{T,N,NK}: per-step setpoints copied out of `storage` and per-step feed-forward Fτ  (lqr_tracking.jl:3-43)"""

    def __init__(self, mechanism, storage, Fτ, eqcids, Q, R, controlfunction=None):
        self.controlfunction = controlfunction          # lqr_tracking.jl:19: a host closure (batch, lqr, k); None = control_trackinglqr! on the device
        nb = len(mechanism.bodies)
        N = storage.steps
        self.mechanism = mechanism
        self.eqcids = [int(e) for e in eqcids]
        self.ctrl_joints = [mechanism.joint_index(e) for e in self.eqcids]
        self.Q = _blockdiag([np.asarray(q, dtype=np.float64) for q in Q]) * mechanism.Δt     # lqr_tracking.jl:22
        self.R = _blockdiag([np.asarray(r, dtype=np.float64) for r in R]) * mechanism.Δt     # lqr_tracking.jl:23
        self.zd = np.ascontiguousarray(storage.z[0])                                          # lqr_tracking.jl:30-37
        mu = len(self.eqcids)
        self.Fd = np.array([[np.asarray(Fτ[k][i], dtype=np.float64).reshape(-1)[0] for i in range(mu)] for k in range(N)]).reshape(N, mu)
        dev = _device_mech(mechanism)
        self.projected = bool(getattr(mechanism, "has_loops", False))
        if self.projected:
            # closed loops: the projected pair (A'_k, D_k) of every knot from the device's constrained step map (see LQR), recursion with
            # no multipliers left
            mx = 12 * nb
            Ap, D = _capi.linearize_projected(dev, self.zd[:N - 1], self.ctrl_joints, self.Fd[:N - 1])
            self.K, self.kbreak = _capi.riccati_tv(Ap, D, np.zeros((N - 1, mx, 0)), np.zeros((N - 1, 0, mx)), self.Q, self.R, N)
        else:
            self.K, self.kbreak = _capi.riccati_tracking(dev, self.ctrl_joints, self.zd, self.Fd, self.Q, self.R, N)   # lqr_tracking.jl:40
        self.N = N
        self.NK = 12 * nb

    def _ctrl_handle(self, dev, fric=None, noise_scale=0.0, noise_seed=None):
        return _capi.CtrlHandle(dev, self.ctrl_joints, K=self.K, N=self.N, zd=self.zd, Fd=self.Fd, fric=fric, noise_scale=noise_scale, noise_seed=noise_seed)


class OpenLoop(Controller):
    """control!(mechanism, k) = setForce!(mechanism, joint, [U[k]])  — the open-loop closure of
    examples/trackingLQR_triple_cartpole.jl:46-48 as a table: U[k][i] for controlled joints eqcids."""

    def __init__(self, mechanism, eqcids, U):
        self.mechanism = mechanism
        self.eqcids = [int(e) for e in eqcids]
        self.ctrl_joints = [mechanism.joint_index(e) for e in self.eqcids]
        U = np.asarray(U, dtype=np.float64)
        self.Fd = U.reshape(U.shape[0], len(self.eqcids))
        self.N = self.Fd.shape[0] + 1          # every recorded step applies its force
        nb = len(mechanism.bodies)
        zd = np.zeros((self.Fd.shape[0], nb, 13))
        zd[..., 3] = 1.0
        self.zd = zd

    def _ctrl_handle(self, dev, fric=None, noise_scale=0.0, noise_seed=None):
        return _capi.CtrlHandle(dev, self.ctrl_joints, K=None, N=self.N, zd=self.zd, Fd=self.Fd, fric=fric, noise_scale=noise_scale, noise_seed=noise_seed)


class PID(Controller):
    """PID{T,N}: P, I, D, eqcids, goals (integratederrors / lasterrors live on the device)   src/control/pid.jl:3-40
    PID(mechanism, eqcid::Int, goal; P, I, D)  /  PID(mechanism, eqcids::Vector, goals::Vector; P, I, D)"""

    def __init__(self, mechanism, eqcids, goals, P=None, I=None, D=None, controlfunction=None):
        self.controlfunction = controlfunction          # pid.jl:16, :27: a host closure (batch, pid, k); None = control_pid! on the device
        scalar = np.ndim(eqcids) == 0
        ids = [int(eqcids)] if scalar else [int(e) for e in eqcids]
        n = len(ids)
        vec = lambda v: np.zeros(n) if v is None else np.asarray(v, dtype=np.float64).reshape(n)
        self.mechanism = mechanism
        self.eqcids = ids
        self.goals, self.P, self.I, self.D = vec(goals), vec(P), vec(I), vec(D)
        for e in ids:
            assert len(mechanism.geteqconstraint(e)) == 5, "Only 1 DOF joints are supported"      # pid.jl:20,36
        self.joints = [mechanism.joint_index(e) for e in ids]

    def _ctrl_handle(self, dev, fric=None, noise_scale=0.0, noise_seed=None):
        return _capi.CtrlHandle(dev, [], K=None, N=0, fric=fric, noise_scale=noise_scale, noise_seed=noise_seed,
                                pid=dict(joint=self.joints, P=self.P, I=self.I, D=self.D, goal=self.goals))


class BatchState:
    """What a `controlfunction(batch, controller, k)` sees at step k: the states of every instance, in the mechanism's body order
    (z [n_inst][nb][13] = x, q, v, ω per body; the reference's closure reads body.state.xsol[2] ... of ONE mechanism, lqr.jl:98-103), and the
    joint inputs it sets with setForce (the reference's setForce!(mechanism, eqc, u), lqr.jl:110)."""

    def __init__(self, mechanism, z, k):
        self.mechanism, self.z, self.k = mechanism, z, int(k)
        self.n_inst = z.shape[0]
        self.u = {}
        self.on_device = not isinstance(z, np.ndarray)          # a torch tensor in HBM: the closure was registered with on_device()

    x = property(lambda self: self.z[:, :, 0:3])
    q = property(lambda self: self.z[:, :, 3:7])
    v = property(lambda self: self.z[:, :, 7:10])
    ω = property(lambda self: self.z[:, :, 10:13])
    w = ω


def on_device(controlfunction=None, graph=False):
    """Marks a `controlfunction(batch, controller, k)` as a DEVICE closure: simulate then hands it the batch's states as a torch tensor in HBM
    (batch.z [n_inst][nb][13], batch.x / .q / .v / .ω views of it) and expects its inputs as tensors (setForce), so the whole loop -- the
    closure's torch kernels, the hand-over of its inputs (cclqr_ctrl_set_feedforward, device to device) and the single-step launch -- runs on
    one stream with no host round trip per step.  control_lqr, state_error and setForce work on either kind of batch.
    graph=True (@on_device(graph=True)): the whole horizon is captured into a hipGraph by the first simulate call and REPLAYED by every later call
    with the same batch size and horizon -- the closure then runs once, at capture time: it must be capturable (no host synchronisation, no
    .item() / .cpu(), the same tensor shapes for every k) and a pure function of (batch, controller, k)."""
    def mark(f):
        f.on_device = True
        f.capture = bool(graph)
        return f
    return mark if controlfunction is None else mark(controlfunction)


def setForce(batch, eqc, u):
    """setForce!(mechanism, eqconstraint, u) on a batch: u a scalar, [n_inst] or [n_inst][1]; eqc an EqualityConstraint or its id"""
    j = batch.mechanism.joint_index(getattr(eqc, "id", eqc))
    if batch.on_device:
        import torch
        u = torch.as_tensor(u, dtype=torch.float64, device=batch.z.device).reshape(-1)
        batch.u[j] = u.expand(batch.n_inst) if u.numel() == 1 else u
        return
    batch.u[j] = np.broadcast_to(np.asarray(u, dtype=np.float64).reshape(-1), (batch.n_inst,)).copy()


def _device_tables(controller, device):
    """the controller's gain / setpoint / feed-forward tables as torch tensors on `device` (made once per controller and device)"""
    import torch
    cache = controller.__dict__.setdefault("_torch_tables", {})
    key = str(device)
    if key not in cache:
        cache[key] = tuple(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device) for a in (controller.K, controller.zd, controller.Fd))
    return cache[key]


def _fused_gain_tables(lqr, device):
    """G [nK][mu][13 nb], h [nK][mu] with u_k = h_k - G_k z (z the state as stored, 13 numbers per body): K_k folded with the affine map z -> Δz of
    the step's setpoint (state_error: identity blocks for x, v, ω, the 3 x 4 matrix of qd' (x) . for the quaternion part)"""
    import torch
    cache = lqr.__dict__.setdefault("_torch_fused", {})
    key = str(device)
    if key not in cache:
        K = np.asarray(lqr.K, dtype=np.float64)
        nK, mu = K.shape[0], K.shape[1]
        nb = K.shape[2] // 12
        Kb = K.reshape(nK, mu, nb, 12)
        zd, Fd = np.asarray(lqr.zd, dtype=np.float64), np.asarray(lqr.Fd, dtype=np.float64)
        d = zd[np.minimum(np.arange(nK), zd.shape[0] - 1)] if zd.shape[0] > 1 else np.repeat(zd[:1], nK, 0)        # setpoint of table row i (step i + 1)
        fd = Fd[np.minimum(np.arange(nK), Fd.shape[0] - 1)] if Fd.shape[0] > 1 else np.repeat(Fd[:1], nK, 0)
        G = np.zeros((nK, mu, nb, 13))
        G[..., 0:3] = Kb[..., 0:3]                   # x
        G[..., 7:10] = Kb[..., 3:6]                  # v
        G[..., 10:13] = Kb[..., 9:12]                # ω
        s0, ax, ay, az = d[:, None, :, 3], -d[:, None, :, 4], -d[:, None, :, 5], -d[:, None, :, 6]
        kx, ky, kz = Kb[..., 6], Kb[..., 7], Kb[..., 8]
        # qe = M (s1, bx, by, bz)' with rows (ax, s0, -az, ay), (ay, az, s0, -ax), (az, -ay, ax, s0): the quaternion columns of G are K_q M
        G[..., 3] = kx * ax + ky * ay + kz * az
        G[..., 4] = kx * s0 + ky * az - kz * ay
        G[..., 5] = -kx * az + ky * s0 + kz * ax
        G[..., 6] = kx * ay - ky * ax + kz * s0
        c = np.zeros((nK, nb, 12))
        c[..., 0:3], c[..., 3:6], c[..., 9:12] = d[..., 0:3], d[..., 7:10], d[..., 10:13]
        h = fd + np.einsum("kmbj,kbj->km", Kb, c)
        cache[key] = (torch.from_numpy(G.reshape(nK, mu, 13 * nb)).to(device), torch.from_numpy(np.ascontiguousarray(h)).to(device))
    return cache[key]


def _state_error_device(batch, controller, k):
    """state_error on a device batch: the same expressions in torch"""
    import torch
    _, zd, _ = _device_tables(controller, batch.z.device)
    d = zd[min(k - 1, zd.shape[0] - 1)] if zd.shape[0] > 1 else zd[0]
    z = batch.z
    s0, a = d[None, :, 3:4], -d[None, :, 4:7]                    # qd' = (s0, a): the conjugate of the setpoint's quaternion
    s1, b = z[:, :, 3:4], z[:, :, 4:7]
    qe = s0 * b + s1 * a + torch.cross(a.expand_as(b), b, dim=2)
    dz = torch.cat([z[:, :, 0:3] - d[None, :, 0:3], z[:, :, 7:10] - d[None, :, 7:10], qe, z[:, :, 10:13] - d[None, :, 10:13]], dim=2)
    return dz.reshape(z.shape[0], -1)


def state_error(batch, controller, k):
    """Δz [n_inst][12 nb] of lqr.jl:92-103 / lqr_tracking.jl:49-62 about the controller's setpoint of step k (bodies in mechanism order)"""
    if getattr(batch, "on_device", False):
        return _state_error_device(batch, controller, k)
    zd = controller.zd
    d = zd[min(k - 1, zd.shape[0] - 1)] if zd.shape[0] > 1 else zd[0]
    zt = batch.z.transpose(1, 2, 0)             # [nb][13][n_inst]
    nb, n = zt.shape[0], zt.shape[2]
    # One [n_inst] plane per component: the batch that simulate hands a closure is stored that way (_simulate_hosted), so every operand below is a
    # contiguous run of n_inst numbers (an instance-major array goes through the same code as strided planes, a few times slower)
    dzt = np.empty((nb, 12, n))
    np.subtract(zt[:, 0:3], d[:, 0:3, None], out=dzt[:, 0:3])
    np.subtract(zt[:, 7:10], d[:, 7:10, None], out=dzt[:, 3:6])
    np.subtract(zt[:, 10:13], d[:, 10:13, None], out=dzt[:, 9:12])
    s0, ax, ay, az = d[:, 3, None], -d[:, 4, None], -d[:, 5, None], -d[:, 6, None]           # qd' = (s0, a): the conjugate of the setpoint's quaternion
    s1, bx, by, bz = zt[:, 3], zt[:, 4], zt[:, 5], zt[:, 6]
    dzt[:, 6] = s0 * bx + s1 * ax + (ay * bz - az * by)                                      # vector part of qd \ q = s0 b + s1 a + a x b
    dzt[:, 7] = s0 * by + s1 * ay + (az * bx - ax * bz)
    dzt[:, 8] = s0 * bz + s1 * az + (ax * by - ay * bx)
    return dzt.reshape(12 * nb, n).T


def control_lqr(batch, lqr, k):
    """control_lqr! (lqr.jl:89-139) / control_trackinglqr! (lqr_tracking.jl:46-71) on the host for a batch: u = Fτd - K[k] Δz on the
    controller's joints, gated by k < N; sets the forces and returns u [n_inst][mu] -- the building block of a custom controlfunction"""
    mu = len(lqr.eqcids)
    if getattr(batch, "on_device", False):
        import torch
        if not (lqr.N <= 0 or k < lqr.N):
            return torch.zeros((batch.n_inst, mu), dtype=torch.float64, device=batch.z.device)
        # Δz is affine in z for a fixed setpoint (the vector part of qd \ q is linear in q), so u = Fd - K Δz = h_k - z G_k' with one row G_k = K_k T_k
        # per input and step, folded on the host once per controller: the whole law is ONE product on the device (a closure's step is bound by
        # the number of torch launches, not by their arithmetic)
        G, h = _fused_gain_tables(lqr, batch.z.device)
        i = 0 if lqr.N <= 0 else min(k - 1, G.shape[0] - 1)
        u = h[i][None, :] - batch.z.reshape(batch.n_inst, -1) @ G[i].T
        for j, e in enumerate(lqr.eqcids):
            setForce(batch, e, u[:, j])
        return u
    u = np.zeros((batch.n_inst, mu))
    if lqr.N <= 0 or k < lqr.N:                                  # LQR{T,Inf}: always; finite: k < N (lqr.jl:106)
        dz = state_error(batch, lqr, k)
        K = lqr.K[0] if lqr.N <= 0 else lqr.K[min(k - 1, lqr.K.shape[0] - 1)]
        Fd = lqr.Fd[min(k - 1, lqr.Fd.shape[0] - 1)] if lqr.Fd.shape[0] > 1 else lqr.Fd[0]
        u = Fd[None, :] - (K.reshape(mu, -1) @ dz.T).T     # (this association: [mu][12 nb] x [12 nb][n_inst] is a row-major product whichever way Δz is stored)
        for i, e in enumerate(lqr.eqcids):
            setForce(batch, e, u[:, i])
    return u


def _simulate_hosted(mechanism, steps, controller, record, z0):
    """simulate! with a host-side controlfunction: one single-step launch per step through the k0 continuation of cclqr_rollout_dev (state and
    multipliers stay on the device), the states copied to the host for the closure and its joint inputs handed back (cclqr_ctrl_set_feedforward)"""
    import torch
    t = mechanism.tables()
    nb, n = t.nb, z0.shape[0]
    dev = _device_mech(mechanism)
    joints = [j for j in range(t.ne) if int(t.type[j]) in (0, 1)]             # every 1-DoF joint can be given an input (revolute, prismatic)
    slot = {j: i for i, j in enumerate(joints)}
    ctrl = _capi.CtrlHandle(dev, joints, K=None, N=0, Fd=np.zeros((n, len(joints))), n_ctrl=n)
    td = torch.device("cuda", torch.cuda.current_device())
    z = torch.from_numpy(np.ascontiguousarray(z0)).to(td)
    zn = torch.empty_like(z)
    lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=td)
    st = torch.zeros(n, dtype=torch.int32, device=td)
    traj = np.zeros((n, steps if record else 0, nb, 13))
    # an instance whose step ended on a non-finite residual is LOST: the fused rollout freezes it at its last pose, at rest, for the rest of
    # the horizon (rollout_chain.hip, LinkC::DEAD).  A launch per step keeps that through CCLQR_ROLLOUT_CARRY_STATUS: the status array travels from
    # launch to launch on the device, a lost instance is not stepped again (the closure still sees it, frozen, as it would in the fused run)
    stream = torch.cuda.current_stream().cuda_stream
    # The states reach the closure through one page-locked buffer, transposed on the device to one [n_inst] plane per state component: BatchState.z
    # is a [n_inst][nb][13] VIEW of it (valid during the call: the next step overwrites it), so that a closure's numpy slices z[:, b, i] are contiguous
    # runs of n_inst numbers.  The recorded trajectory is copied instance-major in a second transfer.
    zt_pinned = torch.empty((nb, 13, n), dtype=torch.float64, pin_memory=True)
    zh = zt_pinned.numpy().transpose(2, 0, 1)
    za_pinned = torch.empty(z.shape, dtype=torch.float64, pin_memory=True) if record else None
    U_pinned = torch.zeros((n, len(joints)), dtype=torch.float64, pin_memory=True)
    U, U_dev = U_pinned.numpy(), torch.zeros((n, len(joints)), dtype=torch.float64, device=td)
    try:
        for k in range(1, steps + 1):
            zt_pinned.copy_(z.permute(1, 2, 0))     # (synchronises the stream: the previous launch no longer reads the feed-forward table set_feedforward rewrites below)
            if record:
                za_pinned.copy_(z)
                traj[:, k - 1] = za_pinned.numpy()
            batch = BatchState(mechanism, zh, k)
            controller.controlfunction(batch, controller, k)
            U[:] = 0.0
            for j, u in batch.u.items():
                if j not in slot:
                    raise ValueError("setForce on a constraint without a degree of freedom")
                U[:, slot[j]] = u
            U_dev.copy_(U_pinned, non_blocking=True)      # (page-locked -> device on the launch's stream; the table is replaced device to device behind it;
                                                          #  U_pinned is rewritten only after the next step's blocking copy of the states has drained the stream)
            ctrl.set_feedforward(dev_ptr=U_dev.data_ptr(), length=U_dev.numel(), stream=stream)
            _capi.rollout_dev(dev, ctrl, n, 1, k, z.data_ptr(), lam.data_ptr(), 0, 0, 0, zn.data_ptr(), st.data_ptr(), stream, flags=_capi.ROLLOUT_CARRY_STATUS)
            z, zn = zn, z
        zT = z.cpu().numpy()
        status = st.cpu().numpy()
    finally:
        ctrl.close()
    return zT, traj, status


class _DeviceClosureRun:
    """One (controller, batch size, horizon) instance of the device-closure loop: its buffers in HBM, the per-instance feed-forward controller, and --
    for a closure registered with on_device(f, graph=True) -- the whole horizon captured once into a hipGraph and replayed by every later simulate
    call with the same shape (a Monte-Carlo sweep, an MPC loop over the same horizon): the replay costs no host time per step at all."""

    def __init__(self, mechanism, steps, controller, record, n):
        import torch
        self.torch = torch
        t = mechanism.tables()
        self.mechanism, self.controller, self.steps, self.record, self.n, self.nb, self.ne = mechanism, controller, steps, record, n, t.nb, t.ne
        self.dev = _device_mech(mechanism)
        self.joints = [j for j in range(t.ne) if int(t.type[j]) in (0, 1)]
        self.slot = {j: i for i, j in enumerate(self.joints)}
        self.ctrl = _capi.CtrlHandle(self.dev, self.joints, K=None, N=0, Fd=np.zeros((n, len(self.joints))), n_ctrl=n)
        td = self.td = torch.device("cuda", torch.cuda.current_device())
        self.z0 = torch.empty((n, t.nb, 13), dtype=torch.float64, device=td)
        self.lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=td)
        self.st = torch.zeros(n, dtype=torch.int32, device=td)
        self.traj = torch.empty((n, steps, t.nb, 13), dtype=torch.float64, device=td) if record else None
        self.U = torch.zeros((n, len(self.joints)), dtype=torch.float64, device=td)
        self.graph = None
        self.zT = None

    def close(self):
        self.graph = None
        self.ctrl.close()

    def _steps(self, flags):
        """the horizon on the current stream, from self.z0; leaves the final state in self.zT"""
        torch, n = self.torch, self.n
        stream = torch.cuda.current_stream().cuda_stream
        st, lam, U = self.st, self.lam, self.U
        st.zero_(); lam.zero_()            # (CCLQR_ROLLOUT_CARRY_STATUS: the status array carries every instance's status from launch to launch -- a lost one stays frozen)
        z, zn = self.z0.clone(), torch.empty_like(self.z0)
        for k in range(1, self.steps + 1):
            if self.record:
                self.traj[:, k - 1] = z
            batch = BatchState(self.mechanism, z, k)
            self.controller.controlfunction(batch, self.controller, k)
            U.zero_()
            for j, u in batch.u.items():
                if j not in self.slot:
                    raise ValueError("setForce on a constraint without a degree of freedom")
                U[:, self.slot[j]] = u
            self.ctrl.set_feedforward(dev_ptr=U.data_ptr(), length=U.numel(), stream=stream)
            _capi.rollout_dev(self.dev, self.ctrl, n, 1, k, z.data_ptr(), lam.data_ptr(), 0, 0, 0, zn.data_ptr(), st.data_ptr(), stream,
                              flags=flags | _capi.ROLLOUT_CARRY_STATUS)
            z, zn = zn, z
        self.zT = z

    def run(self, z0):
        torch = self.torch
        self.z0.copy_(torch.from_numpy(np.ascontiguousarray(z0)))
        if getattr(self.controller.controlfunction, "capture", False):
            if self.graph is None:
                # one step outside the capture first: what the closure builds lazily (device copies of its tables) must exist before a capture opens
                self.controller.controlfunction(BatchState(self.mechanism, self.z0, 1), self.controller, 1)
                torch.cuda.synchronize()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    graph.capture_begin()
                    self._steps(_capi.ROLLOUT_NO_ALLOC)
                    graph.capture_end()
                torch.cuda.current_stream().wait_stream(side)
                self.graph = graph
            self.graph.replay()
        else:
            self._steps(0)
        zT = self.zT.cpu().numpy()
        status = self.st.cpu().numpy()
        trajh = self.traj.cpu().numpy() if self.record else np.zeros((self.n, 0, self.nb, 13))
        return zT, trajh, status


def _simulate_device_closure(mechanism, steps, controller, record, z0):
    """simulate! with a DEVICE closure (on_device): as _simulate_hosted, but nothing leaves HBM between the steps -- the closure reads a torch view of
    the state and returns tensors, its inputs reach the controller's feed-forward table device to device on the launch's stream, and the lost-instance
    bookkeeping (freeze at the last pose, at rest; status) is the library's (CCLQR_ROLLOUT_CARRY_STATUS).  One synchronisation, at the end.  With graph=True the
    run (buffers + captured graph) is kept on the controller and replayed by later calls of the same shape."""
    n = z0.shape[0]
    if not getattr(controller.controlfunction, "capture", False):
        run = _DeviceClosureRun(mechanism, steps, controller, record, n)
        try:
            return run.run(z0)
        finally:
            run.close()
    cache = controller.__dict__.setdefault("_device_closure_runs", {})
    key = (id(mechanism), id(controller.controlfunction), n, steps, bool(record))
    if key not in cache:
        for old in cache.values():          # one captured horizon per controller: a new shape replaces it
            old.close()
        cache.clear()
        cache[key] = _DeviceClosureRun(mechanism, steps, controller, record, n)
    return cache[key].run(z0)


def simulate(mechanism, tend_or_storage, controller, record=True, z0=None, fric=None, noise=None, noise_scale=None, noise_seed=None,
             first_instance=0):
    """simulate!(mechanism, tend::Real | storage::Storage, controller; record)  -> Storage

    z0 [n_inst][nb][13]: batch of initial states (default: the mechanism's current body states, one instance).
    fric [ne], noise [n_inst][steps] (injected samples) or noise_seed (device-side Philox-4x32 stream per instance), noise_scale:
    the friction/noise law of examples/trackingLQR_triple_cartpole.jl:93-111.  first_instance: global index of z0[0] when z0 is one
    rank's shard of a larger batch (the Philox stream of an instance is keyed by its global index).
    After the call the mechanism's bodies hold instance 0's final state (simulate! mutates the mechanism)."""
    if isinstance(tend_or_storage, Storage):
        steps = tend_or_storage.steps
    else:
        steps = int(math.ceil(tend_or_storage / mechanism.Δt))     # steps = 1:ceil(tend/Δt)
    nb = len(mechanism.bodies)
    z0 = mechanism.state()[None] if z0 is None else np.asarray(z0, dtype=np.float64).reshape(-1, nb, 13)
    dev = _device_mech(mechanism)
    if (noise is not None or noise_seed is not None) and noise_scale is None:
        noise_scale = 1.0
    if getattr(controller, "controlfunction", None) is not None:
        # a custom controlfunction (lqr.jl:14): the closure owns the whole law, as in the reference -- the built-in extras do not apply on top of it
        if fric is not None or noise is not None or noise_seed is not None:
            raise ValueError("fric / noise are options of the built-in laws; a custom controlfunction computes its own inputs")
        hosted = _simulate_device_closure if getattr(controller.controlfunction, "on_device", False) else _simulate_hosted
        zT, traj, status = hosted(mechanism, steps, controller, record, z0)
    else:
        ctrl = controller._ctrl_handle(dev, fric=fric, noise_scale=0.0 if noise_scale is None else noise_scale, noise_seed=noise_seed)
        try:
            zT, traj, status = _capi.rollout(dev, ctrl, z0, steps, k0=1, noise=noise, record=record, first_instance=first_instance)
        finally:
            ctrl.close()
    mechanism.set_state(zT[0])
    if isinstance(tend_or_storage, Storage) and record:
        tend_or_storage.z = traj
        tend_or_storage.n_inst = z0.shape[0]
        tend_or_storage.status, tend_or_storage.zT = status, zT
        return tend_or_storage
    return Storage(steps, nb, z0.shape[0], traj if record else np.zeros((z0.shape[0], 0, nb, 13)), status, zT)
