"""oracle/loops.py -- TEST INFRASTRUCTURE ONLY: dense-KKT restatement of one integrator step for mechanisms with CLOSED kinematic
loops (examples/lqr_deltabot.jl:25-33), which the tree solvers (oracle C code and the HIP kernels) do not take.

Same discretisation as oracle/cclqr_oracle.c: unknowns s = (v+, w+) per body and one multiplier per constraint row,

    d_b(s) - sum_j G_{j,b}(z_k)' lambda_j = 0,        g_j(x+, q+) = 0,

with the joint functions and their Jacobians taken from the C oracle itself (`orc_joint_blocks` = the joint_eval / Q_to_phi every tree
joint goes through), so that a statement proved here is a statement about that code.  A loop makes the constraint rows redundant
(the deltabot has 33 rows on 30 body coordinates), so the Newton step solves the KKT system in the minimum-norm least-squares sense
(`numpy.linalg.lstsq`); velocities are unique, multipliers are the minimum-norm representative.  The Newton matrix is a central
finite difference of the residual (63 unknowns: cheap), which only affects convergence speed, not the converged point.

Joint kinds: REVOLUTE / PRISMATIC (5 rows, as in the tree code) and FIXED_ORIENTATION (the three rotational rows of the same
evaluation; `FixedOrientation(origin, platform; qoffset)` of lqr_deltabot.jl:25)."""
import ctypes as C

import numpy as np

from . import orc

REVOLUTE, PRISMATIC, FIXED_ORIENTATION = 0, 1, 2
_dp = C.POINTER(C.c_double)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def rotx(a):
    return np.array([np.cos(a / 2), np.sin(a / 2), 0.0, 0.0])


def qmul(a, b):
    s = a[0] * b[0] - a[1:] @ b[1:]
    v = a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])
    return np.concatenate([[s], v])


def rot(q):
    s, x, y, z = q
    return np.array([[s * s + x * x - y * y - z * z, 2 * (x * y - s * z), 2 * (x * z + s * y)],
                     [2 * (x * y + s * z), s * s - x * x + y * y - z * z, 2 * (y * z - s * x)],
                     [2 * (x * z - s * y), 2 * (y * z + s * x), s * s - x * x - y * y + z * z]])


class Joint:
    def __init__(self, kind, parent, child, axis=(1.0, 0.0, 0.0), p1=(0, 0, 0), p2=(0, 0, 0), qoff=(1, 0, 0, 0)):
        self.kind, self.parent, self.child = kind, parent, child
        self.axis, self.p1, self.p2, self.qoff = (np.array(v, dtype=np.float64) for v in (axis, p1, p2, qoff))
        self.rows = 3 if kind == FIXED_ORIENTATION else 5


class LoopMechanism:
    """bodies: masses m [nb], inertias J [nb][3][3]; joints: list of Joint (parent -1 = origin); any graph, loops allowed"""

    def __init__(self, m, J, joints, dt=0.01, g=-9.81):
        self.m, self.J, self.joints, self.dt, self.g = np.asarray(m, float), np.asarray(J, float), joints, dt, g
        self.nb = len(self.m)
        self.nrows = sum(j.rows for j in joints)

    def _blocks(self, j, z):
        """g [rows], Ga, Gb [rows][6] of joint j at the poses of z [nb][13]"""
        a, b = j.parent, j.child
        g, Ga, Gb = np.zeros(5), np.zeros((5, 6)), np.zeros((5, 6))
        xa = np.ascontiguousarray(z[a, 0:3]) if a >= 0 else None
        qa = np.ascontiguousarray(z[a, 3:7]) if a >= 0 else None
        xb, qb = np.ascontiguousarray(z[b, 0:3]), np.ascontiguousarray(z[b, 3:7])
        kind = PRISMATIC if j.kind == FIXED_ORIENTATION else j.kind        # rows 2..4 of the prismatic evaluation = Rotational3
        orc.lib().orc_joint_blocks(C.c_int32(kind), _p(j.p1), _p(j.p2), _p(j.axis), _p(j.qoff), _p(xa), _p(qa), _p(xb), _p(qb), _p(g), _p(Ga), _p(Gb))
        if j.kind == FIXED_ORIENTATION:
            return g[2:], Ga[2:], Gb[2:]
        return g, Ga, Gb

    def constraints(self, z):
        return np.concatenate([self._blocks(j, z)[0] for j in self.joints])

    def input_wrenches(self, z, u):
        """joint inputs u [njoints] -> world force F [nb][3], body-frame torque tau [nb][3] (SURVEY 8a-bis 'Joint input', revolute/prismatic)"""
        F, tau = np.zeros((self.nb, 3)), np.zeros((self.nb, 3))
        for j, uj in zip(self.joints, u):
            if uj == 0.0 or j.kind == FIXED_ORIENTATION:
                continue
            a, b = j.parent, j.child
            Ra = rot(z[a, 3:7]) if a >= 0 else np.eye(3)
            Rb = rot(z[b, 3:7])
            f = j.axis / np.linalg.norm(j.axis) * uj          # parent frame
            fw = Ra @ f
            fb = Rb.T @ fw
            if j.kind == PRISMATIC:
                F[b] += fw; tau[b] += np.cross(j.p2, fb)
                if a >= 0:
                    F[a] -= fw; tau[a] -= np.cross(j.p1, f)
            else:
                tau[b] += fb
                if a >= 0:
                    tau[a] -= f
        return F, tau

    def _next(self, z, s):
        dt = self.dt
        zn = z.copy()
        for b in range(self.nb):
            w = s[b, 3:]
            sq = np.sqrt(4.0 / dt ** 2 - w @ w)
            zn[b, 0:3] = z[b, 0:3] + s[b, 0:3] * dt
            zn[b, 3:7] = qmul(z[b, 3:7], 0.5 * dt * np.concatenate([[sq], w]))
            zn[b, 7:13] = s[b]
        return zn

    def residual(self, z, s, lam, F, tau, Gk):
        dt = self.dt
        d = np.zeros((self.nb, 6))
        for b in range(self.nb):
            v1, w1, v2, w2 = z[b, 7:10], z[b, 10:13], s[b, 0:3], s[b, 3:6]
            J = self.J[b]
            sq1, sq2 = np.sqrt(4.0 / dt ** 2 - w1 @ w1), np.sqrt(4.0 / dt ** 2 - w2 @ w2)
            d[b, 0:3] = self.m[b] * ((v2 - v1) / dt + np.array([0, 0, -self.g])) - F[b]
            d[b, 3:6] = sq2 * (J @ w2) + np.cross(w2, J @ w2) - (sq1 * (J @ w1) - np.cross(w1, J @ w1)) - 2.0 * tau[b]
        o = 0
        for j, (Ga, Gb) in zip(self.joints, Gk):
            lj = lam[o:o + j.rows]
            d[j.child] -= Gb.T @ lj
            if j.parent >= 0:
                d[j.parent] -= Ga.T @ lj
            o += j.rows
        zn = self._next(z, s)
        return np.concatenate([d.ravel(), self.constraints(zn)])

    def step(self, z, lam, u, tol=1e-10, maxit=50):
        """one integrator step; returns (z_next, lam, newton iterations); raises if Newton does not converge"""
        z = np.asarray(z, float).reshape(self.nb, 13)
        F, tau = self.input_wrenches(z, u)
        Gk = [self._blocks(j, z)[1:] for j in self.joints]
        ns = 6 * self.nb
        x = np.concatenate([z[:, 7:13].ravel(), np.asarray(lam, float)])
        f = lambda xx: self.residual(z, xx[:ns].reshape(self.nb, 6), xx[ns:], F, tau, Gk)
        for it in range(1, maxit + 1):
            r = f(x)
            if np.linalg.norm(r) < tol:
                return self._next(z, x[:ns].reshape(self.nb, 6)), x[ns:], it - 1
            h = 1e-6
            Jm = np.zeros((len(r), len(x)))
            for i in range(len(x)):
                e = np.zeros(len(x)); e[i] = h
                Jm[:, i] = (f(x + e) - f(x - e)) / (2 * h)
            dx = np.linalg.lstsq(Jm, r, rcond=1e-10)[0]      # minimum-norm step: the KKT matrix of a loop is rank deficient
            x = x - dx
        raise RuntimeError("dense-KKT Newton did not converge: |f| = %g" % np.linalg.norm(f(x)))


def state_error(z, z0):
    """error coordinates per body x, v, q~ = vec(q0^-1 q), w (lqr.jl:92-103) of z [nb][13] about z0"""
    e = np.zeros((len(z), 12))
    for b in range(len(z)):
        qe = qmul(np.concatenate([[z0[b, 3]], -z0[b, 4:7]]), z[b, 3:7])
        e[b] = np.concatenate([z[b, 0:3] - z0[b, 0:3], z[b, 7:10] - z0[b, 7:10], qe[1:], z[b, 10:13] - z0[b, 10:13]])
    return e.ravel()


def projected_linear_model(lm, z, u, ctrl, h=1e-6):
    """A' [mx][mx], D [mx][mu]: Jacobians of the constrained one-step map of `lm` about (z, u) in the error coordinates of lqr.jl:92-103, by
    central differences (the quantity the recursion of lqr.jl:151-170 works with once the multipliers are eliminated; for a tree it equals
    A - Bl (G Bl)^-1 G A, Bu - Bl (G Bl)^-1 G Bu of the reference's linearsystem).  ctrl: indices of the controlled joints."""
    nb, mx = lm.nb, 12 * lm.nb
    lam0 = np.zeros(lm.nrows)
    step = lambda zz, uu: lm.step(zz, lam0, uu, tol=1e-13)[0]
    znom = step(z, u)
    Ap, D = np.zeros((mx, mx)), np.zeros((mx, len(ctrl)))
    for col in range(mx + len(ctrl)):
        out = []
        for sgn in (+1.0, -1.0):
            zz, uu = z.copy(), u.copy()
            if col < mx:
                b, e = divmod(col, 12)
                if e < 3:
                    zz[b, e] += sgn * h
                elif e < 6:
                    zz[b, 7 + e - 3] += sgn * h
                elif e < 9:
                    dq = np.array([np.sqrt(1 - h * h), 0.0, 0.0, 0.0]); dq[1 + e - 6] = sgn * h
                    zz[b, 3:7] = qmul(z[b, 3:7], dq)
                else:
                    zz[b, 10 + e - 9] += sgn * h
            else:
                uu[ctrl[col - mx]] += sgn * h
            out.append(state_error(step(zz, uu), znom))
        v = (out[0] - out[1]) / (2 * h)
        if col < mx:
            Ap[:, col] = v
        else:
            D[:, col - mx] = v
    return Ap, D


def place(z, a, b, p1, p2, dq):
    """setPosition!(a, b; p1, p2, Δq) (examples/lqr_deltabot.jl:37-41): q_b = q_a Δq ; x_b = x_a + R(q_a) p1 - R(q_b) p2"""
    xa = z[a, 0:3] if a >= 0 else np.zeros(3)
    qa = z[a, 3:7] if a >= 0 else np.array([1.0, 0, 0, 0])
    qb = qmul(qa, dq)
    z[b, 3:7] = qb
    z[b, 0:3] = xa + rot(qa) @ np.asarray(p1, float) - rot(qb) @ np.asarray(p2, float)


def box_inertia(x, y, z, m):
    return m / 12.0 * np.diag([y * y + z * z, x * x + z * z, x * x + y * y])


def deltabot():
    """the mechanism, pose and feed-forward of examples/lqr_deltabot.jl:7-53 (numbers from the script; bodies in the script's `links`
    order lowerlegl, lowerlegr, upperlegl, upperlegr, platform; joints platl, platr, floor-left, floor-right, platform
    orientation, kneel, kneer -- `constraints[1:2]` = platl, platr are the controlled ones)"""
    L = 1.0
    ax = (1.0, 0.0, 0.0)
    pll, pul, pp = np.array([0, 0, L / 2]), np.array([0, 0, L / 4]), np.array([0, 0, L / 4 * np.sqrt(2)])
    m = [L, L, L / 2, L / 2, L / 2 * np.sqrt(2)]                                               # Box(x, y, z, m): :18-22
    J = [box_inertia(0.1, 0.1, L, L)] * 2 + [box_inertia(0.1, 0.1, L / 2, L / 2)] * 2 + [box_inertia(0.1, 0.1, L / 2 * np.sqrt(2), L / 2 * np.sqrt(2))]
    LL, LR, UL, UR, PL = 0, 1, 2, 3, 4
    joints = [Joint(REVOLUTE, PL, UL, ax, p1=pp, p2=pul),                                       # platl :28
              Joint(REVOLUTE, PL, UR, ax, p1=-pp, p2=pul),                                      # platr :29
              Joint(REVOLUTE, -1, LL, ax, p2=-pll), Joint(REVOLUTE, -1, LR, ax, p2=-pll),       # floorlr :25
              Joint(FIXED_ORIENTATION, -1, PL, ax, qoff=rotx(np.pi / 2)),                       # floorlr :25
              Joint(REVOLUTE, LL, UL, ax, p1=pll, p2=-pul), Joint(REVOLUTE, LR, UR, ax, p1=pll, p2=-pul)]   # kneel, kneer :26-27
    mech = LoopMechanism(m, J, joints, dt=0.01, g=-9.81)                                        # :36
    z = np.zeros((5, 13)); z[:, 3] = 1.0
    place(z, -1, LL, (0, 0, 0), -pll, rotx(np.pi / 4))                                          # :37
    place(z, -1, LR, (0, 0, 0), -pll, rotx(-np.pi / 4))                                         # :38
    place(z, LL, UL, pll, -pul, rotx(-np.pi / 2))                                               # :39
    place(z, LR, UR, pll, -pul, rotx(np.pi / 2))                                                # :40
    place(z, UL, PL, pul, pp, rotx(3 * np.pi / 4))                                              # :41
    u = np.zeros(len(joints)); u[0], u[1] = 6.7879484, -6.7879484                               # Fτd :53
    return mech, z, u


def from_tables(t):
    """the LoopMechanism of a host-mirror mechanism's tables (constrainedcontrol.jl_amd/mechanism.py MechTables: bodies and joints in the caller's
    order, parent -1 = origin) -- so that any mechanism the tests build through the mirror has its dense-KKT reference"""
    joints = [Joint(int(t.type[j]), int(t.parent[j]), int(t.child[j]), t.axis[j], t.p1[j], t.p2[j], t.qoff[j]) for j in range(t.ne)]
    return LoopMechanism(t.mass, t.inertia.reshape(t.nb, 3, 3), joints, dt=t.dt, g=t.g)



# ------------------------------------------------------------------------------------------------ flop count of a loop mechanism's step (bench.py roofline)
def step_with_the_parity_rule(lm, z, lam, u, eps=1e-10, maxit=100, line_maxit=10):
    """LoopMechanism.step with the stopping rule and line search of the tree oracle's newton() (stop when ||f|| < eps AND the step taken
    alpha ||dx|| < eps; halve while ||f|| grows, at most 10 times) -- the rule the device kernels follow -- so that its iteration and evaluation COUNTS
    are the ones a step of the reference algorithm takes.  Returns (z_next, lam, iterations, residual evaluations).  Same converged point as step()."""
    z = np.asarray(z, float).reshape(lm.nb, 13)
    F, tau = lm.input_wrenches(z, u)
    Gk = [lm._blocks(j, z)[1:] for j in lm.joints]
    ns = 6 * lm.nb
    x = np.concatenate([z[:, 7:13].ravel(), np.asarray(lam, float)])
    f = lambda xx: lm.residual(z, xx[:ns].reshape(lm.nb, 6), xx[ns:], F, tau, Gk)
    r = f(x)
    n0, evals = np.linalg.norm(r), 1
    for it in range(1, maxit + 1):
        h = 1e-6
        Jm = np.zeros((len(r), len(x)))
        for i in range(len(x)):
            e = np.zeros(len(x)); e[i] = h
            Jm[:, i] = (f(x + e) - f(x - e)) / (2 * h)
        dx = np.linalg.lstsq(Jm, r, rcond=1e-10)[0]
        evals += 1                                   # the evaluation WITH Jacobians of this iteration (the differences above stand in for the analytic blocks)
        alpha = 1.0
        for ls in range(line_maxit + 1):
            rt = f(x - alpha * dx)
            evals += 1
            n1 = np.linalg.norm(rt)
            if n1 > n0 and ls < line_maxit:
                alpha *= 0.5
            else:
                break
        x, r = x - alpha * dx, rt
        if n1 < eps and alpha * np.linalg.norm(dx) < eps:
            return lm._next(z, x[:ns].reshape(lm.nb, 6)), x[ns:], it, evals
        n0 = n1
    raise RuntimeError("dense-KKT Newton (parity rule) did not converge")


def flops_model(lm):
    """fp64 flops of the pieces of one Newton iteration on a loop mechanism, by the SAME bookkeeping as the instrumented tree oracle (cclqr_oracle.c FL(...)):
    the joint evaluations are MEASURED on liborc_flops.so (orc_constraints = the residual-only evaluation, orc_joint_blocks = the evaluation with Jacobians
    + two Q_to_phi of 120), the body terms are the constants of residual() there (next_pose 20 + residual 40 + two 3x3 products 30 + two cross products 18 =
    108, Jacobian + 30 + 54), and the linear solve -- which the tree oracle does by the body/joint LDU and a loop cannot -- is priced as what it is
    algebraically: D_b^-1 per body, W = G_v D^-1 per (joint, side), one 5 x 5 block of S = G_v D^-1 G_k' per ordered pair of joints that share a body,
    a dense LU of the m x m Schur complement (2/3 m^3 + 2 m^2, m = constraint rows) and the body back-substitution.  -> dict of counts"""
    L = orc.lib(True)
    z = np.zeros((lm.nb, 13)); z[:, 3] = 1.0
    f_eval0 = f_eval1 = 0.0
    sides = 0
    pair_blocks = 0
    around = [0] * lm.nb
    for j in lm.joints:
        hp = j.parent >= 0
        sides += 2 if hp else 1
        around[j.child] += 1
        if hp:
            around[j.parent] += 1
        kind = PRISMATIC if j.kind == FIXED_ORIENTATION else j.kind
        xa = np.zeros(3); qa = np.array([1.0, 0, 0, 0]); xb = np.array([0.1, 0.2, 0.3]); qb = rotx(0.3)
        g, Ga, Gb = np.zeros(5), np.zeros((5, 6)), np.zeros((5, 6))
        orc.flops_reset()
        L.orc_joint_blocks(C.c_int32(kind), _p(j.p1), _p(j.p2), _p(j.axis), _p(j.qoff), _p(xa) if hp else None, _p(qa) if hp else None, _p(xb), _p(qb), _p(g), _p(Ga), _p(Gb))
        fjb = orc.flops_get()
        frac = j.rows / 5.0                                              # a FixedOrientation keeps three of the five rows
        f_eval1 += frac * ((fjb - 240.0) + 36.0 * (2 if hp else 1) + 30.0)  # joint_eval(jac) + Q_to_omega per side + assembly, as residual(jac = 1) does
        f_eval0 += frac * (fjb - 240.0 - 105.0)                          # joint_eval(jac = 0): FL(15) instead of FL(120), same rotations / quaternion products
        f_eval0 += frac * (120.0 if hp else 60.0); f_eval1 += frac * (120.0 if hp else 60.0)      # d -= G_k' lambda
    pair_blocks = sum(n * n for n in around)
    m = lm.nrows
    out = {
        "residual_evaluation": 108.0 * lm.nb + f_eval0 + 2.0 * (6 * lm.nb + m) + 1.0,
        "evaluation_with_jacobians": (108.0 + 84.0) * lm.nb + f_eval1 + 2.0 * (6 * lm.nb + m) + 1.0,
        "trial_point": 2.0 * (6 * lm.nb + m),
        "schur_assembly": 45.0 * lm.nb + 105.0 * sides + 300.0 * pair_blocks + 60.0 * sides,
        "dense_lu_solve": 2.0 / 3.0 * m ** 3 + 2.0 * m * m,
        "back_substitution": 60.0 * sides + 42.0 * lm.nb,
        "step_norm": 2.0 * (6 * lm.nb + m),
        "rows_m": m, "ordered_joint_pairs_sharing_a_body": pair_blocks,
    }
    out["newton_iteration_without_line_search"] = out["evaluation_with_jacobians"] + out["schur_assembly"] + out["dense_lu_solve"] + out["back_substitution"] + out["step_norm"]
    return out


def flops_per_step(lm, z, lam, u, ctrl_flops=0.0):
    """counted iterations / evaluations of step_with_the_parity_rule x flops_model: the algorithmic fp64 flops of ONE step from (z, lam) under inputs u;
    returns (flops, z_next, lam_next, iterations).  Fixed per step: the joint Jacobians at the current knot (one evaluation with Jacobians of every joint), the
    input wrenches (30 per driven joint, as apply_input) and the control law (2 mu 12 nb, passed in)."""
    fm = flops_model(lm)
    zn, ln, its, evals = step_with_the_parity_rule(lm, z, lam, u)
    fixed = fm["evaluation_with_jacobians"] - (108.0 + 84.0) * lm.nb + 30.0 * float(np.count_nonzero(u)) + ctrl_flops
    # evals = 1 (initial) + per iteration: 1 with Jacobians + (1 + halvings) residual-only trials
    trials = evals - 1 - its
    fl = fixed + fm["residual_evaluation"] + its * fm["newton_iteration_without_line_search"] + trials * (fm["residual_evaluation"] + fm["trial_point"])
    return fl, zn, ln, its
