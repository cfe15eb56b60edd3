"""Host-side description of a mechanism: the subset of ConstrainedDynamics' data model the
reference's LQR path touches.  Names follow the Julia call sites:

    Origin{Float64}()                              examples/lqr_cartpole.jl:20
    Box(x, y, z, m)                                examples/lqr_cartpole.jl:21-22
    EqualityConstraint(Prismatic(origin, cart, ey))            :25
    EqualityConstraint(Revolute(cart, pole, ex; p2=-p2))       :26
    Mechanism(origin, links, constraints, g=-9.81[, Δt=0.01])  :32
    setPosition!(origin, cart, Δx=[0;0.5;0])                   :33   -> setPosition(...)
    setPosition!(cart, pole, p2=-p2, Δq=Quaternion(RotX(0.2))) :34
    getid.(links)                                              :41

This is description only (no dynamics arithmetic): it flattens to the `cclqr_mech_desc` tables the
C-ABI takes (include/cclqr.h).  Bodies get ids 1..Nb, joints Nb+1..Nb+Ne (as implied by
geteqconstraint(mechanism, 6..8) with Nb = 4, examples/trackingLQR_triple_cartpole.jl:109-111).
"""
import math

import numpy as np

REVOLUTE, PRISMATIC, FIXED_ORIENTATION = 0, 1, 2


# ------------------------------------------------------------------ quaternions (scalar first)
def Quaternion(*a):
    """Quaternion(RotX(θ)) / Quaternion(s, x, y, z) / Quaternion([s,x,y,z])"""
    if len(a) == 1:
        q = np.asarray(a[0], dtype=np.float64).reshape(4)
    else:
        q = np.asarray(a, dtype=np.float64).reshape(4)
    return q.copy()


def one_quaternion():
    return np.array([1.0, 0.0, 0.0, 0.0])


def RotX(t):
    return np.array([math.cos(t / 2), math.sin(t / 2), 0.0, 0.0])


def RotY(t):
    return np.array([math.cos(t / 2), 0.0, math.sin(t / 2), 0.0])


def RotZ(t):
    return np.array([math.cos(t / 2), 0.0, 0.0, math.sin(t / 2)])


def qmul(a, b):
    s1, v1, s2, v2 = a[0], a[1:], b[0], b[1:]
    return np.concatenate([[s1 * s2 - v1 @ v2], s1 * v2 + s2 * v1 + np.cross(v1, v2)])


def qconj(a):
    return np.array([a[0], -a[1], -a[2], -a[3]])


def vrotate(p, q):
    p = np.asarray(p, dtype=np.float64)
    return qmul(qmul(q, np.concatenate([[0.0], p])), qconj(q))[1:]


def rpy_quaternion(r, p, y):
    """URDF fixed-axis roll-pitch-yaw -> quaternion (Rz(y) Ry(p) Rx(r))"""
    return qmul(RotZ(y), qmul(RotY(p), RotX(r)))


# ------------------------------------------------------------------ components
class State:
    def __init__(self):
        self.xc = np.zeros(3)
        self.qc = one_quaternion()
        self.vc = np.zeros(3)
        self.ωc = np.zeros(3)

    @property
    def wc(self):
        return self.ωc


class Origin:
    def __init__(self, name="origin"):
        self.id = 0
        self.name = name
        self.state = State()


class Body:
    def __init__(self, m, J, name=""):
        self.m = float(m)
        self.J = np.asarray(J, dtype=np.float64).reshape(3, 3)
        self.id = -1
        self.name = name
        self.state = State()


def Box(x, y, z, m, name=""):
    """Box(x,y,z,m): J = m/12 diag(y²+z², x²+z², x²+y²)  (SURVEY 8a-bis 'Box inertia')"""
    J = m / 12.0 * np.diag([y * y + z * z, x * x + z * z, x * x + y * y])
    return Body(m, J, name)


class _Joint:
    def __init__(self, kind, body1, body2, axis, p1=None, p2=None, qoffset=None):
        self.kind = kind
        self.body1, self.body2 = body1, body2
        self.axis = np.asarray(axis, dtype=np.float64).reshape(3)
        self.p1 = np.zeros(3) if p1 is None else np.asarray(p1, dtype=np.float64).reshape(3)
        self.p2 = np.zeros(3) if p2 is None else np.asarray(p2, dtype=np.float64).reshape(3)
        self.qoffset = one_quaternion() if qoffset is None else Quaternion(qoffset)


def Revolute(body1, body2, axis, p1=None, p2=None, qoffset=None):
    return _Joint(REVOLUTE, body1, body2, axis, p1, p2, qoffset)


def Prismatic(body1, body2, axis, p1=None, p2=None, qoffset=None):
    return _Joint(PRISMATIC, body1, body2, axis, p1, p2, qoffset)


def FixedOrientation(body1, body2, qoffset=None):
    """FixedOrientation(body1, body2; qoffset) — the three rotational rows only (examples/lqr_deltabot.jl:25); makes the
    mechanism a closed-loop one for the device (cclqr.h CCLQR_FIXED_ORIENTATION)"""
    return _Joint(FIXED_ORIENTATION, body1, body2, (1.0, 0.0, 0.0), None, None, qoffset)


class EqualityConstraint:
    def __init__(self, joint, name=""):
        self.joint = joint
        self.parentid = None
        self.childid = None
        self.id = -1
        self.name = name

    def __len__(self):
        return 3 if self.joint.kind == FIXED_ORIENTATION else 5  # Revolute and Prismatic both remove 5 DoF


def getid(c):
    return c.id


class MechTables:
    """flat tables = cclqr_mech_desc (include/cclqr.h)"""

    def __init__(self, nb, ne, dt, g, mass, inertia, parent, child, type, p1, p2, axis, qoff):
        self.nb, self.ne, self.dt, self.g = int(nb), int(ne), float(dt), float(g)
        self.mass = np.ascontiguousarray(mass, dtype=np.float64).reshape(nb)
        self.inertia = np.ascontiguousarray(inertia, dtype=np.float64).reshape(nb, 9)
        self.parent = np.ascontiguousarray(parent, dtype=np.int32).reshape(ne)
        self.child = np.ascontiguousarray(child, dtype=np.int32).reshape(ne)
        self.type = np.ascontiguousarray(type, dtype=np.int32).reshape(ne)
        self.p1 = np.ascontiguousarray(p1, dtype=np.float64).reshape(ne, 3)
        self.p2 = np.ascontiguousarray(p2, dtype=np.float64).reshape(ne, 3)
        self.axis = np.ascontiguousarray(axis, dtype=np.float64).reshape(ne, 3)
        self.qoff = np.ascontiguousarray(qoff, dtype=np.float64).reshape(ne, 4)

    @property
    def mx(self):
        return 12 * self.nb

    @property
    def ml(self):
        return 5 * self.ne


class Mechanism:
    """Mechanism(origin, bodies, eqconstraints; g=-9.81, Δt=0.01)"""

    def __init__(self, origin, bodies=None, eqconstraints=None, g=-9.81, dt=None, floating=False, **kw):
        if "Δt" in kw:
            dt = kw.pop("Δt")
        if isinstance(origin, str):      # Mechanism(path, floating=false, g=0.0)   examples/lqr_sawyer.jl:9
            m = mechanism_from_urdf_tables(parse_urdf(origin, keep_fixed=True), floating=floating, g=g, dt=0.01 if dt is None else dt)
            self.__dict__.update(m.__dict__)
            return
        if kw:
            raise TypeError("unexpected keyword(s): %s" % list(kw))
        self.origin = origin
        self.bodies = list(bodies)
        self.eqconstraints = list(eqconstraints)
        self.g = float(g)
        self.Δt = 0.01 if dt is None else float(dt)
        nb = len(self.bodies)
        for i, b in enumerate(self.bodies):
            b.id = i + 1
        for j, e in enumerate(self.eqconstraints):
            e.id = nb + j + 1
            e.parentid = e.joint.body1.id
            e.childid = e.joint.body2.id
        self._byname = {e.name: e for e in self.eqconstraints if e.name}
        self.tables()  # validates the topology

    @property
    def dt(self):
        return self.Δt

    def geteqconstraint(self, id_or_name):
        if isinstance(id_or_name, str):
            return self._byname[id_or_name]
        return self.eqconstraints[int(id_or_name) - len(self.bodies) - 1]

    def getbody(self, id):
        return self.bodies[int(id) - 1]

    def joint_index(self, eqcid):
        j = int(eqcid) - len(self.bodies) - 1
        if not 0 <= j < len(self.eqconstraints):
            raise IndexError("no equality constraint with id %r" % (eqcid,))
        return j

    def tables(self):
        nb, ne = len(self.bodies), len(self.eqconstraints)
        parent = [e.joint.body1.id - 1 for e in self.eqconstraints]
        child = [e.joint.body2.id - 1 for e in self.eqconstraints]
        # closed loops (examples/lqr_deltabot.jl:25-33): more joints than bodies, a body that is the child of two joints, or a
        # FixedOrientation constraint.  The device rolls them out (rollout_loop.hip); LQR / TrackingLQR on them are built from the projected
        # linear model (cclqr_linearize_projected + the recursion with no multipliers left, lqr.py).
        self.has_loops = ne != nb or sorted(child) != list(range(nb)) or any(e.joint.kind == FIXED_ORIENTATION for e in self.eqconstraints)
        if ne < nb or not set(range(nb)) <= set(child) | {p for p in parent if p >= 0}:
            raise ValueError("every body must hang off at least one joint (Nb=%d, Ne=%d)" % (nb, ne))
        return MechTables(nb, ne, self.Δt, self.g, [b.m for b in self.bodies], [b.J.reshape(9) for b in self.bodies], parent, child,
                          [e.joint.kind for e in self.eqconstraints], [e.joint.p1 for e in self.eqconstraints],
                          [e.joint.p2 for e in self.eqconstraints], [e.joint.axis for e in self.eqconstraints],
                          [e.joint.qoffset for e in self.eqconstraints])

    def state(self):
        """current body states as z[nb][13] = x(3) q(4) v(3) ω(3)"""
        z = np.zeros((len(self.bodies), 13))
        for i, b in enumerate(self.bodies):
            z[i, 0:3], z[i, 3:7], z[i, 7:10], z[i, 10:13] = b.state.xc, b.state.qc, b.state.vc, b.state.ωc
        return z

    def set_state(self, z):
        z = np.asarray(z, dtype=np.float64).reshape(len(self.bodies), 13)
        for i, b in enumerate(self.bodies):
            b.state.xc, b.state.qc, b.state.vc, b.state.ωc = z[i, 0:3].copy(), z[i, 3:7].copy(), z[i, 7:10].copy(), z[i, 10:13].copy()


def setPosition(body1, body2=None, p1=None, p2=None, Δx=None, Δq=None, x=None, q=None):
    """setPosition!(body; x, q)  /  setPosition!(body1, body2; p1, p2, Δx, Δq):
    q2 = q1 ⊗ Δq ; x2 = x1 + R(q1)(p1 + Δx) − R(q2) p2      (cross-checked there)"""
    if body2 is None:
        if x is not None:
            body1.state.xc = np.asarray(x, dtype=np.float64).reshape(3).copy()
        if q is not None:
            body1.state.qc = Quaternion(q)
        return
    p1 = np.zeros(3) if p1 is None else np.asarray(p1, dtype=np.float64)
    p2 = np.zeros(3) if p2 is None else np.asarray(p2, dtype=np.float64)
    Δx = np.zeros(3) if Δx is None else np.asarray(Δx, dtype=np.float64)
    Δq = one_quaternion() if Δq is None else Quaternion(Δq)
    q1, x1 = body1.state.qc, body1.state.xc
    q2 = qmul(q1, Δq)
    body2.state.qc = q2
    body2.state.xc = x1 + vrotate(p1 + Δx, q1) - vrotate(p2, q2)


def setVelocity(body, v=None, ω=None, **kw):
    """setVelocity!(body; v, ω) — defaults zero both (examples/trackingLQR_triple_cartpole.jl:144-147)"""
    if "w" in kw:
        ω = kw["w"]
    body.state.vc = np.zeros(3) if v is None else np.asarray(v, dtype=np.float64).reshape(3).copy()
    body.state.ωc = np.zeros(3) if ω is None else np.asarray(ω, dtype=np.float64).reshape(3).copy()


def setJointPosition(mech, eqc, θ):
    """setPosition!(mech, eqc, [θ]) for a 1-DoF joint (examples/lqr_sawyer.jl:11-14): place the child
    relative to its parent at joint coordinate θ (angle for Revolute, offset for Prismatic); descendants
    are re-placed by calling this in root-to-leaf order."""
    j = eqc.joint
    θ = float(np.asarray(θ).reshape(-1)[0])
    a = j.axis / np.linalg.norm(j.axis)
    if j.kind == REVOLUTE:
        dq = qmul(np.concatenate([[math.cos(θ / 2)], math.sin(θ / 2) * a]), j.qoffset)
        setPosition(j.body1, j.body2, p1=j.p1, p2=j.p2, Δq=dq)
    else:
        setPosition(j.body1, j.body2, p1=j.p1, p2=j.p2, Δx=θ * a, Δq=j.qoffset)


def joint_position_states(mech, θ):
    """batch form of `setPosition!(mech, eqc, [θ])` applied to every joint of a TREE mechanism in root-to-leaf order (the loop of
    examples/lqr_sawyer.jl:11-14 for n poses at once): θ [n][ne] joint coordinates in the order of mech.eqconstraints -> z [n][nb][13]
    at rest.  Same arithmetic as setJointPosition / setPosition, vectorised over the batch (workload generator of bench.py)."""
    θ = np.asarray(θ, dtype=np.float64).reshape(-1, len(mech.eqconstraints))
    n, nb = θ.shape[0], len(mech.bodies)
    index = {id(b): i for i, b in enumerate(mech.bodies)}
    z = np.zeros((n, nb, 13))
    z[:, :, 3] = 1.0

    def bq(a, b):          # batched quaternion product, either operand [4] or [n][4]
        a, b = np.broadcast_to(a, (n, 4)), np.broadcast_to(b, (n, 4))
        w = a[:, 0] * b[:, 0] - np.einsum("ij,ij->i", a[:, 1:], b[:, 1:])
        v = a[:, 0:1] * b[:, 1:] + b[:, 0:1] * a[:, 1:] + np.cross(a[:, 1:], b[:, 1:])
        return np.concatenate([w[:, None], v], axis=1)

    def brot(p, q):        # R(q) p, p [3] or [n][3]
        p = np.broadcast_to(p, (n, 3))
        t = 2.0 * np.cross(q[:, 1:], p)
        return p + q[:, 0:1] * t + np.cross(q[:, 1:], t)

    done = set()
    pending = list(enumerate(mech.eqconstraints))
    while pending:
        rest = []
        for k, e in pending:
            j = e.joint
            pa = index.get(id(j.body1), -1)
            if pa >= 0 and pa not in done:
                rest.append((k, e))
                continue
            ch = index[id(j.body2)]
            if pa >= 0:
                q1, x1 = z[:, pa, 3:7], z[:, pa, 0:3]
            else:
                q1, x1 = np.broadcast_to(j.body1.state.qc, (n, 4)), np.broadcast_to(j.body1.state.xc, (n, 3))
            a = j.axis / np.linalg.norm(j.axis)
            if j.kind == REVOLUTE:
                dq = bq(np.concatenate([np.cos(θ[:, k:k + 1] / 2), np.sin(θ[:, k:k + 1] / 2) * a[None]], axis=1), j.qoffset)
                dx = np.zeros((n, 3))
            else:
                dq = np.broadcast_to(j.qoffset, (n, 4))
                dx = θ[:, k:k + 1] * a[None]
            q2 = bq(q1, dq)
            z[:, ch, 3:7] = q2
            z[:, ch, 0:3] = x1 + brot(j.p1 + dx, q1) - brot(j.p2, q2)
            done.add(ch)
        if len(rest) == len(pending):
            raise ValueError("joint_position_states: the mechanism is not a tree hung off the origin")
        pending = rest
    return z


# ------------------------------------------------------------------ URDF subset: Mechanism(path, floating=false, g=0.0)
def parse_urdf(path, keep_fixed=False):
    """links (mass, COM offset, inertia about the COM in the link frame) and revolute/prismatic joints of a URDF file.
    Returns a plain dict of numbers (also the format of tests/golden/sawyer_arm_tables.json).  keep_fixed: `fixed` joints are returned too
    (type "fixed"; mechanism_from_urdf_tables lumps the links they hold together, urdf_lump_fixed) instead of being refused."""
    import xml.etree.ElementTree as ET
    root = ET.parse(path).getroot()

    def vec(s, n=3):
        v = [float(x) for x in s.split()] if s else [0.0] * n
        assert len(v) == n
        return v

    links, joints = {}, []
    for ln in root.findall("link"):
        ine = ln.find("inertial")
        if ine is None:
            links[ln.get("name")] = dict(mass=0.0, com=[0.0, 0.0, 0.0], rpy=[0.0, 0.0, 0.0], inertia=[0.0] * 6)
            continue
        org = ine.find("origin")
        I = ine.find("inertia")
        links[ln.get("name")] = dict(mass=float(ine.find("mass").get("value")), com=vec(org.get("xyz") if org is not None else None),
                                     rpy=vec(org.get("rpy") if org is not None else None),
                                     inertia=[float(I.get(k)) for k in ("ixx", "ixy", "ixz", "iyy", "iyz", "izz")])
    for jn in root.findall("joint"):
        typ = jn.get("type")
        if typ not in ("revolute", "continuous", "prismatic") and not (keep_fixed and typ == "fixed"):
            raise ValueError("URDF joint type %r is outside the supported subset (1-DoF joints)" % typ)
        org = jn.find("origin")
        ax = jn.find("axis")
        joints.append(dict(name=jn.get("name"), type=typ if typ in ("prismatic", "fixed") else "revolute", parent=jn.find("parent").get("link"),
                           child=jn.find("child").get("link"), xyz=vec(org.get("xyz") if org is not None else None),
                           rpy=vec(org.get("rpy") if org is not None else None), axis=vec(ax.get("xyz")) if ax is not None else [1.0, 0.0, 0.0]))
    return dict(links=links, joints=joints)


def _rotmat(q):
    return np.array([vrotate(e, q) for e in np.eye(3)]).T


def _link_inertia(L):
    """3x3 inertia about the COM in the LINK frame (the URDF gives it in the inertial frame, rotated by the inertial origin's rpy)"""
    I = np.array([[L["inertia"][0], L["inertia"][1], L["inertia"][2]], [L["inertia"][1], L["inertia"][3], L["inertia"][4]],
                  [L["inertia"][2], L["inertia"][4], L["inertia"][5]]])
    if any(abs(a) > 0 for a in L["rpy"]):
        Rm = _rotmat(rpy_quaternion(*L["rpy"]))
        I = Rm @ I @ Rm.T
    return I


def urdf_lump_fixed(tab):
    """parse_urdf(keep_fixed=True) tables -> tables without `fixed` joints: every link held by a fixed joint is lumped into the link it is fixed
    to (one rigid body: summed mass, common COM, inertias moved by the parallel-axis theorem), and the joints that hung off it are re-anchored
    on the lumped link (their origin transforms composed; the composed rotation is kept as a quaternion, key "quat").  The motion of the
    remaining bodies is that of the original mechanism.  (The reference's dependency keeps a fixed joint as a 6-row constraint between two
    bodies; the hot path here has 5-row joints only -- SURVEY 8f-2.)"""
    links = {k: dict(v) for k, v in tab["links"].items()}
    joints = [dict(j) for j in tab["joints"]]
    for j in joints:
        j.setdefault("quat", list(rpy_quaternion(*j["rpy"])))
    S = lambda d: float(d @ d) * np.eye(3) - np.outer(d, d)
    while True:
        # a fixed joint whose child carries no further fixed joint below it: lump leaf-first so that every transform is used once
        fixed = [j for j in joints if j["type"] == "fixed"]
        if not fixed:
            break
        j = next(f for f in fixed if not any(g["type"] == "fixed" and g["parent"] == f["child"] for g in joints))
        P, Cn = links[j["parent"]], links[j["child"]]
        q = np.asarray(j["quat"], dtype=np.float64)
        Rpc, xyz = _rotmat(q), np.asarray(j["xyz"], dtype=np.float64)
        mp, mc = P["mass"], Cn["mass"]
        cp, cc = np.asarray(P["com"], dtype=np.float64), xyz + Rpc @ np.asarray(Cn["com"], dtype=np.float64)
        M = mp + mc
        com = (mp * cp + mc * cc) / M if M > 0 else cp
        I = _link_inertia(P) + mp * S(cp - com) + Rpc @ _link_inertia(Cn) @ Rpc.T + mc * S(cc - com)
        links[j["parent"]] = dict(mass=M, com=list(com), rpy=[0.0, 0.0, 0.0], inertia=[I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]])
        for g in joints:                      # what hung off the lumped link now hangs off its holder
            if g is not j and g["parent"] == j["child"]:
                g["parent"] = j["parent"]
                g["xyz"] = list(xyz + Rpc @ np.asarray(g["xyz"], dtype=np.float64))
                g["quat"] = list(qmul(q, np.asarray(g["quat"], dtype=np.float64)))
        del links[j["child"]]
        joints.remove(j)
    return dict(links=links, joints=joints)


def mechanism_from_urdf_tables(tab, floating=False, g=-9.81, dt=0.01):
    """Build the Mechanism from parse_urdf() output.  floating=false welds the root link to the origin (SURVEY 8a-bis 'URDF').
    Body frame = link frame orientation with its origin at the COM; joint vertices are shifted to COM frames; inertia tensors
    are full 3x3.  The bodies are placed at the zero pose."""
    if floating:
        raise NotImplementedError("floating base needs a 6-DoF root joint, outside the 1-DoF scope (lqr.jl:1-2)")
    if any(j["type"] == "fixed" for j in tab["joints"]):
        tab = urdf_lump_fixed(tab)
    links, joints = tab["links"], tab["joints"]
    children = {j["child"] for j in joints}
    roots = [n for n in links if n not in children]
    if len(roots) != 1:
        raise ValueError("URDF must have exactly one root link")
    origin = Origin(roots[0])
    bodies, eqcs, body_of = [], [], {roots[0]: origin}
    com = {roots[0]: np.zeros(3)}    # root link frame = world frame; its inertia is irrelevant when welded
    pending = list(joints)
    while pending:
        progressed = False
        for j in list(pending):
            if j["parent"] not in body_of:
                continue
            L = links[j["child"]]
            Ic = _link_inertia(L)
            b = Body(L["mass"], Ic, name=j["child"])
            com[j["child"]] = np.asarray(L["com"], dtype=np.float64)
            qj = np.asarray(j["quat"], dtype=np.float64) if "quat" in j else rpy_quaternion(*j["rpy"])
            axis_parent = vrotate(np.asarray(j["axis"], dtype=np.float64), qj)
            p1 = np.asarray(j["xyz"], dtype=np.float64) - com[j["parent"]]
            p2 = -com[j["child"]]
            ctor = Prismatic if j["type"] == "prismatic" else Revolute
            eqcs.append(EqualityConstraint(ctor(body_of[j["parent"]], b, axis_parent, p1=p1, p2=p2, qoffset=qj), name=j["name"]))
            bodies.append(b)
            body_of[j["child"]] = b
            pending.remove(j)
            progressed = True
        if not progressed:
            raise ValueError("URDF joints do not form a tree")
    mech = Mechanism(origin, bodies, eqcs, g=g, dt=dt)
    for e in mech.eqconstraints:      # zero pose, root to leaf
        setJointPosition(mech, e, 0.0)
    return mech


def minimal_to_maximal(mech, eqcids, xθ, vω):
    """Maximal-coordinate setpoint (xd, vd, qd, ωd per body) from joint coordinates xθ and joint rates vω of the 1-DoF joints
    `eqcids` (one per body) — what the 6-argument linearsystem returns besides A, Bu, Bλ, G (lqr.jl:80).  Forward kinematics root
    to leaf; velocities by rigid-body kinematics (ω in the body frame, v of the COM in the world frame)."""
    nb = len(mech.bodies)
    coord = {int(e): (float(x), float(v)) for e, x, v in zip(eqcids, xθ, vω)}
    saved = mech.state()
    vel = {mech.origin.id: (np.zeros(3), np.zeros(3))}           # body id -> (v world, ω world)
    todo = list(mech.eqconstraints)
    while todo:
        progressed = False
        for e in list(todo):
            j = e.joint
            if j.body1.id not in vel:
                continue
            θ, θd = coord.get(e.id, (0.0, 0.0))
            setJointPosition(mech, e, θ)
            a, b = j.body1, j.body2
            va, wa = vel[a.id]
            qa, xa = a.state.qc, a.state.xc
            axis_w = vrotate(j.axis / np.linalg.norm(j.axis), qa)
            pj = xa + vrotate(j.p1, qa)                              # joint vertex in the world (parent side)
            if j.kind == REVOLUTE:
                wb = wa + axis_w * θd
                vb = va + np.cross(wa, pj - xa) + np.cross(wb, b.state.xc - pj)
            else:
                wb = wa
                vb = va + np.cross(wa, b.state.xc - xa) + axis_w * θd
            vel[b.id] = (vb, wb)
            b.state.vc = vb
            b.state.ωc = vrotate(wb, qconj(b.state.qc))             # body frame
            todo.remove(e)
            progressed = True
        if not progressed:
            raise ValueError("joints do not form a tree")
    z = mech.state()
    mech.set_state(saved)
    return ([z[i, 0:3] for i in range(nb)], [z[i, 7:10] for i in range(nb)], [z[i, 3:7] for i in range(nb)], [z[i, 10:13] for i in range(nb)])
