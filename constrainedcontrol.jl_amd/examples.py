"""The reference's example mechanisms (= the BASELINE.json configs) rebuilt through the host mirror.
Every number below is read from the cited script; nothing else of the scripts is reproduced."""
import numpy as np

from .mechanism import (Box, EqualityConstraint, Mechanism, Origin, Prismatic, Quaternion, Revolute, RotX, getid, setJointPosition, setPosition)

EX = np.array([1.0, 0.0, 0.0])
EY = np.array([0.0, 1.0, 0.0])


def pendulum(θ0=np.pi - 0.4):
    """examples/lqr_pendulum.jl:8-41 — Box(0.1,0.1,1,1) on a revolute about x, p2=[0,0,0.5]; setpoint RotX(π), xd=[0,0,0.5];
    Q[7,7]=1000, Q[10,10]=100, R=1, horizon Inf."""
    p2 = np.array([0.0, 0.0, 0.5])
    origin = Origin()
    link1 = Box(0.1, 0.1, 1.0, 1.0)
    joint = EqualityConstraint(Revolute(origin, link1, EX, p2=p2))
    mech = Mechanism(origin, [link1], [joint])
    setPosition(origin, link1, p2=p2, Δq=Quaternion(RotX(θ0)))
    Q = [np.zeros((12, 12))]
    Q[0][6, 6] = 1000.0
    Q[0][9, 9] = 100.0
    R = [np.ones((1, 1))]
    xd = [np.array([0.0, 0.0, 0.5])]
    qd = [Quaternion(RotX(np.pi))]
    return dict(mech=mech, bodies=[link1], joints=[joint], Q=Q, R=R, xd=xd, qd=qd, ctrl=[joint], horizon=np.inf, tend=10.0)


def cartpole_n(n=1, y0=0.5, φ0=None):
    """examples/lqr_cartpole.jl:8-44 (n=1) and examples/lqr_cartpole_n_pendulum.jl:8-53 (N links):
    cart Box(0.1,0.5,0.1,0.5) on Prismatic(origin,cart,ey); poles Box(0.1,0.1,1,1) on revolutes about ex;
    xd = [0,0,0] and [0,0,i-0.5]; Q = I12 per body, R = 1, horizon 10 s."""
    p2 = np.array([0.0, 0.0, 0.5])
    if φ0 is None:
        φ0 = [0.2] + [0.0] * (n - 1)
    origin = Origin()
    cart = Box(0.1, 0.5, 0.1, 0.5)
    poles = [Box(0.1, 0.1, 1.0, 1.0) for _ in range(n)]
    bodies = [cart] + poles
    joints = [EqualityConstraint(Prismatic(origin, cart, EY)), EqualityConstraint(Revolute(cart, poles[0], EX, p2=-p2))]
    for i in range(1, n):
        joints.append(EqualityConstraint(Revolute(poles[i - 1], poles[i], EX, p1=p2, p2=-p2)))
    mech = Mechanism(origin, bodies, joints, g=-9.81)
    place_cartpole(mech, y0, φ0)
    xd = [np.zeros(3)] + [np.array([0.0, 0.0, i + 0.5]) for i in range(n)]
    Q = [np.eye(12) for _ in range(n + 1)]
    R = [np.ones((1, 1))]
    return dict(mech=mech, bodies=bodies, joints=joints, Q=Q, R=R, xd=xd, qd=None, ctrl=[joints[0]], horizon=10.0, tend=10.0)


def place_cartpole(mech, y0, φ):
    """setPosition! sequence of lqr_cartpole_n_pendulum.jl:39-43: relative angles φ[i] between consecutive links"""
    p2 = np.array([0.0, 0.0, 0.5])
    b = mech.bodies
    setPosition(mech.origin, b[0], Δx=[0.0, y0, 0.0])
    setPosition(b[0], b[1], p2=-p2, Δq=Quaternion(RotX(φ[0])))
    for i in range(2, len(b)):
        setPosition(b[i - 1], b[i], p1=p2, p2=-p2, Δq=Quaternion(RotX(φ[i - 1])))


def cartpole_states(n, y0, φ):
    """batch of initial states z0[n_inst][n+1][13] for the N-link cartpole, zero velocities (8d 'Synthetic inputs')"""
    y0 = np.atleast_1d(np.asarray(y0, dtype=np.float64))
    φ = np.asarray(φ, dtype=np.float64).reshape(len(y0), n)
    ninst = len(y0)
    z = np.zeros((ninst, n + 1, 13))
    z[:, :, 3] = 1.0
    z[:, 0, 1] = y0
    ang = np.cumsum(φ, axis=1)                      # absolute angle of link i about x
    s, c = np.sin(ang), np.cos(ang)
    # joint of link i sits at the top of link i-1 (cart: its COM); COM is 0.5 along the link's z axis
    jy, jz = y0.copy(), np.zeros(ninst)
    for i in range(n):
        # R(RotX(a)) [0,0,0.5] = [0, -0.5 sin a, 0.5 cos a]
        z[:, i + 1, 0] = 0.0
        z[:, i + 1, 1] = jy - 0.5 * s[:, i]
        z[:, i + 1, 2] = jz + 0.5 * c[:, i]
        z[:, i + 1, 3] = np.cos(ang[:, i] / 2)
        z[:, i + 1, 4] = np.sin(ang[:, i] / 2)
        jy = jy - s[:, i]
        jz = jz + c[:, i]
    return z


def triple_cartpole():
    """examples/trackingLQR_triple_cartpole.jl:11-45,64-73: cart + 3 poles hanging (p2 = +[0,0,0.5] on the child side,
    p1 = -p2 on the parent side), Δt = 0.01; Q entries (2,2)=10,(5,5)=1 on the cart, (7,7)=40,(10,10)=1 on each pole; R = 0.1."""
    p2 = np.array([0.0, 0.0, 0.5])
    origin = Origin()
    cart = Box(0.1, 0.5, 0.1, 0.5)
    poles = [Box(0.1, 0.1, 1.0, 1.0) for _ in range(3)]
    joints = [EqualityConstraint(Prismatic(origin, cart, EY)), EqualityConstraint(Revolute(cart, poles[0], EX, p2=p2)),
              EqualityConstraint(Revolute(poles[0], poles[1], EX, p1=-p2, p2=p2)),
              EqualityConstraint(Revolute(poles[1], poles[2], EX, p1=-p2, p2=p2))]
    bodies = [cart] + poles
    mech = Mechanism(origin, bodies, joints, g=-9.81, Δt=0.01)
    setPosition(origin, cart, Δx=[0.0, 0.0, 0.0])
    setPosition(cart, poles[0], p2=p2, Δq=Quaternion(RotX(0.0)))
    setPosition(poles[0], poles[1], p1=-p2, p2=p2, Δq=Quaternion(RotX(0.0)))
    setPosition(poles[1], poles[2], p1=-p2, p2=p2, Δq=Quaternion(RotX(0.0)))
    Q = [np.zeros((12, 12)) for _ in range(4)]
    Q[0][1, 1] = 10.0
    Q[0][4, 4] = 1.0
    for i in (1, 2, 3):
        Q[i][6, 6] = 40.0
        Q[i][9, 9] = 1.0
    R = [np.ones((1, 1)) * 0.1]
    return dict(mech=mech, bodies=bodies, joints=joints, Q=Q, R=R, ctrl=[joints[0]], fric=np.array([0.1, 0.1, 0.1, 0.1]), noise_scale=2.0)


def sawyer(tables, g=0.0):
    """examples/lqr_sawyer.jl:8-33 from the numeric content of examples_files/sawyer_arm.urdf (tests/golden/sawyer_arm_tables.json):
    7 bodies / 7 revolute joints (base link welded to the origin), g = 0, zero pose as setpoint, Q = 1000 I12 per body,
    R = 1 per joint, all joints controlled, horizon 20 s."""
    from .mechanism import mechanism_from_urdf_tables
    mech = mechanism_from_urdf_tables(tables, floating=False, g=g)
    nb = len(mech.bodies)
    z = mech.state()
    return dict(mech=mech, bodies=mech.bodies, joints=mech.eqconstraints, Q=[np.eye(12) * 1000.0 for _ in range(nb)],
                R=[np.ones((1, 1)) for _ in range(nb)], xd=[z[i, 0:3] for i in range(nb)], qd=[z[i, 3:7] for i in range(nb)],
                ctrl=list(mech.eqconstraints), horizon=20.0, tend=20.0)


def acrobot():
    """examples/lqr_acrobot.jl:8-50 — two links on revolutes about x, only the SECOND joint actuated: link1 Box(0.1,0.1,1,1) with
    p2=[0,0,0.5]; link2 Box(0.1,0.1,2,1) hung at p1=-[0,0,0.5] / p2=[0,0,1]; start RotX(π-0.1), RotX(0.1); setpoint both RotX(π),
    xd=[[0,0,0.5],[0,0,2]]; Q1[7,7]=Q1[10,10]=4, Q2[7,7]=Q2[10,10]=1, R=1, horizon 10 s."""
    p2a, p2b = np.array([0.0, 0.0, 0.5]), np.array([0.0, 0.0, 1.0])
    origin = Origin()
    link1 = Box(0.1, 0.1, 1.0, 1.0)
    link2 = Box(0.1, 0.1, 2.0, 1.0)
    joint1 = EqualityConstraint(Revolute(origin, link1, EX, p2=p2a))
    joint2 = EqualityConstraint(Revolute(link1, link2, EX, p1=-p2a, p2=p2b))
    mech = Mechanism(origin, [link1, link2], [joint1, joint2], g=-9.81)
    setPosition(origin, link1, p2=p2a, Δq=Quaternion(RotX(np.pi - 0.1)))
    setPosition(link1, link2, p1=-p2a, p2=p2b, Δq=Quaternion(RotX(0.1)))
    Q = [np.zeros((12, 12)), np.zeros((12, 12))]
    Q[0][6, 6] = Q[0][9, 9] = 4.0
    Q[1][6, 6] = Q[1][9, 9] = 1.0
    return dict(mech=mech, bodies=[link1, link2], joints=[joint1, joint2], Q=Q, R=[np.ones((1, 1))], ctrl=[joint2],
                xd=[np.array([0.0, 0.0, 0.5]), np.array([0.0, 0.0, 2.0])], qd=[Quaternion(RotX(np.pi))] * 2, horizon=10.0, tend=10.0)


def double_pendulum(φ1=0.0, φ2=0.0):
    """examples/pid_doublependulum.jl:5-38 — two Box(0.1,0.1,1,1) links on revolutes about x (p2=[0,0,0.5]; p1=-p2, p2), default
    gravity; PID goals [π/2, -π/4], P=[10,10], I=[10,10], D=[5,5], 10 s."""
    p2 = np.array([0.0, 0.0, 0.5])
    origin = Origin()
    link1 = Box(0.1, 0.1, 1.0, 1.0)
    link2 = Box(0.1, 0.1, 1.0, 1.0)
    joint1 = EqualityConstraint(Revolute(origin, link1, EX, p2=p2))
    joint2 = EqualityConstraint(Revolute(link1, link2, EX, p1=-p2, p2=p2))
    mech = Mechanism(origin, [link1, link2], [joint1, joint2])
    setPosition(origin, link1, p2=p2, Δq=Quaternion(RotX(φ1)))
    setPosition(link1, link2, p1=-p2, p2=p2, Δq=Quaternion(RotX(φ2)))
    return dict(mech=mech, bodies=[link1, link2], joints=[joint1, joint2], goals=[np.pi / 2, -np.pi / 4], P=[10.0, 10.0], I=[10.0, 10.0],
                D=[5.0, 5.0], tend=10.0)


def prismatic_slider():
    """examples/lqr_prismatic.jl:8-30: Box(0.1,0.1,0.1,1) on Prismatic(origin, link1, ex), g = 0, start at Δx = [1,0,0];
    minimal-coordinate LQR: Q = ones(1), R = ones(1), horizon 10 s (setpoint: joint coordinate 0)."""
    origin = Origin()
    link1 = Box(0.1, 0.1, 0.1, 1.0)
    joint = EqualityConstraint(Prismatic(origin, link1, EX))
    mech = Mechanism(origin, [link1], [joint], g=0.0)
    setPosition(origin, link1, Δx=[1.0, 0.0, 0.0])
    return dict(mech=mech, bodies=[link1], joints=[joint], Q=np.ones(1), R=np.ones(1), horizon=10.0)


def dual_cartpole(φ1=0.1, φ2=-0.15, y0=0.2):
    """Not one of the reference's scripts: the cart of lqr_cartpole.jl carrying TWO poles (lengths 1 and 0.6) on the same cart body --
    the smallest mechanism whose body/joint tree branches (two child joints on one body)."""
    origin = Origin()
    cart = Box(0.1, 0.5, 0.1, 0.5)
    pole1 = Box(0.1, 0.1, 1.0, 1.0)
    pole2 = Box(0.1, 0.1, 0.6, 0.6)
    j0 = EqualityConstraint(Prismatic(origin, cart, EY))
    j1 = EqualityConstraint(Revolute(cart, pole1, EX, p1=np.array([0.0, 0.15, 0.0]), p2=np.array([0.0, 0.0, -0.5])))
    j2 = EqualityConstraint(Revolute(cart, pole2, EX, p1=np.array([0.0, -0.15, 0.0]), p2=np.array([0.0, 0.0, -0.3])))
    mech = Mechanism(origin, [cart, pole1, pole2], [j0, j1, j2], g=-9.81)
    setJointPosition(mech, j0, y0)
    setJointPosition(mech, j1, φ1)
    setJointPosition(mech, j2, φ2)
    return dict(mech=mech, bodies=[cart, pole1, pole2], joints=[j0, j1, j2], Q=[np.eye(12) for _ in range(3)], R=[np.ones((1, 1))], ctrl=[j0],
                horizon=10.0, tend=10.0)


def tree_mechanism(parents, seed=0, g=-9.81, prismatic=()):
    """A random tree of 1-DoF joints for tests: body i hangs off body parents[i] (-1 = origin; parents[i] < i) through a revolute
    (or, for i in `prismatic`, a prismatic) joint with random axis and anchor points; placed at random joint coordinates."""
    rng = np.random.default_rng(seed)
    origin = Origin()
    bodies, joints = [], []
    for i, a in enumerate(parents):
        dims = rng.uniform(0.1, 0.6, 3)
        b = Box(dims[0], dims[1], dims[2], float(rng.uniform(0.3, 1.5)))
        bodies.append(b)
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        par = origin if a < 0 else bodies[a]
        kw = dict(p1=rng.uniform(-0.3, 0.3, 3), p2=rng.uniform(-0.3, 0.3, 3))
        joints.append(EqualityConstraint((Prismatic if i in prismatic else Revolute)(par, b, axis, **kw)))
    mech = Mechanism(origin, bodies, joints, g=g)
    for e in joints:                                    # bodies are listed parents-first, so this places root to leaf
        setJointPosition(mech, e, rng.uniform(-0.6, 0.6))
    return dict(mech=mech, bodies=bodies, joints=joints)


def deltabot():
    """examples/lqr_deltabot.jl:7-53: a planar five-bar mechanism with a platform held level -- five bodies, seven equality
    constraints (two floor revolutes and a FixedOrientation to the origin, two knees, two platform revolutes: 33 constraint rows on 30
    body coordinates, i.e. closed loops), placed at the script's pose; Fτd = +-6.7879484 on the platform joints holds it at rest.
    Returns the mechanism, the controlled constraint ids (platl, platr) and the holding inputs."""
    from .mechanism import FixedOrientation
    L = 1.0
    ex = [1.0, 0.0, 0.0]
    pll, pul, pp = np.array([0, 0, L / 2]), np.array([0, 0, L / 4]), np.array([0, 0, L / 4 * np.sqrt(2)])     # :11-16
    origin = Origin()
    lowerlegl, lowerlegr = Box(0.1, 0.1, L, L, "lowerlegl"), Box(0.1, 0.1, L, L, "lowerlegr")                  # :18-19
    upperlegl, upperlegr = Box(0.1, 0.1, L / 2, L / 2, "upperlegl"), Box(0.1, 0.1, L / 2, L / 2, "upperlegr")  # :20-21
    platform = Box(0.1, 0.1, L / 2 * np.sqrt(2), L / 2 * np.sqrt(2), "platform")                               # :22
    platl = EqualityConstraint(Revolute(platform, upperlegl, ex, p1=pp, p2=pul), "platl")                      # :28
    platr = EqualityConstraint(Revolute(platform, upperlegr, ex, p1=-pp, p2=pul), "platr")                     # :29
    floorl = EqualityConstraint(Revolute(origin, lowerlegl, ex, p2=-pll), "floorl")                             # :25 (floorlr)
    floorr = EqualityConstraint(Revolute(origin, lowerlegr, ex, p2=-pll), "floorr")
    flooro = EqualityConstraint(FixedOrientation(origin, platform, qoffset=RotX(np.pi / 2)), "flooro")
    kneel = EqualityConstraint(Revolute(lowerlegl, upperlegl, ex, p1=pll, p2=-pul), "kneel")                    # :26
    kneer = EqualityConstraint(Revolute(lowerlegr, upperlegr, ex, p1=pll, p2=-pul), "kneer")                    # :27
    links = [lowerlegl, lowerlegr, upperlegl, upperlegr, platform]                                               # :31
    mech = Mechanism(origin, links, [platl, platr, floorl, floorr, flooro, kneel, kneer], g=-9.81, dt=0.01)      # :32-36
    setPosition(origin, lowerlegl, p2=-pll, Δq=RotX(np.pi / 4))                                                  # :37
    setPosition(origin, lowerlegr, p2=-pll, Δq=RotX(-np.pi / 4))                                                 # :38
    setPosition(lowerlegl, upperlegl, p1=pll, p2=-pul, Δq=RotX(-np.pi / 2))                                      # :39
    setPosition(lowerlegr, upperlegr, p1=pll, p2=-pul, Δq=RotX(np.pi / 2))                                       # :40
    setPosition(upperlegl, platform, p1=pul, p2=pp, Δq=RotX(3 * np.pi / 4))                                      # :41
    Q = [np.zeros((12, 12)) for _ in range(5)]                                                                 # :46-50: only the platform's y, z
    Q[4][1, 1] = Q[4][2, 2] = 10.0                                                                             # position and velocity are weighted
    Q[4][4, 4] = Q[4][5, 5] = 1.0
    R = [np.ones((1, 1)) * 0.1 for _ in range(2)]                                                              # :51
    return {"mech": mech, "eqcids": [getid(platl), getid(platr)], "Fd": np.array([6.7879484, -6.7879484]),     # :53
            "Q": Q, "R": R, "links": links, "origin": origin, "L": L}


def deltabot_initial_states(ex, stride=1):
    """the valid initial conditions of examples/lqr_deltabot.jl:56-136: platform positions (y, z) on the script's 101 x 101 grid over
    [-1.5 L, 1.5 L]^2 with z >= 0 whose two hip points lie between 0.5 L and 1.5 L from the floor joints, the leg angles from the two
    triangles (floor joint, knee, hip), placed with the script's setPosition! sequence (:139-143).  The script simulates ONE of them
    (i = 97); here they are a batch.  Returns (z0 [nc][5][13], yz [nc][2]); stride > 1 keeps every stride-th condition."""
    L = ex["L"]
    mech, origin = ex["mech"], ex["origin"]
    lowerlegl, lowerlegr, upperlegl, upperlegr, platform = ex["links"]
    pll, pul, p3 = np.array([0, 0, L / 2]), np.array([0, 0, L / 4]), L / 4 * np.sqrt(2)
    grid = -1.5 * L + 3 * L * np.arange(101) / 100.0
    yz = [(y, z) for z in grid if z >= 0 for y in grid
          if 0.5 * L <= np.hypot(y + p3, z) <= 1.5 * L and 0.5 * L <= np.hypot(y - p3, z) <= 1.5 * L]
    yz = np.array(yz)[::stride]
    a, b = L, L / 2

    def triangle(py, pz):
        c = np.hypot(py, pz)
        beta = np.arccos((a * a + c * c - b * b) / (2 * a * c))
        gamma = np.arccos((a * a + b * b - c * c) / (2 * a * b))
        with np.errstate(divide="ignore"):
            delta = abs(np.arctan(np.float64(py) / np.float64(pz)))
        return beta, gamma, delta

    saved = mech.state()
    out = []
    for y, z in yz:
        ly, ry = y - p3, y + p3
        bl, gl, dl = triangle(ly, z)
        br, gr, dr = triangle(ry, z)
        if ly <= 0 and z >= 0:   al = (dl + bl, -np.pi + gl)                                                   # :117-125
        elif ly >= 0 and z >= 0: al = (-dl + bl, -np.pi + gl)
        elif ly >= 0 and z <= 0: al = (-np.pi + dl - bl, np.pi - gl)
        else:                    al = (-np.pi - dl - bl, np.pi - gl)
        if ry <= 0 and z >= 0:   ar = (dr - br, np.pi - gr)                                                    # :127-135
        elif ry >= 0 and z >= 0: ar = (-dr - br, np.pi - gr)
        elif ry >= 0 and z <= 0: ar = (-np.pi + dr + br, -np.pi + gr)
        else:                    ar = (-np.pi - dr + br, -np.pi + gr)
        setPosition(origin, lowerlegl, p2=-pll, Δq=RotX(al[0]))                                                # :139-143
        setPosition(origin, lowerlegr, p2=-pll, Δq=RotX(ar[0]))
        setPosition(lowerlegl, upperlegl, p1=pll, p2=-pul, Δq=RotX(al[1]))
        setPosition(lowerlegr, upperlegr, p1=pll, p2=-pul, Δq=RotX(ar[1]))
        setPosition(origin, platform, p1=np.array([0.0, y, z]), Δq=RotX(np.pi / 2))
        out.append(mech.state())
    mech.set_state(saved)
    return np.array(out), yz


def fourbar(θ=0.6, a=0.5, b=0.9):
    """a planar parallelogram four-bar linkage in the y-z plane (not one of the reference's scripts: a second closed-loop topology for the
    tests): crank and rocker of length a on floor revolutes b apart, a coupler of length b between their tips, every joint a revolute about
    ex -- three bodies, four joints, 20 constraint rows on 18 body coordinates with one degree of freedom (rank 17), placed at crank angle θ."""
    ex = [1.0, 0.0, 0.0]
    origin = Origin()
    crank, coupler, rocker = Box(0.05, 0.05, a, 0.8, "crank"), Box(0.05, b, 0.05, 1.1, "coupler"), Box(0.05, 0.05, a, 0.7, "rocker")
    ha, hb = np.array([0, 0, a / 2]), np.array([0, b / 2, 0])
    j1 = EqualityConstraint(Revolute(origin, crank, ex, p2=-ha), "floor_crank")
    j2 = EqualityConstraint(Revolute(crank, coupler, ex, p1=ha, p2=-hb), "crank_coupler")
    j3 = EqualityConstraint(Revolute(coupler, rocker, ex, p1=hb, p2=ha), "coupler_rocker")
    j4 = EqualityConstraint(Revolute(origin, rocker, ex, p1=np.array([0, b, 0]), p2=-ha), "floor_rocker")
    mech = Mechanism(origin, [crank, coupler, rocker], [j1, j2, j3, j4], g=-9.81, dt=0.01)
    setPosition(origin, crank, p2=-ha, Δq=RotX(θ))
    setPosition(crank, coupler, p1=ha, p2=-hb, Δq=RotX(-θ))          # the coupler stays level in a parallelogram
    setPosition(coupler, rocker, p1=hb, p2=ha, Δq=RotX(θ))
    return {"mech": mech, "eqcids": [getid(j1)], "bodies": [crank, coupler, rocker]}

