"""GPU parity of the setup path: linearsystem (a8), dlqr (a5), tracking dlqr (a10) and the LQR/TrackingLQR/simulate mirror,
all through the C-ABI, against the CPU oracle.  Tolerances are relative to the largest entry of the compared array."""
import numpy as np
import pytest
import scipy.linalg as sl

from conftest import hanging_setpoint, upright_setpoint

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


@pytest.mark.parametrize("n_links,hanging", [(1, False), (3, False), (16, True)])
def test_linearize_matches_oracle(cclqr, orc, n_links, hanging):
    capi = cclqr._capi
    ex = cclqr.examples.cartpole_n(n_links)
    t = ex["mech"].tables()
    mech = capi.MechHandle(t)
    zd = hanging_setpoint(cclqr, n_links) if hanging else upright_setpoint(n_links)
    # a second, generic knot: a few open-loop steps away from the setpoint, with non-zero feed-forward
    z = zd.copy()
    lam = np.zeros(5 * t.ne)
    uj = np.zeros(t.ne)
    uj[0] = 0.4
    for _ in range(5):
        z, lam, it = orc.step(t, z, lam, uj)
    knots = np.stack([zd, z])
    Fd = np.array([[0.0], [0.7]])
    A, Bu, Bl, G = capi.linearize(mech, knots, [0], Fd)
    for k in range(2):
        ref = orc.linearize(t, knots[k], [0], Fd[k])
        for got, want in zip((A[k], Bu[k], Bl[k], G[k]), ref):
            assert _rel(got, want) < 1e-10


def test_riccati_vs_scipy_dare(cclqr):
    """unconstrained branch (lqr.jl:36) iterated to convergence == util.jl:1-19's dare == scipy.linalg.solve_discrete_are"""
    capi = cclqr._capi
    rng = np.random.default_rng(1)
    n, m = 6, 2
    A = rng.normal(size=(n, n)) * 0.5
    B = rng.normal(size=(n, m))
    Q, R = np.eye(n), np.eye(m)
    K, kb = capi.riccati(A, B, np.zeros((n, 0)), np.zeros((0, n)), Q, R, 2000, tol=1e-13)
    P = sl.solve_discrete_are(A, B, Q, R)
    Kd = np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)
    assert np.abs(K[0] - Kd).max() < 1e-9
    assert kb > 1


@pytest.mark.parametrize("n_links,N", [(1, 1000), (3, 300), (7, 120)])
def test_riccati_matches_oracle(cclqr, orc, n_links, N):
    capi = cclqr._capi
    ex = cclqr.examples.cartpole_n(n_links)
    t = ex["mech"].tables()
    zd = upright_setpoint(n_links)
    A, Bu, Bl, G = orc.linearize(t, zd, [0], np.zeros(1))
    Q = sl.block_diag(*ex["Q"]) * t.dt
    R = sl.block_diag(*ex["R"]) * t.dt
    K, kb = capi.riccati(A, Bu, Bl, G, Q, R, N)
    Ko, kbo = orc.riccati(A, Bu, Bl, G, Q, R, N)
    assert kb == kbo
    assert _rel(K, Ko) < 1e-7
    # early exit + back-fill semantics (lqr.jl:172-181)
    for k2 in range(kb - 1):
        assert np.array_equal(K[k2], K[kb - 1])
    # closed loop keeps the constraint: G (A - Bu Ku - Bl Kl) = 0  is implied by M22/M21 rows; check through D: G D = 0
    D = Bu - Bl @ np.linalg.solve(G @ Bl, G @ Bu)
    assert np.abs(G @ D).max() < 1e-9


@pytest.mark.parametrize("path", [1, 2])
def test_riccati_both_paths(cclqr, orc, path):
    """the LDS-resident and the tiled implementation of dlqr (cclqr_riccati_opts.path) against the oracle: time-invariant with an
    early break, batched, time-varying (lqr_tracking.jl:73-122), multi-input (mu = 3), and a singular G*Bλ"""
    capi = cclqr._capi
    if True:
        ex = cclqr.examples.cartpole_n(3)
        t = ex["mech"].tables()
        zd = upright_setpoint(3)
        A, Bu, Bl, G = orc.linearize(t, zd, [0], np.zeros(1))
        Q, R = sl.block_diag(*ex["Q"]) * t.dt, sl.block_diag(*ex["R"]) * t.dt
        K, kb = capi.riccati(A, Bu, Bl, G, Q, R, 400, path=path)
        Ko, kbo = orc.riccati(A, Bu, Bl, G, Q, R, 400)
        assert kb == kbo and _rel(K, Ko) < 1e-7
        # three actuated joints, batch of 3 distinct problems, a horizon with an early break in some of them
        mats = [orc.linearize(t, cclqr.examples.cartpole_states(3, [0.1 * p], np.full((1, 3), 0.05 * p))[0], [0, 1, 3], np.zeros(3)) for p in range(3)]
        Ab, Bub, Blb, Gb = (np.stack([m[i] for m in mats]) for i in range(4))
        R3 = np.diag([0.01, 0.02, 0.03])
        Kb, kbb = capi.riccati(Ab, Bub, Blb, Gb, Q * 100, R3, 150, tol=1e-3, path=path)
        for p in range(3):
            Ko, kbo = orc.riccati(Ab[p], Bub[p], Blb[p], Gb[p], Q * 100, R3, 150, tol=1e-3)
            assert kbb[p] == kbo and _rel(Kb[p], Ko) < 1e-7
        # time-varying knots
        t3 = t
        N = 60
        rng = np.random.default_rng(3)
        zs = np.stack([cclqr.examples.cartpole_states(3, [0.01 * k], rng.uniform(-0.3, 0.3, (1, 3)))[0] for k in range(N)])
        Fd = rng.normal(size=(N, 1))
        Q3, R3 = np.eye(48) * 0.01, np.eye(1) * 0.01
        mech = capi.MechHandle(t3)
        Kt, kbt = capi.riccati_tracking(mech, [0], zs, Fd, Q3, R3, N, path=path)
        Kto, kbto = orc.riccati_tracking(t3, [0], zs, Fd, Q3, R3, N)
        assert kbt == kbto and _rel(Kt, Kto) < 1e-7
        # singular G*Bλ (a duplicated constraint row) is reported, lqr.jl:151
        G2, Bl2 = G.copy(), Bl.copy()
        G2[1] = G2[0]
        with pytest.raises(capi.CclqrError) as e:
            capi.riccati(A, Bu, Bl2, G2, Q, R, 50, path=path)
        assert e.value.code == capi.ESINGULAR
        # N = 1 and N = 2 edge cases (empty / single gain)
        K1, kb1 = capi.riccati(A, Bu, Bl, G, Q, R, 1, path=path)
        assert K1.shape[0] == 0 and kb1 == 0
        K2, kb2 = capi.riccati(A, Bu, Bl, G, Q, R, 2, path=path)
        Ko2, kbo2 = orc.riccati(A, Bu, Bl, G, Q, R, 2)
        assert kb2 == kbo2 and _rel(K2, Ko2) < 1e-7


def test_riccati_time_varying_on_the_register_fragment_kernel(cclqr, orc):
    """lqr_tracking.jl:73-122 at the Sawyer's shape (mx = 84, mu = 7: `riccati_resident_kernel<7, 21, 0>`) and the triple cartpole's
    (mx = 48, mu = 1) with D = Bu - Bλ (G Bλ)⁻¹ G Bu changing from knot to knot (poses a radian apart: ~10 % of |D| per knot, 1e6 x the tolerance): the fp64 register-fragment form has no barrier
    between the update phase (reads the step's D from LDS) and the Pkp1 tiles, so the next knot's D must not land in LDS before every
    wavefront has left the update phase (it is held in registers until the norm's barrier).  Several problems per launch and
    repeated launches give the wavefronts room to drift apart; gains against the oracle's statement-by-statement recursion."""
    import json
    import os
    capi = cclqr._capi
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sawyer_arm_tables.json")))
    ex = cclqr.examples.sawyer(tab)
    mech = ex["mech"]
    t = mech.tables()
    rng = np.random.default_rng(11)
    N = 24
    zs = []
    for k in range(N):
        for e in mech.eqconstraints:
            cclqr.setJointPosition(mech, e, rng.uniform(-1.2, 1.2))       # poses a radian apart: D differs in every entry between knots
        zs.append(mech.state())
    zs = np.stack(zs)
    Fd = rng.normal(size=(N, 7))
    Q, R = np.eye(84) * 10.0, np.diag(rng.uniform(0.5, 2.0, 7)) * 0.01
    h = capi.MechHandle(t)
    Kto, kbto = orc.riccati_tracking(t, list(range(7)), zs, Fd, Q, R, N)
    Dn = []
    for k in range(N - 1):
        A, Bu, Bl, G = orc.linearize(t, zs[k], list(range(7)), Fd[k])
        Dn.append(Bu - Bl @ np.linalg.solve(G @ Bl, G @ Bu))
    assert max(np.abs(Dn[k] - Dn[k + 1]).max() / np.abs(Dn[k]).max() for k in range(N - 2)) > 0.05     # the knots' D really differ
    for rep in range(3):
        Kt, kbt = capi.riccati_tracking(h, list(range(7)), zs, Fd, Q, R, N, path=1)
        assert kbt == kbto and _rel(Kt, Kto) < 1e-7
    ex3 = cclqr.examples.cartpole_n(3)
    t3 = ex3["mech"].tables()
    N3 = 50
    zs3 = np.stack([cclqr.examples.cartpole_states(3, [0.02 * k], rng.uniform(-1.0, 1.0, (1, 3)))[0] for k in range(N3)])
    Fd3 = rng.normal(size=(N3, 1))
    h3 = capi.MechHandle(t3)
    Kt3, kb3 = capi.riccati_tracking(h3, [0], zs3, Fd3, np.eye(48) * 0.01, np.eye(1) * 0.01, N3, path=1)
    Kto3, kbo3 = orc.riccati_tracking(t3, [0], zs3, Fd3, np.eye(48) * 0.01, np.eye(1) * 0.01, N3)
    assert kb3 == kbo3 and _rel(Kt3, Kto3) < 1e-7


def test_riccati_batched(cclqr, orc):
    capi = cclqr._capi
    ex = cclqr.examples.cartpole_n(1)
    t = ex["mech"].tables()
    rng = np.random.default_rng(5)
    mats = []
    for p in range(5):
        zd = cclqr.examples.cartpole_states(1, [rng.uniform(-0.3, 0.3)], rng.uniform(-0.2, 0.2, (1, 1)))[0]
        mats.append(orc.linearize(t, zd, [0], np.zeros(1)))
    A, Bu, Bl, G = (np.stack([m[i] for m in mats]) for i in range(4))
    Q = np.eye(24) * t.dt
    R = np.eye(1) * t.dt
    K, kb = capi.riccati(A, Bu, Bl, G, Q, R, 200)
    for p in range(5):
        Ko, kbo = orc.riccati(A[p], Bu[p], Bl[p], G[p], Q, R, 200)
        assert kb[p] == kbo and _rel(K[p], Ko) < 1e-7


def test_lqr_pipeline_cartpole(cclqr, orc):
    """examples/lqr_cartpole.jl end to end through the mirror: LQR(...) then simulate!(mech, 10, lqr)"""
    ex = cclqr.examples.cartpole_n(1)
    mech = ex["mech"]
    lqr = cclqr.LQR(mech, [cclqr.getid(b) for b in ex["bodies"]], [cclqr.getid(ex["ctrl"][0])], ex["Q"], ex["R"], 10.0, xd=ex["xd"])
    assert lqr.K.shape == (999, 1, 24) and lqr.N == 1000
    t = mech.tables()
    Ao, Buo, Blo, Go = orc.linearize(t, lqr.zd[0], [0], np.zeros(1))
    Ko, kbo = orc.riccati(Ao, Buo, Blo, Go, lqr.Q, lqr.R, 1000)
    assert lqr.kbreak == kbo and _rel(lqr.K, Ko) < 1e-7
    z0 = mech.state()
    storage = cclqr.simulate(mech, 10, lqr, record=True)
    assert storage.z.shape == (1, 1000, 2, 13)
    octrl = orc.ctrl_desc(2, [0], K=lqr.K, N=lqr.N, zd=lqr.zd)
    zT, traj, st = orc.rollout(t, octrl, z0[None], 1000, record=True)
    assert np.abs(storage.z - traj).max() < 1e-9
    # the controller does its job: cart back at the origin, pole upright
    assert abs(storage.zT[0, 0, 1]) < 5e-3 and abs(storage.zT[0, 1, 4]) < 1e-4
    assert np.allclose(mech.bodies[0].state.xc, storage.zT[0, 0, 0:3])


def test_custom_controlfunction_closure(cclqr, orc):
    """`controlfunction` (lqr.jl:14, :56; used at examples/trackingLQR_triple_cartpole.jl:117): a host closure (batch, lqr, k) that ends in
    setForce.  (i) a closure that is control_lqr! itself reproduces the fused device rollout of the same controller (the closure's u is
    formed in numpy, the device's in the kernel: 1e-9 over 300 steps of a batch of 16) AND the oracle; (ii) the script's friction law
    (trackingLQR_triple_cartpole.jl:93-111: LQR input on the cart, -fric x relative joint velocity on every joint) written as a closure
    equals the built-in `fric=` option of simulate; (iii) fric / noise options on top of a closure are refused (the closure owns the law)"""
    ex = cclqr.examples.cartpole_n(1)
    mech = ex["mech"]
    ids = [cclqr.getid(b) for b in ex["bodies"]]
    rng = np.random.default_rng(5)
    z0 = cclqr.examples.cartpole_states(1, rng.uniform(-0.5, 0.5, 16), rng.uniform(0.0, 0.3, (16, 1)))
    calls = []

    def law(batch, lqr, k):
        calls.append(k)
        cclqr.control_lqr(batch, lqr, k)

    lqr_dev = cclqr.LQR(mech, ids, [cclqr.getid(ex["ctrl"][0])], ex["Q"], ex["R"], 10.0, xd=ex["xd"])
    lqr_host = cclqr.LQR(mech, ids, [cclqr.getid(ex["ctrl"][0])], ex["Q"], ex["R"], 10.0, xd=ex["xd"], controlfunction=law)
    assert np.array_equal(lqr_dev.K, lqr_host.K)
    sd = cclqr.simulate(mech, 3.0, lqr_dev, z0=z0)
    sh = cclqr.simulate(mech, 3.0, lqr_host, z0=z0)
    assert calls == list(range(1, 301))
    assert (sh.status > 0).all() and np.array_equal(sh.status, sd.status)
    assert np.abs(sh.z - sd.z).max() < 1e-9 and np.abs(sh.zT - sd.zT).max() < 1e-9
    t = mech.tables()
    zTo, trajo, sto = orc.rollout(t, orc.ctrl_desc(2, [0], K=lqr_dev.K, N=lqr_dev.N, zd=lqr_dev.zd), z0, 300, record=True)
    assert np.abs(sh.z - trajo).max() < 1e-9
    fric = np.array([0.3, 0.05])
    axes = [np.asarray(e.joint.axis, dtype=np.float64) / np.linalg.norm(e.joint.axis) for e in mech.eqconstraints]

    def law_fric(batch, lqr, k):
        u = cclqr.control_lqr(batch, lqr, k)[:, 0]
        cart_v = batch.v[:, 0] @ axes[0]                                       # prismatic on the origin: velocity along the axis
        pole_w = (batch.ω[:, 1] - batch.ω[:, 0]) @ axes[1]                     # revolute: axis . relative angular velocity
        cclqr.setForce(batch, mech.eqconstraints[0], u - fric[0] * cart_v)
        cclqr.setForce(batch, mech.eqconstraints[1], -fric[1] * pole_w)

    lqr_host.controlfunction = law_fric
    shf = cclqr.simulate(mech, 2.0, lqr_host, z0=z0)
    sdf = cclqr.simulate(mech, 2.0, lqr_dev, z0=z0, fric=fric)
    assert (shf.status > 0).all() and np.abs(shf.z - sdf.z).max() < 1e-9
    assert np.abs(sdf.z - sd.z[:, :200]).max() > 1e-3                         # (the friction did something)
    with pytest.raises(ValueError):
        cclqr.simulate(mech, 0.1, lqr_host, z0=z0, fric=fric)


def test_device_closure_controlfunction(cclqr, orc):
    """cclqr.on_device(controlfunction): the same closure hook with the batch's states as a torch tensor in HBM and the inputs as tensors -- nothing
    leaves the device between the steps.  (i) control_lqr as a device closure reproduces the fused rollout, the host closure and the oracle (1e-9,
    same Newton counts); (ii) the script's friction law written in torch equals simulate's built-in `fric=`; (iii) an instance poisoned with a NaN
    input at step 5 is lost and frozen exactly as the fused rollout freezes it, the others are untouched bit for bit"""
    import torch
    ex = cclqr.examples.cartpole_n(1)
    mech = ex["mech"]
    ids = [cclqr.getid(b) for b in ex["bodies"]]
    rng = np.random.default_rng(5)
    z0 = cclqr.examples.cartpole_states(1, rng.uniform(-0.5, 0.5, 16), rng.uniform(0.0, 0.3, (16, 1)))
    seen = []

    @cclqr.on_device
    def law(batch, lqr, k):
        assert batch.on_device and batch.z.is_cuda and batch.z.shape == (16, 2, 13)
        seen.append(k)
        cclqr.control_lqr(batch, lqr, k)

    def law_host(batch, lqr, k):
        cclqr.control_lqr(batch, lqr, k)

    mk = lambda f: cclqr.LQR(mech, ids, [cclqr.getid(ex["ctrl"][0])], ex["Q"], ex["R"], 10.0, xd=ex["xd"], controlfunction=f)
    fused, dev, host = cclqr.simulate(mech, 3.0, mk(None), z0=z0), cclqr.simulate(mech, 3.0, mk(law), z0=z0), cclqr.simulate(mech, 3.0, mk(law_host), z0=z0)
    assert seen == list(range(1, 301))
    assert (dev.status > 0).all() and np.array_equal(dev.status, fused.status) and np.array_equal(dev.status, host.status)
    assert np.abs(dev.z - fused.z).max() < 1e-9 and np.abs(dev.zT - fused.zT).max() < 1e-9 and np.abs(dev.z - host.z).max() < 1e-9
    t = mech.tables()
    lq = mk(None)
    _, trajo, _ = orc.rollout(t, orc.ctrl_desc(2, [0], K=lq.K, N=lq.N, zd=lq.zd), z0, 300, record=True)
    assert np.abs(dev.z - trajo).max() < 1e-9
    fric = np.array([0.3, 0.05])
    axes = [torch.tensor(np.asarray(e.joint.axis, dtype=np.float64) / np.linalg.norm(e.joint.axis), device="cuda") for e in mech.eqconstraints]

    @cclqr.on_device
    def law_fric(batch, lqr, k):
        u = cclqr.control_lqr(batch, lqr, k)[:, 0]
        cart_v = batch.v[:, 0] @ axes[0]
        pole_w = (batch.ω[:, 1] - batch.ω[:, 0]) @ axes[1]
        cclqr.setForce(batch, mech.eqconstraints[0], u - fric[0] * cart_v)
        cclqr.setForce(batch, mech.eqconstraints[1], -fric[1] * pole_w)

    df = cclqr.simulate(mech, 2.0, mk(law_fric), z0=z0)
    ff = cclqr.simulate(mech, 2.0, mk(None), z0=z0, fric=fric)
    assert (df.status > 0).all() and np.abs(df.z - ff.z).max() < 1e-9

    @cclqr.on_device
    def law_poisoned(batch, lqr, k):
        u = cclqr.control_lqr(batch, lqr, k)[:, 0].clone()
        if k == 5:
            u[3] = float("nan")
        cclqr.setForce(batch, mech.eqconstraints[0], u)

    # graph=True: the horizon is captured once and replayed for every later batch of the same shape -- the closure itself runs at capture time only
    ran = []

    @cclqr.on_device(graph=True)
    def law_graph(batch, lqr, k):
        ran.append(k)
        cclqr.control_lqr(batch, lqr, k)

    lg = mk(law_graph)
    z1 = cclqr.examples.cartpole_states(1, rng.uniform(-0.5, 0.5, 16), rng.uniform(0.0, 0.3, (16, 1)))
    g0 = cclqr.simulate(mech, 3.0, lg, z0=z0)
    n_first = len(ran)
    g1 = cclqr.simulate(mech, 3.0, lg, z0=z1)
    assert n_first == 301 and len(ran) == 301                                   # (one warm-up call + the 300 captured steps; the replay calls nothing)
    assert np.array_equal(g0.z, dev.z) and np.array_equal(g0.zT, dev.zT) and np.array_equal(g0.status, dev.status)
    e1 = cclqr.simulate(mech, 3.0, mk(law), z0=z1)
    assert np.array_equal(g1.z, e1.z) and np.array_equal(g1.status, e1.status) and np.abs(g1.z - g0.z).max() > 1e-3

    clean, lost = cclqr.simulate(mech, 0.4, mk(law), z0=z0), cclqr.simulate(mech, 0.4, mk(law_poisoned), z0=z0)
    others = [i for i in range(16) if i != 3]
    assert np.array_equal(lost.z[others], clean.z[others]) and np.array_equal(lost.status[others], clean.status[others])
    assert lost.status[3] < 0 and (clean.status > 0).all()
    for k in range(5, 40):
        assert np.array_equal(lost.z[3, k, :, 0:7], lost.z[3, 4, :, 0:7]) and not lost.z[3, k, :, 7:].any()
    assert np.array_equal(lost.zT[3, :, 0:7], lost.z[3, 4, :, 0:7]) and not lost.zT[3, :, 7:].any()


def test_custom_controlfunction_keeps_a_lost_instance_frozen(cclqr, orc):
    """the fused rollout freezes an instance whose step ended on a non-finite residual (at its last pose, at rest, flagged) for the rest of the
    horizon; the step-per-launch path of a host `controlfunction` forgets that flag between launches, so lqr.py carries it: the lost instance's frozen
    state is restored after every later launch and its status stands.  A closure that hands instance 3 a NaN input at step 5 loses exactly that instance;
    the others equal the run without the poison bit for bit."""
    ex = cclqr.examples.cartpole_n(1)
    mech = ex["mech"]
    ids = [cclqr.getid(b) for b in ex["bodies"]]
    rng = np.random.default_rng(6)
    z0 = cclqr.examples.cartpole_states(1, rng.uniform(-0.5, 0.5, 8), rng.uniform(0.0, 0.3, (8, 1)))

    def law(batch, lqr, k):
        cclqr.control_lqr(batch, lqr, k)

    def law_poisoned(batch, lqr, k):
        u = cclqr.control_lqr(batch, lqr, k)[:, 0].copy()
        if k == 5:
            u[3] = np.nan
        cclqr.setForce(batch, mech.eqconstraints[0], u)

    mk = lambda f: cclqr.LQR(mech, ids, [cclqr.getid(ex["ctrl"][0])], ex["Q"], ex["R"], 10.0, xd=ex["xd"], controlfunction=f)
    clean = cclqr.simulate(mech, 0.4, mk(law), z0=z0)
    lost = cclqr.simulate(mech, 0.4, mk(law_poisoned), z0=z0)
    others = [i for i in range(8) if i != 3]
    assert np.array_equal(lost.z[others], clean.z[others]) and np.array_equal(lost.status[others], clean.status[others])
    assert lost.status[3] < 0 and (clean.status > 0).all()
    # frozen from the step it was lost in: the pose of knot 5 (what step 5 started from), at rest, in every later record and in the final state
    for k in range(5, 40):
        assert np.array_equal(lost.z[3, k, :, 0:7], lost.z[3, 4, :, 0:7]) and not lost.z[3, k, :, 7:].any()
    assert np.array_equal(lost.zT[3, :, 0:7], lost.z[3, 4, :, 0:7]) and not lost.zT[3, :, 7:].any()


def test_lqr_pendulum_inf_horizon(cclqr, orc):
    """examples/lqr_pendulum.jl: horizon = Inf -> K = [Ku[1]] (lqr.jl:40-43), always-on feedback (lqr.jl:116-139)"""
    ex = cclqr.examples.pendulum()
    mech = ex["mech"]
    lqr = cclqr.LQR(mech, [1], [2], ex["Q"], ex["R"], np.inf, xd=ex["xd"], qd=ex["qd"])
    assert lqr.K.shape == (1, 1, 12) and lqr.N == 0 and lqr.converged
    storage = cclqr.simulate(mech, 10, lqr)
    q = storage.zT[0, 0, 3:7]
    # ends at the upright setpoint RotX(pi) (up to the quaternion double cover)
    assert abs(abs(q[1]) - 1.0) < 1e-3


def test_tracking_pipeline(cclqr, orc):
    """TrackingLQR about an open-loop trajectory (examples/trackingLQR_triple_cartpole.jl:46-53,117), short horizon"""
    ex = cclqr.examples.triple_cartpole()
    mech = ex["mech"]
    N = 60
    U = 8.0 * np.sin(np.arange(N) * 0.15)
    z00 = mech.state()
    joint1 = ex["ctrl"][0]
    storage0 = cclqr.simulate(mech, cclqr.Storage(N, 4), cclqr.OpenLoop(mech, [joint1.id], U.reshape(N, 1)))
    t = mech.tables()
    ol = orc.ctrl_desc(4, [0], K=None, N=N + 1, zd=np.tile(z00, (N, 1, 1)), Fd=U.reshape(N, 1))
    _, ref, _ = orc.rollout(t, ol, z00[None], N, record=True)
    assert np.abs(storage0.z - ref).max() < 1e-9
    tl = cclqr.TrackingLQR(mech, storage0, [[[U[k]]] for k in range(N)], [joint1.id], ex["Q"], ex["R"])
    Ko, kbo = orc.riccati_tracking(t, [0], ref[0], U.reshape(N, 1), tl.Q, tl.R, N)
    assert tl.kbreak == kbo
    assert _rel(tl.K, Ko) < 1e-6
    # closed loop with friction and injected noise follows the oracle
    rng = np.random.default_rng(2)
    ninst = 4
    noise = rng.normal(size=(ninst, N))
    mech.set_state(z00)
    st = cclqr.simulate(mech, cclqr.Storage(N, 4), tl, z0=np.tile(z00, (ninst, 1, 1)), fric=ex["fric"], noise=noise, noise_scale=2.0)
    oc = orc.ctrl_desc(4, [0], K=tl.K, N=N, zd=tl.zd, Fd=tl.Fd, fric=ex["fric"], noise_scale=2.0, noise=noise)
    _, traj, _ = orc.rollout(t, oc, np.tile(z00, (ninst, 1, 1)), N, record=True)
    assert np.abs(st.z - traj).max() < 1e-9


def test_error_codes(cclqr):
    capi = cclqr._capi
    ex = cclqr.examples.cartpole_n(2)
    t = ex["mech"].tables()
    big = cclqr.examples.tree_mechanism([-1, 0, 0, 0, 0, 0])["mech"].tables()
    with pytest.raises(capi.CclqrError) as e:
        capi.MechHandle(big)   # five child joints on one body
    assert e.value.code == capi.EUNSUPPORTED
    mech = capi.MechHandle(t)
    with pytest.raises(capi.CclqrError) as e:
        capi.CtrlHandle(mech, [7])
    assert e.value.code == capi.EINVAL


def test_sawyer_config4_pipeline(cclqr, orc):
    """examples/lqr_sawyer.jl through the mirror: URDF numbers -> Mechanism -> LQR (mx = 84, mu = 7, ml = 35) -> batched simulate!"""
    import json
    import os
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sawyer_arm_tables.json")))
    ex = cclqr.examples.sawyer(tab)
    mech = ex["mech"]
    t = mech.tables()
    ids = [cclqr.getid(b) for b in mech.bodies]
    eids = [cclqr.getid(e) for e in mech.eqconstraints]
    lqr = cclqr.LQR(mech, ids, eids, ex["Q"], ex["R"], 2.0, xd=ex["xd"], qd=ex["qd"])       # 200-step horizon keeps the oracle fast
    assert lqr.K.shape == (199, 7, 84)
    Ao, Buo, Blo, Go = orc.linearize(t, lqr.zd[0], list(range(7)), np.zeros(7))
    for a, b in zip((lqr.A, lqr.Bu, lqr.Bλ, lqr.G), (Ao, Buo, Blo, Go)):
        assert _rel(a, b) < 1e-10
    Ko, kbo = orc.riccati(Ao, Buo, Blo, Go, lqr.Q, lqr.R, 200)
    assert lqr.kbreak == kbo and _rel(lqr.K, Ko) < 1e-7
    rng = np.random.default_rng(1)
    z0 = []
    for n in range(9):
        for e in mech.eqconstraints:
            cclqr.setJointPosition(mech, e, rng.uniform(-0.05, 0.05))      # q_j ~ U(-0.05, 0.05) about the zero pose
        z0.append(mech.state())
    z0 = np.stack(z0)
    st = cclqr.simulate(mech, 1.5, lqr, z0=z0)
    oc = orc.ctrl_desc(7, list(range(7)), K=lqr.K, N=lqr.N, zd=lqr.zd)
    _, traj, sto = orc.rollout(t, oc, z0, 150, record=True)
    assert (st.status > 0).all() and (sto > 0).all()
    assert np.abs(st.z - traj).max() < 1e-9


def test_projected_linear_model_and_lqr_on_the_deltabot(cclqr, orc):
    """examples/lqr_deltabot.jl:47-53 end to end.  The redundant constraint rows of a closed loop make G Bλ (lqr.jl:151) singular, but the
    pair the recursion works with -- A' = A - Bλ (G Bλ)^-1 G A, D = Bu - Bλ (G Bλ)^-1 G Bu -- is the Jacobian of the constrained one-step
    map and stays unique; `cclqr_linearize_projected` forms it on the device ANALYTICALLY (h <= 0: exact Jacobians in the loop's own
    bookkeeping, G Bλ eliminated with complete pivoting up to its numerical rank) or, as a cross-check, by central differences (h > 0).
    (i) on a TREE the analytic pair equals the numpy projection of the `cclqr_linearize` output to 1e-9 and the differences to 1e-7;
    (ii) on the deltabot it equals central differences (h = 1e-5) of the ORACLE's dense-KKT step (oracle/loops.py) to 2e-8 at |A'| = 200
    (the device's own differences: 1e-6), and the gains of the recursion (lqr.jl:141-184 with no multipliers left)
    equal the oracle's to 1e-7 relative;
    (iii) the LQR holds the mechanism: from a state 0.13 off the setpoint the closed loop returns to it, the open loop falls."""
    from oracle import loops
    capi = cclqr._capi
    ex = cclqr.examples.cartpole_n(3)
    t = ex["mech"].tables()
    zd = hanging_setpoint(cclqr, 3)
    mh = capi.MechHandle(t)
    A, Bu, Bl, G = (M[0] for M in capi.linearize(mh, zd[None], [0], np.zeros((1, 1))))
    AD = np.hstack([A, Bu]) - Bl @ np.linalg.solve(G @ Bl, G @ np.hstack([A, Bu]))
    Ap, D = capi.linearize_projected(mh, zd[None], [0], np.zeros((1, 1)))
    assert np.abs(Ap[0] - AD[:, :48]).max() < 1e-9 and np.abs(D[0] - AD[:, 48:]).max() < 1e-11 and np.abs(AD).max() > 100
    Af, Df = capi.linearize_projected(mh, zd[None], [0], np.zeros((1, 1)), h=1e-6)          # the differences of the device's step map
    assert np.abs(Af[0] - AD[:, :48]).max() < 1e-7 and np.abs(Df[0] - AD[:, 48:]).max() < 1e-9
    # deltabot
    ex = cclqr.examples.deltabot()
    mech = ex["mech"]
    z0 = mech.state()
    ids = [cclqr.getid(b) for b in mech.bodies]
    Q, R = [np.eye(12) for _ in ids], [np.eye(1) * 0.1 for _ in ex["eqcids"]]
    lq = cclqr.LQR(mech, ids, ex["eqcids"], Q, R, 10.0, xd=[z0[i, 0:3] for i in range(5)], qd=[z0[i, 3:7] for i in range(5)],
                   Fτd=[[ex["Fd"][0]], [ex["Fd"][1]]])
    assert lq.projected and lq.K.shape == (999, 2, 60) and lq.G.shape == (0, 60)
    lm, z, u = loops.deltabot()
    Ao, Do = loops.projected_linear_model(lm, z, u, [0, 1], h=1e-5)
    assert np.abs(lq.A - Ao).max() < 2e-8 and np.abs(lq.Bu - Do).max() < 2e-8 and np.abs(Ao).max() > 100
    Afd, Dfd = capi.linearize_projected(mech._cclqr_handle, z0[None], lq.ctrl_joints, ex["Fd"].reshape(1, 2), h=1e-6)
    assert np.abs(Afd[0] - lq.A).max() < 1e-6 and np.abs(Dfd[0] - lq.Bu).max() < 1e-6
    Ko, kbo = orc.riccati(Ao, Do, np.zeros((60, 0)), np.zeros((0, 60)), lq.Q, lq.R, 1000)
    assert int(lq.kbreak) == kbo and np.abs(lq.K - Ko).max() < 1e-7 * np.abs(Ko).max()
    # closed loop vs open loop from a consistent perturbed state (eight steps with 90 % of the holding torque)
    weak = capi.CtrlHandle(mech._cclqr_handle, lq.ctrl_joints, K=None, N=0, zd=z0[None], Fd=0.9 * ex["Fd"].reshape(1, 2))
    zp, _, st = capi.rollout(mech._cclqr_handle, weak, z0[None], 8)
    assert st[0] > 0 and 0.05 < np.abs(zp[0] - z0).max() < 0.3
    res = cclqr.simulate(mech, 3.0, lq, record=False, z0=zp)
    assert res.status[0] > 0 and np.abs(res.zT[0] - z0).max() < 5e-3
    hold = cclqr.OpenLoop(mech, ex["eqcids"], np.tile(ex["Fd"], (300, 1)))
    free = cclqr.simulate(mech, cclqr.Storage(300, 5), hold, record=False, z0=zp)
    assert free.status[0] < 0 or np.abs(free.zT[0] - z0).max() > 20 * np.abs(res.zT[0] - z0).max()      # the holding inputs alone do not bring it back


@pytest.mark.parametrize("n_links", [15, 16])
def test_projected_linear_model_default_call_on_the_long_chains(cclqr, n_links):
    """cclqr_linearize_projected(h <= 0) keeps [G Bλ | G A | G Bu] of a knot in one CU's LDS -- 175 KB for a 16-body mechanism, 198 KB for
    the 17-body headline chain -- so the default call used to come back with CCLQR_EUNSUPPORTED on exactly those; it now differences what does
    not fit (h = 1e-6) and is defined for every mechanism.  Checked against the numpy elimination of the multipliers from cclqr_linearize's
    four matrices (the reference's own D and A' of lqr.jl:151)."""
    capi = cclqr._capi
    ex = cclqr.examples.cartpole_n(n_links)
    t = ex["mech"].tables()
    zd = hanging_setpoint(cclqr, n_links)
    mh = capi.MechHandle(t)
    A, Bu, Bl, G = (M[0] for M in capi.linearize(mh, zd[None], [0], np.zeros((1, 1))))
    AD = np.hstack([A, Bu]) - Bl @ np.linalg.solve(G @ Bl, G @ np.hstack([A, Bu]))
    mx = 12 * t.nb
    Ap, D = capi.linearize_projected(mh, zd[None], [0], np.zeros((1, 1)))          # default h = 0
    scale = np.abs(AD).max()
    assert scale > 100
    assert np.abs(Ap[0] - AD[:, :mx]).max() < 1e-8 * scale and np.abs(D[0] - AD[:, mx:]).max() < 1e-8 * scale


def test_tracking_lqr_on_the_deltabot(cclqr, orc):
    """TrackingLQR (lqr_tracking.jl:17-43) on a closed-loop mechanism: per-knot projected models from the device (cclqr_linearize_projected,
    119 knots linearised analytically in one launch), the time-varying recursion on them (cclqr_riccati_tv, no multipliers left).
    The model of a mid-trajectory knot equals the oracle's dense-KKT differences about the same (state, input); the gains equal a numpy
    restatement of the recursion (lqr_tracking.jl:73-122) on the same models; and the controller pulls a perturbed start back onto the
    recorded trajectory where the replayed inputs alone drift away."""
    from oracle import loops
    capi = cclqr._capi
    ex = cclqr.examples.deltabot()
    mech = ex["mech"]
    z00 = mech.state()
    N = 120
    U = ex["Fd"][None, :] * (1.0 + 0.03 * np.sin(2 * np.pi * np.arange(N) / 60.0))[:, None]
    s0 = cclqr.simulate(mech, cclqr.Storage(N, 5), cclqr.OpenLoop(mech, ex["eqcids"], U))
    assert s0.status[0] > 0 and np.abs(s0.z[0, -1] - z00).max() > 1e-3                     # the platform is driven around
    mech.set_state(z00)
    Q, R = [np.eye(12) for _ in range(5)], [np.eye(1) * 0.1 for _ in range(2)]
    tl = cclqr.TrackingLQR(mech, s0, [[[U[k, 0]], [U[k, 1]]] for k in range(N)], ex["eqcids"], Q, R)
    assert tl.projected and tl.K.shape == (N - 1, 2, 60)
    # one knot's model against the oracle
    lm, z, u = loops.deltabot()
    k = 40
    uk = np.zeros(7); uk[:2] = U[k]
    Ao, Do = loops.projected_linear_model(lm, s0.z[0, k].copy(), uk, [0, 1], h=1e-5)
    Ap, D = capi.linearize_projected(mech._cclqr_handle, s0.z[0, k:k + 1], tl.ctrl_joints, U[k:k + 1])
    assert np.abs(Ap[0] - Ao).max() < 5e-8 and np.abs(D[0] - Do).max() < 5e-8
    # the recursion on the device's models, restated in numpy (ml = 0: D = Bu, lqr.jl:151-176 with the knot's matrices)
    Aall, Dall = capi.linearize_projected(mech._cclqr_handle, s0.z[0, :N - 1], tl.ctrl_joints, U[:N - 1])
    P, Kref = tl.Q.copy(), np.zeros_like(tl.K)
    kb = 0
    for kk in range(N - 1, 0, -1):
        A, B = Aall[kk - 1], Dall[kk - 1]
        Kk = np.linalg.solve(tl.R + B.T @ P @ B, B.T @ P @ A)
        Kref[kk - 1] = Kk
        Abar = A - B @ Kk
        Pn = tl.Q + Kk.T @ tl.R @ Kk + Abar.T @ P @ Abar
        kb = kk
        if np.linalg.norm(P - Pn) < 1e-5:
            break
        P = Pn
    for k2 in range(kb - 1, 0, -1):
        Kref[k2 - 1] = Kref[k2]
    assert int(tl.kbreak) == kb and np.abs(tl.K - Kref).max() < 1e-7 * np.abs(Kref).max()
    # tracking: start two steps ahead on the same motion (a consistent perturbed state), follow the recorded trajectory
    zp = s0.z[0, 2:3].copy()
    track = cclqr.simulate(mech, cclqr.Storage(N - 1, 5), tl, z0=zp)
    replay = cclqr.simulate(mech, cclqr.Storage(N - 1, 5), cclqr.OpenLoop(mech, ex["eqcids"], U), z0=zp)
    e_track = np.abs(track.z[0, -1] - s0.z[0, N - 2]).max()
    e_replay = np.abs(replay.z[0, -1] - s0.z[0, N - 2]).max()
    assert track.status[0] > 0 and e_track < 0.2 * e_replay and e_track < 5e-3, (e_track, e_replay)


def test_minimal_coordinate_lqr_prismatic(cclqr, orc):
    """examples/lqr_prismatic.jl: LQR(mech, getid.(constraints), getid.(constraints), Q::Vector, R::Vector, 10.)  (lqr.jl:68-86)"""
    ex = cclqr.examples.prismatic_slider()
    mech = ex["mech"]
    ids = [cclqr.getid(j) for j in ex["joints"]]
    lqr = cclqr.LQR(mech, ids, ids, ex["Q"], ex["R"], 10.0)
    assert lqr.K.shape == (999, 1, 12) and np.allclose(lqr.zd[0, 0, 0:3], 0) and np.allclose(lqr.Q, np.eye(12) * 0.01)
    z0 = mech.state()
    st = cclqr.simulate(mech, 10.0, lqr)
    assert abs(st.zT[0, 0, 0]) < 1e-2 and abs(st.z[0, 0, 0, 0] - 1.0) < 1e-12          # slides from x = 1 to the setpoint
    t = mech.tables()
    oc = orc.ctrl_desc(1, [0], K=lqr.K, N=lqr.N, zd=lqr.zd)
    _, traj, _ = orc.rollout(t, oc, z0[None], 1000, record=True)
    assert np.abs(st.z - traj).max() < 1e-9


def test_pid_controller(cclqr, orc):
    """examples/pid_pendulum.jl through the mirror: PID(mech, joint.id, pi/2, P=10, I=10, D=5); batch of start angles"""
    ex = cclqr.examples.pendulum(θ0=0.0)
    mech = ex["mech"]
    pid = cclqr.PID(mech, cclqr.getid(ex["joints"][0]), np.pi / 2, P=10.0, I=10.0, D=5.0)
    z0 = np.stack([cclqr.examples.pendulum(θ0=a)["mech"].state() for a in (0.0, 0.5, -1.0, 3.0, -3.0)])
    st = cclqr.simulate(mech, 10.0, pid, z0=z0)
    t = mech.tables()
    oc = orc.ctrl_desc(1, [], K=None, N=0, pid=dict(joint=[0], P=[10.0], I=[10.0], D=[5.0], goal=[np.pi / 2]))
    zo, traj, sto = orc.rollout(t, oc, z0, 1000, record=True)
    assert (st.status > 0).all() and (sto > 0).all()
    assert np.abs(st.z - traj).max() < 1e-9
    assert all(abs(np.angle(np.exp(1j * (orc.minimal_coordinates(t, st.zT[i])[0] - np.pi / 2)))) < 1e-2 for i in range(5))   # modulo 2π


def test_acrobot_underactuated_lqr(cclqr, orc):
    """examples/lqr_acrobot.jl through the mirror: only the elbow joint is actuated (eqcids = [constraints[2]]); gains against the
    oracle's pipeline, closed-loop rollout against the oracle, and the upright setpoint is held"""
    ex = cclqr.examples.acrobot()
    mech = ex["mech"]
    lqr = cclqr.LQR(mech, [cclqr.getid(b) for b in ex["bodies"]], [cclqr.getid(j) for j in ex["ctrl"]], ex["Q"], ex["R"], ex["horizon"],
                    xd=ex["xd"], qd=ex["qd"])
    assert lqr.K.shape == (999, 1, 24) and lqr.ctrl_joints == [1]
    t = mech.tables()
    A, Bu, Bl, G = orc.linearize(t, lqr.zd[0], [1], np.zeros(1))
    K_o, kb = orc.riccati(A, Bu, Bl, G, lqr.Q, lqr.R, 1000)
    assert np.abs(lqr.K - K_o).max() < 1e-6 * max(1.0, np.abs(K_o).max())
    z0 = mech.state()
    st = cclqr.simulate(mech, 10, lqr, record=True)
    oc = orc.ctrl_desc(2, [1], K=lqr.K, N=lqr.N, zd=lqr.zd)
    _, traj, sto = orc.rollout(t, oc, z0[None], 1000, record=True)
    assert (st.status > 0).all() and (sto > 0).all()
    assert np.abs(st.z - traj).max() < 1e-8
    th = orc.minimal_coordinates(t, st.zT[0])
    assert abs(np.angle(np.exp(1j * (th[0] - np.pi)))) < 0.05 and abs(th[1]) < 0.05       # balanced upright from 0.1 rad off


def test_pid_double_pendulum(cclqr, orc):
    """examples/pid_doublependulum.jl: PID(mech, getfield.(constraints,:id), [pi/2;-pi/4], P=[10,10], I=[10,10], D=[5,5])"""
    ex = cclqr.examples.double_pendulum()
    mech = ex["mech"]
    pid = cclqr.PID(mech, [cclqr.getid(j) for j in ex["joints"]], ex["goals"], P=ex["P"], I=ex["I"], D=ex["D"])
    z0 = np.stack([cclqr.examples.double_pendulum(a, b)["mech"].state() for a, b in ((0.0, 0.0), (0.4, -0.3), (-0.8, 0.5))])
    st = cclqr.simulate(mech, ex["tend"], pid, z0=z0)
    t = mech.tables()
    oc = orc.ctrl_desc(2, [], K=None, N=0, pid=dict(joint=[0, 1], P=ex["P"], I=ex["I"], D=ex["D"], goal=ex["goals"]))
    zo, traj, sto = orc.rollout(t, oc, z0, 1000, record=True)
    assert (st.status > 0).all() and (sto > 0).all()
    assert np.abs(st.z - traj).max() < 1e-8
    for i in range(3):
        th = orc.minimal_coordinates(t, st.zT[i])
        assert abs(th[0] - np.pi / 2) < 0.05 and abs(th[1] + np.pi / 4) < 0.05


def test_plain_c_program_through_the_abi(cclqr, orc, tmp_path):
    """examples/c_abi_cartpole.c (plain C, no Python in the process): lqr_cartpole.jl through mech_create -> linearize -> riccati ->
    ctrl_create -> rollout; its printed gains and final states against the oracle"""
    import subprocess
    from test_host import _build_c_example
    exe = _build_c_example(tmp_path)
    n = 5
    out = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "abi %d layout ok" % cclqr._capi.ABI_VERSION        # cclqr_abi_layout against the C compiler's own sizeof / offsetof
    lines = lines[1:]
    kb, kmax = int(lines[0].split()[1]), float(lines[0].split()[3])
    zT = np.array([[float(v) for v in ln.split()[5:]] for ln in lines[1:]]).reshape(n, 2, 13)
    st = [int(ln.split()[3]) for ln in lines[1:]]
    ex = cclqr.examples.cartpole_n(1)
    t = ex["mech"].tables()
    zd = upright_setpoint(1)
    A, Bu, Bl, G = orc.linearize(t, zd, [0], np.zeros(1))
    Ko, kbo = orc.riccati(A, Bu, Bl, G, np.eye(24) * t.dt, np.eye(1) * t.dt, 1000)
    assert kb == kbo and abs(kmax - np.abs(Ko).max()) < 1e-7 * np.abs(Ko).max()
    y0 = np.array([-0.4 + 0.8 * i / (n - 1) for i in range(n)])
    phi = np.array([[0.05 + 0.25 * i / (n - 1)] for i in range(n)])
    z0 = cclqr.examples.cartpole_states(1, y0, phi)
    zo, _, sto = orc.rollout(t, orc.ctrl_desc(2, [0], K=Ko, N=1000, zd=zd), z0, 1000)
    assert all(s > 0 for s in st) and (sto > 0).all()
    assert np.abs(zT - zo).max() < 1e-8


def test_riccati_bf16_split_mode_error_is_measured(cclqr, orc):
    """BASELINE configs[3] ("dense Riccati on MFMA bf16 -> fp32 accumulate") as a measured-error option (cclqr_riccati_opts.bf16_terms):
    the two mx^3 products of every backward step on v_mfma_f32_16x16x16_bf16 with fp32 accumulation, fp64 operands split into 1..3
    bf16 terms.  The fp64 MFMA path stays the parity mode; this test records what the option costs in accuracy on the Sawyer problem
    (mx 84, mu 7, ml 35): bf16x3 reproduces the gains to ~5e-6, bf16x2 to ~2e-4, plain bf16 does not reproduce them at all (the
    recursion amplifies its 4e-3 rounding to O(10)), and the 1e-5 break test of lqr.jl:172 never fires in any of them."""
    import json
    import os
    capi = cclqr._capi
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sawyer_arm_tables.json")))
    ex = cclqr.examples.sawyer(tab)
    t = ex["mech"].tables()
    mech = capi.MechHandle(t)
    A, Bu, Bl, G = (m[0] for m in capi.linearize(mech, ex["mech"].state()[None], list(range(7)), np.zeros((1, 7))))
    Q, R, N = np.eye(84) * 1000.0 * t.dt, np.eye(7) * t.dt, 600
    Kref, kbref = capi.riccati(A, Bu, Bl, G, Q, R, N, path=2)
    Ko, kbo = orc.riccati(A, Bu, Bl, G, Q, R, N)
    assert kbref == kbo and np.abs(Kref - Ko).max() < 1e-7 * np.abs(Ko).max()           # the parity mode
    err = {}
    for terms in (3, 2, 1):
        K, kb = capi.riccati(A, Bu, Bl, G, Q, R, N, bf16_terms=terms)
        assert np.isfinite(K).all() and kb == 1                                         # fp32 noise in P: the break test never fires
        err[terms] = np.abs(K - Kref).max() / np.abs(Kref).max()
    assert err[3] < 1e-4 and err[3] < err[2] < 1e-2 and err[1] > 1e-2, err
    with pytest.raises(capi.CclqrError):
        capi.riccati(A, Bu, Bl, G, Q, R, N, bf16_terms=4)


def test_bench_line_contract(tmp_path):
    """`python bench.py` prints ONE JSON line with the fields the driver reads: metric/value/unit, n_gpus, steps, warmup, ms_per_step,
    scaling, dtype, config.workload, roofline {bound, achieved, peak, unit, frac, traffic}, cpu_baseline {value, unit, cores, kind, sample}"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--instances", "512", "--sim-steps", "40",
                        "--cpu-sample", "64"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-12
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] in ("port", "reference")
    assert abs(d["value"] - 512 * 40 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_riccati_bf16_mode_on_the_batched_resident_kernel(cclqr, orc):
    """BASELINE configs[3] as written -- "dense Riccati on MFMA bf16 -> fp32 accumulate" on the BATCHED recursion: the
    LDS-resident kernel (one workgroup per problem, path = 1) with cclqr_riccati_opts.bf16_terms = 1..3 on distinct Sawyer setpoints.
    Measured, not promised: bf16x3 reproduces the fp64 gains of the same kernel to ~1e-5, bf16x2 to ~1e-3, plain bf16 not at all; the
    fp64 MFMA mode stays the parity mode (= oracle to 1e-7, checked here for one problem)."""
    import json
    import os
    capi = cclqr._capi
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sawyer_arm_tables.json")))
    ex = cclqr.examples.sawyer(tab)
    mech = ex["mech"]
    t = mech.tables()
    n, N = 12, 400
    rng = np.random.default_rng(8)
    zd = cclqr.joint_position_states(mech, rng.uniform(-0.6, 0.6, (n, 7)))
    mh = capi.MechHandle(t)
    A, Bu, Bl, G = capi.linearize(mh, zd, list(range(7)), np.zeros((n, 7)))
    Q, R = np.eye(84) * 1000.0 * t.dt, np.eye(7) * t.dt
    Kref, kbref = capi.riccati(A, Bu, Bl, G, Q, R, N, path=1, keep_last=True)
    Ko, kbo = orc.riccati(A[0], Bu[0], Bl[0], G[0], Q, R, N)
    assert kbo == kbref[0] and np.abs(Kref[0, 0] - Ko[0]).max() < 1e-7 * np.abs(Ko[0]).max()
    err = {}
    for terms in (3, 2, 1):
        K, kb = capi.riccati(A, Bu, Bl, G, Q, R, N, path=1, keep_last=True, bf16_terms=terms)
        assert np.isfinite(K).all()
        err[terms] = float(np.abs(K - Kref).max() / np.abs(Kref).max())
    print("batched resident Riccati, bf16 split terms -> relative gain error vs fp64 MFMA:", err)
    assert err[3] < 1e-3 and err[3] < err[2] < 5e-2 and err[1] > err[2], err
