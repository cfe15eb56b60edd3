"""GPU parity: HIP rollout kernel (through the C-ABI) vs the CPU oracle on the same seeded inputs.
Tolerance: fp64, max |state difference| over the whole trajectory < 1e-9 (north star asks < 1e-8)."""
import numpy as np
import pytest
import scipy.linalg as sl

from conftest import hanging_setpoint, upright_setpoint

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _gains(orc, ex, t, zd, N=1000):
    A, Bu, Bl, G = orc.linearize(t, zd, [0], np.zeros(1))
    Q = sl.block_diag(*ex["Q"]) * t.dt
    R = sl.block_diag(*ex["R"]) * t.dt
    K, _ = orc.riccati(A, Bu, Bl, G, Q, R, N)
    return K


# one case per instantiation of the chain kernel's (lanes, layout links): (16, 8) x2, (16, 8) full, (32, 16), (32, 17), (32, 32); odd instance
# counts leave lane groups of the last wavefront without an instance
@pytest.mark.parametrize("n_links,ninst,steps,hanging", [(1, 37, 300, False), (3, 9, 200, False), (7, 5, 150, True), (11, 5, 100, True),
                                                          (16, 3, 120, True), (22, 3, 60, True)])
def test_rollout_matches_oracle(cclqr, orc, n_links, ninst, steps, hanging):
    capi = cclqr._capi
    ex = cclqr.examples.cartpole_n(n_links)
    t = ex["mech"].tables()
    zd = hanging_setpoint(cclqr, n_links) if hanging else upright_setpoint(n_links)
    K = _gains(orc, ex, t, zd, N=steps + 50)
    rng = np.random.default_rng(7)
    phi = rng.uniform(-1, 1, (ninst, n_links)) * (0.3 if hanging else 0.3 / 3 ** n_links)
    if hanging:
        phi[:, 0] += np.pi
    z0 = cclqr.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, ninst), phi)
    octrl = orc.ctrl_desc(t.nb, [0], K=K, N=steps + 50, zd=zd)
    zT_o, traj_o, st_o = orc.rollout(t, octrl, z0, steps, record=True)
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, [0], K=K, N=steps + 50, zd=zd)
    zT, traj, st = capi.rollout(mech, ctrl, z0, steps, record=True)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < TOL
    assert np.abs(zT - zT_o).max() < TOL
    # record=False gives the same final state; two launches continue exactly where one left off
    zT2, _, _ = capi.rollout(mech, ctrl, z0, steps, record=False)
    assert np.array_equal(zT2, zT)


def test_tracking_friction_noise(cclqr, orc):
    """TrackingLQR-style controller tables (per-step setpoints/feed-forward), friction and injected noise"""
    capi = cclqr._capi
    ex = cclqr.examples.triple_cartpole()
    t = ex["mech"].tables()
    N, ninst = 80, 6
    rng = np.random.default_rng(11)
    U = 5.0 * np.sin(np.arange(N) * 0.1)
    z00 = ex["mech"].state()
    ol = orc.ctrl_desc(t.nb, [0], K=None, N=N + 1, zd=np.tile(z00, (N, 1, 1)), Fd=U.reshape(N, 1))
    _, ref, _ = orc.rollout(t, ol, z00[None], N, record=True)
    K = rng.normal(size=(N - 1, 1, 12 * t.nb)) * 0.5
    noise = rng.normal(size=(ninst, N))
    kw = dict(K=K, N=N, zd=ref[0], Fd=U.reshape(N, 1), fric=ex["fric"], noise_scale=2.0)
    z0 = np.tile(z00, (ninst, 1, 1))
    zT_o, traj_o, st_o = orc.rollout(t, orc.ctrl_desc(t.nb, [0], noise=noise, **kw), z0, N, record=True)
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, [0], **kw)
    zT, traj, st = capi.rollout(mech, ctrl, z0, N, noise=noise, record=True)
    assert (st > 0).all()
    assert np.abs(traj - traj_o).max() < TOL


def test_philox_noise_and_instance_offset(cclqr, orc):
    """noise_philox: device-generated Philox-4x32 / Box-Muller stream per GLOBAL instance == the oracle's; a shard launched with
    cclqr_rollout_opts.first_instance reproduces its slice of the whole batch bit for bit"""
    capi = cclqr._capi
    ex = cclqr.examples.triple_cartpole()
    t = ex["mech"].tables()
    N, ninst = 120, 50
    z00 = ex["mech"].state()
    zd = np.tile(z00, (N, 1, 1))
    K = np.random.default_rng(2).normal(size=(N - 1, 1, 48)) * 0.3
    kw = dict(K=K, N=N, zd=zd, fric=ex["fric"], noise_scale=2.0, noise_seed=0xC0FFEE)
    z0 = np.tile(z00, (ninst, 1, 1))
    _, traj_o, _ = orc.rollout(t, orc.ctrl_desc(t.nb, [0], **kw), z0, N, record=True)
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, [0], **kw)
    zT, traj, st = capi.rollout(mech, ctrl, z0, N, record=True)
    assert (st > 0).all()
    assert np.abs(traj[0] - traj[1]).max() > 1e-3
    assert np.abs(traj - traj_o).max() < TOL
    zT_s, _, _ = capi.rollout(mech, ctrl, z0[20:35], N, first_instance=20)
    assert np.array_equal(zT_s, zT[20:35])


def test_chained_device_launches_equal_one_launch(cclqr, orc):
    """step-per-launch / MPC-style use of cclqr_rollout_dev: state and multipliers round-trip HBM between launches (k0 continuation);
    40 + 1 + 59 steps in three launches == 100 steps in one launch, bit for bit, on device pointers and a non-default stream"""
    import torch
    capi = cclqr._capi
    ex = cclqr.examples.cartpole_n(3)
    t = ex["mech"].tables()
    zd = upright_setpoint(3)
    K = _gains(orc, ex, t, zd, N=150)
    rng = np.random.default_rng(5)
    n = 37
    z0 = cclqr.examples.cartpole_states(3, rng.uniform(-0.5, 0.5, n), rng.uniform(-1, 1, (n, 3)) * 0.01)
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, [0], K=K, N=150, zd=zd)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        z0_d = torch.from_numpy(z0).to(dev)
        one = torch.empty_like(z0_d)
        st = torch.zeros(n, dtype=torch.int32, device=dev)
        traj = torch.empty((n, 100, t.nb, 13), dtype=torch.float64, device=dev)
        capi.rollout_dev(mech, ctrl, n, 100, 1, z0_d.data_ptr(), 0, 0, 0, traj.data_ptr(), one.data_ptr(), st.data_ptr(), stream.cuda_stream)
        lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=dev)
        a, b, c = torch.empty_like(z0_d), torch.empty_like(z0_d), torch.empty_like(z0_d)
        capi.rollout_dev(mech, ctrl, n, 40, 1, z0_d.data_ptr(), lam.data_ptr(), 0, 0, 0, a.data_ptr(), st.data_ptr(), stream.cuda_stream)
        capi.rollout_dev(mech, ctrl, n, 1, 41, a.data_ptr(), lam.data_ptr(), 0, 0, 0, b.data_ptr(), st.data_ptr(), stream.cuda_stream)
        capi.rollout_dev(mech, ctrl, n, 59, 42, b.data_ptr(), lam.data_ptr(), 0, 0, 0, c.data_ptr(), st.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert torch.equal(c, one)
    assert torch.equal(a, traj[:, 40]) and torch.equal(b, traj[:, 41])
    octrl = orc.ctrl_desc(t.nb, [0], K=K, N=150, zd=zd)
    zo, _, _ = orc.rollout(t, octrl, z0, 100)
    assert np.abs(one.cpu().numpy() - zo).max() < TOL


def test_pid_state_carried_across_launches(cclqr, orc):
    """PID (pid.jl:69-88) with step-per-launch use: with cclqr_rollout_opts.pid_state_dev the integrated / last errors survive between launches,
    so 30 + 70 steps in two launches equal 100 steps in one (and equal the oracle)"""
    import torch
    capi = cclqr._capi
    ex = cclqr.examples.double_pendulum(0.2, -0.1)
    t = ex["mech"].tables()
    pid = dict(joint=[0, 1], P=ex["P"], I=ex["I"], D=ex["D"], goal=ex["goals"])
    z0 = np.stack([cclqr.examples.double_pendulum(a, b)["mech"].state() for a, b in ((0.2, -0.1), (-0.5, 0.3), (0.0, 0.0))])
    n = len(z0)
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, [], K=None, N=0, pid=pid)
    dev = torch.device("cuda", 0)
    z0_d = torch.from_numpy(z0).to(dev)
    one, a, b = torch.empty_like(z0_d), torch.empty_like(z0_d), torch.empty_like(z0_d)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    capi.rollout_dev(mech, ctrl, n, 100, 1, z0_d.data_ptr(), 0, 0, 0, 0, one.data_ptr(), st.data_ptr())
    lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=dev)
    pstate = torch.zeros((n, t.nb, 2), dtype=torch.float64, device=dev)
    # explicit options (cclqr_rollout_ex): the buffer is an argument of the launch, its length is checked
    capi.rollout_dev(mech, ctrl, n, 30, 1, z0_d.data_ptr(), lam.data_ptr(), 0, 0, 0, a.data_ptr(), st.data_ptr(), pid_state=pstate.data_ptr())
    capi.rollout_dev(mech, ctrl, n, 70, 31, a.data_ptr(), lam.data_ptr(), 0, 0, 0, b.data_ptr(), st.data_ptr(), pid_state=pstate.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(b, one)
    # a controller without a PID law never touches a PID buffer handed to its launch
    ex2 = cclqr.examples.cartpole_n(1)
    t2 = ex2["mech"].tables()
    m2 = capi.MechHandle(t2)
    c2 = capi.CtrlHandle(m2, [0], K=np.zeros((5, 1, 24)), N=6, zd=None)
    zc = torch.from_numpy(cclqr.examples.cartpole_states(1, np.linspace(-0.3, 0.3, 40), np.full((40, 1), 0.1))).to(dev)
    pst2 = torch.full((40, t2.nb, 2), 7.0, dtype=torch.float64, device=dev)
    keep = pst2.clone()
    capi.rollout_dev(m2, c2, 40, 3, 1, zc.data_ptr(), 0, 0, 0, 0, torch.empty_like(zc).data_ptr(), torch.zeros(40, dtype=torch.int32, device=dev).data_ptr(),
                     pid_state=pst2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(keep, pst2)
    zo, _, _ = orc.rollout(t, orc.ctrl_desc(2, [], K=None, N=0, pid=pid), z0, 100)
    assert np.abs(one.cpu().numpy() - zo).max() < TOL
    # without the buffer the integrators restart: a different trajectory
    capi.rollout_dev(mech, ctrl, n, 70, 31, a.data_ptr(), lam.data_ptr(), 0, 0, 0, b.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    assert not torch.equal(b, one)


def test_step_per_launch_chain_captured_in_a_hip_graph(cclqr, orc):
    """configs[4] asks for a hipGraph-captured step: 20 single-step launches of cclqr_rollout_dev (state and multipliers round-trip
    HBM between them) captured once on a stream into a graph and replayed == one 20-step launch, bit for bit"""
    import torch
    capi = cclqr._capi
    ex = cclqr.examples.triple_cartpole()
    t = ex["mech"].tables()
    N = 40
    rng = np.random.default_rng(2)
    z00 = ex["mech"].state()
    zd = np.tile(z00, (N, 1, 1))
    K = rng.normal(size=(N - 1, 1, 48)) * 0.3
    n = 256
    z0 = np.tile(z00, (n, 1, 1))
    z0[:, 0, 1] += rng.uniform(-0.1, 0.1, n)
    z0[:, 1:, 1] += (z0[:, 0, 1] - z00[0, 1])[:, None]          # shift the whole mechanism with the cart: still consistent
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, [0], K=K, N=N, zd=zd, fric=ex["fric"])
    dev = torch.device("cuda", 0)
    z0_d = torch.from_numpy(z0).to(dev)
    ref = torch.empty_like(z0_d)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    capi.rollout_dev(mech, ctrl, n, 20, 1, z0_d.data_ptr(), 0, 0, 0, 0, ref.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    za, zb = z0_d.clone(), torch.empty_like(z0_d)
    lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=dev)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        capi.rollout_dev(mech, ctrl, n, 1, 1, za.data_ptr(), lam.data_ptr(), 0, 0, 0, zb.data_ptr(), st.data_ptr(), side.cuda_stream)   # warm (loads the code object)
        side.synchronize()
        za.copy_(z0_d)
        lam.zero_()
        side.synchronize()
        graph.capture_begin()
        src, dst = za, zb
        for k in range(1, 21):
            capi.rollout_dev(mech, ctrl, n, 1, k, src.data_ptr(), lam.data_ptr(), 0, 0, 0, dst.data_ptr(), st.data_ptr(), side.cuda_stream)
            src, dst = dst, src
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(src, ref)
    # replay on fresh inputs through the same captured buffers
    za.copy_(z0_d)
    lam.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(src, ref)
    zo, _, _ = orc.rollout(t, orc.ctrl_desc(t.nb, [0], K=K, N=N, zd=zd, fric=ex["fric"]), z0[:16], 20)
    assert np.abs(ref[:16].cpu().numpy() - zo).max() < TOL


def test_per_instance_controller_tables(cclqr, orc):
    """SURVEY 8d cfg4 / VERDICT r1 item 7: a batch in which every instance has its OWN gains, setpoint and feed-forward
    (cclqr_ctrl_desc.n_ctrl), chain kernel (triple cartpole) and tree kernel (dual-pole cart); a shard launched with
    first_instance reads its slice of the tables"""
    capi = cclqr._capi
    rng = np.random.default_rng(21)
    for ex in (cclqr.examples.triple_cartpole(), cclqr.examples.dual_cartpole()):
        if ex is None:
            continue
        t = ex["mech"].tables()
        n, N = 24, 30
        z00 = ex["mech"].state()
        zd = np.tile(z00, (n, 1, 1, 1))                      # [n][nsp = 1][nb][13]
        zd[:, 0, 0, 1] += rng.uniform(-0.2, 0.2, n)          # every instance regulates the cart to its own position
        K = rng.normal(size=(n, N - 1, 1, 12 * t.nb)) * 0.3
        Fd = rng.normal(size=(n, 1, 1)) * 0.5
        z0 = np.tile(z00, (n, 1, 1))
        mech = capi.MechHandle(t)
        ctrl = capi.CtrlHandle(mech, [0], K=K, N=N, zd=zd, Fd=Fd, n_ctrl=n)
        zT, traj, st = capi.rollout(mech, ctrl, z0, N, record=True)
        oc = orc.ctrl_desc(t.nb, [0], K=K, N=N, zd=zd, Fd=Fd, n_ctrl=n)
        zo, trajo, sto = orc.rollout(t, oc, z0, N, record=True)
        assert (st > 0).all() and (sto > 0).all()
        assert np.abs(traj - trajo).max() < TOL
        assert np.abs(traj[0] - traj[1]).max() > 1e-3      # the tables really differ
        # shard [8, 20): first_instance selects the tables
        import torch
        dev = torch.device("cuda", 0)
        zs = torch.from_numpy(z0[8:20]).to(dev)
        out = torch.empty_like(zs)
        sts = torch.zeros(12, dtype=torch.int32, device=dev)
        capi.rollout_dev(mech, ctrl, 12, N, 1, zs.data_ptr(), 0, 0, 0, 0, out.data_ptr(), sts.data_ptr(), first_instance=8)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), zT[8:20])
        with pytest.raises(capi.CclqrError):
            capi.rollout_dev(mech, ctrl, 12, N, 1, zs.data_ptr(), 0, 0, 0, 0, out.data_ptr(), sts.data_ptr(), first_instance=20)


def test_closed_loop_free_fall_energy_error_is_first_order(cclqr):
    """a physics pin for the closed-loop kernel that needs no oracle: the deltabot released with no joint inputs falls for 0.3 s (a large
    motion: velocities of several m/s); the total mechanical energy of the recorded states stays within 1.2 % at dt = 0.01 and the
    error halves with the step (0.27, 0.135, 0.068 J at dt, dt/2, dt/4: the first-order scheme of SURVEY 8a-bis), i.e. the constraint
    forces of the loops do no net work"""
    capi = cclqr._capi

    def drift(dt, steps):
        ex = cclqr.examples.deltabot()
        mech = ex["mech"]
        mech.Δt = dt
        t = mech.tables()
        z0 = mech.state()
        mh = capi.MechHandle(t)
        c = capi.CtrlHandle(mh, [0, 1], K=None, N=0, zd=z0[None], Fd=np.zeros((1, 2)))
        zT, traj, st = capi.rollout(mh, c, z0[None], steps, record=True)
        assert st[0] > 0
        E = np.array([sum(0.5 * t.mass[b] * z[b, 7:10] @ z[b, 7:10] + 0.5 * z[b, 10:13] @ t.inertia[b].reshape(3, 3) @ z[b, 10:13] + t.mass[b] * 9.81 * z[b, 2]
                          for b in range(t.nb)) for z in traj[0]])
        assert np.abs(traj[0, -1, :, 7:10]).max() > 2.0          # it really falls
        return np.abs(E - E[0]).max(), E[0]

    d1, E0 = drift(0.01, 30)
    d2, _ = drift(0.005, 60)
    d3, _ = drift(0.0025, 120)
    assert d1 < 0.015 * E0 and 1.8 < d1 / d2 < 2.2 and 1.8 < d2 / d3 < 2.2


def test_forest_with_a_long_and_a_short_chain(cclqr, orc):
    """one mechanism, two chains: 13 links (odd-even reduction level ahead of the sweep) and 3 links (plain sweep), interleaved body
    numbering, one controlled joint each"""
    from conftest import long_and_short_chain_forest
    capi = cclqr._capi
    t2, z0, zd, K, cj = long_and_short_chain_forest(cclqr)
    z0 = np.repeat(z0, 3, 0)
    z0[1, :, 7:10] += 0.01
    oc = orc.ctrl_desc(t2.nb, cj, K=K, N=21, zd=zd)
    zT_o, traj_o, st_o = orc.rollout(t2, oc, z0[:1], 20, record=True)
    mech = capi.MechHandle(t2)
    ctrl = capi.CtrlHandle(mech, cj, K=K, N=21, zd=zd)
    zT, traj, st = capi.rollout(mech, ctrl, z0, 20, record=True)
    assert (st_o > 0).all() and (st > 0).all() and mech.geometry()[0] == 32
    assert np.abs(traj[0] - traj_o[0]).max() < TOL and np.array_equal(traj[0], traj[2])


def test_closed_loop_deltabot_rollout_matches_oracle(cclqr, orc):
    """examples/lqr_deltabot.jl:25-53 through the C-ABI (`rollout_loop_kernel`, csrc/rollout_loop.hip): 33 constraint rows on 30 body
    coordinates.  (i) the reference's own number: Fτd = +-6.7879484 on the platform joints holds the script's pose at rest; (ii) with
    80 % of it, open loop and with a feedback law u = Fτd - K dz on top, every instance's trajectory equals the oracle's dense-KKT
    minimum-norm solution (oracle/loops.py) to 1e-9; (iii) the batch is bitwise independent of the instance's position; (iv) linearsystem
    on the loop mechanism returns A, Bu, Bλ, G with rank(G Bλ) = 28 of 35 -- the reference's own recursion form is refused on it
    (CCLQR_ESINGULAR), the projected pair is what LQR construction uses"""
    from oracle import loops
    from test_emulated_kernel import loop_feedback_reference
    capi = cclqr._capi
    ex = cclqr.examples.deltabot()
    mech_py = ex["mech"]
    t = mech_py.tables()
    cj = [mech_py.joint_index(e) for e in ex["eqcids"]]
    lm, z, u = loops.deltabot()
    z0 = mech_py.state()
    zd = z0[None].copy()
    mech = capi.MechHandle(t)
    assert mech.geometry()[0] == 64
    hold = capi.CtrlHandle(mech, cj, K=None, N=0, zd=zd, Fd=ex["Fd"].reshape(1, 2))
    zT, _, st = capi.rollout(mech, hold, z0[None], 50)            # (the torque is given to eight digits: the residual acceleration ~5e-8 m/s2 grows from there)
    assert st[0] > 0 and np.abs(zT[0] - z0).max() < 1e-6
    steps, n = 25, 5
    rng = np.random.default_rng(3)
    K = rng.normal(size=(n, 1, 2, 12 * t.nb)) * 2.0
    K[0] = 0.0                                                      # instance 0: open loop
    scale = np.linspace(0.8, 0.6, n)
    Fd = scale[:, None, None] * ex["Fd"].reshape(1, 1, 2)
    ctrl = capi.CtrlHandle(mech, cj, K=K, N=0, zd=np.repeat(zd[None], n, 0), Fd=Fd, n_ctrl=n)
    zb = np.repeat(z0[None], n, 0)
    zT, traj, st = capi.rollout(mech, ctrl, zb, steps, record=True)
    assert (st > 0).all()
    for i in range(n):
        ref, zref = loop_feedback_reference(lm, z.copy(), Fd[i, 0], None if i == 0 else K[i, 0], zd[0], steps)
        assert np.abs(traj[i] - ref).max() < TOL and np.abs(zT[i] - zref).max() < TOL, i
        assert np.abs(lm.constraints(zT[i])).max() < 1e-12
    assert np.abs(zT[0] - z0).max() > 0.5
    one = capi.CtrlHandle(mech, cj, K=K[3], N=0, zd=zd, Fd=Fd[3])
    zT1, traj1, _ = capi.rollout(mech, one, z0[None], steps, record=True)
    assert np.array_equal(zT1[0], zT[3]) and np.array_equal(traj1[0], traj[3])
    # linearsystem of the loop mechanism: A, Bu, Bλ, G with the multipliers exogenous come out (35 constraint rows, two of them the
    # FixedOrientation's null rows), G Bλ has rank 28, and eliminating λ in numpy gives the pair cclqr_linearize_projected forms on the device
    A, Bu, Bl, G = (M[0] for M in capi.linearize(mech, zd, cj, Fd[3].reshape(1, -1)))
    assert G.shape == (35, 60) and Bl.shape == (60, 35)
    sv = np.linalg.svd(G @ Bl, compute_uv=False)
    assert int((sv > 1e-9 * sv[0]).sum()) == 28
    XY = np.linalg.lstsq(G @ Bl, G @ np.hstack([A, Bu]), rcond=1e-11)[0]
    AD = np.hstack([A, Bu]) - Bl @ XY
    Ap, D = capi.linearize_projected(mech, zd, cj, Fd[3].reshape(1, -1))
    assert np.abs(Ap[0] - AD[:, :60]).max() < 1e-9 * np.abs(AD).max() and np.abs(D[0] - AD[:, 60:]).max() < 1e-9 * np.abs(AD).max()
    with pytest.raises(capi.CclqrError) as e:
        capi.riccati(A, Bu, Bl, G, np.eye(60) * 0.01, np.eye(len(cj)) * 0.01, 20)      # the reference's own form divides by the singular G Bλ (lqr.jl:151)
    assert e.value.code == capi.ESINGULAR
    # round 4: the friction / noise law and newton_mode 1 on a loop mechanism.  Friction per joint in the caller's joint order,
    # noise injected per (instance, step) on the controlled joints: every instance's trajectory equals the dense-KKT reference under the same law
    fric = np.array([0.5, 0.3, 0.4, 0.2, 0.6, 0.0, 0.0])
    noise = np.random.default_rng(8).normal(size=(n, steps))
    fn = capi.CtrlHandle(mech, cj, K=K, N=0, zd=np.repeat(zd[None], n, 0), Fd=Fd, n_ctrl=n, fric=fric, noise_scale=0.7)
    zTf, trajf, stf = capi.rollout(mech, fn, zb, steps, record=True, noise=noise)
    assert (stf > 0).all()
    for i in (0, 2, 4):
        ref, zref = loop_feedback_reference(lm, z.copy(), Fd[i, 0], None if i == 0 else K[i, 0], zd[0], steps, fric=fric, noise=noise[i], noise_scale=0.7)
        assert np.abs(trajf[i] - ref).max() < TOL and np.abs(zTf[i] - zref).max() < TOL, i
    assert np.abs(zTf - zT).max() > 1e-3
    zT1, _, st1 = capi.rollout(mech, ctrl, zb, steps, newton_mode=1, newton_eps_alone=1e-12)      # the measured-error stop on the loop kernel
    assert (st1 > 0).all() and (st1 <= st).all() and 0.0 <= np.abs(zT1 - zT).max() < 1e-8
    # PID on a loop mechanism (pid.jl:69-88 on the two actuated platform joints, lp_pid): against the dense-KKT reference under the same law, and
    # the integrators carried between two launches through pid_state_dev ([n_inst][joints][2] for a loop mechanism) equal one launch
    from tests.test_emulated_kernel import loop_joint_coordinate
    pidd = dict(joint=cj, P=[8.0, 6.0], I=[3.0, 2.0], D=[0.4, 0.3], goal=[loop_joint_coordinate(lm.joints[j], z) + d for j, d in zip(cj, (0.15, -0.1))])
    pc = capi.CtrlHandle(mech, cj, K=None, N=0, zd=zd, Fd=ex["Fd"].reshape(1, 2), pid=pidd)
    zTp, trajp, stp = capi.rollout(mech, pc, z0[None], steps, record=True)
    refp, zrefp = loop_feedback_reference(lm, z.copy(), ex["Fd"], None, zd[0], steps, pid=pidd)
    assert stp[0] > 0 and np.abs(trajp[0] - refp).max() < TOL and np.abs(zTp[0] - zrefp).max() < TOL
    assert np.abs(zTp[0] - z0).max() > 1e-2
    import torch
    dev = torch.device("cuda", 0)
    zt = torch.from_numpy(z0[None].copy()).to(dev); zo = torch.empty_like(zt)
    lamt = torch.zeros((1, 5 * t.ne), dtype=torch.float64, device=dev); stt = torch.zeros(1, dtype=torch.int32, device=dev)
    pst = torch.zeros((1, t.ne, 2), dtype=torch.float64, device=dev)
    h = steps // 2
    capi.rollout_dev(mech, pc, 1, h, 1, zt.data_ptr(), lamt.data_ptr(), 0, 0, 0, zo.data_ptr(), stt.data_ptr(), 0, pid_state=pst.data_ptr())
    capi.rollout_dev(mech, pc, 1, steps - h, h + 1, zo.data_ptr(), lamt.data_ptr(), 0, 0, 0, zt.data_ptr(), stt.data_ptr(), 0, pid_state=pst.data_ptr())
    torch.cuda.synchronize()
    assert np.abs(zt.cpu().numpy()[0] - zTp[0]).max() < 1e-12



def test_fourbar_linkage_rollout_matches_oracle(cclqr, orc):
    """a second closed-loop topology on `rollout_loop_kernel<3>` (20 constraint rows of rank 17: three redundant directions, another pattern
    than the deltabot's): a batch of parallelogram four-bar linkages at different crank angles under a feedback law on the crank, every
    instance against the dense-KKT reference built from the same tables (oracle/loops.py from_tables)"""
    from oracle import loops
    capi = cclqr._capi
    steps, n = 40, 4
    exs = [cclqr.examples.fourbar(θ=th) for th in (0.6, 0.3, -0.4, 1.1)]
    t = exs[0]["mech"].tables()
    lm = loops.from_tables(t)
    z0 = np.array([e["mech"].state() for e in exs])
    rng = np.random.default_rng(11)
    K = rng.normal(size=(n, 1, 1, 12 * t.nb)) * 1.5
    Fd = rng.uniform(-1.0, 1.0, (n, 1, 1))
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, [0], K=K, N=0, zd=z0[:, None], Fd=Fd, n_ctrl=n)
    zT, traj, st = capi.rollout(mech, ctrl, z0, steps, record=True)
    assert (st > 0).all()
    for i in range(n):
        z, lam = z0[i].copy(), np.zeros(lm.nrows)
        for k in range(steps):
            assert np.abs(traj[i, k] - z).max() < 1e-8, (i, k)        # north_star's tolerance: the linkage swings at up to 5 rad/s and the
            u = np.zeros(4)                                             # reference's own Newton stops at |f| < 1e-10 per step (1.2e-9 after 24 steps)
            u[0] = Fd[i, 0, 0] - K[i, 0, 0] @ loops.state_error(z, z0[i])
            z, lam, _ = lm.step(z, lam, u)
        assert np.abs(zT[i] - z).max() < 1e-8 and np.abs(lm.constraints(zT[i])).max() < 1e-11


_SHARDED_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
rank, world, local = pkg.dist.init_from_env(backend="gloo")       # two ranks share the box's one GPU: RCCL refuses that, gloo carries the bytes
torch.cuda.set_device(0); capi.set_device(0)
dev = torch.device("cuda", 0)
n_links, n_local, T, H = 16, 48, 64, 4
ex = pkg.examples.cartpole_n(n_links); t = ex["mech"].tables(); nb = t.nb
zd = pkg.examples.cartpole_states(n_links, [0.0], np.array([[np.pi] + [0.0] * (n_links - 1)]))[0]
rng = np.random.default_rng(7)
K = rng.normal(size=(T - 1, 1, 12 * nb)) * 0.05
phi = rng.uniform(-0.2, 0.2, (n_local * world, n_links)); phi[:, 0] += np.pi
z0_all = pkg.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, n_local * world), phi)
mh = capi.MechHandle(t); ctrl = capi.CtrlHandle(mh, [0], K=K, N=T, zd=zd)
lo, hi = pkg.dist.shard_bounds(n_local * world, rank, world)
z0 = torch.from_numpy(z0_all[lo:hi]).to(dev)
zT = torch.empty_like(z0); st = torch.zeros(n_local, dtype=torch.int32, device=dev)
lam = torch.zeros((n_local, 5 * nb), dtype=torch.float64, device=dev)
tg = pkg.dist.TrajectoryGather(rank, world, n_local, T, nb, H, dev)
Tc = T // H
for c in range(H):
    tg.wait_slab_free(c)
    capi.rollout_dev(mh, ctrl, n_local, Tc, c * Tc + 1, (z0 if c == 0 else zT).data_ptr(), lam.data_ptr(), 0, 0, tg.slab(c).data_ptr(), zT.data_ptr(), st.data_ptr(),
                     torch.cuda.current_stream().cuda_stream)
    tg.submit(c)
traj = tg.finish()
zall = pkg.dist.gather_to_root(zT, n_local * world, rank, world)
torch.cuda.synchronize()
if rank == 0:
    zT1, traj1, st1 = capi.rollout(mh, ctrl, z0_all, T, record=True)          # the whole batch, one launch, one process
    assert (st1 > 0).all()
    assert np.array_equal(traj.cpu().numpy(), traj1), "chunked + collected trajectory differs from the single launch"
    assert np.array_equal(zall.cpu().numpy(), zT1)
    print("SHARDED_OK", tg.bytes_gathered)
"""


def test_two_ranks_chunked_rollout_and_trajectory_collection(tmp_path):
    """SURVEY 8e on the one GPU of the box: two processes roll out their instance shards in 4 launches of 16 steps (k0 continuation, state and
    multipliers carried), every chunk's trajectory slab travels to rank 0 on a second stream while the next chunk is computed
    (dist.TrajectoryGather; gloo here, RCCL on a multi-GPU node), and what rank 0 holds is bit for bit the trajectory of ONE launch of
    the whole batch in one process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_SHARDED_WORKER % {"root": root})
    from conftest import free_port
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "SHARDED_OK" in r.stdout


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO torchrun environment starts its two ranks itself (fresh child processes; the
    parent never touches the GPU) and the one JSON line says n_gpus = 2.  On the one-GPU box the ranks share cuda:0 and collect over gloo
    (--rehearse-shared-gpu; RCCL refuses two ranks on one device)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--instances", "256", "--sim-steps", "64",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [x for x in r.stdout.splitlines() if x.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rehearsal_shared_gpu"] and d["collective_backend"] == "gloo"
    assert d["config"]["launches_per_rollout"] == 8 and d["newton"]["failed_instances"] == 0
    assert abs(d["value"] - 2 * 256 * 64 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["collection"]["trajectory_bytes_gathered_to_rank0_per_rollout"] == 256 * 64 * 17 * 13 * 8
    assert d["collection"]["mode"] == "trajectory"
    # the fallback collection mode -- only final states travel, every rank keeps its own trajectory, one launch per rollout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--collect", "final", "--instances", "256",
                        "--sim-steps", "64", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    f = json.loads([x for x in r.stdout.splitlines() if x.strip().startswith("{")][0])
    assert f["n_gpus"] == 2 and f["collection"]["mode"] == "final" and f["collection"]["trajectory_bytes_gathered_to_rank0_per_rollout"] == 0
    assert f["config"]["launches_per_rollout"] == 1 and f["config"]["record"] and f["newton"]["failed_instances"] == 0
    assert "allocation plan (--collect final" in r.stderr
    # item 4b: a run whose collection buffers cannot fit the free HBM is refused on every rank with a sentence and exit code 3, before any large
    # allocation (2 x 65536 instances x 4000 steps x 17 bodies: 927 GB of assembled trajectories on rank 0)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--instances", "65536", "--sim-steps", "4000",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode != 0 and "refusing to start" in r.stderr and "--collect final" in r.stderr, r.stderr[-3000:]
    assert not [x for x in r.stdout.splitlines() if x.strip().startswith("{")]


_RCCL_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
import torch.distributed as dist
pkg = g.load_package(); capi = pkg._capi
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r)
rank, world, local = pkg.dist.init_from_env(force_init=True)          # backend None -> "nccl" = RCCL on a GPU box; no fallback
assert dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", 0); capi.set_device(0)
n_links, n, T, H = 3, 64, 32, 4
ex = pkg.examples.cartpole_n(n_links); t = ex["mech"].tables(); nb = t.nb
zd = pkg.examples.cartpole_states(n_links, [0.0], np.array([[np.pi] + [0.0] * (n_links - 1)]))[0]
rng = np.random.default_rng(9)
K = rng.normal(size=(T - 1, 1, 12 * nb)) * 0.05
phi = rng.uniform(-0.2, 0.2, (n, n_links)); phi[:, 0] += np.pi
z0h = pkg.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, n), phi)
mh = capi.MechHandle(t); ctrl = capi.CtrlHandle(mh, [0], K=K, N=T, zd=zd)
z0 = torch.from_numpy(z0h).to(dev); zT = torch.empty_like(z0); st = torch.zeros(n, dtype=torch.int32, device=dev)
lam = torch.zeros((n, 5 * nb), dtype=torch.float64, device=dev)
# the single rank really calls the collective: librccl is loaded, its communicator exists, the gather runs on the second stream
tg = pkg.dist.TrajectoryGather(rank, world, n, T, nb, H, dev, force_collective=True)
fin = pkg.dist.RootGather(n, (nb, 13), torch.float64, dev, rank, world, force_collective=True)
Tc = T // H
for c in range(H):
    tg.wait_slab_free(c)
    capi.rollout_dev(mh, ctrl, n, Tc, c * Tc + 1, (z0 if c == 0 else zT).data_ptr(), lam.data_ptr(), 0, 0, tg.slab(c).data_ptr(), zT.data_ptr(), st.data_ptr(),
                     torch.cuda.current_stream().cuda_stream)
    tg.submit(c)
traj = tg.finish()
zall = fin(zT)
torch.cuda.synchronize()
zT1, traj1, st1 = capi.rollout(mh, ctrl, z0h, T, record=True)
assert (st1 > 0).all()
assert np.array_equal(traj.cpu().numpy(), traj1) and np.array_equal(zall.cpu().numpy(), zT1)
libs = [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l]
assert libs, "librccl is not mapped into the process"
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", sorted(set(libs))[0])
"""


def test_rccl_executes_once_with_one_rank(tmp_path):
    """the `nccl` backend (= RCCL) is initialised with world_size 1 and ONE chunked rollout goes through
    dist.TrajectoryGather / dist.RootGather with the collective forced, so that librccl's load, its communicator, the stream ordering of
    the gather behind the rollout launch and HSA_ENABLE_IPC_MODE_LEGACY are exercised once on the GPU box (a hardware scaling curve is
    still unmeasured: one GPU).  The collected trajectory equals one launch of the batch bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from conftest import free_port
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER % {"root": root, "port": str(free_port())})
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL_OK" in r.stdout


def test_graph_captured_steps_with_friction_and_philox_noise(cclqr, orc):
    """configs[4] as written -- the friction + noise law of trackingLQR_triple_cartpole.jl:93-111 AND a hipGraph-captured step: 20
    single-step launches with device-generated Philox noise captured into a graph == one fused 20-step launch (the samples are keyed by
    (global instance, step), so the split does not change them), and == the oracle.  Round 5: a launch of up to CCLQR_PHILOX_INKERNEL_STEPS
    steps generates its samples INSIDE the rollout kernel (one graph node per step, no workspace: a fresh controller handle is captured as
    it is), every captured launch carries CCLQR_ROLLOUT_NO_ALLOC, and a launch that WOULD have to grow the handle's workspace is refused before
    anything synchronises or allocates -- on the capturing stream without the flag, on ANY stream with it.  The same
    steps captured as TWO independent chains of half the batch each (parallel branches of one graph: what bench.py's pipelined form does)
    give the same bits again."""
    import torch
    capi = cclqr._capi
    ex = cclqr.examples.triple_cartpole()
    t = ex["mech"].tables()
    N, n = 40, 192
    rng = np.random.default_rng(12)
    z00 = ex["mech"].state()
    zd = np.tile(z00, (N, 1, 1))
    K = rng.normal(size=(N - 1, 1, 48)) * 0.3
    kw = dict(K=K, N=N, zd=zd, fric=ex["fric"], noise_scale=2.0, noise_seed=0xC0FFEE)
    z0 = np.tile(z00, (n, 1, 1))
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, [0], **kw)
    dev = torch.device("cuda", 0)
    z0_d = torch.from_numpy(z0).to(dev)
    ref = torch.empty_like(z0_d)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    capi.rollout_dev(mech, ctrl, n, 20, 1, z0_d.data_ptr(), 0, 0, 0, 0, ref.data_ptr(), st.data_ptr())       # 20 steps: samples from the workspace
    torch.cuda.synchronize()
    assert (st > 0).all() and float((ref[0] - ref[1]).abs().max()) > 1e-4        # the noise is there and differs per instance
    ctrl2 = capi.CtrlHandle(mech, [0], **kw)          # a fresh handle: no workspace, and none is needed for single-step launches
    NO_ALLOC = capi.ROLLOUT_NO_ALLOC
    # (outside any capture) the flag alone refuses a launch that would have to grow the workspace: 9 steps > CCLQR_PHILOX_INKERNEL_STEPS
    tmp = torch.empty_like(z0_d)
    with pytest.raises(capi.CclqrError) as err:
        capi.rollout_dev(mech, ctrl2, n, capi.PHILOX_INKERNEL_STEPS + 1, 1, z0_d.data_ptr(), 0, 0, 0, 0, tmp.data_ptr(), st.data_ptr(), flags=NO_ALLOC)
    assert err.value.code == capi.EINVAL and "cclqr_ctrl_reserve_noise" in str(err.value)
    za, zb = z0_d.clone(), torch.empty_like(z0_d)
    lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=dev)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph.capture_begin()
        src, dst = za, zb
        for k in range(1, 21):
            capi.rollout_dev(mech, ctrl2, n, 1, k, src.data_ptr(), lam.data_ptr(), 0, 0, 0, dst.data_ptr(), st.data_ptr(), side.cuda_stream, flags=NO_ALLOC)
            src, dst = dst, src
            if k == 10:
                # a launch that would have to GROW the handle's workspace while its stream is being captured comes back as CCLQR_EINVAL before
                # anything synchronises or allocates (the capture stays valid: the replays below are still bit-identical to the fused launch)
                with pytest.raises(capi.CclqrError) as err:
                    capi.rollout_dev(mech, ctrl2, n, 9, k + 1, src.data_ptr(), lam.data_ptr(), 0, 0, 0, dst.data_ptr(), st.data_ptr(), side.cuda_stream)
                assert err.value.code == capi.EINVAL
        graph.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(2):                                  # the second replay runs on fresh inputs through the same captured buffers
        za.copy_(z0_d)
        lam.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(src, ref)
    # two independent chains of half the batch each, forked and joined inside ONE capture (first_instance keys the second half's samples)
    h = n // 2
    nz, nl = t.nb * 13 * 8, 5 * t.ne * 8
    graph2 = torch.cuda.CUDAGraph()
    s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph2.capture_begin()
        for b in range(2):
            s2[b].wait_stream(side)
            srcb, dstb = za, zb
            for k in range(1, 21):
                capi.rollout_dev(mech, ctrl2, h, 1, k, srcb.data_ptr() + b * h * nz, lam.data_ptr() + b * h * nl, 0, 0, 0, dstb.data_ptr() + b * h * nz,
                                 st.data_ptr() + b * h * 4, s2[b].cuda_stream, first_instance=b * h, flags=NO_ALLOC)
                srcb, dstb = dstb, srcb
        for b in range(2):
            side.wait_stream(s2[b])
        graph2.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    za.copy_(z0_d)
    lam.zero_()
    graph2.replay()
    torch.cuda.synchronize()
    assert torch.equal(srcb, ref)
    # caller-owned workspace (two launches sharing one controller on different streams would each bring their own)
    ws = torch.empty(n * 20, dtype=torch.float64, device=dev)
    out = torch.empty_like(z0_d)
    capi.rollout_dev(mech, ctrl, n, 20, 1, z0_d.data_ptr(), 0, 0, 0, 0, out.data_ptr(), st.data_ptr(), noise_ws=ws.data_ptr(), noise_ws_len=n * 20)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # ... and a single step with the caller's workspace takes the fill-kernel path: the same sample bits as the in-kernel generation
    za.copy_(z0_d)
    capi.rollout_dev(mech, ctrl, n, 1, 1, za.data_ptr(), 0, 0, 0, 0, out.data_ptr(), st.data_ptr(), noise_ws=ws.data_ptr(), noise_ws_len=n)
    capi.rollout_dev(mech, ctrl, n, 1, 1, za.data_ptr(), 0, 0, 0, 0, tmp.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, tmp)
    with pytest.raises(capi.CclqrError):
        capi.rollout_dev(mech, ctrl, n, 20, 1, z0_d.data_ptr(), 0, 0, 0, 0, out.data_ptr(), st.data_ptr(), noise_ws=ws.data_ptr(), noise_ws_len=n * 20 - 1)
    with pytest.raises(capi.CclqrError):        # unknown flag bits are refused
        capi.rollout_dev(mech, ctrl, n, 1, 1, z0_d.data_ptr(), 0, 0, 0, 0, out.data_ptr(), st.data_ptr(), flags=8)
    zo, _, _ = orc.rollout(t, orc.ctrl_desc(t.nb, [0], **kw), z0[:16], 20)
    assert np.abs(ref[:16].cpu().numpy() - zo).max() < TOL


@pytest.mark.parametrize("n_links,ninst,steps,extra", [(1, 37, 120, False), (3, 21, 100, True), (7, 9, 80, False), (16, 5, 60, False), (22, 3, 40, False)])
def test_spread_and_packed_launches_agree_bitwise(cclqr, n_links, ninst, steps, extra):
    """A batch too small to give every SIMD a wavefront is spread over more wavefronts (rollout_chain.hip::chain_instances_per_wavefront: the lane
    groups without an instance work on their neighbours' line searches); CCLQR_ROLLOUT_PACK_WAVEFRONTS packs 64 / lanes-per-instance instances
    into every wavefront as a device-filling batch is.  Same arithmetic in the same order: trajectories, final states and Newton counts are
    bitwise equal -- so every small parity test of this suite, which runs spread, also stands for the packed layout of the full-size ones.
    Both a persistent launch and single-step launches (whose spreading target is a quarter of the device)."""
    import torch
    capi = cclqr._capi
    rng = np.random.default_rng(40 + n_links)
    if extra:
        ex = cclqr.examples.triple_cartpole()
        z0 = np.tile(ex["mech"].state(), (ninst, 1, 1))
    else:
        ex = cclqr.examples.cartpole_n(n_links)
        phi = rng.uniform(-0.3, 0.3, (ninst, n_links))
        phi[:, 0] += np.pi
        z0 = cclqr.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, ninst), phi)
    t = ex["mech"].tables()
    mech = capi.MechHandle(t)
    kw = dict(K=rng.normal(size=(steps + 5, 1, 12 * t.nb)) * 0.05, N=steps + 6, zd=z0[0], Fd=np.array([[0.3]]))
    if extra:
        kw.update(fric=ex["fric"], noise_scale=0.5, noise_seed=77)
    ctrl = capi.CtrlHandle(mech, [0], **kw)
    full = 64 // mech.geometry()[0]
    assert mech.instances_per_wavefront(ninst, steps) == 1 and mech.instances_per_wavefront(ninst, steps, capi.ROLLOUT_PACK_WAVEFRONTS) == full
    assert mech.instances_per_wavefront(4096, 1000) == min(4, full) and mech.instances_per_wavefront(4096, 1) == full       # (1024 SIMDs; short launches: a quarter)
    a = capi.rollout(mech, ctrl, z0, steps, record=True)
    b = capi.rollout(mech, ctrl, z0, steps, record=True, flags=capi.ROLLOUT_PACK_WAVEFRONTS)
    assert (a[2] > 0).all()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    dev = torch.device("cuda", 0)
    outs = []
    for flags in (0, capi.ROLLOUT_PACK_WAVEFRONTS):
        z = torch.from_numpy(z0).to(dev)
        zn = torch.empty_like(z)
        lam = torch.zeros((ninst, 5 * t.ne), dtype=torch.float64, device=dev)
        s = torch.zeros(ninst, dtype=torch.int32, device=dev)
        for k in range(1, 11):
            capi.rollout_dev(mech, ctrl, ninst, 1, k, z.data_ptr(), lam.data_ptr(), 0, 0, 0, zn.data_ptr(), s.data_ptr(), 0, flags=flags)
            z, zn = zn, z
        torch.cuda.synchronize()
        outs.append((z.cpu().numpy(), lam.cpu().numpy(), s.cpu().numpy()))
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
    assert np.array_equal(outs[0][0], a[1][:, 10])         # ten single-step launches = the first ten steps of the persistent one (traj[:, k] = the state after k steps)


def test_lanes_per_link_of_the_chain_instantiations(cclqr):
    """cclqr_rollout_lanes_per_link: mechanisms of 1-2 links (2 of their 8 lanes own a link) run three lanes per link (rollout_chain_kernel<8, 4, law, relax, 3, 2>,
    round 5: the rows of an evaluation with Jacobians dealt to a link's lanes at unchanged occupancy); everything else one lane per link"""
    capi = cclqr._capi
    for n_links, want in ((1, (3, 2)), (3, (1, 8)), (7, (1, 16)), (15, (1, 32)), (16, (1, 32)), (40, (1, 64))):
        mech = capi.MechHandle(cclqr.examples.cartpole_n(n_links)["mech"].tables())
        assert mech.lanes_per_link() == want, (n_links, mech.lanes_per_link())
        mech.close()
    pend = capi.MechHandle(cclqr.examples.pendulum()["mech"].tables())
    assert pend.lanes_per_link() == (3, 2)
    pend.close()


def test_newton_mode_residual_only_is_a_measured_error_option(cclqr, orc):
    """cclqr_rollout_opts.newton_mode = 1 (stop on ||f|| < eps alone): NOT the parity mode.  The default stays the exact rule (= the
    oracle, checked here once more); the option's deviation from it is MEASURED and printed (2e-8 over these 200 steps under random
    gains -- already past the north star's 1e-8, which is why it is an option; the headline workload's figure is in DESIGN.md 4.1d);
    its Newton iteration counts are never larger.  Refused under the friction / noise / PID laws."""
    capi = cclqr._capi
    n_links, steps, n = 7, 200, 96
    ex = cclqr.examples.cartpole_n(n_links)
    t = ex["mech"].tables()
    zd = hanging_setpoint(cclqr, n_links)
    rng = np.random.default_rng(31)
    K = rng.normal(size=(steps + 19, 1, 12 * t.nb)) * 0.05
    phi = rng.uniform(-0.2, 0.2, (n, n_links))
    phi[:, 0] += np.pi
    z0 = cclqr.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, n), phi)
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, [0], K=K, N=steps + 20, zd=zd)
    zT0, tr0, st0 = capi.rollout(mech, ctrl, z0, steps, record=True)
    zT1, tr1, st1 = capi.rollout(mech, ctrl, z0, steps, record=True, newton_mode=1)
    _, tro, sto = orc.rollout(t, orc.ctrl_desc(t.nb, [0], K=K, N=steps + 20, zd=zd), z0[:8], steps, record=True)
    assert np.abs(tr0[:8] - tro).max() < TOL and np.array_equal(st0[:8], sto)
    dev = np.abs(tr1 - tr0).max()
    print("newton_mode 1: max |state - exact rule| over %d steps = %.3g; max Newton iterations %d -> %d" % (steps, dev, st0.max(), st1.max()))
    assert (st1 > 0).all() and 0.0 < dev < 1e-6 and (st1 <= st0).all() and st1.max() < st0.max()
    # branching trees take the option since round 4 (tests/test_gpu_treereg.py); the friction / noise / PID laws keep the exact rule only
    cf = capi.CtrlHandle(mech, [0], K=K, N=steps + 20, zd=zd, fric=np.full(t.ne, 0.01))
    with pytest.raises(capi.CclqrError) as e:
        capi.rollout(mech, cf, z0[:2], 3, newton_mode=1)
    assert e.value.code == capi.EUNSUPPORTED


def _mechanism_for(cclqr, kind):
    """(tables, initial state, controlled joints) of a chain, a branching tree or the closed-loop deltabot"""
    if kind == "chain":
        ex = cclqr.examples.cartpole_n(2)
        return ex["mech"].tables(), cclqr.examples.cartpole_states(2, [0.1], [[0.2, -0.1]])[0], [0]
    if kind == "tree":
        ex = cclqr.examples.dual_cartpole()
        return ex["mech"].tables(), ex["mech"].state(), [0]
    ex = cclqr.examples.deltabot()
    return ex["mech"].tables(), ex["mech"].state(), [ex["mech"].joint_index(e) for e in ex["eqcids"]]


@pytest.mark.parametrize("kind", ["chain", "tree", "loop"])
def test_carried_status_keeps_a_lost_instance_frozen_across_launches(cclqr, kind):
    """CCLQR_ROLLOUT_CARRY_STATUS: `status` is read and written, so a batch stepped one launch at a time behaves like ONE launch over the horizon -- an instance
    that a non-finite input loses at step 5 stays frozen at its pose of knot 5, at rest, through every later launch and keeps its (negative) status; the
    others carry their largest Newton count along.  Final states, multipliers and statuses of twelve chained single-step launches equal the persistent
    launch's bit for bit; without the flag every launch steps the lost instance again (here it is lost again at once: its multipliers are NaN) and the status
    array only speaks for the last step."""
    import torch
    capi = cclqr._capi
    t, z00, cj = _mechanism_for(cclqr, kind)
    n, T = 6, 12
    rng = np.random.default_rng(8)
    Fd = rng.normal(size=(n, T, len(cj))) * 0.2
    Fd[3, 4] = np.nan                                            # instance 3, step 5
    zd = np.tile(z00, (n, T, 1, 1))
    mech = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mech, cj, K=None, N=0, zd=zd, Fd=Fd, n_ctrl=n)
    z0 = np.tile(z00, (n, 1, 1))
    zT, traj, st = capi.rollout(mech, ctrl, z0, T, record=True)
    assert st[3] < 0 and abs(st[3]) < capi.NEWTON_MAXIT and (np.delete(st, 3) > 0).all()
    assert np.array_equal(zT[3, :, 0:7], traj[3, 4, :, 0:7]) and not zT[3, :, 7:].any()
    dev = torch.device("cuda", 0)
    out = {}
    for flags in (capi.ROLLOUT_CARRY_STATUS, 0):
        z = torch.from_numpy(z0).to(dev)
        zn = torch.empty_like(z)
        lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=dev)
        s = torch.zeros(n, dtype=torch.int32, device=dev)
        for k in range(1, T + 1):
            capi.rollout_dev(mech, ctrl, n, 1, k, z.data_ptr(), lam.data_ptr(), 0, 0, 0, zn.data_ptr(), s.data_ptr(), 0, flags=flags)
            z, zn = zn, z
        torch.cuda.synchronize()
        out[flags] = (z.cpu().numpy(), s.cpu().numpy())
    zc, sc = out[capi.ROLLOUT_CARRY_STATUS]
    assert np.array_equal(zc, zT) and np.array_equal(sc, st)
    z_plain, s_plain = out[0]
    others = [i for i in range(n) if i != 3]
    assert np.array_equal(z_plain[others], zT[others])
    assert (s_plain[others] <= sc[others]).all()                             # (without the flag the array only speaks for the last launch's step)
    with pytest.raises(capi.CclqrError):                                     # the flag needs the status array
        capi.rollout_dev(mech, ctrl, n, 1, 1, z.data_ptr(), lam.data_ptr(), 0, 0, 0, zn.data_ptr(), 0, 0, flags=capi.ROLLOUT_CARRY_STATUS)
