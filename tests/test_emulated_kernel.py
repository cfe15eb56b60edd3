"""CPU tests of the HIP kernels' arithmetic: the __host__ __device__ phase functions of csrc/cclqr_dev.h / cclqr_lin_dev.h
are run serially by tests/emu (same phase order as the kernels) and compared with the oracle.  This checks the LDS
layout, indexing, the Schur/block-tridiagonal solve and the linearisation without a GPU; the GPU tests then check the
real kernels through the C-ABI."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sl

from conftest import hanging_setpoint, long_and_short_chain_forest, upright_setpoint

dp = C.POINTER(C.c_double)


def emu_rollout(emu, orc, t, ctrl, z0, steps, noise=None, G=0):
    m = orc.mech_desc(t)
    z0 = np.ascontiguousarray(z0, dtype=np.float64).reshape(-1, t.nb, 13)
    n = z0.shape[0]
    traj = np.zeros((n, steps, t.nb, 13))
    zT = np.zeros_like(z0)
    st = np.zeros(n, dtype=np.int32)
    args = (C.byref(m.desc), C.byref(ctrl.desc), C.c_int64(n), C.c_int(steps), C.c_int(1), z0.ctypes.data_as(dp),
            None if noise is None else noise.ctypes.data_as(dp), traj.ctypes.data_as(dp), zT.ctypes.data_as(dp),
            st.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(G))
    # forests of chains run the register-resident chain kernel (csrc/rollout_chain.hip), branched mechanisms the tree kernel
    rc = emu.emu_chain_rollout(*args)
    if rc == -5:
        rc = emu.emu_rollout(*args)
    assert rc == 0
    return zT, traj, st


@pytest.mark.parametrize("n_links,steps,hanging,G", [(1, 150, False, 0), (1, 60, False, 64), (3, 100, False, 0), (7, 60, True, 0), (16, 25, True, 0),
                                                   (11, 20, True, 0), (12, 20, True, 0), (15, 20, True, 0)])   # 12/13/16 links: the reduction level with an even / odd count
def test_emulated_rollout_matches_oracle(cclqr, orc, emu, n_links, steps, hanging, G):
    ex = cclqr.examples.cartpole_n(n_links)
    t = ex["mech"].tables()
    zd = hanging_setpoint(cclqr, n_links) if hanging else upright_setpoint(n_links)
    rng = np.random.default_rng(4)
    if n_links <= 3:
        A, Bu, Bl, G_ = orc.linearize(t, zd, [0], np.zeros(1))
        K, _ = orc.riccati(A, Bu, Bl, G_, sl.block_diag(*ex["Q"]) * t.dt, sl.block_diag(*ex["R"]) * t.dt, steps + 20)
    else:
        K = rng.normal(size=(steps + 19, 1, 12 * t.nb)) * 0.05
    phi = rng.uniform(-1, 1, (2, n_links)) * (0.3 if hanging else 0.3 / 3 ** n_links)
    if hanging:
        phi[:, 0] += np.pi
    z0 = cclqr.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, 2), phi)
    octrl = orc.ctrl_desc(t.nb, [0], K=K, N=steps + 20, zd=zd)
    zT_o, traj_o, st_o = orc.rollout(t, octrl, z0, steps, record=True)
    zT, traj, st = emu_rollout(emu, orc, t, octrl, z0, steps, G=G)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-10
    assert np.abs(zT - zT_o).max() < 1e-10


@pytest.mark.parametrize("n_links,kl,hanging", [(1, 3, False), (1, 2, False), (3, 2, False), (7, 2, True), (15, 2, True), (16, 3, True)])
def test_emulated_rollout_with_several_lanes_per_link(cclqr, orc, emu, n_links, kl, hanging):
    """round 5: mechanisms that leave lanes of their lane group idle give a link KL = 2 or 3 lanes and deal the rows of the joint evaluation and of the
    Schur complement to them (cclqr_chain.h joint_eval_rows / ck_schur_rows_sub).  The emulator runs every evaluation with Jacobians through those forms --
    all sub-lanes of a link one after the other, into an LDS image that starts as signalling NaNs, so an entry that no sub-lane writes poisons the
    solve -- and the rollout must equal the oracle's as the one-lane form does (prismatic cart + revolute links: both row layouts; with a parent and without)."""
    ex = cclqr.examples.cartpole_n(n_links)
    t = ex["mech"].tables()
    zd = hanging_setpoint(cclqr, n_links) if hanging else upright_setpoint(n_links)
    rng = np.random.default_rng(14)
    steps = 40 if n_links < 8 else 15
    K = rng.normal(size=(steps + 19, 1, 12 * t.nb)) * 0.05
    phi = rng.uniform(-1, 1, (2, n_links)) * (0.3 if hanging else 0.3 / 3 ** n_links)
    if hanging:
        phi[:, 0] += np.pi
    z0 = cclqr.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, 2), phi)
    octrl = orc.ctrl_desc(t.nb, [0], K=K, N=steps + 20, zd=zd)
    zT_o, traj_o, st_o = orc.rollout(t, octrl, z0, steps, record=True)
    emu.emu_chain_set_lanes_per_link(C.c_int(kl))
    try:
        zT, traj, st = emu_rollout(emu, orc, t, octrl, z0, steps)
    finally:
        emu.emu_chain_set_lanes_per_link(C.c_int(1))
    # (Newton iteration counts: at the residual's round-off floor one decision of the stopping rule may move with the rows' order of summation)
    assert (st_o > 0).all() and (st > 0).all() and np.abs(st - st_o).max() <= 1
    assert np.abs(traj - traj_o).max() < 1e-10 and np.abs(zT - zT_o).max() < 1e-10


def test_emulated_tracking_friction_noise(cclqr, orc, emu):
    ex = cclqr.examples.triple_cartpole()
    t = ex["mech"].tables()
    N, ninst = 40, 3
    rng = np.random.default_rng(11)
    z00 = ex["mech"].state()
    zd = np.tile(z00, (N, 1, 1))
    zd[:, 0, 1] = 0.01 * np.arange(N)
    K = rng.normal(size=(N - 1, 1, 48)) * 0.5
    noise = rng.normal(size=(ninst, N))
    c = orc.ctrl_desc(4, [0], K=K, N=N, zd=zd, Fd=rng.normal(size=(N, 1)), fric=ex["fric"], noise_scale=2.0, noise=noise)
    z0 = np.tile(z00, (ninst, 1, 1))
    _, traj_o, _ = orc.rollout(t, c, z0, N, record=True)
    _, traj, _ = emu_rollout(emu, orc, t, c, z0, N, noise=noise)
    assert np.abs(traj - traj_o).max() < 1e-10


def test_emulated_philox_noise(cclqr, orc, emu):
    """counter-based noise (noise_philox): the kernels' Philox/Box-Muller (csrc/cclqr_dev.h, run on the host here) against the oracle's"""
    ex = cclqr.examples.triple_cartpole()
    t = ex["mech"].tables()
    N, ninst = 30, 4
    z00 = ex["mech"].state()
    zd = np.tile(z00, (N, 1, 1))
    K = np.random.default_rng(2).normal(size=(N - 1, 1, 48)) * 0.3
    c = orc.ctrl_desc(4, [0], K=K, N=N, zd=zd, fric=ex["fric"], noise_scale=2.0, noise_seed=0xC0FFEE)
    z0 = np.tile(z00, (ninst, 1, 1))
    _, traj_o, _ = orc.rollout(t, c, z0, N, record=True)
    _, traj, _ = emu_rollout(emu, orc, t, c, z0, N)
    assert np.abs(traj_o[0] - traj_o[1]).max() > 1e-3       # instances got different streams
    assert np.abs(traj - traj_o).max() < 1e-10
    # and the stream is what an injected array of orc.philox_normal samples gives
    noise = np.array([[orc.philox_normal(0xC0FFEE, n, k) for k in range(1, N + 1)] for n in range(ninst)])
    c2 = orc.ctrl_desc(4, [0], K=K, N=N, zd=zd, fric=ex["fric"], noise_scale=2.0, noise=noise)
    _, traj_2, _ = orc.rollout(t, c2, z0, N, record=True)
    assert np.array_equal(traj_2, traj_o)


def test_body_order_permutation(cclqr, orc, emu):
    """bodies listed leaf-first: the kernels' breadth-first link order differs from the caller's body/joint numbering"""
    ex = cclqr.examples.cartpole_n(2)
    t = ex["mech"].tables()
    perm = [2, 0, 1]          # new body i = old body perm[i]
    inv = np.argsort(perm)
    jperm = [1, 2, 0]         # new joint j = old joint jperm[j]
    t2 = cclqr.MechTables(3, 3, t.dt, t.g, t.mass[perm], t.inertia[perm], [(-1 if t.parent[j] < 0 else inv[t.parent[j]]) for j in jperm],
                          [inv[t.child[j]] for j in jperm], t.type[jperm], t.p1[jperm], t.p2[jperm], t.axis[jperm], t.qoff[jperm])
    rng = np.random.default_rng(0)
    K = rng.normal(size=(30, 1, 36)) * 0.3
    zd = upright_setpoint(2)
    z0 = cclqr.examples.cartpole_states(2, [0.2], [[0.05, -0.03]])
    K2 = K.reshape(30, 1, 3, 12)[:, :, perm].reshape(30, 1, 36)
    c1 = orc.ctrl_desc(3, [0], K=K, N=31, zd=zd)
    c2 = orc.ctrl_desc(3, [jperm.index(0)], K=K2, N=31, zd=zd[perm])
    _, tr1, _ = orc.rollout(t, c1, z0, 30, record=True)
    _, tr2o, _ = orc.rollout(t2, c2, z0[:, perm], 30, record=True)
    assert np.abs(tr2o[:, :, inv] - tr1).max() < 1e-11
    _, tr2, _ = emu_rollout(emu, orc, t2, c2, z0[:, perm], 30)
    assert np.abs(tr2 - tr2o).max() < 1e-10


def test_emulated_linearize_matches_oracle(cclqr, orc, emu):
    ex = cclqr.examples.triple_cartpole()
    t = ex["mech"].tables()
    z = ex["mech"].state()
    lam = np.zeros(20)
    for _ in range(20):
        z, lam, it = orc.step(t, z, lam, np.array([5.0, 0.2, -0.1, 0.05]))
    cj = np.array([0, 2], dtype=np.int32)
    Fd = np.array([1.7, -0.4])
    m = orc.mech_desc(t)
    mx, ml = 48, 20
    A, Bu, Bl, G = np.zeros((mx, mx)), np.zeros((mx, 2)), np.zeros((mx, ml)), np.zeros((ml, mx))
    rc = emu.emu_linearize(C.byref(m.desc), np.ascontiguousarray(z).ctypes.data_as(dp), C.c_int(2), cj.ctypes.data_as(C.POINTER(C.c_int32)),
                           Fd.ctypes.data_as(dp), A.ctypes.data_as(dp), Bu.ctypes.data_as(dp), Bl.ctypes.data_as(dp), G.ctypes.data_as(dp))
    assert rc == 0
    for got, want in zip((A, Bu, Bl, G), orc.linearize(t, z, cj, Fd)):
        assert np.abs(got - want).max() < 1e-10 * max(1.0, np.abs(want).max())


def test_emulated_sawyer_rollout_and_linearize(cclqr, orc, emu):
    """config 4 mechanism: 7 revolute joints with rpy offsets, full inertia tensors, 7 controlled joints (mu = 7)"""
    import json
    import os
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sawyer_arm_tables.json")))
    ex = cclqr.examples.sawyer(tab, g=-9.81)
    mech = ex["mech"]
    t = mech.tables()
    rng = np.random.default_rng(3)
    zd = mech.state()
    K = rng.normal(size=(49, 7, 84)) * 0.004
    z0 = []
    for n in range(2):
        for e in mech.eqconstraints:
            cclqr.setJointPosition(mech, e, rng.uniform(-0.3, 0.3))
        z0.append(mech.state())
    z0 = np.stack(z0)
    c = orc.ctrl_desc(7, list(range(7)), K=K, N=50, zd=zd, Fd=0.02 * rng.normal(size=(1, 7)))
    zT_o, traj_o, st_o = orc.rollout(t, c, z0, 40, record=True)
    zT, traj, st = emu_rollout(emu, orc, t, c, z0, 40)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-10
    cj = np.arange(7, dtype=np.int32)
    Fd = rng.normal(size=7)
    m = orc.mech_desc(t)
    A, Bu, Bl, G = np.zeros((84, 84)), np.zeros((84, 7)), np.zeros((84, 35)), np.zeros((35, 84))
    rc = emu.emu_linearize(C.byref(m.desc), np.ascontiguousarray(z0[1]).ctypes.data_as(dp), C.c_int(7), cj.ctypes.data_as(C.POINTER(C.c_int32)),
                           Fd.ctypes.data_as(dp), A.ctypes.data_as(dp), Bu.ctypes.data_as(dp), Bl.ctypes.data_as(dp), G.ctypes.data_as(dp))
    assert rc == 0
    for got, want in zip((A, Bu, Bl, G), orc.linearize(t, z0[1], cj, Fd)):
        assert np.abs(got - want).max() < 1e-10 * max(1.0, np.abs(want).max())


def test_emulated_two_chains(cclqr, orc, emu):
    """forest of chains: two independent cartpoles hanging off the same origin, bodies interleaved in the caller's numbering"""
    e1 = cclqr.examples.cartpole_n(2)
    t1 = e1["mech"].tables()
    nb = 6
    # caller's numbering: bodies 0,2,4 = chain A (cart, pole, pole), 1,3,5 = chain B
    ia, ib = [0, 2, 4], [1, 3, 5]
    mass, inertia = np.zeros(nb), np.zeros((nb, 9))
    parent, child, typ = np.zeros(nb, dtype=np.int32), np.zeros(nb, dtype=np.int32), np.zeros(nb, dtype=np.int32)
    p1, p2, axis, qoff = np.zeros((nb, 3)), np.zeros((nb, 3)), np.zeros((nb, 3)), np.zeros((nb, 4))
    for ids in (ia, ib):
        for k in range(3):
            mass[ids[k]], inertia[ids[k]] = t1.mass[k], t1.inertia[k]
            j = ids[k]                      # joint j hangs body ids[k]
            parent[j] = -1 if t1.parent[k] < 0 else ids[t1.parent[k]]
            child[j], typ[j], p1[j], p2[j], axis[j], qoff[j] = ids[k], t1.type[k], t1.p1[k], t1.p2[k], t1.axis[k], t1.qoff[k]
    t2 = cclqr.MechTables(nb, nb, t1.dt, t1.g, mass, inertia, parent, child, typ, p1, p2, axis, qoff)
    rng = np.random.default_rng(9)
    za = cclqr.examples.cartpole_states(2, [0.2], [[0.1, -0.2]])[0]
    zb = cclqr.examples.cartpole_states(2, [-0.3], [[-0.15, 0.05]])[0]
    z0 = np.zeros((1, nb, 13))
    z0[0, ia], z0[0, ib] = za, zb
    zd = np.zeros((nb, 13))
    zd[:, 3] = 1
    K = rng.normal(size=(30, 2, 12 * nb)) * 0.2
    c = orc.ctrl_desc(nb, [0, 1], K=K, N=31, zd=zd)
    _, traj_o, st_o = orc.rollout(t2, c, z0, 30, record=True)
    _, traj, st = emu_rollout(emu, orc, t2, c, z0, 30)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-10


@pytest.mark.parametrize("short_first", [False, True])
def test_emulated_forest_long_and_short_chain(cclqr, orc, emu, short_first):
    """a 13-link chain (takes the odd-even reduction level: 32 lanes, 16-link layout) next to a 3-link chain (plain two-front sweep) in
    one mechanism, bodies interleaved in the caller's numbering: the level's lane map must leave the second chain alone"""
    t2, z0, zd, K, cj = long_and_short_chain_forest(cclqr, short_first)
    c = orc.ctrl_desc(t2.nb, cj, K=K, N=21, zd=zd)
    _, traj_o, st_o = orc.rollout(t2, c, z0, 20, record=True)
    _, traj, st = emu_rollout(emu, orc, t2, c, z0, 20)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-9


def test_emulated_pid(cclqr, orc, emu):
    """control_pid! (pid.jl:69-88) as a device law: cart + double pendulum, all three joints under PID"""
    ex = cclqr.examples.cartpole_n(2)
    t = ex["mech"].tables()
    z0 = cclqr.examples.cartpole_states(2, [0.1, -0.2], [[0.3, -0.2], [3.0, 0.4]])
    pid = dict(joint=[0, 1, 2], P=[20.0, 30.0, 15.0], I=[5.0, 10.0, 2.0], D=[4.0, 6.0, 1.5], goal=[0.0, 0.4, -3.0])
    c = orc.ctrl_desc(3, [], K=None, N=0, pid=pid)
    _, traj_o, st_o = orc.rollout(t, c, z0, 120, record=True)
    _, traj, st = emu_rollout(emu, orc, t, c, z0, 120)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-10


def test_emulated_acrobot_and_pid_double_pendulum(cclqr, orc, emu):
    """examples/lqr_acrobot.jl (second joint actuated only) and examples/pid_doublependulum.jl (two PID joints) through the kernels' phases"""
    ex = cclqr.examples.acrobot()
    t = ex["mech"].tables()
    zd = np.zeros((2, 13))
    zd[:, 0:3], zd[:, 3:7] = np.array(ex["xd"]), np.array(ex["qd"])
    A, Bu, Bl, G = orc.linearize(t, zd, [1], np.zeros(1))
    K, _ = orc.riccati(A, Bu, Bl, G, sl.block_diag(*ex["Q"]) * t.dt, sl.block_diag(*ex["R"]) * t.dt, 300)
    oc = orc.ctrl_desc(2, [1], K=K, N=300, zd=zd)
    z0 = ex["mech"].state()[None]
    zo, traj_o, _ = orc.rollout(t, oc, z0, 250, record=True)
    _, traj, st = emu_rollout(emu, orc, t, oc, z0, 250)
    assert (st > 0).all() and np.abs(traj - traj_o).max() < 1e-9

    ex = cclqr.examples.double_pendulum()
    t = ex["mech"].tables()
    z0 = np.stack([cclqr.examples.double_pendulum(a, b)["mech"].state() for a, b in ((0.0, 0.0), (0.4, -0.3))])
    oc = orc.ctrl_desc(2, [], K=None, N=0, pid=dict(joint=[0, 1], P=ex["P"], I=ex["I"], D=ex["D"], goal=ex["goals"]))
    _, traj_o, _ = orc.rollout(t, oc, z0, 300, record=True)
    _, traj, st = emu_rollout(emu, orc, t, oc, z0, 300)
    assert (st > 0).all() and np.abs(traj - traj_o).max() < 1e-9


def test_emulated_maximum_size_32_bodies(cclqr, orc, emu):
    """the largest mechanism of a 32-lane group (cart + 31 links: two instances per wavefront, the 32-link image)"""
    ex = cclqr.examples.cartpole_n(31)
    t = ex["mech"].tables()
    zd = hanging_setpoint(cclqr, 31)
    rng = np.random.default_rng(8)
    K = rng.normal(size=(1, 1, 12 * 32)) * 0.02
    phi = rng.uniform(-1, 1, (1, 31)) * 0.15
    phi[:, 0] += np.pi
    z0 = cclqr.examples.cartpole_states(31, [0.2], phi)
    oc = orc.ctrl_desc(32, [0], K=K, N=0, zd=zd)
    _, traj_o, st_o = orc.rollout(t, oc, z0, 12, record=True)
    _, traj, st = emu_rollout(emu, orc, t, oc, z0, 12)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-10


@pytest.mark.parametrize("n_links", [39, 63])
def test_emulated_chains_beyond_32_links(cclqr, orc, emu, n_links):
    """chains of 33 .. 64 bodies (CCLQR_MAXL = 64 since round 4): one instance per wavefront, the 64-link image, the plain two-front sweep
    (no reduction level beyond 17 links) -- a 40-body and a 64-body hanging chain under a feedback law, against the oracle"""
    nb = n_links + 1
    ex = cclqr.examples.cartpole_n(n_links)
    t = ex["mech"].tables()
    zd = hanging_setpoint(cclqr, n_links)
    rng = np.random.default_rng(8)
    K = rng.normal(size=(1, 1, 12 * nb)) * 0.01
    phi = rng.uniform(-1, 1, (1, n_links)) * 0.1
    phi[:, 0] += np.pi
    z0 = cclqr.examples.cartpole_states(n_links, [0.2], phi)
    oc = orc.ctrl_desc(nb, [0], K=K, N=0, zd=zd)
    _, traj_o, st_o = orc.rollout(t, oc, z0, 8, record=True)
    _, traj, st = emu_rollout(emu, orc, t, oc, z0, 8)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-10


def test_chain_emulator_under_address_and_undefined_sanitizers(tmp_path):
    """the emulator's LDS image is allocated at its exact size and NaN-poisoned (tests/emu/emu_chain.cpp); here it is
    built with -fsanitize=address,undefined and run on cartpole chains of 2, 8, 17 and 32 bodies (both lane-group sizes, every
    layout instantiation): an out-of-range LDS offset or an uninitialised read in the kernel's phase functions fails this test
    instead of reading a neighbour's zeros."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emu_dir = os.path.join(root, "tests", "emu")
    exe = str(tmp_path / "asan_emu")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-x", "hip", "--offload-host-only", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-I/opt/rocm/include", "-o", exe,
                           os.path.join(emu_dir, "emu_chain.cpp"), os.path.join(emu_dir, "emu_loop.cpp"), os.path.join(emu_dir, "asan_main.cpp")],
                          stderr=subprocess.DEVNULL)
    # the closed-loop kernel's phases on the deltabot (tables, pose and 90 % of the holding inputs handed over as a flat file)
    import __graft_entry__ as g
    pkg = g.load_package()
    ex = pkg.examples.deltabot()
    t = ex["mech"].tables()
    flat = np.concatenate([[t.nb, t.ne, t.dt, t.g], t.mass, t.inertia.ravel(), t.parent, t.child, t.type, t.p1.ravel(), t.p2.ravel(), t.axis.ravel(),
                           t.qoff.ravel(), ex["mech"].state().ravel(), 0.9 * ex["Fd"]]).astype(np.float64)
    flat.tofile(str(tmp_path / "deltabot.bin"))
    runs = [[exe, str(n_links), "12"] for n_links in (1, 7, 16, 31)] + [[exe, "loop", str(tmp_path / "deltabot.bin"), "12"]]
    for cmd in runs:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "rc 0 status" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stdout + r.stderr
        status = int(r.stdout.split("status")[1].split()[0])
        assert status > 0


def emu_loop_rollout(emu, orc, t, ctrl, z0, steps):
    m = orc.mech_desc(t)
    z0 = np.ascontiguousarray(z0, dtype=np.float64).reshape(-1, t.nb, 13)
    n = z0.shape[0]
    traj, zT, st, lam = np.zeros((n, steps, t.nb, 13)), np.zeros_like(z0), np.zeros(n, dtype=np.int32), np.zeros((n, 5 * t.ne))
    rc = emu.emu_loop_rollout(C.byref(m.desc), C.byref(ctrl.desc), C.c_int64(n), C.c_int(steps), C.c_int(1), z0.ctypes.data_as(dp), lam.ctypes.data_as(dp),
                              traj.ctypes.data_as(dp), zT.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0
    return zT, traj, st


def loop_joint_coordinate(j, z):
    """minimalCoordinates of a 1-DoF joint of oracle/loops.py from its two bodies' poses, as oracle/cclqr_oracle.c joint_coordinate states it for
    trees (pid.jl:43-57 reads it): revolute -> angle of qa^-1 qb qoff^-1 about the axis, prismatic -> offset along the axis"""
    from oracle import loops
    conj = lambda q: np.concatenate([[q[0]], -q[1:]])
    ax = j.axis / np.linalg.norm(j.axis)
    qa = z[j.parent, 3:7] if j.parent >= 0 else np.array([1.0, 0, 0, 0])
    xa = z[j.parent, 0:3] if j.parent >= 0 else np.zeros(3)
    if j.kind == loops.REVOLUTE:
        e = loops.qmul(loops.qmul(conj(qa), z[j.child, 3:7]), conj(j.qoff))
        return 2.0 * np.arctan2(ax @ e[1:], e[0])
    w = z[j.child, 0:3] + loops.rot(z[j.child, 3:7]) @ j.p2 - xa
    return ax @ (loops.rot(qa).T @ w - j.p1)


def loop_feedback_reference(lm, z, Fd, K, zd, steps, fric=None, noise=None, noise_scale=0.0, pid=None):
    """oracle side of a closed-loop rollout: u = Fd - K dz (lqr.jl:92-111, error order x, v, q~, w per body) on the controlled joints
    0, 1 of oracle/loops.py's deltabot, stepped by its dense-KKT minimum-norm Newton; fric [njoints]: viscous friction -fric * (axis . relative
    angular velocity, each body's in its own frame) on every revolute joint and noise [steps] * noise_scale on the controlled ones
    (examples/trackingLQR_triple_cartpole.jl:93-111 as ck_friction / lp_friction state it)"""
    from oracle import loops
    traj, lam = [], np.zeros(lm.nrows)
    integ, last = (np.zeros(len(pid["joint"])), np.zeros(len(pid["joint"]))) if pid else (None, None)
    for k in range(steps):
        traj.append(z.copy())
        u = np.zeros(len(lm.joints))
        if pid:                                    # control_pid!, pid.jl:69-88 (k is 0-based here: the reference's k == 1 is the first step)
            for i, ji in enumerate(pid["joint"]):
                e = pid["goal"][i] - loop_joint_coordinate(lm.joints[ji], z)
                if lm.joints[ji].kind == loops.REVOLUTE:
                    e = e - 2 * np.pi if e > np.pi else (e + 2 * np.pi if e < -np.pi else e)
                if k == 0:
                    last[i] = e
                integ[i] += e * lm.dt
                u[ji] += pid["P"][i] * e + pid["I"][i] * integ[i] + pid["D"][i] * (e - last[i]) / lm.dt
                last[i] = e
        if fric is not None:
            for ji, j in enumerate(lm.joints):
                if j.kind == loops.REVOLUTE and fric[ji] != 0.0:
                    ax = j.axis / np.linalg.norm(j.axis)
                    rel = ax @ z[j.child, 10:13] - (ax @ z[j.parent, 10:13] if j.parent >= 0 else 0.0)
                    u[ji] += -fric[ji] * rel
        if noise is not None:
            u[:2] += noise_scale * noise[k]
        if K is None:
            u[:2] += Fd
        else:
            dz = np.zeros((lm.nb, 12))
            for b in range(lm.nb):
                qe = loops.qmul(np.concatenate([[zd[b, 3]], -zd[b, 4:7]]), z[b, 3:7])
                dz[b] = np.concatenate([z[b, 0:3] - zd[b, 0:3], z[b, 7:10] - zd[b, 7:10], qe[1:], z[b, 10:13] - zd[b, 10:13]])
            u[:2] += Fd - K @ dz.ravel()
        z, lam, _ = lm.step(z, lam, u)
    return np.array(traj), z


def test_emulated_closed_loop_deltabot(cclqr, orc, emu):
    """examples/lqr_deltabot.jl:25-53 on the closed-loop kernel's phase functions (csrc/cclqr_loop.h): the reference's holding torque
    keeps the script's pose at rest; with 80 % of it, and with a feedback law on top, the trajectory equals the oracle's dense-KKT
    minimum-norm solution (oracle/loops.py) although the device solves the singular Schur complement by rank-truncated pivoting --
    velocities and poses are unique, multipliers are not"""
    from oracle import loops
    ex = cclqr.examples.deltabot()
    mech = ex["mech"]
    t = mech.tables()
    assert mech.has_loops and (t.nb, t.ne) == (5, 7)
    cj = [mech.joint_index(e) for e in ex["eqcids"]]
    lm, z, u = loops.deltabot()
    z0 = mech.state()
    assert np.abs(z0 - z).max() == 0.0 and cj == [0, 1]
    zd = z0[None].copy()
    hold = orc.ctrl_desc(t.nb, cj, K=None, N=0, zd=zd, Fd=ex["Fd"].reshape(1, 2))
    zT, _, st = emu_loop_rollout(emu, orc, t, hold, z0, 50)
    assert st[0] > 0 and np.abs(zT[0] - z0).max() < 1e-6            # 6.7879484 is the torque to eight digits: at rest to 1e-6
    steps = 25
    weak = orc.ctrl_desc(t.nb, cj, K=None, N=0, zd=zd, Fd=0.8 * ex["Fd"].reshape(1, 2))
    zT, traj, st = emu_loop_rollout(emu, orc, t, weak, z0, steps)
    ref, zref = loop_feedback_reference(lm, z.copy(), 0.8 * ex["Fd"], None, zd[0], steps)
    assert st[0] > 0 and np.abs(zT[0] - z0).max() > 0.5             # it falls a long way
    assert np.abs(traj[0] - ref).max() < 1e-10 and np.abs(zT[0] - zref).max() < 1e-10
    rng = np.random.default_rng(3)
    K = rng.normal(size=(1, 2, 12 * t.nb)) * 2.0
    fb = orc.ctrl_desc(t.nb, cj, K=K, N=0, zd=zd, Fd=0.8 * ex["Fd"].reshape(1, 2))
    zT, traj, st = emu_loop_rollout(emu, orc, t, fb, z0, steps)
    ref, zref = loop_feedback_reference(lm, z.copy(), 0.8 * ex["Fd"], K[0], zd[0], steps)
    assert st[0] > 0 and np.abs(traj[0] - ref).max() < 1e-9 and np.abs(zT[0] - zref).max() < 1e-9
    # joint friction on a loop mechanism (round 4: lp_friction, the law of trackingLQR_triple_cartpole.jl:93-101 per joint in the caller's order)
    fric = np.array([0.5, 0.3, 0.4, 0.2, 0.6, 0.0, 0.0])
    fr = orc.ctrl_desc(t.nb, cj, K=K, N=0, zd=zd, Fd=0.8 * ex["Fd"].reshape(1, 2), fric=fric)
    zTf, trajf, st = emu_loop_rollout(emu, orc, t, fr, z0, steps)
    reff, zreff = loop_feedback_reference(lm, z.copy(), 0.8 * ex["Fd"], K[0], zd[0], steps, fric=fric)
    assert st[0] > 0 and np.abs(trajf[0] - reff).max() < 1e-9 and np.abs(zTf[0] - zreff).max() < 1e-9
    assert np.abs(zTf[0] - zT[0]).max() > 1e-3                      # (the friction did something)
    # PID on a loop mechanism (round 4: lp_pid, pid.jl:69-88 on the two actuated platform joints; the joint coordinate from the joint's two
    # bodies, goals 0.15 rad off the start): on top of the holding torques, against the dense-KKT reference under the same law
    pid = dict(joint=cj, P=[8.0, 6.0], I=[3.0, 2.0], D=[0.4, 0.3], goal=[loop_joint_coordinate(lm.joints[j], z) + d for j, d in zip(cj, (0.15, -0.1))])
    pc = orc.ctrl_desc(t.nb, cj, K=None, N=0, zd=zd, Fd=ex["Fd"].reshape(1, 2), pid=pid)
    zTp, trajp, st = emu_loop_rollout(emu, orc, t, pc, z0, steps)
    refp, zrefp = loop_feedback_reference(lm, z.copy(), ex["Fd"], None, zd[0], steps, pid=pid)
    assert st[0] > 0 and np.abs(trajp[0] - refp).max() < 1e-9 and np.abs(zTp[0] - zrefp).max() < 1e-9
    assert np.abs(zTp[0] - z0).max() > 1e-2                         # (the PID moved the platform)


def test_emulated_fourbar_linkage(cclqr, orc, emu):
    """a second closed-loop topology for the column-order Gauss-Jordan solve of round 4 (csrc/cclqr_loop.h): a parallelogram four-bar linkage --
    three bodies, four revolutes about one axis, 20 constraint rows of rank 17 (every joint repeats the two out-of-plane rotational rows and the
    loop closes one planar row pair twice).  The mechanism is closed at its pose, falls under gravity with a PID-free feedback law on the crank,
    and equals the dense-KKT minimum-norm reference (oracle/loops.py, built from the same tables) step by step"""
    from oracle import loops
    ex = cclqr.examples.fourbar()
    mech = ex["mech"]
    t = mech.tables()
    assert mech.has_loops and (t.nb, t.ne) == (3, 4)
    lm = loops.from_tables(t)
    z0 = mech.state()
    assert np.abs(lm.constraints(z0)).max() < 1e-14                          # the loop is closed at the placed pose
    steps = 40
    rng = np.random.default_rng(11)
    K = rng.normal(size=(1, 1, 12 * t.nb)) * 1.5
    Fd = np.array([0.7])
    c = orc.ctrl_desc(t.nb, [0], K=K, N=0, zd=z0[None], Fd=Fd.reshape(1, 1))
    zT, traj, st = emu_loop_rollout(emu, orc, t, c, z0, steps)
    traj_ref, lam, z = [], np.zeros(lm.nrows), z0.copy()
    for k in range(steps):
        traj_ref.append(z.copy())
        u = np.zeros(4)
        u[0] = Fd[0] - K[0, 0] @ loops.state_error(z, z0)
        z, lam, _ = lm.step(z, lam, u)
    assert st[0] > 0
    assert np.abs(traj[0] - np.array(traj_ref)).max() < 1e-9 and np.abs(zT[0] - z).max() < 1e-9
    assert np.abs(lm.constraints(zT[0])).max() < 1e-11 and np.abs(zT[0] - z0).max() > 0.05      # still closed, and it moved


def emu_loop_linearize(emu, orc, t, zd, cj, Fd, force_loop):
    m = orc.mech_desc(t)
    nb, nj, mu = t.nb, t.ne, len(cj)
    mx, ml = 12 * nb, 5 * nj
    A, Bu, Bl, G = np.zeros((mx, mx)), np.zeros((mx, max(mu, 1))), np.zeros((mx, ml)), np.zeros((ml, mx))
    cja = np.ascontiguousarray(cj, dtype=np.int32)
    Fd = np.ascontiguousarray(Fd, dtype=np.float64)
    zd = np.ascontiguousarray(zd, dtype=np.float64)
    st = C.c_int(0)
    rc = emu.emu_loop_linearize(C.byref(m.desc), C.c_int(1 if force_loop else 0), zd.ctypes.data_as(dp), C.c_int(mu), cja.ctypes.data_as(C.POINTER(C.c_int32)),
                                Fd.ctypes.data_as(dp), A.ctypes.data_as(dp), Bu.ctypes.data_as(dp), Bl.ctypes.data_as(dp), G.ctypes.data_as(dp), C.byref(st))
    assert rc == 0 and st.value > 0, (rc, st.value)
    return A, Bu[:, :mu].copy() if mu else Bu[:, :0], Bl, G


def projected_pair(A, Bu, Bl, G):
    """A' = A - Bl X, D = Bu - Bl Y with (G Bl) [X | Y] = G [A | Bu] in the minimum-norm least-squares sense (a loop makes G Bl singular;
    Bl X is the same for every solution)"""
    XY = np.linalg.lstsq(G @ Bl, G @ np.concatenate([A, Bu], axis=1), rcond=1e-11)[0]
    P = np.concatenate([A, Bu], axis=1) - Bl @ XY
    return P[:, :A.shape[1]], P[:, A.shape[1]:]


def test_emulated_loop_linearisation(cclqr, orc, emu):
    """csrc/cclqr_lin_loop.h (linearsystem on the closed-loop layout, lqr.jl:63 for examples/lqr_deltabot.jl:47-53), run by the emulator:
      * TREES sent through the closed-loop tables (a three-link cartpole off its equilibrium with a feed-forward input, the 7-body Sawyer arm with dense inertias, at
        moving setpoints): A, Bu, Bl, G equal the oracle's analytic tree linearisation to 1e-10 -- same model, other bookkeeping;
      * the deltabot at its holding torque: the projected pair built from (A, Bu, Bl, G) equals central differences of the oracle's dense-KKT
        step map (oracle/loops.py) to the accuracy of those differences, G [A' | D] = 0, and rank(G Bl) = 28 of 35."""
    import json
    import os
    from oracle import loops
    ex = cclqr.examples.cartpole_n(3)
    t = ex["mech"].tables()
    rng = np.random.default_rng(5)
    zd = cclqr.examples.cartpole_states(3, [0.13], np.array([[0.4, -0.3, 0.25]]))[0]
    Fd = np.array([1.7])
    Ao, Buo, Blo, Go = orc.linearize(t, zd, [0], Fd)
    A, Bu, Bl, G = emu_loop_linearize(emu, orc, t, zd, [0], Fd, force_loop=True)
    for X, Xo, name in ((A, Ao, "A"), (Bu, Buo, "Bu"), (Bl, Blo, "Bl"), (G, Go, "G")):
        assert np.abs(X - Xo).max() < 1e-10 * max(1.0, np.abs(Xo).max()), (name, np.abs(X - Xo).max())
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sawyer_arm_tables.json")))
    exs = cclqr.examples.sawyer(tab)
    ts = exs["mech"].tables()
    zs = cclqr.joint_position_states(exs["mech"], rng.uniform(-0.6, 0.6, (1, 7)))[0]
    cj = [0, 2, 5]
    Fs = np.array([0.3, -0.2, 0.1])
    Ao, Buo, Blo, Go = orc.linearize(ts, zs, cj, Fs)
    A, Bu, Bl, G = emu_loop_linearize(emu, orc, ts, zs, cj, Fs, force_loop=True)
    for X, Xo, name in ((A, Ao, "A"), (Bu, Buo, "Bu"), (Bl, Blo, "Bl"), (G, Go, "G")):
        assert np.abs(X - Xo).max() < 1e-10 * max(1.0, np.abs(Xo).max()), (name, np.abs(X - Xo).max())
    # the deltabot
    exd = cclqr.examples.deltabot()
    mech = exd["mech"]
    td = mech.tables()
    cjd = [mech.joint_index(e) for e in exd["eqcids"]]
    lm, z, u = loops.deltabot()
    A, Bu, Bl, G = emu_loop_linearize(emu, orc, td, mech.state(), cjd, exd["Fd"].reshape(-1), force_loop=False)
    sv = np.linalg.svd(G @ Bl, compute_uv=False)
    assert int((sv > 1e-9 * sv[0]).sum()) == 28
    Ap, D = projected_pair(A, Bu, Bl, G)
    assert np.abs(G @ Ap).max() < 1e-8 and np.abs(G @ D).max() < 1e-8
    Apo, Do = loops.projected_linear_model(lm, z, u, cjd)
    assert np.abs(Ap - Apo).max() < 2e-6 * max(1.0, np.abs(Apo).max()) and np.abs(D - Do).max() < 2e-6 * max(1.0, np.abs(Do).max())
