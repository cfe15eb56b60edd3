"""General trees (a body with several child joints): the oracle's tree LDU / linearisation on branched mechanisms, and the
kernels' sibling-coupled Schur complement + table-driven elimination (csrc/cclqr_dev.h ph_*_tree) run through the CPU emulator."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sl

from test_emulated_kernel import emu_rollout
from test_oracle import _err_state, _perturb

dp = C.POINTER(C.c_double)

TREES = {
    "dual_cartpole": None,
    "small4": [-1, 0, 0, 1],                   # four bodies, two joints on body 0: the 16-lane tree kernel like the dual cartpole
    "y": [-1, 0, 1, 2, 1, 4],                  # chain 0-1-2-3 with a branch 1-4-5
    "three_children": [-1, 0, 0, 0, 2, 2],     # three joints on body 0, two on body 2
    "four_children": [-1, 0, 0, 0, 0, -1, 5, 5],
    "deep": [-1, 0, 1, 2, 3, 2, 5, 6, 1, 8, 8, 10, 0, 12],
}


def build(cclqr, name):
    if name == "dual_cartpole":
        return cclqr.examples.dual_cartpole()
    return cclqr.examples.tree_mechanism(TREES[name], seed=len(name), prismatic=(0,) if name != "deep" else (0, 5))


@pytest.mark.parametrize("name", ["dual_cartpole", "y", "three_children"])
def test_oracle_tree_step_and_linearisation(cclqr, orc, name):
    """branched mechanisms in the oracle: constraints hold along a free fall, and A, Bu, Bl, G are the Jacobians of its own step map"""
    ex = build(cclqr, name)
    t = ex["mech"].tables()
    z = ex["mech"].state()
    assert np.abs(orc.constraints(t, z)).max() < 1e-12
    lam = np.zeros(5 * t.ne)
    uj0 = np.linspace(-0.3, 0.4, t.ne)
    for _ in range(15):
        z, lam, it = orc.step(t, z, lam, uj0)
        assert it > 0
    assert np.abs(orc.constraints(t, z)).max() < 1e-9
    cj, Fd = [0, t.ne - 1], np.array([0.7, -0.4])
    A, Bu, Bl, G = orc.linearize(t, z, cj, Fd)
    uj = np.zeros(t.ne)
    for i, j in enumerate(cj):
        uj[j] += Fd[i]
    zn, lam, it = orc.step(t, z, np.zeros(5 * t.ne), uj)
    mx, h = 12 * t.nb, 1e-6
    Afd, Gfd = np.zeros((mx, mx)), np.zeros((5 * t.ne, mx))
    for c in range(mx):
        e = np.zeros(mx)
        e[c] = h
        zp = orc.step_fixed_lambda(t, _perturb(cclqr, t, z, e), lam, uj)
        zm = orc.step_fixed_lambda(t, _perturb(cclqr, t, z, -e), lam, uj)
        Afd[:, c] = (_err_state(cclqr, t, zp, zn) - _err_state(cclqr, t, zm, zn)) / (2 * h)
        Gfd[:, c] = (orc.constraints(t, _perturb(cclqr, t, zn, e)) - orc.constraints(t, _perturb(cclqr, t, zn, -e))) / (2 * h)
    assert np.abs(A - Afd).max() < 5e-8 * max(1.0, np.abs(A).max())
    assert np.abs(G - Gfd).max() < 1e-7
    for i, j in enumerate(cj):
        up, um = uj.copy(), uj.copy()
        up[j] += h
        um[j] -= h
        col = (_err_state(cclqr, t, orc.step_fixed_lambda(t, z, lam, up), zn) - _err_state(cclqr, t, orc.step_fixed_lambda(t, z, lam, um), zn)) / (2 * h)
        assert np.abs(Bu[:, i] - col).max() < 1e-7
    for c in range(5 * t.ne):
        lp, lm = lam.copy(), lam.copy()
        lp[c] += h
        lm[c] -= h
        col = (_err_state(cclqr, t, orc.step_fixed_lambda(t, z, lp, uj), zn) - _err_state(cclqr, t, orc.step_fixed_lambda(t, z, lm, uj), zn)) / (2 * h)
        assert np.abs(Bl[:, c] - col).max() < 1e-7


@pytest.mark.parametrize("name", list(TREES))
def test_emulated_tree_rollout_matches_oracle(cclqr, orc, emu, name):
    """the kernels' tree path (sibling blocks, elimination program, child lists) against the oracle's tree LDU, with feedback"""
    ex = build(cclqr, name)
    t = ex["mech"].tables()
    z0 = ex["mech"].state()[None]
    rng = np.random.default_rng(5)
    steps = 40
    cj = [0, t.ne - 1]
    K = rng.normal(size=(steps + 5, 2, 12 * t.nb)) * 0.05
    Fd = rng.normal(size=(1, 2)) * 0.3
    oc = orc.ctrl_desc(t.nb, cj, K=K, N=steps + 6, zd=z0[0], Fd=Fd)
    zo, traj_o, st_o = orc.rollout(t, oc, z0, steps, record=True)
    zT, traj, st = emu_rollout(emu, orc, t, oc, z0, steps)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-9
    assert np.abs(zT - zo).max() < 1e-9


def emu_treereg_rollout(emu, orc, t, ctrl, z0, steps):
    """the register-resident tree kernel's phases and schedule on the CPU (tests/emu/emu_treereg.cpp)"""
    m = orc.mech_desc(t)
    z0 = np.ascontiguousarray(z0, dtype=np.float64).reshape(-1, t.nb, 13)
    n = z0.shape[0]
    traj, zT, st = np.zeros((n, steps, t.nb, 13)), np.zeros_like(z0), np.zeros(n, dtype=np.int32)
    rc = emu.emu_treereg_rollout(C.byref(m.desc), C.byref(ctrl.desc), C.c_int64(n), C.c_int(steps), C.c_int(1), z0.ctypes.data_as(dp), None,
                                 traj.ctypes.data_as(dp), zT.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(0))
    assert rc == 0
    return zT, traj, st


@pytest.mark.parametrize("name", list(TREES))
def test_emulated_register_resident_tree_rollout_matches_oracle(cclqr, orc, emu, name):
    """csrc/cclqr_treereg.h: sparse sibling blocks, scheduled elimination records, child sums -- against the oracle's tree LDU, with feedback and friction"""
    ex = build(cclqr, name)
    t = ex["mech"].tables()
    z0 = ex["mech"].state()[None]
    rng = np.random.default_rng(5)
    steps = 40
    cj = [0, t.ne - 1]
    K = rng.normal(size=(steps + 5, 2, 12 * t.nb)) * 0.05
    Fd = rng.normal(size=(1, 2)) * 0.3
    oc = orc.ctrl_desc(t.nb, cj, K=K, N=steps + 6, zd=z0[0], Fd=Fd, fric=rng.uniform(0, 0.05, t.ne))
    zo, traj_o, st_o = orc.rollout(t, oc, z0, steps, record=True)
    zT, traj, st = emu_treereg_rollout(emu, orc, t, oc, z0, steps)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-9
    assert np.abs(zT - zo).max() < 1e-9


@pytest.mark.parametrize("nb,seed", [(40, 1), (64, 4)])
def test_emulated_trees_of_33_to_64_links(cclqr, orc, emu, nb, seed):
    """the 64-lane tree tables (cclqr_treereg_tables.h: eight lane groups, schedules of up to 64 steps) and phase functions against the oracle on
    random forests of 40 and 64 bodies -- the CPU side of tests/test_gpu_treereg.py::test_trees_of_33_to_64_links"""
    rng = np.random.default_rng(7000 + seed)
    parents = _random_parents(rng, nb)
    prism = tuple(int(i) for i in range(nb) if rng.uniform() < 0.15)
    ex = cclqr.examples.tree_mechanism(parents, seed=seed, prismatic=prism, g=-9.81 if seed % 2 else 0.0)
    t = ex["mech"].tables()
    z0 = ex["mech"].state()[None]
    steps = 8
    cj = sorted(set(int(j) for j in rng.integers(0, t.ne, 2)))
    oc = orc.ctrl_desc(t.nb, cj, K=rng.normal(size=(steps + 3, len(cj), 12 * t.nb)) * 0.02, N=steps + 4, zd=z0[0], Fd=rng.normal(size=(1, len(cj))) * 0.2)
    zo, traj_o, st_o = orc.rollout(t, oc, z0, steps, record=True)
    zT, traj, st = emu_treereg_rollout(emu, orc, t, oc, z0, steps)
    assert (st_o > 0).all() and np.array_equal(st, st_o)
    assert np.abs(traj - traj_o).max() < 1e-10


def test_emulated_whole_sawyer_robot(cclqr, orc, emu):
    """examples_files/sawyer.urdf with its fixed joints lumped (tests/golden/sawyer_full_tables.json): eight bodies, the head and the arm both on
    the first link -- a real robot that branches, under gravity, through the register-resident tree kernel's phases"""
    import json
    import os
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sawyer_full_tables.json")))
    ex = cclqr.examples.sawyer(tab, g=-9.81)
    mech = ex["mech"]
    t = mech.tables()
    rng = np.random.default_rng(3)
    zd = mech.state()
    K = rng.normal(size=(49, 8, 96)) * 0.004
    z0 = []
    for n in range(2):
        for e in mech.eqconstraints:
            cclqr.setJointPosition(mech, e, rng.uniform(-0.3, 0.3))
        z0.append(mech.state())
    z0 = np.stack(z0)
    c = orc.ctrl_desc(8, list(range(8)), K=K, N=50, zd=zd, Fd=0.02 * rng.normal(size=(1, 8)))
    zT_o, traj_o, st_o = orc.rollout(t, c, z0, 40, record=True)
    zT, traj, st = emu_treereg_rollout(emu, orc, t, c, z0, 40)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-10


@pytest.mark.parametrize("name", ["dual_cartpole", "y", "four_children"])
def test_emulated_tree_linearisation_matches_oracle(cclqr, orc, emu, name):
    ex = build(cclqr, name)
    t = ex["mech"].tables()
    m = orc.mech_desc(t)
    z = ex["mech"].state()
    cj = np.array([0, t.ne - 1], dtype=np.int32)
    Fd = np.array([0.4, -0.2])
    mx, ml = 12 * t.nb, 5 * t.ne
    A, Bu, Bl, G = np.zeros((mx, mx)), np.zeros((mx, 2)), np.zeros((mx, ml)), np.zeros((ml, mx))
    rc = emu.emu_linearize(C.byref(m.desc), z.ctypes.data_as(dp), 2, cj.ctypes.data_as(C.POINTER(C.c_int32)), Fd.ctypes.data_as(dp),
                           A.ctypes.data_as(dp), Bu.ctypes.data_as(dp), Bl.ctypes.data_as(dp), G.ctypes.data_as(dp))
    assert rc == 0
    Ao, Buo, Blo, Go = orc.linearize(t, z, list(cj), Fd)
    for X, Xo in ((A, Ao), (Bu, Buo), (Bl, Blo), (G, Go)):
        assert np.abs(X - Xo).max() < 1e-8 * max(1.0, np.abs(Xo).max())


def test_dual_cartpole_lqr_balances_both_poles(cclqr, orc, emu):
    """LQR on the branched cart: one input (the cart force) balances two poles of different lengths -- the classic dual-pole problem"""
    ex = cclqr.examples.dual_cartpole(0.03, -0.02, 0.1)
    t = ex["mech"].tables()
    zd = cclqr.examples.dual_cartpole(0.0, 0.0, 0.0)["mech"].state()
    A, Bu, Bl, G = orc.linearize(t, zd, [0], np.zeros(1))
    Q = sl.block_diag(*ex["Q"]) * t.dt
    R = sl.block_diag(*ex["R"]) * t.dt
    K, kb = orc.riccati(A, Bu, Bl, G, Q, R, 1000)
    oc = orc.ctrl_desc(3, [0], K=K, N=1000, zd=zd)
    z0 = ex["mech"].state()[None]
    zo, traj_o, st_o = orc.rollout(t, oc, z0, 600, record=True)
    assert (st_o > 0).all()
    th = orc.minimal_coordinates(t, zo[0])
    assert abs(th[0]) < 0.05 and abs(th[1]) < 0.02 and abs(th[2]) < 0.02
    zT, traj, st = emu_rollout(emu, orc, t, oc, z0, 600)
    assert (st > 0).all() and np.abs(traj - traj_o).max() < 1e-8


def _random_parents(rng, nb, max_children=4):
    """random forest: each body picks a parent among the earlier bodies (or the origin) that still has room"""
    parents, count = [], {}
    for i in range(nb):
        cands = [-1] + [a for a in range(i) if count.get(a, 0) < max_children]
        a = int(rng.choice(cands)) if rng.uniform() > 0.15 else -1
        parents.append(a)
        if a >= 0:
            count[a] = count.get(a, 0) + 1
    return parents


@pytest.mark.parametrize("seed", range(24))
def test_random_topologies_emulator_vs_oracle(cclqr, orc, emu, seed):
    """random forests of 2..12 bodies (up to 4 child joints per body, several roots, revolute / prismatic joints with random axes and
    anchors, bodies listed in an order unrelated to the kernels' link order): rollout and linearisation of the kernel phases == oracle"""
    rng = np.random.default_rng(1000 + seed)
    nb = int(rng.integers(2, 13))
    parents = _random_parents(rng, nb)
    prism = tuple(int(i) for i in range(nb) if rng.uniform() < 0.2)
    ex = cclqr.examples.tree_mechanism(parents, seed=seed, prismatic=prism, g=-9.81 if seed % 3 else 0.0)
    t = ex["mech"].tables()
    z0 = ex["mech"].state()[None]
    assert np.abs(orc.constraints(t, z0[0])).max() < 1e-12
    steps = 12
    cj = sorted(set(int(j) for j in rng.integers(0, t.ne, 2)))
    K = rng.normal(size=(steps + 3, len(cj), 12 * t.nb)) * 0.05
    Fd = rng.normal(size=(1, len(cj))) * 0.3
    oc = orc.ctrl_desc(t.nb, cj, K=K, N=steps + 4, zd=z0[0], Fd=Fd, fric=rng.uniform(0, 0.05, t.ne))
    zo, traj_o, st_o = orc.rollout(t, oc, z0, steps, record=True)
    zT, traj, st = emu_rollout(emu, orc, t, oc, z0, steps)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < 1e-9 and np.abs(zT - zo).max() < 1e-9
    if any(parents.count(a) > 1 for a in set(parents) if a >= 0):     # branched: the register-resident tree kernel's phases too
        zT2, traj2, st2 = emu_treereg_rollout(emu, orc, t, oc, z0, steps)
        assert (st2 > 0).all() and np.abs(traj2 - traj_o).max() < 1e-9 and np.abs(zT2 - zo).max() < 1e-9
    m = orc.mech_desc(t)
    cja = np.array(cj, dtype=np.int32)
    mx, ml, mu = 12 * t.nb, 5 * t.ne, len(cj)
    A, Bu, Bl, G = np.zeros((mx, mx)), np.zeros((mx, mu)), np.zeros((mx, ml)), np.zeros((ml, mx))
    rc = emu.emu_linearize(C.byref(m.desc), z0[0].ctypes.data_as(dp), mu, cja.ctypes.data_as(C.POINTER(C.c_int32)), Fd[0].ctypes.data_as(dp),
                           A.ctypes.data_as(dp), Bu.ctypes.data_as(dp), Bl.ctypes.data_as(dp), G.ctypes.data_as(dp))
    assert rc == 0
    Ao, Buo, Blo, Go = orc.linearize(t, z0[0], cj, Fd[0])
    for X, Xo in ((A, Ao), (Bu, Buo), (Bl, Blo), (G, Go)):
        assert np.abs(X - Xo).max() < 1e-8 * max(1.0, np.abs(Xo).max())
