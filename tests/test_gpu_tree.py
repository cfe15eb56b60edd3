"""GPU tests of the general-tree path through the C-ABI: rollout_kernel<G, TREE=true> / linearize_kernel<true> against the oracle."""
import numpy as np
import pytest
import scipy.linalg as sl

from test_tree import TREES, build, _random_parents

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.mark.parametrize("name", list(TREES))
def test_tree_rollout_matches_oracle(cclqr, orc, name):
    capi = cclqr._capi
    ex = build(cclqr, name)
    mech = ex["mech"]
    t = mech.tables()
    rng = np.random.default_rng(3)
    # a batch of consistent start poses: random joint coordinates, root to leaf
    z0 = []
    for n in range(9):
        for e in ex["joints"]:
            cclqr.setJointPosition(mech, e, rng.uniform(-0.5, 0.5))
        z0.append(mech.state())
    z0 = np.stack(z0)
    steps = 60
    cj = [0, t.ne - 1]
    K = rng.normal(size=(steps + 5, 2, 12 * t.nb)) * 0.05
    Fd = rng.normal(size=(1, 2)) * 0.3
    kw = dict(K=K, N=steps + 6, zd=z0[0], Fd=Fd)
    zo, traj_o, st_o = orc.rollout(t, orc.ctrl_desc(t.nb, cj, **kw), z0, steps, record=True)
    h = capi.MechHandle(t)
    lanes, lds = h.geometry()
    assert lanes in (16, 32, 64) and lanes >= t.nb
    zT, traj, st = capi.rollout(h, capi.CtrlHandle(h, cj, **kw), z0, steps, record=True)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < TOL and np.abs(zT - zo).max() < TOL
    for i in range(len(z0)):
        assert np.abs(orc.constraints(t, zT[i])).max() < 1e-9


@pytest.mark.parametrize("name", ["dual_cartpole", "three_children", "deep"])
def test_tree_linearise_matches_oracle(cclqr, orc, name):
    capi = cclqr._capi
    ex = build(cclqr, name)
    t = ex["mech"].tables()
    z = ex["mech"].state()
    cj, Fd = [0, t.ne - 1], np.array([[0.4, -0.2]])
    A, Bu, Bl, G = (m[0] for m in capi.linearize(capi.MechHandle(t), z[None], cj, Fd))
    Ao, Buo, Blo, Go = orc.linearize(t, z, cj, Fd[0])
    for X, Xo in ((A, Ao), (Bu, Buo), (Bl, Blo), (G, Go)):
        assert np.abs(X - Xo).max() < 1e-8 * max(1.0, np.abs(Xo).max())


def test_dual_cartpole_lqr_pipeline(cclqr, orc):
    """LQR(...) + simulate! through the host mirror on the branched cart: one cart force balances two poles of different lengths"""
    ex = cclqr.examples.dual_cartpole(0.03, -0.02, 0.1)
    mech = ex["mech"]
    zd = cclqr.examples.dual_cartpole(0.0, 0.0, 0.0)["mech"].state()
    lqr = cclqr.LQR(mech, [cclqr.getid(b) for b in ex["bodies"]], [cclqr.getid(ex["ctrl"][0])], ex["Q"], ex["R"], 10.0,
                    xd=[zd[i, 0:3] for i in range(3)], qd=[zd[i, 3:7] for i in range(3)])
    t = mech.tables()
    Ao, Buo, Blo, Go = orc.linearize(t, zd, [0], np.zeros(1))
    Ko, kbo = orc.riccati(Ao, Buo, Blo, Go, lqr.Q, lqr.R, 1000)
    assert lqr.kbreak == kbo and np.abs(lqr.K - Ko).max() < 1e-7 * np.abs(Ko).max()
    rng = np.random.default_rng(0)
    z0 = np.stack([cclqr.examples.dual_cartpole(*rng.uniform(-0.04, 0.04, 2), rng.uniform(-0.2, 0.2))["mech"].state() for _ in range(64)])
    st = cclqr.simulate(mech, 8.0, lqr, z0=z0, record=False)
    assert (st.status > 0).all()
    for i in range(0, 64, 7):
        th = orc.minimal_coordinates(t, st.zT[i])
        assert abs(th[0]) < 0.05 and abs(th[1]) < 0.02 and abs(th[2]) < 0.02
    zo, _, sto = orc.rollout(t, orc.ctrl_desc(3, [0], K=lqr.K, N=lqr.N, zd=lqr.zd), z0[:8], 800)
    assert np.abs(st.zT[:8] - zo).max() < 1e-8


@pytest.mark.parametrize("seed", range(10))
def test_random_topologies_gpu_vs_oracle(cclqr, orc, seed):
    """random forests (up to 4 child joints per body, several roots, mixed joint types): batched rollout + linearisation on the GPU"""
    capi = cclqr._capi
    rng = np.random.default_rng(2000 + seed)
    nb = int(rng.integers(2, 15))
    parents = _random_parents(rng, nb)
    prism = tuple(int(i) for i in range(nb) if rng.uniform() < 0.2)
    ex = cclqr.examples.tree_mechanism(parents, seed=seed, prismatic=prism, g=-9.81 if seed % 3 else 0.0)
    mech = ex["mech"]
    t = mech.tables()
    z0 = []
    for n in range(7):
        for e in ex["joints"]:
            cclqr.setJointPosition(mech, e, rng.uniform(-0.5, 0.5))
        z0.append(mech.state())
    z0 = np.stack(z0)
    steps = 25
    cj = sorted(set(int(j) for j in rng.integers(0, t.ne, 2)))
    K = rng.normal(size=(steps + 3, len(cj), 12 * t.nb)) * 0.05
    Fd = rng.normal(size=(1, len(cj))) * 0.3
    kw = dict(K=K, N=steps + 4, zd=z0[0], Fd=Fd, fric=rng.uniform(0, 0.05, t.ne))
    zo, traj_o, st_o = orc.rollout(t, orc.ctrl_desc(t.nb, cj, **kw), z0, steps, record=True)
    h = capi.MechHandle(t)
    zT, traj, st = capi.rollout(h, capi.CtrlHandle(h, cj, **kw), z0, steps, record=True)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < TOL
    A, Bu, Bl, G = capi.linearize(h, z0[:2], cj, np.tile(Fd, (2, 1)))
    for k in range(2):
        Ao, Buo, Blo, Go = orc.linearize(t, z0[k], cj, Fd[0])
        for X, Xo in ((A[k], Ao), (Bu[k], Buo), (Bl[k], Blo), (G[k], Go)):
            assert np.abs(X - Xo).max() < 1e-8 * max(1.0, np.abs(Xo).max())
