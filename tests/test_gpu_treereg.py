"""GPU tests of the register-resident tree kernel (csrc/rollout_treereg.hip) beyond tests/test_gpu_tree.py: the control laws that came with it
(PID, injected and Philox noise, friction; newton_mode 1), launches chained through multipliers and PID state, every image size up to 32 links,
and the launch geometry.  Everything goes through the C-ABI and is compared with the oracle."""
import numpy as np
import pytest

from test_tree import TREES, build, _random_parents

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _starts(cclqr, ex, rng, n):
    z0 = []
    for _ in range(n):
        for e in ex["joints"]:
            cclqr.setJointPosition(ex["mech"], e, rng.uniform(-0.5, 0.5))
        z0.append(ex["mech"].state())
    return np.stack(z0)


@pytest.mark.parametrize("name", ["dual_cartpole", "three_children", "deep"])
def test_tree_pid_friction_and_injected_noise_match_oracle(cclqr, orc, name):
    """PID on two joints (pid.jl:69-88) on top of the LQR law with joint friction and injected noise (trackingLQR_triple_cartpole.jl:93-111) on
    branching mechanisms: the tree kernel's EXTRA = 2 instantiation against the oracle"""
    capi = cclqr._capi
    ex = build(cclqr, name)
    t = ex["mech"].tables()
    rng = np.random.default_rng(17)
    z0 = _starts(cclqr, ex, rng, 5)
    steps = 50
    cj = [0, t.ne - 1]
    K = rng.normal(size=(steps + 5, 2, 12 * t.nb)) * 0.05
    noise = rng.normal(size=(len(z0), steps))
    pid = dict(joint=[1, t.ne - 1], P=[2.0, 1.0], I=[0.5, 0.2], D=[0.1, 0.05], goal=[0.2, -0.1])
    kw = dict(K=K, N=steps + 6, zd=z0[0], Fd=rng.normal(size=(1, 2)) * 0.3, fric=rng.uniform(0, 0.05, t.ne), noise_scale=0.3, pid=pid)
    zo, traj_o, st_o = orc.rollout(t, orc.ctrl_desc(t.nb, cj, noise=noise, **kw), z0, steps, record=True)
    h = capi.MechHandle(t)
    zT, traj, st = capi.rollout(h, capi.CtrlHandle(h, cj, **kw), z0, steps, noise=noise, record=True)
    assert (st_o > 0).all() and np.array_equal(st > 0, st_o > 0)
    assert np.abs(traj - traj_o).max() < TOL and np.abs(zT - zo).max() < TOL


def test_tree_philox_noise_and_instance_offset(cclqr, orc):
    """device-generated Philox noise on a tree == the oracle's stream per GLOBAL instance; a shard launched with first_instance reproduces its slice"""
    capi = cclqr._capi
    ex = build(cclqr, "y")
    t = ex["mech"].tables()
    rng = np.random.default_rng(3)
    steps, n = 40, 21
    z0 = np.tile(ex["mech"].state(), (n, 1, 1))
    K = rng.normal(size=(steps + 5, 1, 12 * t.nb)) * 0.05
    kw = dict(K=K, N=steps + 6, zd=z0[0], fric=rng.uniform(0, 0.05, t.ne), noise_scale=0.5, noise_seed=0xBEEF)
    _, traj_o, st_o = orc.rollout(t, orc.ctrl_desc(t.nb, [0], **kw), z0, steps, record=True)
    h = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(h, [0], **kw)
    zT, traj, st = capi.rollout(h, ctrl, z0, steps, record=True)
    assert (st > 0).all() and np.abs(traj[0] - traj[1]).max() > 1e-4
    assert np.abs(traj - traj_o).max() < TOL
    zT_s, _, _ = capi.rollout(h, ctrl, z0[9:16], steps, first_instance=9)
    assert np.array_equal(zT_s, zT[9:16])


def test_tree_chained_launches_equal_one_launch(cclqr, orc):
    """step-per-launch use on a tree: multipliers (lam) and PID state carried between launches make 20 + 35 steps equal 55 steps in one, bit for bit"""
    import torch
    capi = cclqr._capi
    ex = build(cclqr, "deep")
    t = ex["mech"].tables()
    rng = np.random.default_rng(5)
    z0 = _starts(cclqr, ex, rng, 7)
    n, steps = len(z0), 55
    K = rng.normal(size=(steps + 5, 1, 12 * t.nb)) * 0.05
    pid = dict(joint=[2], P=[1.5], I=[0.4], D=[0.05], goal=[0.1])
    h = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(h, [0], K=K, N=steps + 6, zd=z0[0], pid=pid)
    dev = torch.device("cuda", 0)
    z0_d = torch.from_numpy(z0).to(dev)
    one, a, b = torch.empty_like(z0_d), torch.empty_like(z0_d), torch.empty_like(z0_d)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    capi.rollout_dev(h, ctrl, n, steps, 1, z0_d.data_ptr(), 0, 0, 0, 0, one.data_ptr(), st.data_ptr())
    lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=dev)
    pstate = torch.zeros((n, t.nb, 2), dtype=torch.float64, device=dev)
    capi.rollout_dev(h, ctrl, n, 20, 1, z0_d.data_ptr(), lam.data_ptr(), 0, 0, 0, a.data_ptr(), st.data_ptr(), pid_state=pstate.data_ptr())
    capi.rollout_dev(h, ctrl, n, 35, 21, a.data_ptr(), lam.data_ptr(), 0, 0, 0, b.data_ptr(), st.data_ptr(), pid_state=pstate.data_ptr())
    torch.cuda.synchronize()
    assert (st.cpu().numpy() > 0).all()
    assert torch.equal(b, one)
    zo, _, _ = orc.rollout(t, orc.ctrl_desc(t.nb, [0], K=K, N=steps + 6, zd=z0[0], pid=pid), z0, steps)
    assert np.abs(one.cpu().numpy() - zo).max() < TOL


def test_tree_newton_mode_residual_only(cclqr, orc):
    """newton_mode 1 on a tree (the RELAX instantiation): never more iterations than the exact rule, a small measured deviation from it"""
    capi = cclqr._capi
    ex = build(cclqr, "deep")
    t = ex["mech"].tables()
    rng = np.random.default_rng(8)
    z0 = _starts(cclqr, ex, rng, 24)
    steps = 80
    K = rng.normal(size=(steps + 5, 1, 12 * t.nb)) * 0.05
    h = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(h, [0], K=K, N=steps + 6, zd=z0[0])
    zT0, tr0, st0 = capi.rollout(h, ctrl, z0, steps, record=True)
    zT1, tr1, st1 = capi.rollout(h, ctrl, z0, steps, record=True, newton_mode=1)
    _, tro, sto = orc.rollout(t, orc.ctrl_desc(t.nb, [0], K=K, N=steps + 6, zd=z0[0]), z0[:4], steps, record=True)
    assert np.abs(tr0[:4] - tro).max() < TOL and np.array_equal(st0[:4], sto)
    dev = np.abs(tr1 - tr0).max()
    print("tree, newton_mode 1: max |state - exact rule| over %d steps = %.3g; max Newton iterations %d -> %d" % (steps, dev, st0.max(), st1.max()))
    assert (st1 > 0).all() and dev < 1e-6 and (st1 <= st0).all()


@pytest.mark.parametrize("nb,seed", [(9, 1), (11, 2), (13, 3), (16, 4), (20, 5), (24, 6), (29, 7), (32, 8)])
def test_every_image_size_up_to_32_links(cclqr, orc, nb, seed):
    """random forests of 9 .. 32 bodies (up to 4 child joints per body): the 32-lane instantiations for images of 10, 12, 14, 16, 24 and 32 links"""
    capi = cclqr._capi
    rng = np.random.default_rng(4000 + seed)
    parents = _random_parents(rng, nb)
    if not any(parents.count(a) > 1 for a in set(parents) if a >= 0):
        parents[-1] = parents[-2] if parents[-2] >= 0 else 0      # make sure it branches
    prism = tuple(int(i) for i in range(nb) if rng.uniform() < 0.15)
    ex = cclqr.examples.tree_mechanism(parents, seed=seed, prismatic=prism, g=-9.81 if seed % 2 else 0.0)
    t = ex["mech"].tables()
    z0 = _starts(cclqr, ex, rng, 5)
    steps = 15
    cj = sorted(set(int(j) for j in rng.integers(0, t.ne, 2)))
    K = rng.normal(size=(steps + 3, len(cj), 12 * t.nb)) * 0.03
    kw = dict(K=K, N=steps + 4, zd=z0[0], Fd=rng.normal(size=(1, len(cj))) * 0.2)
    zo, traj_o, st_o = orc.rollout(t, orc.ctrl_desc(t.nb, cj, **kw), z0, steps, record=True)
    h = capi.MechHandle(t)
    lanes, lds = h.geometry()
    assert lanes == 32 and lds <= 160 * 1024
    zT, traj, st = capi.rollout(h, capi.CtrlHandle(h, cj, **kw), z0, steps, record=True)
    assert (st_o > 0).all() and (st > 0).all()
    assert np.abs(traj - traj_o).max() < TOL


@pytest.mark.parametrize("nb,seed", [(33, 1), (41, 2), (48, 3), (57, 4), (64, 5)])
def test_trees_of_33_to_64_links(cclqr, orc, nb, seed):
    """round 5: a branching tree of 33 .. 64 bodies is one instance per wavefront on rollout_treereg_kernel<64, 48 | 64, law> (eight lane groups for the
    scheduled elimination), as chains of that size have been since round 4; 65 bodies are refused (CCLQR_MAXL)"""
    capi = cclqr._capi
    rng = np.random.default_rng(6000 + seed)
    parents = _random_parents(rng, nb)
    if not any(parents.count(a) > 1 for a in set(parents) if a >= 0):
        parents[-1] = parents[-2] if parents[-2] >= 0 else 0
    prism = tuple(int(i) for i in range(nb) if rng.uniform() < 0.15)
    ex = cclqr.examples.tree_mechanism(parents, seed=seed, prismatic=prism, g=-9.81 if seed % 2 else 0.0)
    t = ex["mech"].tables()
    z0 = _starts(cclqr, ex, rng, 3)
    steps = 10
    cj = sorted(set(int(j) for j in rng.integers(0, t.ne, 2)))
    K = rng.normal(size=(steps + 3, len(cj), 12 * t.nb)) * 0.02
    kw = dict(K=K, N=steps + 4, zd=z0[0], Fd=rng.normal(size=(1, len(cj))) * 0.2, fric=rng.uniform(0, 0.05, t.ne))
    zo, traj_o, st_o = orc.rollout(t, orc.ctrl_desc(t.nb, cj, **kw), z0, steps, record=True)
    h = capi.MechHandle(t)
    lanes, lds = h.geometry()
    assert lanes == 64 and lds <= 160 * 1024 and h.layout_links() == (48 if nb <= 48 else 64)
    assert h.instances_per_wavefront(10 ** 6, steps) == 1
    zT, traj, st = capi.rollout(h, capi.CtrlHandle(h, cj, **kw), z0, steps, record=True)
    assert (st_o > 0).all() and np.array_equal(st, st_o)
    assert np.abs(traj - traj_o).max() < TOL and np.abs(zT - zo).max() < TOL
    # chained single-step launches continue bit for bit (multipliers through HBM)
    import torch
    dev = torch.device("cuda", 0)
    z = torch.from_numpy(z0).to(dev)
    zn = torch.empty_like(z)
    lam = torch.zeros((len(z0), 5 * t.ne), dtype=torch.float64, device=dev)
    s = torch.zeros(len(z0), dtype=torch.int32, device=dev)
    ctrl = capi.CtrlHandle(h, cj, **kw)
    for k in range(1, 4):
        capi.rollout_dev(h, ctrl, len(z0), 1, k, z.data_ptr(), lam.data_ptr(), 0, 0, 0, zn.data_ptr(), s.data_ptr(), 0)
        z, zn = zn, z
    torch.cuda.synchronize()
    assert np.array_equal(z.cpu().numpy(), traj[:, 3])
    if nb <= 41:        # linearsystem on the big tree (its LDS-resident kernel holds ~3 KB per link: up to ~50 links with sibling blocks)
        A, Bu, Bl, G = capi.linearize(h, z0[:1], cj, np.zeros((1, len(cj))))
        Ao, Buo, Blo, Go = orc.linearize(t, z0[0], cj, np.zeros(len(cj)))
        for x, y in ((A[0], Ao), (Bu[0], Buo), (Bl[0], Blo), (G[0], Go)):
            assert np.abs(x - y).max() < 1e-8 * max(1.0, np.abs(y).max())


def test_tree_launch_geometry(cclqr):
    """two instances of the 14-body tree per wavefront in an image that lets four workgroups share a CU's 160 KB (the LDS-resident kernel of
    rounds 1-3: one instance per wavefront); four instances of the dual-pole cart per wavefront"""
    capi = cclqr._capi
    ex = build(cclqr, "deep")
    lanes, lds = capi.MechHandle(ex["mech"].tables()).geometry()
    assert lanes == 32 and 4 * lds <= 160 * 1024
    lanes, lds = capi.MechHandle(build(cclqr, "dual_cartpole")["mech"].tables()).geometry()
    assert lanes == 16


def test_whole_sawyer_robot_lqr_pipeline(cclqr, orc):
    """examples_files/sawyer.urdf (fixed joints lumped: eight bodies, head and arm both on the first link) through the mirror like
    examples/lqr_sawyer.jl does for the arm alone: URDF numbers -> Mechanism -> LQR (mx = 96, mu = 8, ml = 40: tree linearisation + Riccati on
    the device) -> batched simulate! on the register-resident tree kernel; gains and trajectories against the oracle"""
    import json
    import os
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sawyer_full_tables.json")))
    ex = cclqr.examples.sawyer(tab)
    mech = ex["mech"]
    t = mech.tables()
    assert t.nb == 8 and sum(1 for a in t.parent if a == 0) == 2
    ids = [cclqr.getid(b) for b in mech.bodies]
    eids = [cclqr.getid(e) for e in mech.eqconstraints]
    lqr = cclqr.LQR(mech, ids, eids, ex["Q"], ex["R"], 2.0, xd=ex["xd"], qd=ex["qd"])
    assert lqr.K.shape == (199, 8, 96)
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    Ao, Buo, Blo, Go = orc.linearize(t, lqr.zd[0], list(range(8)), np.zeros(8))
    for a, b in zip((lqr.A, lqr.Bu, lqr.Bλ, lqr.G), (Ao, Buo, Blo, Go)):
        assert rel(a, b) < 1e-10
    Ko, kbo = orc.riccati(Ao, Buo, Blo, Go, lqr.Q, lqr.R, 200)
    assert lqr.kbreak == kbo and rel(lqr.K, Ko) < 1e-7
    rng = np.random.default_rng(1)
    z0 = []
    for n in range(9):
        for e in mech.eqconstraints:
            cclqr.setJointPosition(mech, e, rng.uniform(-0.02, 0.02))
        z0.append(mech.state())
    z0 = np.stack(z0)
    st = cclqr.simulate(mech, 1.5, lqr, z0=z0)
    oc = orc.ctrl_desc(8, list(range(8)), K=lqr.K, N=lqr.N, zd=lqr.zd)
    _, traj, sto = orc.rollout(t, oc, z0, 150, record=True)
    assert (sto > 0).all() and np.array_equal(st.status > 0, sto > 0)
    assert np.abs(st.z - traj).max() < 1e-9


@pytest.mark.parametrize("name", ["dual_cartpole", "deep"])
def test_tree_spread_and_packed_launches_agree_bitwise(cclqr, name):
    """the tree kernel spreads a small batch over more wavefronts by the chain kernels' rule (rollout_chain.hip::spread_instances_per_wavefront);
    CCLQR_ROLLOUT_PACK_WAVEFRONTS packs them as a device-filling batch is: bitwise the same trajectories and Newton counts"""
    capi = cclqr._capi
    ex = build(cclqr, name)
    t = ex["mech"].tables()
    rng = np.random.default_rng(9)
    z0 = _starts(cclqr, ex, rng, 7)
    steps, cj = 50, [0, t.ne - 1]
    h = capi.MechHandle(t)
    full = 64 // h.geometry()[0]
    assert h.instances_per_wavefront(7, steps) == 1 and h.instances_per_wavefront(7, steps, capi.ROLLOUT_PACK_WAVEFRONTS) == full
    assert h.instances_per_wavefront(10 ** 6, steps) == full
    ctrl = capi.CtrlHandle(h, cj, K=rng.normal(size=(steps + 5, 2, 12 * t.nb)) * 0.05, N=steps + 6, zd=z0[0], Fd=rng.normal(size=(1, 2)) * 0.3)
    a = capi.rollout(h, ctrl, z0, steps, record=True)
    b = capi.rollout(h, ctrl, z0, steps, record=True, flags=capi.ROLLOUT_PACK_WAVEFRONTS)
    assert (a[2] > 0).all()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
