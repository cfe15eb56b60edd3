"""GPU: no kernel's result depends on what the previous kernel left in LDS.

LDS is not cleared between kernels.  Since round 5 the chain rollout's prologue initialises the multiplier block only (the clear of the whole image
was 2.3 of a single-step launch's 6.5 us), on the argument that no phase reads an LDS word before writing it -- shown by the CPU emulator on an image
of signalling NaNs (tests/emu).  This is the same statement on the hardware, for every kernel family: tests/gpu/poison_lds.hip fills the LDS of every
compute unit with a pattern (a signalling NaN; 1e300, which survives a multiplication by a mask where NaN would also be caught), then the kernel
runs; outputs must be BITWISE those of a run without poisoning."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SNAN = 0x7FF4000000000001
HUGE = int(np.array([1e300]).view(np.uint64)[0])


@pytest.fixture(scope="module")
def poison(tmp_path_factory):
    import torch
    torch.zeros(1, device="cuda")      # torch's HIP runtime first: a library loaded later binds to the runtime already in the process (the order every other GPU test has)
    so = str(tmp_path_factory.mktemp("poison") / "libpoison_lds.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(HERE, "gpu", "poison_lds.hip"), "-o", so])
    lib = C.CDLL(so)
    lib.poison_lds.argtypes = [C.c_ulonglong, C.c_int, C.c_int]

    lib.peek_lds.argtypes = [C.c_int, C.c_int, C.c_void_p]

    def run(pattern):
        rc = lib.poison_lds(pattern, 160 * 1024, 8 * 256)          # the whole 160 KB of a unit per workgroup, eight workgroups per unit
        assert rc == 0, "poison_lds failed: %d" % rc

    def peek(bytes_per_workgroup, workgroups):
        out = np.zeros((workgroups, 256), dtype=np.uint64)
        rc = lib.peek_lds(bytes_per_workgroup, workgroups, out.ctypes.data)
        assert rc == 0, "peek_lds failed: %d" % rc
        return out
    run.peek = peek
    return run


def test_the_poison_reaches_every_compute_unit(poison):
    """the harness itself: after poisoning, a kernel that reads LDS without writing it sees the pattern -- in every workgroup of a grid that covers
    the device four times over with the chain kernels' allocation size (38.4 KB, four workgroups per unit) and with the 64-lane kernels' 76.8 KB"""
    for pattern in (SNAN, HUGE):
        for size in (38464, 76800, 160 * 1024):
            poison(pattern)
            seen = poison.peek(size, 4 * 1024)
            assert (seen == np.uint64(pattern)).all(), (hex(pattern), size, float((seen == np.uint64(pattern)).mean()))


def _chain_case(cclqr, n_links, ninst, steps, extra=False):
    capi = cclqr._capi
    rng = np.random.default_rng(100 + n_links)
    if extra:
        ex = cclqr.examples.triple_cartpole()
        z0 = np.tile(ex["mech"].state(), (ninst, 1, 1))
        z0[:, :, 7:] += rng.normal(size=(ninst, 4, 6)) * 0.0     # (consistent start: at rest)
    else:
        ex = cclqr.examples.cartpole_n(n_links)
        phi = rng.uniform(-0.3, 0.3, (ninst, n_links))
        phi[:, 0] += np.pi
        z0 = cclqr.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, ninst), phi)
    t = ex["mech"].tables()
    mech = capi.MechHandle(t)
    K = rng.normal(size=(steps + 5, 1, 12 * t.nb)) * 0.05
    kw = dict(K=K, N=steps + 6, zd=z0[0], Fd=np.array([[0.3]]))
    if extra:
        kw.update(fric=ex["fric"], noise_scale=0.5, noise_seed=77)
    ctrl = capi.CtrlHandle(mech, [0], **kw)

    def run(prep):
        prep()
        zT, traj, st = capi.rollout(mech, ctrl, z0, steps, record=True)
        assert (st > 0).all()
        # the same horizon as single-step launches (multipliers and state round-trip HBM: the k0 > 1 prologue)
        import torch
        dev = torch.device("cuda", 0)
        z = torch.from_numpy(z0).to(dev)
        zn = torch.empty_like(z)
        lam = torch.zeros((ninst, 5 * t.ne), dtype=torch.float64, device=dev)
        s = torch.zeros(ninst, dtype=torch.int32, device=dev)
        for k in range(1, 6):
            prep()
            capi.rollout_dev(mech, ctrl, ninst, 1, k, z.data_ptr(), lam.data_ptr(), 0, 0, 0, zn.data_ptr(), s.data_ptr(), 0)
            z, zn = zn, z
        torch.cuda.synchronize()
        return [zT, traj, st, z.cpu().numpy(), lam.cpu().numpy()]
    return run


def _tree_case(cclqr, name, nb=0):
    from test_tree import build, _random_parents
    capi = cclqr._capi
    if nb:          # a random forest of nb bodies (33 .. 64: the tree kernel's 64-lane instantiations)
        r = np.random.default_rng(nb)
        ex = cclqr.examples.tree_mechanism(_random_parents(r, nb), seed=nb, prismatic=(0, 5))
        ex["joints"] = ex["mech"].eqconstraints
    else:
        ex = build(cclqr, name)
    mech_py = ex["mech"]
    t = mech_py.tables()
    rng = np.random.default_rng(3)
    z0 = []
    for n in range(5):
        for e in ex["joints"]:
            cclqr.setJointPosition(mech_py, e, rng.uniform(-0.5, 0.5))
        z0.append(mech_py.state())
    z0 = np.stack(z0)
    steps, cj = 40, [0, t.ne - 1]
    K = rng.normal(size=(steps + 5, 2, 12 * t.nb)) * 0.05
    h = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(h, cj, K=K, N=steps + 6, zd=z0[0], Fd=rng.normal(size=(1, 2)) * 0.3)

    def run(prep):
        prep()
        zT, traj, st = capi.rollout(h, ctrl, z0, steps, record=True)
        assert (st > 0).all()
        prep()
        if nb:
            return [zT, traj, st]
        A, Bu, Bl, G = capi.linearize(h, z0[:2], cj, np.zeros((2, 2)))
        return [zT, traj, st, A, Bu, Bl, G]
    return run


def _loop_case(cclqr):
    capi = cclqr._capi
    ex = cclqr.examples.deltabot()
    mech_py = ex["mech"]
    t = mech_py.tables()
    cj = [mech_py.joint_index(e) for e in ex["eqcids"]]
    z0 = mech_py.state()
    mech = capi.MechHandle(t)
    rng = np.random.default_rng(5)
    n = 4
    K = rng.normal(size=(n, 1, 2, 12 * t.nb)) * 2.0
    Fd = np.linspace(0.8, 0.6, n)[:, None, None] * ex["Fd"].reshape(1, 1, 2)
    ctrl = capi.CtrlHandle(mech, cj, K=K, N=0, zd=np.repeat(z0[None, None], n, 0), Fd=Fd, n_ctrl=n)

    def run(prep):
        prep()
        zT, traj, st = capi.rollout(mech, ctrl, np.repeat(z0[None], n, 0), 25, record=True)
        assert (st > 0).all()
        prep()
        Ap, D = capi.linearize_projected(mech, z0[None], cj, ex["Fd"].reshape(1, 2))
        return [zT, traj, st, Ap, D]
    return run


def _riccati_case(cclqr):
    capi = cclqr._capi
    ex = cclqr.examples.cartpole_n(6)
    t = ex["mech"].tables()
    mech = capi.MechHandle(t)
    zd = ex["mech"].state()
    A, Bu, Bl, G = capi.linearize(mech, zd[None], [0], np.zeros((1, 1)))
    mx = 12 * t.nb

    def run(prep):
        out = []
        for path in (1, 2):         # resident (P and W in LDS) and tiled
            prep()
            K, kb = capi.riccati(np.repeat(A, 3, 0), np.repeat(Bu, 3, 0), np.repeat(Bl, 3, 0), np.repeat(G, 3, 0), np.eye(mx) * t.dt, np.eye(1) * t.dt, 60, path=path)
            out += [K, np.asarray(kb)]
        return out
    return run


CASES = {
    "cartpole (8 lanes, three lanes per link)": lambda c: _chain_case(c, 1, 37, 60),
    "triple cartpole, friction + Philox noise (8 lanes)": lambda c: _chain_case(c, 3, 19, 60, extra=True),
    "7 links (16 lanes)": lambda c: _chain_case(c, 7, 9, 50),
    "16 links (32 lanes, 17-link image, reduction level)": lambda c: _chain_case(c, 16, 5, 40),
    "22 links (32 lanes, 32-link image)": lambda c: _chain_case(c, 22, 3, 30),
    "40 links (64 lanes)": lambda c: _chain_case(c, 40, 3, 20),
    "tree: dual cartpole": lambda c: _tree_case(c, "dual_cartpole"),
    "tree: deep": lambda c: _tree_case(c, "deep"),
    "tree: 40 bodies (64 lanes)": lambda c: _tree_case(c, None, nb=40),
    "closed loops: deltabot": _loop_case,
    "riccati resident + tiled": _riccati_case,
}


@pytest.mark.parametrize("name", list(CASES))
def test_results_do_not_depend_on_what_lds_held(cclqr, poison, name):
    run = CASES[name](cclqr)
    ref = run(lambda: None)
    for pattern in (SNAN, HUGE):
        got = run(lambda: poison(pattern))       # (before every launch: a kernel leaves its own image behind for the next)
        for a, b in zip(ref, got):
            assert np.asarray(a).tobytes() == np.asarray(b).tobytes(), (name, hex(pattern), float(np.nanmax(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)))))
