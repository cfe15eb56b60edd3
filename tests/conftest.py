import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """a GPU test that does not come back within three minutes is a hang, not a slow test.  pytest-timeout's "thread" method cannot
    interrupt a thread that sits in a HIP call: it dumps the stacks and ENDS THE WHOLE SESSION with os._exit -- the run reports the
    hanging test and stops there (later tests do not run), instead of blocking the box until gpurun's own limit."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(180, method="thread"))


@pytest.fixture(scope="session")
def cclqr():
    return graft.load_package()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.build()
    return o


@pytest.fixture(scope="session")
def emu():
    """CPU emulation of the rollout kernel's phase functions (tests/emu, test infrastructure only)"""
    import ctypes as C
    d = os.path.join(ROOT, "tests", "emu")
    so = os.path.join(d, "libemu.so")
    srcs = [os.path.join(d, "emu_rollout.cpp"), os.path.join(d, "emu_chain.cpp"), os.path.join(d, "emu_loop.cpp"), os.path.join(d, "emu_treereg.cpp")]
    csrc = os.path.join(ROOT, "constrainedcontrol.jl_amd", "csrc")
    deps = srcs + [os.path.join(csrc, h) for h in ("cclqr_dev.h", "cclqr_chain.h", "cclqr_loop.h", "cclqr_lin_loop.h", "cclqr_lin_dev.h", "cclqr_tables.h", "cclqr_internal.h", "cclqr_treereg.h", "cclqr_treereg_tables.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-shared",
                               "-o", so] + srcs)
    return C.CDLL(so)


def hanging_setpoint(cclqr, n):
    return cclqr.examples.cartpole_states(n, [0.0], np.array([[np.pi] + [0.0] * (n - 1)]))[0]


def upright_setpoint(n):
    zd = np.zeros((n + 1, 13))
    zd[:, 3] = 1.0
    for i in range(1, n + 1):
        zd[i, 2] = i - 0.5
    return zd


def free_port():
    """a TCP port nobody is listening on right now (rendezvous of the two-rank tests): a fixed port can still be held by an earlier run"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def long_and_short_chain_forest(cclqr, short_first=False):
    """a 13-link cartpole chain and a 3-link one hanging off the same origin, bodies interleaved in the caller's numbering; returns
    (tables, z0 [1][nb][13], zd [nb][13], K [20][2][12 nb], controlled joints)"""
    ea, eb = cclqr.examples.cartpole_n(12), cclqr.examples.cartpole_n(2)
    ta, tb = ea["mech"].tables(), eb["mech"].tables()
    nb = ta.nb + tb.nb
    order = list(np.random.default_rng(5).permutation(nb))
    ia, ib = order[:ta.nb], order[ta.nb:]
    # the chain whose root joint has the smaller index comes first in the kernels' link order
    if (min(ia[0], ib[0]) == ib[0]) != short_first:
        j0, j1 = ia[0], ib[0]
        ia = [j1 if x == j0 else x for x in ia]
        ib = [j0 if x == j1 else x for x in ib]
    mass, inertia = np.zeros(nb), np.zeros((nb, 9))
    parent, child, typ = np.zeros(nb, dtype=np.int32), np.zeros(nb, dtype=np.int32), np.zeros(nb, dtype=np.int32)
    p1, p2, axis, qoff = np.zeros((nb, 3)), np.zeros((nb, 3)), np.zeros((nb, 3)), np.zeros((nb, 4))
    for ids, tt in ((ia, ta), (ib, tb)):
        for k in range(tt.nb):
            j = ids[k]
            mass[j], inertia[j] = tt.mass[k], tt.inertia[k]
            parent[j] = -1 if tt.parent[k] < 0 else ids[tt.parent[k]]
            child[j], typ[j], p1[j], p2[j], axis[j], qoff[j] = ids[k], tt.type[k], tt.p1[k], tt.p2[k], tt.axis[k], tt.qoff[k]
    t2 = cclqr.MechTables(nb, nb, ta.dt, ta.g, mass, inertia, parent, child, typ, p1, p2, axis, qoff)
    rng = np.random.default_rng(11)
    phi = rng.uniform(-0.2, 0.2, (1, 12)); phi[:, 0] += np.pi
    za = cclqr.examples.cartpole_states(12, [0.1], phi)[0]
    zb = cclqr.examples.cartpole_states(2, [-0.3], [[-0.15, 0.05]])[0]
    z0 = np.zeros((1, nb, 13))
    z0[0, ia], z0[0, ib] = za, zb
    zd = z0[0].copy(); zd[:, 7:] = 0
    K = rng.normal(size=(20, 2, 12 * nb)) * 0.05
    return t2, z0, zd, K, [int(ia[0]), int(ib[0])]
