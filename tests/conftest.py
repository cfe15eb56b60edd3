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
    srcs = [os.path.join(d, "emu_rollout.cpp"), os.path.join(d, "emu_chain.cpp"), os.path.join(d, "emu_loop.cpp")]
    csrc = os.path.join(ROOT, "constrainedcontrol.jl_amd", "csrc")
    deps = srcs + [os.path.join(csrc, h) for h in ("cclqr_dev.h", "cclqr_chain.h", "cclqr_loop.h", "cclqr_lin_dev.h", "cclqr_tables.h", "cclqr_internal.h")]
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
