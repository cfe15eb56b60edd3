#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/).  No reference implementation can run in this pipeline
(Julia is absent, ConstrainedDynamics is not vendored), so these vectors pin OUR restatement against drift; the one
piece of reference-held data, the 1000-sample open-loop input `U` of examples/trackingLQR_triple_cartpole.jl:1, is stored
as triple_cartpole_U.npy (extracted as numbers by this script when /root/reference is present) and is used as an INPUT.

    python tests/golden/make_golden.py
"""
import os
import re
import sys

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
from oracle import orc  # noqa: E402

pkg = graft.load_package()
KS = [0, 1, 9, 99, 999]   # storage indices of steps k = 1, 2, 10, 100, 1000 (SURVEY 8c "what to capture")


def extract_U():
    ref = "/root/reference/examples/trackingLQR_triple_cartpole.jl"
    out = os.path.join(HERE, "triple_cartpole_U.npy")
    if os.path.exists(ref):
        line = open(ref).readline()
        U = np.array([float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", line[line.index("["):])])
        assert len(U) == 1000
        np.save(out, U)
    return np.load(out)


def lqr_case(ex, zd, cj, N, z0, name, ks):
    t = ex["mech"].tables()
    A, Bu, Bl, G = orc.linearize(t, zd, cj, np.zeros(len(cj)))
    Q = sl.block_diag(*ex["Q"]) * t.dt
    R = sl.block_diag(*ex["R"]) * t.dt
    Ntemp = N if N > 0 else 1000
    K, kb = orc.riccati(A, Bu, Bl, G, Q, R, Ntemp)
    Kc = K[:1] if N <= 0 else K
    ctrl = orc.ctrl_desc(t.nb, cj, K=Kc, N=N, zd=zd)
    steps = max(ks) + 1
    zT, traj, st = orc.rollout(t, ctrl, z0, steps, record=True)
    assert (st > 0).all()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), A=A, Bu=Bu, Bl=Bl, G=G, K_first=K[0], K_last=K[-1], kbreak=kb, zd=zd, z0=z0,
                        ks=np.array(ks), traj_samples=traj[:, ks], zT=zT, N=N)
    print(name, "kbreak", kb, "|K|", np.abs(K).max())


def main():
    rng = np.random.default_rng(0xC0FFEE)
    # cfg1 pendulum (lqr_pendulum.jl): horizon Inf
    ex = pkg.examples.pendulum()
    zd = np.zeros((1, 13))
    zd[0, 0:3], zd[0, 3:7] = ex["xd"][0], ex["qd"][0]
    lqr_case(ex, zd, [0], 0, ex["mech"].state()[None], "pendulum_cfg1", KS)
    # cfg2 cartpole (lqr_cartpole.jl): nominal instance + 2 random-init ones
    ex = pkg.examples.cartpole_n(1)
    zd = np.zeros((2, 13))
    zd[:, 3] = 1
    zd[1, 2] = 0.5
    z0 = np.concatenate([ex["mech"].state()[None], pkg.examples.cartpole_states(1, rng.uniform(-0.5, 0.5, 2), rng.uniform(0, 1 / 3, (2, 1)))])
    lqr_case(ex, zd, [0], 1000, z0, "cartpole_cfg2", KS)
    # cfg3 mechanism (N = 16 links) about the hanging equilibrium, short rollout
    n = 16
    ex = pkg.examples.cartpole_n(n)
    zd = pkg.examples.cartpole_states(n, [0.0], np.array([[np.pi] + [0.0] * (n - 1)]))[0]
    phi = rng.uniform(-0.2, 0.2, (2, n))
    phi[:, 0] += np.pi
    z0 = pkg.examples.cartpole_states(n, rng.uniform(-0.5, 0.5, 2), phi)
    lqr_case(ex, zd, [0], 1000, z0, "chain16_hanging_cfg3", [0, 1, 9, 99])
    # cfg5 triple cartpole: open-loop replay of the reference's U, then TrackingLQR gains about that trajectory
    U = extract_U()
    ex = pkg.examples.triple_cartpole()
    t = ex["mech"].tables()
    z00 = ex["mech"].state()
    N = 1000
    ol = orc.ctrl_desc(4, [0], K=None, N=N + 1, zd=np.tile(z00, (N, 1, 1)), Fd=U.reshape(N, 1))
    zT, traj, st = orc.rollout(t, ol, z00[None], N, record=True)
    Q = sl.block_diag(*ex["Q"]) * t.dt
    R = sl.block_diag(*ex["R"]) * t.dt
    K, kb = orc.riccati_tracking(t, [0], traj[0], U.reshape(N, 1), Q, R, N)
    np.savez_compressed(os.path.join(HERE, "triple_tracking_cfg5.npz"), open_loop_samples=traj[0, KS], open_loop_zT=zT[0], ks=np.array(KS),
                        K_samples=K[[0, 499, 997, 998]], K_idx=np.array([0, 499, 997, 998]), kbreak=kb, Kabsmax=np.abs(K).max())
    print("triple tracking kbreak", kb, "|K|", np.abs(K).max(), "swing-up final angles (deg)",
          np.degrees(2 * np.arctan2(zT[0, 1:, 4], zT[0, 1:, 3])))


if __name__ == "__main__":
    main()
