"""diagnostic (not a test): persistent-workgroup (1) vs tiled (2) Riccati path, wall time of cclqr_riccati incl. upload/download"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
sys.path.insert(0, os.path.join(g.ROOT, "oracle"))


def timed(f, reps=2):
    f()
    best = 1e9
    for _ in range(reps):
        t0 = time.time(); out = f(); best = min(best, time.time() - t0)
    return best, out


def case(name, A, Bu, Bl, G, Q, R, N, tol=1e-5):
    res = {}
    for path in (1, 2):
        dt, (K, kb) = timed(lambda: capi.riccati(A, Bu, Bl, G, Q, R, N, tol=tol, path=path))
        res[path] = (dt, K, np.atleast_1d(kb))
    steps = (N - np.maximum(res[1][2], 1)).max() + 1
    d = np.abs(res[1][1] - res[2][1]).max() / max(1.0, np.abs(res[1][1]).max())
    print("%-34s steps %4d: resident %.4fs (%.1f us/step)  tiled %.4fs (%.1f us/step)  speedup %.2fx  |dK|rel %.1e kb %s" % (
        name, steps, res[1][0], 1e6 * res[1][0] / steps, res[2][0], 1e6 * res[2][0] / steps, res[1][0] / res[2][0], d,
        "same" if np.array_equal(res[1][2], res[2][2]) else "DIFFER"))


gd = np.load(os.path.join(g.ROOT, "tests", "golden", "chain16_hanging_cfg3.npz"))
case("chain16 mx=204 ml=85 N=1000", gd["A"], gd["Bu"], gd["Bl"], gd["G"], np.eye(204) * 0.01, np.eye(1) * 0.01, 1000, tol=0.0)
import orc
for n in (1, 3, 7):
    ex = pkg.examples.cartpole_n(n); t = ex["mech"].tables()
    zd = np.zeros((n + 1, 13)); zd[:, 3] = 1.0
    zd = pkg.examples.cartpole_states(n, [0.0], np.zeros((1, n)))[0]
    A, Bu, Bl, G = orc.linearize(t, zd, [0], np.zeros(1))
    case("cartpole_n(%d) mx=%d N=1000" % (n, 12 * (n + 1)), A, Bu, Bl, G, np.eye(12 * (n + 1)) * 0.01, np.eye(1) * 0.01, 1000, tol=0.0)
tab = json.load(open(os.path.join(g.ROOT, "tests", "golden", "sawyer_arm_tables.json")))
ex = pkg.examples.sawyer(tab); mech = ex["mech"]; t = mech.tables()
rng = np.random.default_rng(0)
for nprob in (1, 64, 1024):
    zs = []
    for n in range(nprob):
        for e in mech.eqconstraints:
            pkg.setJointPosition(mech, e, rng.uniform(-0.05, 0.05))
        zs.append(mech.state())
    mh = capi.MechHandle(t)
    A, Bu, Bl, G = capi.linearize(mh, np.stack(zs), list(range(7)), np.zeros((nprob, 7)))
    case("sawyer mx=84 mu=7 nprob=%d N=200" % nprob, A, Bu, Bl, G, np.eye(84) * 1000 * t.dt, np.eye(7) * t.dt, 200, tol=0.0)
# tracking (time-varying): triple cartpole swing-up reference
gt = np.load(os.path.join(g.ROOT, "tests", "golden", "triple_tracking_cfg5.npz"))
print(list(gt.keys()))
