"""diagnostic (not a test): one Riccati case for rocprofv3 --kernel-trace --stats.  argv: case path"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
which = sys.argv[1] if len(sys.argv) > 1 else "chain16"
path = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if which == "chain16":
    gd = np.load(os.path.join(g.ROOT, "tests", "golden", "chain16_hanging_cfg3.npz"))
    args = (gd["A"], gd["Bu"], gd["Bl"], gd["G"], np.eye(204) * 0.01, np.eye(1) * 0.01, 1000)
else:
    nprob = int(which)
    tab = json.load(open(os.path.join(g.ROOT, "tests", "golden", "sawyer_arm_tables.json")))
    ex = pkg.examples.sawyer(tab); mech = ex["mech"]; t = mech.tables()
    rng = np.random.default_rng(0); zs = []
    for n in range(nprob):
        for e in mech.eqconstraints:
            pkg.setJointPosition(mech, e, rng.uniform(-0.05, 0.05))
        zs.append(mech.state())
    A, Bu, Bl, G = capi.linearize(capi.MechHandle(t), np.stack(zs), list(range(7)), np.zeros((nprob, 7)))
    args = (A, Bu, Bl, G, np.eye(84) * 1000 * t.dt, np.eye(7) * t.dt, 200)
capi.riccati(*args, tol=0.0, path=path)
t0 = time.time(); capi.riccati(*args, tol=0.0, path=path); print(which, "riccati %.4f s" % (time.time() - t0))
