"""diagnostic (not a test): batched Riccati throughput at config 4's shape (Sawyer: mx = 84, mu = 7, ml = 35), distinct setpoints"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tab = json.load(open(os.path.join(g.ROOT, "tests", "golden", "sawyer_arm_tables.json")))
ex = pkg.examples.sawyer(tab); mech = ex["mech"]; t = mech.tables()
rng = np.random.default_rng(0)
zs = []
for n in range(nprob):
    for e in mech.eqconstraints:
        pkg.setJointPosition(mech, e, rng.uniform(-0.05, 0.05))
    zs.append(mech.state())
zs = np.stack(zs)
mh = capi.MechHandle(t)
t0 = time.time(); A, Bu, Bl, G = capi.linearize(mh, zs, list(range(7)), np.zeros((nprob, 7))); tl = time.time() - t0
Q = np.eye(84) * 1000 * t.dt; R = np.eye(7) * t.dt
capi.riccati(A[:2], Bu[:2], Bl[:2], G[:2], Q, R, 3)
t0 = time.time(); K, kb = capi.riccati(A, Bu, Bl, G, Q, R, N); tr = time.time() - t0
mx, mu, ml = 84, 7, 35; m = mu + ml
F = 4 * mx**3 + 4 * mx**2 * m + 2 * mx * (ml**2 + m**2) + 2 / 3 * m**3 + 2 / 3 * ml**3
steps = (N - np.maximum(kb, 1) + 1).sum()
print("linearize %d setpoints: %.3fs (%.0f/s)" % (nprob, tl, nprob / tl))
print("riccati nprob=%d N=%d: %.3fs -> %.0f backward steps/s, %.1f gains(problems)/s, %.2f TFLOP/s (F_ric=%.3g), kbreak min %d max %d" % (
    nprob, N, tr, steps / tr, nprob / tr, F * steps / tr / 1e12, F, kb.min(), kb.max()))
