"""diagnostic (not a test): section shares of the LDS-resident Riccati kernel from the -DCCLQR_PROFILE build (problems with mx <= ~96)"""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libcclqr_prof.so")
name = sys.argv[1] if len(sys.argv) > 1 else "cartpole_cfg2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nprob = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if name == "sawyer":
    import json
    tab = json.load(open(os.path.join(g.ROOT, "tests", "golden", "sawyer_arm_tables.json")))
    ex = pkg.examples.sawyer(tab); t = ex["mech"].tables()
    A, Bu, Bl, G = (m[0] for m in capi.linearize(capi.MechHandle(t), ex["mech"].state()[None], list(range(7)), np.zeros((1, 7))))
else:
    gd = np.load(os.path.join(g.ROOT, "tests", "golden", name + ".npz"))
    A, Bu, Bl, G = gd["A"], gd["Bu"], gd["Bl"], gd["G"]
mx, mu, ml = A.shape[0], Bu.shape[1], Bl.shape[1]
Q = np.eye(mx) * 0.01; R = np.eye(mu) * 0.01
rep = lambda M: np.tile(M[None], (nprob, 1, 1))
buf = (C.c_ulonglong * 24)()
capi.riccati(rep(A), rep(Bu), rep(Bl), rep(G), Q, R, 3)   # warm
capi.lib().cclqr_ric_prof_read(buf, 1)
t0 = time.time(); K, kb = capi.riccati(rep(A), rep(Bu), rep(Bl), rep(G), Q, R, N); dt = time.time() - t0
capi.lib().cclqr_ric_prof_read(buf, 1)
v = np.array(list(buf), dtype=np.float64); steps = max(v[5], 1); tot = v[:5].sum()
m = mu + ml
F = 4 * mx**3 + 4 * mx**2 * m + 2 * mx * (ml**2 + m**2) + 2 / 3 * m**3 + 2 / 3 * ml**3
kbs = np.atleast_1d(kb); done = (N - np.maximum(kbs, 1) + 1).sum()
print("%s mx=%d mu=%d ml=%d nprob=%d N=%d: %.3fs total, %.3f ms/backward-step/problem-wave, kbreak %s, %.1f GFLOP/s (F_ric=%.3g)" % (name, mx, mu, ml, nprob, N, dt, 1e3 * dt / max(1, N - kbs.min()), kbs[:3], F * done / dt / 1e9, F))
for i, n in enumerate(["W = P [A'|D]", "D'W, mu x mu solve, Ku", "Abar, P Abar updates", "prefetch, norm, barriers", "Pkp1 tiles of wavefront 0"]):
    print("  %-24s %6.2f%%  %9.0f cycles/step" % (n, 100 * v[i] / tot, v[i] / steps))
if v.size >= 24:
    print("  per wavefront, cycles per step inside its own tiles:  W " + " ".join("%5.0f" % (x / steps) for x in v[8:16]) + "   Pkp1 " + " ".join("%5.0f" % (x / steps) for x in v[16:24]))
