"""diagnostic (not a test): examples/lqr_deltabot.jl as a batch -- the script's LQR (its Q, R, infinite horizon, holding inputs) driving ALL
valid initial conditions of its grid (the script simulates one, i = 97) for 10 s on the closed-loop rollout kernel"""
import sys, os, time, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
ex = pkg.examples.deltabot(); mech = ex["mech"]
z00 = mech.state()
ids = [pkg.getid(b) for b in mech.bodies]
t0 = time.time()
lq = pkg.LQR(mech, ids, ex["eqcids"], ex["Q"], ex["R"], math.inf, xd=[z00[i, 0:3] for i in range(5)], qd=[z00[i, 3:7] for i in range(5)],
             Fτd=[[ex["Fd"][0]], [ex["Fd"][1]]])
print("LQR (projected model + recursion): %.3f s, K %s, kbreak %d, converged %s, |K|max %.3g" % (time.time() - t0, lq.K.shape, lq.kbreak, lq.converged, np.abs(lq.K).max()))
z0, yz = pkg.examples.deltabot_initial_states(ex)
t0 = time.time()
st = pkg.simulate(mech, 10.0, lq, record=False, z0=z0)
dt = time.time() - t0
ok = st.status > 0
dev = np.abs(st.zT[:, 4, 1:3] - z00[4, 1:3]).max(axis=1)
home = ok & (dev < 1e-2)
print("%d initial conditions x %d steps: %.2f s; converged Newton on %d, platform back within 1 cm of the setpoint on %d" % (len(z0), st.steps, dt, ok.sum(), home.sum()))
i = 96
print("the script's case i = 97 (platform at y %.2f z %.2f): status %d, final platform offset %.2e" % (yz[i, 0], yz[i, 1], st.status[i], dev[i]))
rad = np.hypot(yz[:, 0] - z00[4, 1], yz[:, 1] - z00[4, 2])
for r0, r1 in ((0, 0.1), (0.1, 0.2), (0.2, 0.4), (0.4, 0.8), (0.8, 2.0)):
    m = (rad >= r0) & (rad < r1)
    if m.any():
        print("  start %.1f-%.1f m from the setpoint: %4d conditions, %4d brought home" % (r0, r1, m.sum(), home[m].sum()))
