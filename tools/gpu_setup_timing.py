"""diagnostic (not a test): where the LQR construction time of the headline workload goes (linearise, Riccati sweep, copies)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
import bench
ex, mech, zd, z0 = bench.build_workload(pkg, 16, 8, 0, 0)
t = mech.tables(); nb = t.nb
mh = capi.MechHandle(t)
for rep in range(3):
    t0 = time.time()
    A, Bu, Bl, G = (m[0] for m in capi.linearize(mh, zd[None], [0], np.zeros((1, 1))))
    t1 = time.time()
    K, kb = capi.riccati(A, Bu, Bl, G, np.eye(204) * t.dt, np.eye(1) * t.dt, 1000)
    t2 = time.time()
    lqr = pkg.LQR(mech, [pkg.getid(b) for b in ex["bodies"]], [pkg.getid(ex["ctrl"][0])], ex["Q"], ex["R"], 10.0,
                  xd=[zd[i, 0:3] for i in range(nb)], qd=[zd[i, 3:7] for i in range(nb)])
    t3 = time.time()
    print("rep %d: linearize %.1f ms, riccati (999 steps, mx 204) %.1f ms, whole LQR(...) constructor %.1f ms" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
