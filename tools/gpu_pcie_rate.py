"""diagnostic (not a test): PCIe-inclusive rate of the headline workload through the HOST-pointer entry point cclqr_rollout"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
import bench
ex, mech, zd, z0 = bench.build_workload(pkg, 16, 8192, 0, 0)
t = mech.tables()
K = np.tile(np.load(os.path.join(g.ROOT, "tests", "golden", "chain16_hanging_cfg3.npz"))["K_first"][None], (999, 1, 1))
mh = capi.MechHandle(t); ctrl = capi.CtrlHandle(mh, [0], K=K, N=1000, zd=zd)
capi.rollout(mh, ctrl, z0[:64], 10)
for steps, rec in ((1000, False), (200, True)):
    t0 = time.time(); zT, traj, st = capi.rollout(mh, ctrl, z0, steps, record=rec); dt = time.time() - t0
    gb = (traj.nbytes if rec else 0) / 1e9
    print("host-pointer cclqr_rollout 8192 x %d steps record=%s: %.3f s -> %s (trajectory %.2f GB to pageable host memory)" % (steps, rec, dt, capi.rate_or_refusal(8192 * steps, dt, st), gb))
