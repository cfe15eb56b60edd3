"""diagnostic (not a test): per-phase cycle shares of the rollout kernel from the -DCCLQR_PROFILE build"""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libcclqr_prof.so")
n_links = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ninst = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
import bench
ex, mech, zd, z0 = bench.build_workload(pkg, n_links, ninst, 0, 0)
t = mech.tables()
rng = np.random.default_rng(0)
gold = os.path.join(g.ROOT, "tests", "golden", "chain16_hanging_cfg3.npz")
K = np.tile(np.load(gold)["K_first"][None], (999, 1, 1)) if n_links == 16 else rng.normal(size=(999, 1, 12 * t.nb)) * 0.05
mh = capi.MechHandle(t); ctrl = capi.CtrlHandle(mh, [0], K=K, N=1000, zd=zd)
names = ["control", "forces+knotjac", "eval_body", "eval_joint", "eval_map+norm", "schur_w", "schur_s", "tri_fwd", "tri_bwd", "body_solve", "trial", "accept", "io"]
_nchild = np.bincount(np.asarray(t.parent)[np.asarray(t.parent) >= 0], minlength=t.nb)
read = capi.lib().cclqr_prof_read if (_nchild > 1).any() else capi.lib().cclqr_prof_read_chain
buf = (C.c_ulonglong * 16)()
read(buf, 1)
t0 = time.time(); zT, _, st = capi.rollout(mh, ctrl, z0, steps); dt = time.time() - t0
read(buf, 1)
v = np.array(list(buf), dtype=np.float64)
tot = v[:13].sum()
print("n_links %d inst %d steps %d: %.3fs %s; newton iters/step %.2f evals/step %.2f" % (n_links, ninst, steps, dt, capi.rate_or_refusal(ninst * steps, dt, st), v[13] / v[15], v[14] / v[15]))
for i, n in enumerate(names):
    print("  %-16s %6.2f%%  %9.0f cycles/step" % (n, 100 * v[i] / tot, v[i] / v[15]))
print("  total cycles/step (per wave) %.0f" % (tot / v[15]))
