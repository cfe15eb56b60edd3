import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libcclqr_prof.so")
import torch, bench
mech, lq, z0, _, _ = bench.sawyer_cfg4_workload(pkg, 0.002, 8192)
mh = mech._cclqr_handle; ctrl = lq._ctrl_handle(mh)
names = ["control", "forces+knotjac", "eval_body", "eval_joint", "eval_map+norm", "schur_w", "schur_s", "tri_fwd", "tri_bwd", "body_solve", "trial", "accept", "io"]
buf = (C.c_ulonglong * 16)()
read = capi.lib().cclqr_prof_read_chain
read(buf, 1)
zT, _, st = capi.rollout(mh, ctrl, z0, 400)
read(buf, 1)
v = np.array(list(buf), dtype=np.float64); tot = v[:13].sum()
print("sawyer cfg4: newton iters/step %.2f evals/step %.2f failed %d" % (v[13] / v[15], v[14] / v[15], int((st <= 0).sum())))
for i, n in enumerate(names): print("  %-16s %6.2f%%  %9.0f cycles/step" % (n, 100 * v[i] / tot, v[i] / v[15]))
print("  total %.0f" % (tot / v[15]))
