"""diagnostic (not a test): per-phase cycle shares of the register-resident tree kernel (rollout_treereg.hip) from the -DCCLQR_PROFILE build"""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libcclqr_prof.so")
parents = [-1, 0, 1, 2, 3, 2, 5, 6, 1, 8, 8, 10, 0, 12]
ex = pkg.examples.tree_mechanism(parents, seed=4, prismatic=(0, 5))
t, z0 = ex["mech"].tables(), ex["mech"].state()
n, steps = 8192, 300
rng = np.random.default_rng(0)
K = rng.normal(size=(steps + 5, 1, 12 * t.nb)) * 0.02
mh = capi.MechHandle(t); ctrl = capi.CtrlHandle(mh, [0], K=K, N=steps + 6, zd=z0)
zz = np.tile(z0[None], (n, 1, 1))
names = ["control", "forces+knotjac", "eval_body", "eval_joint", "eval_map+norm", "schur_w", "schur_s", "tri_fwd", "tri_bwd", "body_solve", "trial", "accept", "io"]
read = capi.lib().cclqr_prof_read_treereg
buf = (C.c_ulonglong * 16)()
capi.rollout(mh, ctrl, zz[:64], 10)
read(buf, 1)
t0 = time.time(); zT, _, st = capi.rollout(mh, ctrl, zz, steps); dt = time.time() - t0
read(buf, 1)
v = np.array(list(buf), dtype=np.float64)
tot = v[:13].sum()
print("14-body tree, %d inst x %d steps: %.3fs %s; newton iters/step %.2f evals/step %.2f" % (n, steps, dt, capi.rate_or_refusal(n * steps, dt, st), v[13] / v[15], v[14] / v[15]))
for i, nm in enumerate(names):
    print("  %-16s %6.2f%%  %9.0f cycles/step" % (nm, 100 * v[i] / tot, v[i] / v[15]))
print("  total cycles/step (per wave) %.0f" % (tot / v[15]))
