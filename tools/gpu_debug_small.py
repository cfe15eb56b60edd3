"""bring-up helper (not a test): small rollout on the GPU vs the oracle, prints where they differ"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
from oracle import orc
n_links = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ninst = int(sys.argv[2]) if len(sys.argv) > 2 else 5
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ex = pkg.examples.cartpole_n(n_links); t = ex["mech"].tables()
zd = np.zeros((n_links + 1, 13)); zd[:, 3] = 1
for i in range(1, n_links + 1): zd[i, 2] = i - 0.5
rng = np.random.default_rng(0)
K = rng.normal(size=(steps + 5, 1, 12 * t.nb)) * 0.1
z0 = pkg.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, ninst), rng.uniform(-0.1, 0.1, (ninst, n_links)))
mech = capi.MechHandle(t); ctrl = capi.CtrlHandle(mech, [0], K=K, N=steps + 6, zd=zd)
zT, traj, st = capi.rollout(mech, ctrl, z0, steps, record=True)
zo, trajo, sto = orc.rollout(t, orc.ctrl_desc(t.nb, [0], K=K, N=steps + 6, zd=zd), z0, steps, record=True)
print("status", st, sto)
d = np.abs(traj - trajo)
print("traj max diff", d.max(), "zT max diff", np.abs(zT - zo).max())
np.set_printoptions(linewidth=200, precision=3)
for n in range(ninst):
    for k in range(steps):
        if d[n, k].max() > 1e-9:
            print("inst", n, "step", k, "\n gpu", traj[n, k].ravel(), "\n orc", trajo[n, k].ravel())
            break
print("zT gpu\n", zT.reshape(ninst, -1), "\nzT orc\n", zo.reshape(ninst, -1))
np.set_printoptions(linewidth=220, precision=9)
print("---- instance 0, all steps, body 1 (gpu / orc)")
for k in range(steps):
    print(k, traj[0, k, -1]); print(k, trajo[0, k, -1])
print("---- instance 0, all steps, body 0 (gpu / orc)")
for k in range(steps):
    print(k, traj[0, k, 0]); print(k, trajo[0, k, 0])
