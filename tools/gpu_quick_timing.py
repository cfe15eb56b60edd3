"""ad-hoc timing helper used during bring-up (not a test)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
n_links = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ninst = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ex = pkg.examples.cartpole_n(n_links); t = ex["mech"].tables()
zd = pkg.examples.cartpole_states(n_links, [0.0], np.array([[np.pi] + [0.0] * (n_links - 1)]))[0]
rng = np.random.default_rng(0)
K = rng.normal(size=(999, 1, 12 * t.nb)) * 0.1
phi = rng.uniform(-0.3, 0.3, (ninst, n_links)); phi[:, 0] += np.pi
z0 = pkg.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, ninst), phi)
mech = capi.MechHandle(t); ctrl = capi.CtrlHandle(mech, [0], K=K, N=1000, zd=zd)
print("geometry", mech.geometry())
for rep in range(3):
    t0 = time.time(); zT, _, st = capi.rollout(mech, ctrl, z0, steps); dt = time.time() - t0
    print("n_links %d inst %d steps %d: %.3fs -> %s (host-pointer API incl. copies); status min %d max %d mean %.2f" % (n_links, ninst, steps, dt, capi.rate_or_refusal(ninst * steps, dt, st), st.min(), st.max(), st.mean()))
