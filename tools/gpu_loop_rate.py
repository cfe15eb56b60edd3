"""diagnostic (not a test): throughput of the closed-loop rollout kernel on the deltabot (examples/lqr_deltabot.jl), feedback law on"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ex = pkg.examples.deltabot(); mech_py = ex["mech"]; t = mech_py.tables()
cj = [mech_py.joint_index(e) for e in ex["eqcids"]]
z0 = mech_py.state()
rng = np.random.default_rng(0)
K = rng.normal(size=(1, 2, 12 * t.nb)) * 0.05
mech = capi.MechHandle(t)
scale = rng.uniform(0.97, 1.03, n)
ctrl = capi.CtrlHandle(mech, cj, K=np.repeat(K[None], n, 0), N=0, zd=np.repeat(z0[None, None], n, 0), Fd=scale[:, None, None] * ex["Fd"].reshape(1, 1, 2), n_ctrl=n)
zb = np.repeat(z0[None], n, 0)
print("geometry", mech.geometry())
for rep in range(3):
    t0 = time.time(); zT, _, st = capi.rollout(mech, ctrl, zb, steps); dt = time.time() - t0
    print("deltabot inst %d steps %d: %.3fs -> %s (host-pointer API incl. copies); newton iterations max per instance: min %d max %d" % (
        n, steps, dt, capi.rate_or_refusal(n * steps, dt, st), st.min(), st.max()))
