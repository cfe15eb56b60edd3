"""diagnostic (not a test): throughput of the closed-loop rollout kernel on the deltabot (examples/lqr_deltabot.jl), feedback law on"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
PROF = "--phases" in sys.argv
if PROF:
    sys.argv.remove("--phases")
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libcclqr_prof.so")       # diagnostic build with phase stamps
for a in list(sys.argv):
    if a.startswith("--lib="):
        sys.argv.remove(a); capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), a[6:])       # an experiment build next to the shipped library
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
if PROF:
    import ctypes as C
    names = ["control", "forces+knotjac+map", "eval_body", "eval_joint", "-", "-", "schur rows", "elimination", "back substitution", "body_solve", "trial", "accept", "io"]
    buf = (C.c_ulonglong * 16)()
    capi.lib().cclqr_prof_read_loop(buf, 1)
    zT, _, st = capi.rollout(mech, ctrl, zb, steps)
    capi.lib().cclqr_prof_read_loop(buf, 1)
    v = np.array(list(buf), dtype=np.float64)
    tot = v[:13].sum()
    print("newton iterations/step %.2f, evaluations/step %.2f" % (v[13] / v[15], v[14] / v[15]))
    for i, nm in enumerate(names):
        if nm != "-":
            print("  %-20s %6.2f%%  %9.0f cycles/step" % (nm, 100 * v[i] / tot, v[i] / v[15]))
    print("  total cycles/step (per wavefront = instance) %.0f" % (tot / v[15]))
