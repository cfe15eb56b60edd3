"""diagnostic (not a test): the LDS-resident batched Riccati kernel at config 4's shape (Sawyer: mx = 84, mu = 7, ml = 35), distinct setpoints,
with keep_last (only Ku[1] leaves the device, so the wall clock of the host-pointer call is upload + kernels).  For kernel time run it
under `rocprofv3 --kernel-trace --stats`.  argv: nprob [N] [bf16_terms]"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
nprob = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
terms = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tab = json.load(open(os.path.join(g.ROOT, "tests", "golden", "sawyer_arm_tables.json")))
ex = pkg.examples.sawyer(tab); mech = ex["mech"]; t = mech.tables()
rng = np.random.default_rng(0)
zs = pkg.joint_position_states(mech, rng.uniform(-0.8, 0.8, (nprob, 7)))
mh = capi.MechHandle(t)
t0 = time.time(); A, Bu, Bl, G = capi.linearize(mh, zs, list(range(7)), np.zeros((nprob, 7))); tl = time.time() - t0
Q = np.eye(84) * 1000 * t.dt; R = np.eye(7) * t.dt
capi.riccati(A[:2], Bu[:2], Bl[:2], G[:2], Q, R, 3, path=1)
K0, kb0 = capi.riccati(A[:64], Bu[:64], Bl[:64], G[:64], Q, R, N, tol=0.0, path=1, keep_last=True)                       # fp64 reference for the error figure
t0 = time.time(); K, kb = capi.riccati(A, Bu, Bl, G, Q, R, N, tol=0.0, path=1, keep_last=True, bf16_terms=terms); tr = time.time() - t0
mx, mu, ml = 84, 7, 35; m = mu + ml
F = 4 * mx**3 + 4 * mx**2 * m + 2 * mx * (ml**2 + m**2) + 2 / 3 * m**3 + 2 / 3 * ml**3
steps = nprob * (N - 1)
err = float(np.abs(K[:64] - K0).max() / np.abs(K0).max())
print(json.dumps({"nprob": nprob, "N": N, "bf16_terms": terms, "linearize_s": tl, "riccati_wall_s": tr, "backward_steps": steps,
                  "tflops_by_reference_count_wall": F * steps / tr / 1e12, "gain_rel_err_vs_fp64_first64": err, "finite": bool(np.isfinite(K).all())}))
