"""diagnostic (not a test): gain error and time of the split-bf16 Riccati modes (cclqr_riccati_opts.bf16_terms) against the fp64
MFMA parity mode -- BASELINE configs[3] (Sawyer, mx = 84, mu = 7, N = 2000) and the headline chain (mx = 204, mu = 1, N = 1000)"""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
from oracle import orc
tab = json.load(open(os.path.join(g.ROOT, "tests", "golden", "sawyer_arm_tables.json")))
cases = []
ex = pkg.examples.sawyer(tab); t = ex["mech"].tables()
cases.append(("sawyer mx=84 mu=7 N=2000", capi.MechHandle(t), ex["mech"].state()[None], list(range(7)), np.eye(84) * 1000.0 * t.dt, np.eye(7) * t.dt, 2000))
ex = pkg.examples.cartpole_n(16); t = ex["mech"].tables()
zh = pkg.examples.cartpole_states(16, [0.0], np.array([[np.pi] + [0.0] * 15]))
cases.append(("chain mx=204 mu=1 N=1000", capi.MechHandle(t), zh, [0], np.eye(204) * t.dt, np.eye(1) * t.dt, 1000))
out = {}
for name, mh, zd, cj, Q, R, N in cases:
    A, Bu, Bl, G = (m[0] for m in capi.linearize(mh, zd, cj, np.zeros((1, len(cj)))))
    ref = None
    for terms in (0, 3, 2, 1):
        capi.riccati(A, Bu, Bl, G, Q, R, 50, path=2, bf16_terms=terms)      # warm
        t0 = time.time(); K, kb = capi.riccati(A, Bu, Bl, G, Q, R, N, path=2, bf16_terms=terms); dt = time.time() - t0
        if terms == 0:
            ref = K
        steps = N - max(kb, 1)
        err = np.abs(K - ref).max() / np.abs(ref).max()
        err0 = np.abs(K[0] - ref[0]).max() / np.abs(ref[0]).max()
        print("%-26s bf16_terms %d: %.3f s for %4d backward steps (%.1f us/step), kbreak %4d, max rel gain error %.2e (first gain %.2e), finite %s" % (
            name, terms, dt, steps, 1e6 * dt / max(steps, 1), kb, err, err0, np.isfinite(K).all()), flush=True)
        out["%s/%d" % (name, terms)] = dict(seconds=dt, backward_steps=steps, kbreak=int(kb), max_rel_gain_error=float(err), first_gain_rel_error=float(err0))
json.dump(out, open(os.path.join(g.ROOT, "gpurun_out", "riccati_bf16.json"), "w"), indent=1)
