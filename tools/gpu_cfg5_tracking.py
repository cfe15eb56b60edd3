"""diagnostic: config 5 at full size -- TrackingLQR about the reference's swing-up input U, 16384 instances, friction + noise law"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
U = np.load(os.path.join(g.ROOT, "tests", "golden", "triple_cartpole_U.npy"))
ex = pkg.examples.triple_cartpole(); mech = ex["mech"]; j1 = ex["ctrl"][0]
z00 = mech.state()
t0 = time.time(); s0 = pkg.simulate(mech, pkg.Storage(1000, 4), pkg.OpenLoop(mech, [j1.id], U.reshape(1000, 1))); t1 = time.time()
tl = pkg.TrackingLQR(mech, s0, [[[U[k]]] for k in range(1000)], [j1.id], ex["Q"], ex["R"]); t2 = time.time()
print("open loop %.3fs, TrackingLQR (999 linearisations + time-varying Riccati) %.3fs, kbreak %d |K| %.1f" % (t1 - t0, t2 - t1, tl.kbreak, np.abs(tl.K).max()))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0xC0FFEE)
noise = rng.normal(size=(n, 1000))
z0 = np.tile(z00, (n, 1, 1))
for label, ctrl, ns in (("open loop + friction + noise (the script's uncontrol!)", pkg.OpenLoop(mech, [j1.id], U.reshape(1000, 1)), 2.0),
                        ("TrackingLQR + friction + noise (owncontrol_trackinglqr!)", tl, 2.0), ("TrackingLQR + friction, no noise", tl, 0.0)):
    mech.set_state(z00)
    t0 = time.time(); st = pkg.simulate(mech, pkg.Storage(1000, 4), ctrl, record=False, z0=z0, fric=ex["fric"], noise=noise, noise_scale=ns); dt = time.time() - t0
    ang = np.degrees(2 * np.arctan2(st.zT[:, 1:, 4], st.zT[:, 1:, 3]))
    err = np.abs((ang - 180 + 180) % 360 - 180)
    print("%s: %.2fs (%s incl. copies); status ok %d/%d; final |angle-180| median %s  90%% %s; cart y median %.3f" % (
        label, dt, pkg._capi.rate_or_refusal(n * 1000, dt, st.status), (st.status > 0).sum(), n, np.median(err, axis=0).round(1), np.percentile(err, 90, axis=0).round(1), np.median(np.abs(st.zT[:, 0, 1]))))
