import sys, os, json, numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
pkg = g.load_package()
from oracle import orc
tab = json.load(open('/root/repo/tests/golden/sawyer_arm_tables.json'))
ex = pkg.examples.sawyer(tab); mech = ex["mech"]
lqr = pkg.LQR(mech, [pkg.getid(b) for b in mech.bodies], [pkg.getid(e) for e in mech.eqconstraints], ex["Q"], ex["R"], 20.0, xd=ex["xd"], qd=ex["qd"])
print("K abs max first/last", np.abs(lqr.K[0]).max(), np.abs(lqr.K[-1]).max(), "kbreak", lqr.kbreak)
zd = mech.state(); rng = np.random.default_rng(4); base = []
for n in range(64):
    for e in mech.eqconstraints: pkg.setJointPosition(mech, e, rng.uniform(-0.05, 0.05))
    base.append(mech.state())
z0 = np.stack(base)
st = pkg.simulate(mech, 20.0, lqr, record=True, z0=z0)
bad = np.where(st.status <= 0)[0]; print("failed", len(bad), "of 64:", bad[:10], st.status[bad][:10])
t = mech.tables(); oc = orc.ctrl_desc(7, list(range(7)), K=lqr.K, N=lqr.N, zd=lqr.zd)
idx = list(bad[:2]) + [i for i in range(64) if i not in bad][:2]
zo, traj, sto = orc.rollout(t, oc, z0[idx], 2000, record=True)
print("oracle status for", idx, sto)
for q, i in enumerate(idx):
    d = np.abs(st.z[i] - traj[q]); k = np.argmax(~np.isfinite(d).all(axis=(1,2)) | (d.max(axis=(1,2)) > 1e-6)) if ((~np.isfinite(d)).any() or d.max() > 1e-6) else -1
    print(" inst", i, "first step with |diff|>1e-6:", k, "max diff before", np.nanmax(d[:k]) if k > 0 else np.nanmax(d), "max |w| gpu", np.nanmax(np.abs(st.z[i,:,:,10:13])))
