"""CPU model (instrumented oracle, no GPU) behind the frozen-factorisation iterations of the chain kernel:
 * the residual at the START of every Newton iteration, binned -- how a threshold on it predicts "this factorisation is the last one";
 * what freezing the Jacobians once ||f|| < eps (orc.set_newton_variant(1), a MODEL of the device option) does to the iteration counts
   and to the states against the reference rule.
python tools/cpu_chord_model.py [instances] [steps]   (headline workload: 17-body chain about the hanging equilibrium, golden gains)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
from oracle import orc
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n_links = 16
ex = pkg.examples.cartpole_n(n_links)
t = ex["mech"].tables()
zd = pkg.examples.cartpole_states(n_links, [0.0], np.array([[np.pi] + [0.0] * (n_links - 1)]))[0]
gold = np.load(os.path.join(ROOT, "tests", "golden", "chain16_hanging_cfg3.npz"))
K = np.tile(gold["K_first"][None], (999, 1, 1))
rng = np.random.default_rng(0)
phi = rng.uniform(-0.2, 0.2, (n, n_links)); phi[:, 0] += np.pi
z0 = pkg.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, n), phi)
ctrl = orc.ctrl_desc(t.nb, [0], K=K, N=1000, zd=zd)
res = {}
for variant in (0, 1):
    orc.set_newton_variant(variant, flops=True)
    orc.newton_stats(); orc.newton_hist()
    zT, tr, st = orc.rollout(t, ctrl, z0, steps, record=True, nthreads=1, flops=True)
    halv, reach, _, _ = orc.newton_stats()
    last, notlast, noise = orc.newton_hist()
    res[variant] = (zT, tr, st)
    its = reach.sum() / (n * steps)
    print("variant %d: %.3f Newton iterations per instance-step; solves reaching iteration i: %s" % (variant, its, (reach[:10] / (n * steps)).round(3)))
    print("   halvings per iteration index: %s" % (halv[:10] / np.maximum(reach[:10], 1)).round(2))
    if variant == 0:
        print("   -log10 ||f|| at the start of an iteration | becomes the last real one | does not | entered below eps")
        for b in range(32):
            if last[b] + notlast[b] + noise[b] > 0:
                print("   %2d  %9d %9d %9d" % (b, last[b], notlast[b], noise[b]))
orc.set_newton_variant(0, flops=True)
d = np.abs(res[0][1] - res[1][1]).reshape(n, steps, -1).max(axis=2)
print("frozen Jacobians vs the reference rule: max |state| deviation over %d steps = %.3g (median over instances of the final-step deviation %.3g); max-iteration status equal: %s"
      % (steps, d.max(), np.median(d[:, -1]), np.array_equal(res[0][2], res[1][2])))
