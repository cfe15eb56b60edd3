"""diagnostic (not a test): tree path vs chain path throughput at equal body count"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
if os.environ.get("CCLQR_LIB"):      # an experiment build next to the shipped library (e.g. libcclqr_g16.so)
    capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ["CCLQR_LIB"])


def rate(name, t, z0, n=32768, steps=300):
    rng = np.random.default_rng(0)
    K = rng.normal(size=(steps + 5, 1, 12 * t.nb)) * 0.02
    mh = capi.MechHandle(t); ctrl = capi.CtrlHandle(mh, [0], K=K, N=steps + 6, zd=z0)
    zz = np.tile(z0[None], (n, 1, 1))
    capi.rollout(mh, ctrl, zz[:64], 10)
    t0 = time.time(); zT, _, st = capi.rollout(mh, ctrl, zz, steps); dt = time.time() - t0
    print("%-28s nb=%2d lanes/LDS %s: %s, Newton iters max %d" % (name, t.nb, mh.geometry(), capi.rate_or_refusal(n * steps, dt, st), st.max()))


ex = pkg.examples.dual_cartpole(); rate("dual cartpole (tree)", ex["mech"].tables(), ex["mech"].state())
ex = pkg.examples.cartpole_n(2); rate("cartpole_n(2) (chain)", ex["mech"].tables(), ex["mech"].state())
parents_tree = [-1, 0, 1, 2, 3, 2, 5, 6, 1, 8, 8, 10, 0, 12]
ex = pkg.examples.tree_mechanism(parents_tree, seed=4, prismatic=(0, 5)); rate("random tree, 14 bodies", ex["mech"].tables(), ex["mech"].state())
ex = pkg.examples.cartpole_n(13); rate("cartpole_n(13) (chain), 14 bodies", ex["mech"].tables(), ex["mech"].state())
