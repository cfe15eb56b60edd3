"""diagnostic (not a test): MPC-style step-per-launch on config 5's shape (triple cartpole, 16384 instances): plain stream launches vs
the same 100 launches captured in a hipGraph (torch.cuda.CUDAGraph) vs one 100-step launch"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package(); capi = pkg._capi
ex = pkg.examples.triple_cartpole(); t = ex["mech"].tables()
n, S, N = 16384, 100, 1000
rng = np.random.default_rng(0)
z00 = ex["mech"].state(); zd = np.tile(z00, (N, 1, 1)); K = rng.normal(size=(N - 1, 1, 48)) * 0.3
mech = capi.MechHandle(t); ctrl = capi.CtrlHandle(mech, [0], K=K, N=N, zd=zd, fric=ex["fric"])
dev = torch.device("cuda", 0)
z0 = torch.from_numpy(np.tile(z00, (n, 1, 1))).to(dev)
za, zb = z0.clone(), torch.empty_like(z0)
lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()


def chain(stream):
    src, dst = za, zb
    for k in range(1, S + 1):
        capi.rollout_dev(mech, ctrl, n, 1, k, src.data_ptr(), lam.data_ptr(), 0, 0, 0, dst.data_ptr(), st.data_ptr(), stream)
        src, dst = dst, src


with torch.cuda.stream(side):
    chain(side.cuda_stream); side.synchronize()
    t0 = time.time(); chain(side.cuda_stream); side.synchronize(); t_plain = time.time() - t0
    graph = torch.cuda.CUDAGraph(); graph.capture_begin(); chain(side.cuda_stream); graph.capture_end()
torch.cuda.synchronize()
graph.replay(); torch.cuda.synchronize()
t0 = time.time(); graph.replay(); torch.cuda.synchronize(); t_graph = time.time() - t0
capi.rollout_dev(mech, ctrl, n, S, 1, za.data_ptr(), 0, 0, 0, 0, zb.data_ptr(), st.data_ptr()); torch.cuda.synchronize()
t0 = time.time(); capi.rollout_dev(mech, ctrl, n, S, 1, za.data_ptr(), 0, 0, 0, 0, zb.data_ptr(), st.data_ptr()); torch.cuda.synchronize(); t_one = time.time() - t0
for name, dt in (("100 stream launches", t_plain), ("hipGraph replay of the 100 launches", t_graph), ("one 100-step launch", t_one)):
    print("%-38s %.2f ms  -> %.1f us/step, %.3g inst-steps/s" % (name, 1e3 * dt, 1e6 * dt / S, n * S / dt))
