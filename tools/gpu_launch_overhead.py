"""diagnostic (not a test): what a single-step launch of the chain kernel costs besides its step -- prologue (constants, state, multipliers, LDS image) + epilogue
(final state, multipliers, status) -- measured as hipGraph replays of ZERO-step launches on configs[4]'s shape (16384 triple cartpoles, friction law), 1024 and 2048 wavefronts"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch
pkg = g.load_package(); capi = pkg._capi
ex = pkg.examples.triple_cartpole(); t = ex["mech"].tables()
N = 1000
rng = np.random.default_rng(0)
z00 = ex["mech"].state(); zd = np.tile(z00, (N, 1, 1)); K = rng.normal(size=(N - 1, 1, 48)) * 0.3
mech = capi.MechHandle(t); ctrl = capi.CtrlHandle(mech, [0], K=K, N=N, zd=zd, fric=ex["fric"])
dev = torch.device("cuda", 0)
for n in (8192, 16384):
    z0 = torch.from_numpy(np.tile(z00, (n, 1, 1))).to(dev)
    za, zb = z0.clone(), torch.empty_like(z0)
    lam = torch.zeros((n, 5 * t.ne), dtype=torch.float64, device=dev); st = torch.zeros(n, dtype=torch.int32, device=dev)
    for steps in (0, 1):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            graph.capture_begin()
            src, dst = za, zb
            for k in range(1, 501):
                capi.rollout_dev(mech, ctrl, n, steps, 2, src.data_ptr(), lam.data_ptr(), 0, 0, 0, dst.data_ptr(), st.data_ptr(), side.cuda_stream)
                src, dst = dst, src
            graph.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter(); graph.replay(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%5d instances (%4d wavefronts), %d-step launches: %.2f us per launch" % (n, n // 8, steps, 1e6 * dt / 500))
