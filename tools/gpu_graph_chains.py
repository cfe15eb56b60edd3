"""diagnostic (not a test): configs[4]'s hipGraph-captured step on the bench workload (16384 tracking triple cartpoles, friction + Philox noise, 1000 steps) as
B independent chains of single-step launches inside one graph, B = argv: bench.py::_graph_captured_steps for each B.  python tools/gpu_graph_chains.py 1 2 4 8 16"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
import torch
pkg = g.load_package(); capi = pkg._capi
Bs = tuple(int(a) for a in sys.argv[1:]) or (1, 2, 4, 8, 16)
mech, tl, ex, octrl5, setup, z00 = bench.tracking_cfg5_workload(pkg)
mh = mech._cclqr_handle
ctrl = tl._ctrl_handle(mh, fric=ex["fric"], noise_scale=2.0, noise_seed=0xC0FFEE)
dev = torch.device("cuda", 0)
r = bench._graph_captured_steps(capi, torch, dev, mh, ctrl, np.tile(z00, (16384, 1, 1)), 1000, mech.tables().ne, branches=Bs)
print(json.dumps({b: {k: v[k] for k in ("value", "ms_per_rollout", "same_bits_as_one_persistent_launch", "capture_s")} for b, v in r["by_independent_chains"].items()}, indent=1))
