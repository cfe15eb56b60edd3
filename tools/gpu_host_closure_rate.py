"""diagnostic (not a test): the host-closure `controlfunction` leg of bench.py alone (bench.py::_host_closure_rate: 16384 tracking triple cartpoles, the script's law in
numpy on the host, one launch per step), with cProfile's view of where a step's host time goes.  python tools/gpu_host_closure_rate.py [steps]"""
import cProfile, json, os, pstats, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
import torch
pkg = g.load_package(); capi = pkg._capi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mech, tl, ex, octrl5, setup, z00 = bench.tracking_cfg5_workload(pkg)
z0 = np.tile(z00, (16384, 1, 1))
bench._host_closure_rate(pkg, capi, torch, mech, tl, ex, z0, 5)          # warm-up
r = bench._host_closure_rate(pkg, capi, torch, mech, tl, ex, z0, steps)
rd = bench._device_closure_rate(pkg, capi, torch, mech, tl, ex, z0, 1000)
print("device closure:", json.dumps({k: rd[k] for k in ("instances", "sim_steps", "value", "s_per_run", "captured_in_a_hip_graph")}))
print(json.dumps({k: r[k] for k in ("instances", "sim_steps", "value", "s_per_run")}))
cProfile.run("bench._host_closure_rate(pkg, capi, torch, mech, tl, ex, z0, steps)", "/tmp/host_closure.prof")
pstats.Stats("/tmp/host_closure.prof").sort_stats("tottime").print_stats(14)
