"""diagnostic (not a test): time the headline workload (17-body chain, 8192 instances, record = true) through alternative builds of the
library and check them against the shipped build bit for bit.  argv: steps lib1.so [lib2.so ...]  (each runs in its own process)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, os, json, numpy as np
sys.path.insert(0, %(root)r)
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
capi.LIB_PATH = sys.argv[1]
import torch, bench
steps = int(sys.argv[2]); links = int(os.environ.get("LINKS", "16"))
ex, mech, zd, z0 = bench.build_workload(pkg, links, 8192, 0, 0)
t = mech.tables(); nb = t.nb
K = np.tile(np.load(os.path.join(g.ROOT, "tests", "golden", "chain16_hanging_cfg3.npz"))["K_first"][None], (999, 1, 1)) if links == 16 else np.random.default_rng(0).normal(size=(999, 1, 12 * nb)) * 0.05
mh = capi.MechHandle(t); ctrl = capi.CtrlHandle(mh, [0], K=K, N=1000, zd=zd)
dev = torch.device("cuda", 0)
z0_d = torch.from_numpy(z0).to(dev); zT = torch.empty_like(z0_d); st = torch.zeros(8192, dtype=torch.int32, device=dev)
traj = torch.empty((8192, steps, nb, 13), dtype=torch.float64, device=dev)
run = lambda: capi.rollout_dev(mh, ctrl, 8192, steps, 1, z0_d.data_ptr(), 0, 0, 0, traj.data_ptr(), zT.data_ptr(), st.data_ptr(), torch.cuda.current_stream().cuda_stream)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 2
h = float(zT.double().sum().item()), float(traj[:, -1].abs().sum().item())
print(json.dumps({"lib": os.path.basename(sys.argv[1]), "ms": ms, "minst_steps_per_s": 8192 * steps / ms / 1e3, "failed": int((st <= 0).sum().item()), "checksum": h}))
'''
steps = sys.argv[1]
ref = None
for lib in sys.argv[2:]:
    r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, os.path.abspath(lib), steps], capture_output=True, text=True)
    line = [x for x in r.stdout.splitlines() if x.startswith("{")]
    if not line:
        print(lib, "FAILED", r.stderr[-800:]); continue
    d = json.loads(line[-1])
    if ref is None:
        ref = d["checksum"]
    d["same_bits_as_first"] = d["checksum"] == ref
    print(json.dumps(d), flush=True)
