"""diagnostic (not a test): ONE BASELINE config's rollout launched a few times and nothing else, so that a rocprofv3 --pmc pass attributes
its counters to that config's kernel (tools/profile_configs.sh).  python tools/gpu_config_rollout.py <cartpole_cfg2|cartpole_cfg2_filled|
sawyer_cfg4|tracking_cfg5|tree14|deltabot> [launches]   -> one JSON line: kernel, instances, steps, wavefronts, ms per launch"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
import bench
pkg = g.load_package(); capi = pkg._capi
cfg = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
extra = 0
if cfg.startswith("cartpole_cfg2"):
    mech, lq, z0, _ = bench.cartpole_cfg2_workload(pkg, 65536 if cfg.endswith("filled") else 4096)
    steps, record = 1000, not cfg.endswith("filled")
    mh = mech._cclqr_handle; ctrl = lq._ctrl_handle(mh)
elif cfg == "sawyer_cfg4":
    mech, lq, z0, _, _ = bench.sawyer_cfg4_workload(pkg, 0.002, 8192)
    steps, record = 2000, False
    mh = mech._cclqr_handle; ctrl = lq._ctrl_handle(mh)
elif cfg == "tracking_cfg5":
    mech, tl, ex, _, _, z00 = bench.tracking_cfg5_workload(pkg)
    z0 = np.tile(z00, (16384, 1, 1))
    steps, record, extra = 1000, True, 1
    mh = mech._cclqr_handle; ctrl = tl._ctrl_handle(mh, fric=ex["fric"], noise_scale=2.0, noise_seed=0xC0FFEE)
elif cfg == "tree14":
    t, octrl, z0, steps = bench.tree14_workload(pkg)
    record = False
    mh = capi.MechHandle(t); ctrl = capi.CtrlHandle(mh, octrl["ctrl_joint"], K=octrl["K"], N=octrl["N"], zd=octrl["zd"])
elif cfg == "deltabot":
    mh, ctrl, z0, kern = bench.deltabot_workload(pkg, capi)
    steps, record = 200, False
else:
    raise SystemExit("unknown config " + cfg)
r = bench._timed_rollout(capi, torch, dev, mh, ctrl, z0, steps, record, reps=reps, kernel=kern if cfg == "deltabot" else bench.kernel_name(mh, extra))
lanes, lds = mh.geometry()
ipw = mh.instances_per_wavefront(len(z0), steps)
r.update(config=cfg, lanes_per_instance=lanes, instances_per_wavefront=ipw, wavefronts=(len(z0) + ipw - 1) // ipw,
         lds_bytes_per_workgroup=lds, workgroups_per_cu_by_lds=int(160 * 1024 // lds))
print(json.dumps(r))
