"""diagnostic (not a test): instances per wavefront against the batch size -- the chain kernels at batches that leave SIMDs without a wavefront when every
wavefront is packed full (64 / lanes-per-instance instances): the launch as shipped (spread by rollout_chain.hip::spread_instances_per_wavefront) against
CCLQR_ROLLOUT_PACK_WAVEFRONTS.  (profiles/r05/batch_density.txt came from an experiment build whose density an environment variable forced: every density per batch size.)
python tools/gpu_batch_density.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
import bench
pkg = g.load_package(); capi = pkg._capi
dev = torch.device("cuda", 0)
out = {}


def run(tag, mh, ctrl, z0, steps, extra, ipws=None):
    for n in sorted({min(len(z0), m) for m in (64, 256, 1024, 2048, 4096, 8192)}):
        row = {}
        for name, flags in (("spread", 0), ("packed", capi.ROLLOUT_PACK_WAVEFRONTS)):
            r = bench._timed_rollout(capi, torch, dev, mh, ctrl, z0[:n], steps, False, reps=2, kernel=bench.kernel_name(mh, extra), flags=flags)
            row["%s (%d per wavefront)" % (name, mh.instances_per_wavefront(n, steps, flags))] = round(r["ms_per_rollout"], 3)
        out.setdefault(tag, {})[n] = row
        print(tag, n, row, flush=True)


mech, lq, z0, _ = bench.cartpole_cfg2_workload(pkg, 8192)
run("cartpole <8, 4, 3, 2>", mech._cclqr_handle, lq._ctrl_handle(mech._cclqr_handle), z0, 1000, 0, (8, 4, 2, 1))
mech, tl, ex, _, _, z00 = bench.tracking_cfg5_workload(pkg)
mh = mech._cclqr_handle
run("tracking triple cartpole <8, 4>", mh, tl._ctrl_handle(mh, fric=ex["fric"], noise_scale=2.0, noise_seed=0xC0FFEE), np.tile(z00, (8192, 1, 1)), 1000, 1, (8, 4, 2, 1))
mech, lq, z0, _, _ = bench.sawyer_cfg4_workload(pkg, 0.002, 4096)
run("sawyer <16, 8>", mech._cclqr_handle, lq._ctrl_handle(mech._cclqr_handle), z0, 1000, 0, (4, 2, 1))
ex, mech, zd, z0 = bench.build_workload(pkg, 16, 2048, 0, 0)
lqr = pkg.LQR(mech, [pkg.getid(b) for b in ex["bodies"]], [pkg.getid(ex["ctrl"][0])], ex["Q"], ex["R"], 1000 * mech.tables().dt,
              xd=[zd[i, 0:3] for i in range(17)], qd=[zd[i, 3:7] for i in range(17)])
run("17-body chain <32, 17>", mech._cclqr_handle, lqr._ctrl_handle(mech._cclqr_handle), z0, 300, 0, (2, 1))
json.dump(out, open(os.path.join(g.ROOT, "gpurun_out", "batch_density.json"), "w"), indent=1)
