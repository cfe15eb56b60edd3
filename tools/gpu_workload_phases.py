"""diagnostic (not a test): per-phase cycle shares of the rollout kernel ON A BENCH WORKLOAD (bench.py's cfg2 / cfg4 / cfg5 builders), from the
-DCCLQR_PROFILE build (make -C constrainedcontrol.jl_amd/csrc prof).  python tools/gpu_workload_phases.py <cartpole_cfg2|sawyer_cfg4|tracking_cfg5> [steps]
(tools/gpu_phase_profile.py profiles hanging chains of a given length instead: another Newton regime -- the Sawyer arm's control law, 25 % of ITS step,
was invisible there)"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); capi = pkg._capi
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), os.environ.get("CCLQR_PROF_LIB", "libcclqr_prof.so"))
import torch, bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "sawyer_cfg4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
noise_kw = {}
if cfg == "cartpole_cfg2":
    mech, lq, z0, _ = bench.cartpole_cfg2_workload(pkg, 4096)
    mh = mech._cclqr_handle; ctrl = lq._ctrl_handle(mh)
elif cfg == "sawyer_cfg4":
    mech, lq, z0, _, _ = bench.sawyer_cfg4_workload(pkg, 0.002, 8192)
    mh = mech._cclqr_handle; ctrl = lq._ctrl_handle(mh)
elif cfg == "tracking_cfg5":
    mech, tl, ex, _, _, z00 = bench.tracking_cfg5_workload(pkg)
    z0 = np.tile(z00, (16384, 1, 1))
    mh = mech._cclqr_handle; ctrl = tl._ctrl_handle(mh, fric=ex["fric"], noise_scale=2.0, noise_seed=0xC0FFEE)
else:
    raise SystemExit("unknown workload " + cfg)
names = ["control", "forces+knotjac", "eval_body", "eval_joint", "eval_map+norm", "schur_w", "schur_s", "tri_fwd", "tri_bwd", "body_solve", "trial", "accept", "io"]
buf = (C.c_ulonglong * 16)()
read = capi.lib().cclqr_prof_read_chain
read(buf, 1)
zT, _, st = capi.rollout(mh, ctrl, z0, steps)
read(buf, 1)
v = np.array(list(buf), dtype=np.float64); tot = v[:13].sum()
print("%s, %d instances x %d steps: newton iters/step %.2f evals/step %.2f failed %d" % (cfg, len(z0), steps, v[13] / v[15], v[14] / v[15], int((st <= 0).sum())))
for i, n in enumerate(names):
    print("  %-16s %6.2f%%  %9.0f cycles/step" % (n, 100 * v[i] / tot, v[i] / v[15]))
print("  total cycles/step (per wave) %.0f" % (tot / v[15]))
