"""diagnostic (not a test): cclqr_rollout_opts.newton_mode = 1 (a Newton solve stops on ||f|| < eps alone) against the exact rule on the
headline workload (17-body chain, hanging-equilibrium LQR, 8192 instances x 1000 steps, record = true): kernel time of both modes and the
max state deviation over the whole recorded trajectories.  Writes profiles-style JSON to stdout (argv[1] = output path, optional)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
import bench
pkg = g.load_package(); capi = pkg._capi
n_links = int(os.environ.get("LINKS", "16")); n = int(os.environ.get("INSTANCES", "8192")); T = int(os.environ.get("SIM_STEPS", "1000"))
ex, mech, zd, z0 = bench.build_workload(pkg, n_links, n, 0, 0)
t = mech.tables(); nb = t.nb
lqr = pkg.LQR(mech, [pkg.getid(b) for b in ex["bodies"]], [pkg.getid(ex["ctrl"][0])], ex["Q"], ex["R"], T * t.dt,
              xd=[zd[i, 0:3] for i in range(nb)], qd=[zd[i, 3:7] for i in range(nb)])
mh = mech._cclqr_handle; ctrl = lqr._ctrl_handle(mh)
dev = torch.device("cuda", 0)
z0_d = torch.from_numpy(z0).to(dev)
out = {}
trajs = []
EPS = [float(x) for x in os.environ.get("EPS_ALONE", "1e-10,1e-11,1e-12,1e-13").split(",")]
ref = None
for mode, eps in [(0, 0.0)] + [(1, e) for e in EPS]:
    zT = torch.empty_like(z0_d); st = torch.zeros(n, dtype=torch.int32, device=dev)
    traj = torch.empty((n, T, nb, 13), dtype=torch.float64, device=dev)
    run = lambda: capi.rollout_dev(mh, ctrl, n, T, 1, z0_d.data_ptr(), 0, 0, 0, traj.data_ptr(), zT.data_ptr(), st.data_ptr(), torch.cuda.current_stream().cuda_stream,
                                   newton_mode=mode, newton_eps_alone=eps)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    s = st.cpu().numpy()
    key = "mode0_exact_rule" if mode == 0 else "mode1_eps_alone_%g" % eps
    out[key] = {"ms_per_rollout": ms, "instance_steps_per_s": n * T / (ms * 1e-3), "failed_instances": int((s <= 0).sum()),
                            "max_newton_iterations_mean": float(np.abs(s).mean()), "max_newton_iterations_max": int(np.abs(s).max())}
    if mode == 0:
        ref, ref_ms = traj, ms
    else:
        d = (ref - traj).abs()
        out[key]["max_state_deviation_vs_exact_rule"] = float(d.max().item())
        out[key]["max_state_deviation_by_step_quartile"] = [float(d[:, a:b].max().item()) for a, b in ((0, T // 4), (T // 4, T // 2), (T // 2, 3 * T // 4), (3 * T // 4, T))]
        out[key]["speedup"] = ref_ms / ms
        del d, traj
out["workload"] = "lqr_cartpole_n_pendulum N=%d (%d bodies), hanging-equilibrium LQR, %d instances x %d steps, record=true" % (n_links, nb, n, T)
out["note"] = "mode 0 = exact stopping rule (parity mode, bench default); mode 1 = measured-error option, never the bench headline"
txt = json.dumps(out, indent=1)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
