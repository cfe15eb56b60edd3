#!/usr/bin/env python3
"""bench.py -- LQR-controlled simulation steps/s on the N-link cartpole batch (BASELINE.json metric).

A bench "step" = ONE rollout of the whole per-GPU batch over the full horizon through the HIP hot path
(control_lqr! + variational-integrator Newton solve fused, one persistent kernel launch), record=true like the
reference's simulate!(mech, 10, lqr, record = true).  Inputs (initial states, gains, setpoints) are resident in HBM
before the timed region; outputs stay in HBM.  The LQR itself (linearsystem + 999-step Riccati, both HIP) is setup and is
timed separately (reported under "setup").

Workload (config.workload): the mechanism of examples/lqr_cartpole_n_pendulum.jl with N = 16 links (17 bodies,
187 Newton unknowns), 8192 instances per GPU (configs[2]: 65536 over 8 GPUs), 1000 steps, Q = I, R = 1, horizon 10 s,
regulated about its HANGING equilibrium: about the upright one the reference's own recursion yields |K| ~ 1e11 and every
fp64 rollout diverges (DESIGN.md "Workloads"), so upright chains are parity-tested up to N = 8 only.

Multi-GPU: one process per GPU (torchrun), instances sharded with no data-path collective.  Inside the timed region the
recorded trajectories of all ranks are collected on rank 0 in time chunks (--chunks launches per rollout through the k0
continuation of cclqr_rollout_dev; chunk c travels over RCCL on a second stream while chunk c+1 is computed) and the final
states with one more gather ("scaling": "weak").  RCCL is mandatory for --gpus > 1: gloo has to be asked for (--allow-gloo).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

# RCCL shares device buffers between the ranks' processes through dmabuf IPC on this driver; the legacy IPC mode fails with
# `hipIpcGetMemHandle: invalid argument` (already exported on the pool's images: kept here for any other launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # vector fp64 = matrix fp64 on MI355X (256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz)


def b_step(nb, ml, mu, record):
    """algorithmic bytes per instance-step, step-per-launch form"""
    return 2 * (104 * nb + 8 * ml) + (104 * nb if record else 0) + 8 * mu


def kernel_source_sha():
    """fingerprint of the rollout kernel's sources: PMC-measured numbers in profiles/ are only quoted for the sources they were measured on"""
    h = hashlib.sha256()
    for f in ("rollout_chain.hip", "cclqr_chain.h", "cclqr_dev.h", "cclqr_newton.h", "cclqr_internal.h"):
        h.update(open(os.path.join(ROOT, "constrainedcontrol.jl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def host_cpu_allowance():
    """cores this process may use: the scheduler affinity mask, cut down by the cgroup CPU quota when there is one"""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    cores = aff if quota is None else max(1, min(aff, int(quota)))
    return cores, aff, quota


def self_launch(argv, n):
    """`python bench.py --gpus N` outside a torchrun environment: start the N ranks as FRESH child processes (one per GPU, RCCL over
    xGMI) BEFORE anything in this process has touched the GPU -- this parent never imports torch or the HIP library, never re-execs;
    it passes rank 0's JSON line through (the children inherit stdout/stderr) and exits with the job's exit code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def build_workload(pkg, n_links, n_inst, seed, rank):
    ex = pkg.examples.cartpole_n(n_links)
    mech = ex["mech"]
    zd = pkg.examples.cartpole_states(n_links, [0.0], np.array([[np.pi] + [0.0] * (n_links - 1)]))[0]
    rng = np.random.default_rng(seed + 1000 * rank)
    phi = rng.uniform(-0.2, 0.2, (n_inst, n_links))
    phi[:, 0] += np.pi
    z0 = pkg.examples.cartpole_states(n_links, rng.uniform(-0.5, 0.5, n_inst), phi)
    return ex, mech, zd, z0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--links", type=int, default=16)
    ap.add_argument("--instances", type=int, default=8192, help="instances PER GPU")
    ap.add_argument("--sim-steps", type=int, default=1000)
    ap.add_argument("--no-record", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline workload only (the counter passes of tools/profile_round.sh: counters serialise every dispatch)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="instances of the CPU baseline sample (0 = auto)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="launches per rollout (trajectory collected chunk by chunk); 0 = 1 on one GPU, 8 on several")
    ap.add_argument("--collect", choices=("trajectory", "final"), default="trajectory",
                    help="what rank 0 collects inside the timed region of a multi-GPU run: 'trajectory' = the recorded trajectories of ALL ranks, "
                         "assembled in the Storage layout (116 GB on rank 0 at 8 x 8192 x 1000 steps, gathered chunk by chunk behind the compute) + the "
                         "final states; 'final' = the final states only, every rank keeps its own recorded trajectory in its HBM -- the fallback "
                         "if the root buffer or the overlap misbehaves on a node (the mode is written into the JSON line)")
    ap.add_argument("--allow-gloo", action="store_true", help="accept the gloo backend for --gpus > 1 (never a scaling measurement)")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="rehearsal of the N>1 path on a 1-GPU box: every rank uses cuda:0 and the collective runs over gloo "
                         "(RCCL refuses two ranks on one device); the JSON line is marked and is not a scaling measurement")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    pkg = graft.load_package()
    capi = pkg._capi
    rank, world, local = pkg.dist.init_from_env(backend="gloo" if (args.rehearse_shared_gpu or args.allow_gloo) else None)
    if world != args.gpus and world > 1:
        args.gpus = world
    if world > 1 and not (args.rehearse_shared_gpu or args.allow_gloo):
        assert dist.get_backend() == "nccl", "multi-GPU runs collect over RCCL; pass --allow-gloo to run anything else"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP only (no CPU fallback)")
    if args.rehearse_shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    capi.set_device(local)
    dev = torch.device("cuda", local)
    record = not args.no_record
    n_links, n_inst, T = args.links, args.instances, args.sim_steps

    ex, mech, zd, z0 = build_workload(pkg, n_links, n_inst, 0, rank)
    t = mech.tables()
    nb, mx, ml, mu = t.nb, 12 * t.nb, 5 * t.ne, 1

    # ---- setup (untimed for the metric): LQR(mech, bodyids, eqcids, Q, R, 10.; xd, qd) on the device
    t0 = time.time()
    lqr = pkg.LQR(mech, [pkg.getid(b) for b in ex["bodies"]], [pkg.getid(ex["ctrl"][0])], ex["Q"], ex["R"], T * t.dt,
                  xd=[zd[i, 0:3] for i in range(nb)], qd=[zd[i, 3:7] for i in range(nb)])
    setup_s = time.time() - t0
    t0 = time.time()            # the same construction again: without the first call's code-object load and workspace allocation
    pkg.LQR(mech, [pkg.getid(b) for b in ex["bodies"]], [pkg.getid(ex["ctrl"][0])], ex["Q"], ex["R"], T * t.dt,
            xd=[zd[i, 0:3] for i in range(nb)], qd=[zd[i, 3:7] for i in range(nb)])
    setup_warm_s = time.time() - t0
    mh = mech._cclqr_handle
    ctrl = lqr._ctrl_handle(mh)
    lanes, lds_bytes = mh.geometry()

    gather_traj = record and args.collect == "trajectory"
    chunks = args.chunks if args.chunks > 0 else (8 if (world > 1 and gather_traj) else 1)
    assert T % chunks == 0, "--sim-steps must be a multiple of --chunks"
    Tc = T // chunks
    # ---- allocation plan of this rank, checked against the free HBM BEFORE anything large is allocated: a run that cannot fit ends here, on
    # every rank, with a sentence and exit code 3 (rank 0 of an 8-GPU trajectory collection holds ~150 GB: DESIGN.md 5)
    plan = pkg.dist.collection_plan(rank, world, n_inst, T, nb, chunks, record, args.collect)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    fits = plan["total"] <= 0.92 * free_b
    all_fit = pkg.dist.max_over_ranks(0.0 if fits else 1.0, dev) == 0.0
    if rank == 0 or not fits:
        print("[bench rank %d] allocation plan (--collect %s, %d chunk(s)): %.2f GB of %.2f GB free (%.0f GB HBM)%s" %
              (rank, args.collect, chunks, plan["total"] / 1e9, free_b / 1e9, total_b / 1e9, "" if fits else "  -- DOES NOT FIT"), file=sys.stderr)
        for k, v in plan.items():
            if k != "total":
                print("[bench rank %d]   %-92s %9.3f GB" % (rank, k, v / 1e9), file=sys.stderr)
    if not all_fit:
        if rank == 0:
            print("[bench] refusing to start: the collection buffers of at least one rank exceed its free HBM; use --collect final (every rank keeps "
                  "its own trajectory, only final states travel), fewer --instances, or --no-record", file=sys.stderr)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3)
    z0_d = torch.from_numpy(z0).to(dev)
    zT_d = torch.empty_like(z0_d)
    # one status per instance for the whole rollout: chunk 0 writes it, the later chunks carry it (CCLQR_ROLLOUT_CARRY_STATUS: an instance lost in one chunk
    # stays frozen in the next, as in one launch over the horizon)
    st_d = torch.zeros(n_inst, dtype=torch.int32, device=dev)
    lam_d = torch.zeros((n_inst, ml), dtype=torch.float64, device=dev) if chunks > 1 else None
    collect = gather_traj and (world > 1 or chunks > 1)          # trajectories leave the rank (or are re-assembled) chunk by chunk
    tg = pkg.dist.TrajectoryGather(rank, world, n_inst, T, nb, chunks, dev) if collect else None
    traj_d = torch.empty((n_inst, T, nb, 13), dtype=torch.float64, device=dev) if (record and not collect) else None
    stream = torch.cuda.current_stream().cuda_stream
    kern_ev = []
    # final states of all ranks on rank 0: one fixed-size RCCL fan-in, every buffer allocated here (nothing inside the timed region)
    zT_gather = pkg.dist.RootGather(n_inst * world, (nb, 13), torch.float64, dev, rank, world) if world > 1 else None

    def one_rollout(timed):
        """one bench step: the whole horizon for this rank's instances (+ the collection on rank 0 when there is one)"""
        for c in range(chunks):
            if tg is not None:
                tg.wait_slab_free(c)
                traj_ptr = tg.slab(c).data_ptr()
            else:       # the rank's own trajectory buffer (chunk c writes its T/chunks steps of every instance: row stride = the launch's steps)
                assert chunks == 1 or not record, "a rank-local trajectory is written by ONE launch per rollout (use --chunks 1 with --collect final)"
                traj_ptr = traj_d.data_ptr() if record else 0
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            capi.rollout_dev(mh, ctrl, n_inst, Tc, c * Tc + 1, (z0_d if c == 0 else zT_d).data_ptr(), lam_d.data_ptr() if lam_d is not None else 0,
                             0, 0, traj_ptr, zT_d.data_ptr(), st_d.data_ptr(), stream, flags=(capi.ROLLOUT_CARRY_STATUS if c > 0 else 0))
            if timed:
                e1.record()
                kern_ev.append((e0, e1))
            if tg is not None:
                tg.submit(c)
        if tg is not None:
            tg.finish()
        if world > 1:
            return zT_gather(zT_d)
        return zT_d

    if world > 1:   # open the RCCL peer connections the gathers use, whatever --warmup says (communicator set-up is not a step)
        zT_gather(zT_d)
    for _ in range(args.warmup):
        one_rollout(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_rollout(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = pkg.dist.max_over_ranks(elapsed, dev)
    kern_ms_launch = float(np.mean([a.elapsed_time(b) for a, b in kern_ev])) if kern_ev else float("nan")   # average launch
    kern_ms = kern_ms_launch * chunks                                                                      # kernel time of one rollout
    status = st_d.cpu().numpy()
    n_bad = int((status <= 0).sum())
    status = np.abs(status)
    gathered = tg.bytes_gathered / max(1, args.steps + args.warmup) if tg is not None else 0

    backend_name = dist.get_backend() if world > 1 else None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()     # every rank leaves the job cleanly before rank 0 formats the line
    if rank != 0:
        return
    total_units = float(n_inst) * world * T * args.steps
    value = total_units / elapsed
    bs = b_step(nb, ml, mu, record)
    alg_bytes_per_launch = bs * float(n_inst) * Tc            # one launch = Tc steps of every instance of this rank
    achieved = alg_bytes_per_launch / (kern_ms_launch * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            same_shape = tj.get("config") == {"links": n_links, "instances_per_gpu": n_inst, "sim_steps": Tc, "record": record}
            if same_shape and tj.get("kernel_source_sha") == kernel_source_sha():
                traffic = tj.get("hbm_bytes_per_launch")      # PMC-measured for exactly this launch shape AND these kernel sources
        except Exception:
            traffic = None
    out = {
        "metric": "LQR sim steps/sec (whole node) on N-link cartpole batch", "value": value, "unit": "instance-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "rehearsal_shared_gpu": bool(args.rehearse_shared_gpu), "collective_backend": backend_name,
        "config": {"workload": "lqr_cartpole_n_pendulum N=%d links (%d bodies), hanging-equilibrium LQR, Q=I R=1 horizon %gs, "
                               "y0~U(-0.5,0.5) phi_i~U(-0.2,0.2)" % (n_links, nb, T * t.dt),
                   "instances_per_gpu": n_inst, "sim_steps": T, "record": record, "parallelism": "instances sharded x%d" % world,
                   "lanes_per_instance": lanes, "lds_bytes_per_workgroup": lds_bytes, "launches_per_rollout": chunks},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel": kernel_name(mh),
                     "kernel_ms": kern_ms_launch, "launches_per_rollout": chunks, "algorithmic_bytes_per_instance_step": bs,
                     "kernel_source_sha": kernel_source_sha()},
        "collection": {"mode": (args.collect if world > 1 else "single GPU: nothing leaves the device"),
                       "trajectory_bytes_gathered_to_rank0_per_rollout": gathered, "chunks": chunks,
                       "rank0_allocation_plan_gb": round(plan["total"] / 1e9, 3),
                       "exposed_ms_per_rollout": 1e3 * elapsed / max(1, args.steps) - kern_ms},
        "newton": {"max_iters_mean": float(status.mean()), "max_iters_max": int(status.max()), "failed_instances": n_bad},
        "setup": {"lqr_construct_s": setup_s, "lqr_construct_warm_s": setup_warm_s, "riccati_kbreak": int(lqr.kbreak)},
    }
    if not args.no_cpu_baseline:
        # The kernel is bound by the fp64 vector ALU, not by HBM: with the flops of one instance-step counted by the
        # instrumented oracle on this workload, `roofline` carries THAT bound and keeps the HBM figures beside it.
        f_step = count_flops(pkg, t, lqr, z0, T)
        tf = f_step * (float(n_inst) * Tc / (kern_ms_launch * 1e-3)) / 1e12          # per launch, by the kernel's own duration
        hbm = out["roofline"]
        out["roofline"] = {"bound": "fp64_valu", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                           "traffic": hbm["traffic"], "flops_per_instance_step_counted_by_oracle": f_step,
                           "hbm_achieved_gbs": hbm["achieved"], "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": hbm["frac"],
                           **{k: hbm[k] for k in ("kernel", "kernel_ms", "launches_per_rollout", "algorithmic_bytes_per_instance_step", "kernel_source_sha")},
                           "note": "fp64 vector ALU is the binding unit (256 CU x 4 SIMD x 16 fp64 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s); "
                                   "HBM carries only the recorded trajectory (traffic < algorithmic bytes: state never leaves the chip)"}
    if world == 1 and not args.no_cpu_baseline:
        out.update(cpu_baseline(pkg, t, lqr, z0, T))
    if world == 1 and not args.no_extra:
        out["extra"] = extra_workloads(pkg, capi, torch, dev, setup_s, lqr, mx, mu, ml, T, args)
        if record and chunks == 1:
            try:
                out["extra"]["headline_workload_newton_mode1"] = newton_mode1_line(capi, torch, dev, mh, ctrl, z0_d, traj_d, n_inst, T, nb, value)
            except Exception as e:
                out["extra"]["headline_workload_newton_mode1"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)


def newton_mode1_line(capi, torch, dev, mh, ctrl, z0_d, traj_exact, n_inst, T, nb, exact_value, eps=1e-12):
    """NOT the headline: the same workload with the measured-error Newton mode of cclqr_rollout_opts (newton_mode 1: a solve also stops as soon
    as ||f|| < eps_alone, instead of iterating on until the step it takes is below 1e-10 too), with its max state deviation from the
    exact-rule trajectories of the timed headline run measured here, over all instances and steps"""
    zT = torch.empty_like(z0_d)
    st = torch.zeros(n_inst, dtype=torch.int32, device=dev)
    traj = torch.empty_like(traj_exact)
    stream = torch.cuda.current_stream().cuda_stream
    run = lambda: capi.rollout_dev(mh, ctrl, n_inst, T, 1, z0_d.data_ptr(), 0, 0, 0, traj.data_ptr(), zT.data_ptr(), st.data_ptr(), stream,
                                   newton_mode=1, newton_eps_alone=eps)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    dev_max = float((traj - traj_exact).abs().max().item())
    bad = int((st <= 0).sum().item())
    return {"value": (n_inst * T / (ms * 1e-3)) if bad == 0 else None, "unit": "instance-steps/s", "ms_per_rollout": ms, "failed_instances": bad,
            "newton_eps_alone": eps, "max_state_deviation_vs_exact_rule": dev_max, "speedup_vs_exact_rule": (n_inst * T / (ms * 1e-3)) / exact_value,
            "workload": "the headline workload with cclqr_rollout_opts.newton_mode = 1 (stop on ||f|| < %g alone): a measured-error OPTION; "
                        "the headline value above is the exact stopping rule" % eps}


def chain_rate(pkg, capi, torch, dev, n_links, n_inst, T, count=True):
    """the headline workload's recipe for another chain length (north_star: "16-body chain mechanisms" = N = 15 links)"""
    ex, mech, zd, z0 = build_workload(pkg, n_links, n_inst, 0, 0)
    t = mech.tables()
    t0 = time.time()
    lq = pkg.LQR(mech, [pkg.getid(b) for b in ex["bodies"]], [pkg.getid(ex["ctrl"][0])], ex["Q"], ex["R"], T * t.dt,
                 xd=[zd[i, 0:3] for i in range(t.nb)], qd=[zd[i, 3:7] for i in range(t.nb)])
    setup = time.time() - t0
    mh = mech._cclqr_handle
    ctrl = lq._ctrl_handle(mh)
    f = flops_per_instance_step(t, dict(ctrl_joint=lq.ctrl_joints, K=lq.K, N=lq.N, zd=lq.zd), z0, T) if count else None
    out = dict(_timed_rollout(capi, torch, dev, mh, ctrl, z0, T, True, reps=2, f_step=f, kernel=kernel_name(mh)), lqr_construct_s=setup, riccati_kbreak=int(lq.kbreak),
               workload="lqr_cartpole_n_pendulum N=%d links (%d bodies), hanging-equilibrium LQR, Q=I R=1 horizon %gs, y0~U(-0.5,0.5) phi_i~U(-0.2,0.2), "
                        "%d instances, record=true" % (n_links, t.nb, T * t.dt, n_inst))
    ctrl.close()
    return out


def cartpole_cfg2_workload(pkg, n=4096):
    """configs[1]: lqr_cartpole.jl, n random-init instances, the LQR of the script; -> (mech, lqr, z0, seconds the construction took)"""
    ex = pkg.examples.cartpole_n(1)
    mech = ex["mech"]
    t0 = time.time()
    lq = pkg.LQR(mech, [1, 2], [3], ex["Q"], ex["R"], 10.0, xd=ex["xd"])
    setup = time.time() - t0
    rng = np.random.default_rng(0xC0FFEE)
    z0 = pkg.examples.cartpole_states(1, rng.uniform(-0.5, 0.5, 4096), rng.uniform(0, 1 / 3, (4096, 1)))
    if n != 4096:
        z0 = np.tile(z0, ((n + 4095) // 4096, 1, 1))[:n]
    return mech, lq, z0, setup


def sawyer_cfg4_workload(pkg, spread, n=8192):
    """configs[3]: lqr_sawyer.jl, n arms with joint angles ~ U(-spread, spread) about the zero pose, the script's LQR"""
    tab = json.load(open(os.path.join(ROOT, "tests", "golden", "sawyer_arm_tables.json")))
    ex = pkg.examples.sawyer(tab)
    mech = ex["mech"]
    t0 = time.time()
    lq = pkg.LQR(mech, [pkg.getid(b) for b in mech.bodies], [pkg.getid(e) for e in mech.eqconstraints], ex["Q"], ex["R"], 20.0, xd=ex["xd"], qd=ex["qd"])
    setup = time.time() - t0
    rng = np.random.default_rng(4)
    z_script = pkg.joint_position_states(mech, rng.uniform(-0.05, 0.05, (n, 7)))
    z_in = pkg.joint_position_states(mech, rng.uniform(-0.002, 0.002, (n, 7)))
    return mech, lq, (z_script if spread >= 0.05 else z_in), setup, rng


def tracking_cfg5_workload(pkg):
    """configs[4]: trackingLQR_triple_cartpole.jl -- swing-up replay of the script's U, TrackingLQR about it, friction + Philox noise"""
    U = np.load(os.path.join(ROOT, "tests", "golden", "triple_cartpole_U.npy"))
    ex = pkg.examples.triple_cartpole()
    mech = ex["mech"]
    j1 = ex["ctrl"][0]
    z00 = mech.state()          # the zero pose every instance starts from (taken BEFORE the swing-up replay moves the mechanism)
    t0 = time.time()
    s0 = pkg.simulate(mech, pkg.Storage(1000, 4), pkg.OpenLoop(mech, [j1.id], U.reshape(1000, 1)))
    tl = pkg.TrackingLQR(mech, s0, [[[U[k]]] for k in range(1000)], [j1.id], ex["Q"], ex["R"])
    setup = time.time() - t0
    octrl = dict(ctrl_joint=tl.ctrl_joints, K=tl.K, N=tl.N, zd=tl.zd, Fd=tl.Fd, fric=ex["fric"], noise_scale=2.0, noise_seed=0xC0FFEE)
    return mech, tl, ex, octrl, setup, z00


def extra_workloads(pkg, capi, torch, dev, setup_s, lqr, mx, mu, ml, T, args):
    """not the headline: the other BASELINE configs through the same C-ABI, each with its workload string, its kernel, the kernel's duration by
    HIP events and -- unless --no-cpu-baseline -- the fp64 roofline fraction from flops counted by the instrumented oracle on that workload
    (VERDICT r3 item 2); and the rate of the setup path measured while building the headline LQR"""
    count = not args.no_cpu_baseline
    mech, lq, z0, setup2 = cartpole_cfg2_workload(pkg)
    t = mech.tables()
    mh = mech._cclqr_handle
    ctrl = lq._ctrl_handle(mh)
    f2 = flops_per_instance_step(t, dict(ctrl_joint=lq.ctrl_joints, K=lq.K, N=lq.N, zd=lq.zd), z0, 1000, n_sample=16) if count else None
    n = len(z0)
    c2 = _timed_rollout(capi, torch, dev, mh, ctrl, z0, 1000, True, reps=5, f_step=f2, kernel=kernel_name(mh))
    # the same mechanism with the device filled (4096 instances at eight per wavefront are 512 wavefronts on 1024 SIMDs)
    c2f = _timed_rollout(capi, torch, dev, mh, ctrl, np.tile(z0, (16, 1, 1)), 1000, False, reps=3, f_step=f2, kernel=kernel_name(mh))
    ctrl.close()
    more = {}
    try:
        if args.links == 16 and args.instances >= 1024:     # the default headline run: the 16-BODY chain (N = 15) next to the 17-body one
            more["chain_16_bodies_N15"] = chain_rate(pkg, capi, torch, dev, 15, args.instances, args.sim_steps, count)
        more.update(other_configs(pkg, capi, torch, dev, count))
        more["branching_tree_14_bodies"] = tree14_rate(pkg, capi, torch, dev, count)
        more["deltabot_closed_loops"] = deltabot_rate(pkg, capi, torch, dev, count=count)
    except Exception as e:        # the extra lines never take the headline line down with them
        more["other_configs_error"] = repr(e)
    m = mu + ml
    f_ric = 4 * mx ** 3 + 4 * mx ** 2 * m + 2 * mx * (ml ** 2 + m ** 2) + 2 / 3 * m ** 3 + 2 / 3 * ml ** 3
    nsteps = T - max(int(lqr.kbreak), 1)
    return {"cartpole_cfg2": dict(c2, workload="lqr_cartpole.jl (configs[1]): 4096 cartpoles, y0~U(-0.5,0.5) phi0~U(0,1/3), Q=I R=1 horizon 10 s, record=true",
                                  lqr_construct_s=setup2, riccati_kbreak=int(lq.kbreak),
                                  device_filled=dict(c2f, workload="the same 4096 starts tiled 16 times: 65536 instances, record=false")),
            **more,
            "riccati_setup": {"mx": mx, "backward_steps": nsteps, "flops_per_step": f_ric,
                              "note": "LQR construction of the headline workload = linearize + %d-step recursion (projected form, tiled over the device, fp64 MFMA); "
                                      "wall time incl. host<->device copies" % nsteps,
                              "gflops_lower_bound": f_ric * nsteps / setup_s / 1e9, "fp64_mfma_peak_tflops": FP64_PEAK_TFLOPS}}


def kernel_name(mh, extra=0):
    """the instantiation a mechanism's rollouts run (extra: 0 plain LQR / TrackingLQR law, 1 + friction and noise, 2 + PID)"""
    lanes, _ = mh.geometry()
    par = np.asarray(mh.tables.parent)
    branching = mh.tables.ne == mh.tables.nb and (np.bincount(par[par >= 0], minlength=1) > 1).any()
    if branching:
        return "rollout_treereg_kernel<%d, %d, %d, false>" % (lanes, mh.layout_links(), extra)
    return "rollout_chain_kernel<%d, %d, %d, false, %d, %d>" % ((lanes, mh.layout_links(), extra) + mh.lanes_per_link())


def flops_per_instance_step(t, octrl_kw, z0, steps, n_sample=4):
    """fp64 flops of one instance-step of a workload, counted by the instrumented build of the oracle (the checker, never the product) on an
    n_sample-instance x <= 200-step sample of the same inputs (F_step is counted, not estimated)"""
    from oracle import orc
    sample_steps = min(steps, 200)
    orc.flops_reset()
    orc.rollout(t, orc.ctrl_desc(t.nb, **octrl_kw), z0[:n_sample], sample_steps, nthreads=1, flops=True)
    return orc.flops_get() / (float(min(n_sample, len(z0))) * sample_steps)


def roofline_fp64(f_step, n, steps, kernel_ms, kernel):
    """the binding roofline of the rollout kernels (fp64 vector ALU): counted flops per launch / the launch's duration by HIP events"""
    tf = f_step * float(n) * steps / (kernel_ms * 1e-3) / 1e12
    return {"bound": "fp64_valu", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
            "flops_per_instance_step_counted_by_oracle": f_step, "kernel": kernel, "kernel_ms": kernel_ms}


def _timed_rollout(capi, torch, dev, mh, ctrl, z0, steps, record, reps=3, allow_failed=False, f_step=None, kernel=None, flags=0):
    n, nb = z0.shape[0], z0.shape[1]
    z0_d = torch.from_numpy(np.ascontiguousarray(z0)).to(dev)
    zT_d = torch.empty_like(z0_d)
    st_d = torch.zeros(n, dtype=torch.int32, device=dev)
    traj_d = torch.empty((n, steps, nb, 13), dtype=torch.float64, device=dev) if record else None
    stream = torch.cuda.current_stream().cuda_stream
    run = lambda: capi.rollout_dev(mh, ctrl, n, steps, 1, z0_d.data_ptr(), 0, 0, 0, traj_d.data_ptr() if record else 0, zT_d.data_ptr(), st_d.data_ptr(), stream, flags=flags)
    run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(); run(); b.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))      # (a noise_philox controller's launch includes its Philox fill kernel)
    bad = int((st_d <= 0).sum().item())
    out = {"instances": n, "sim_steps": steps, "record": bool(record), "value": (n * steps / dt) if bad == 0 else None, "unit": "instance-steps/s",
           "ms_per_rollout": 1e3 * dt, "failed_instances": bad, "kernel_ms": kernel_ms}
    if kernel:
        out["kernel"] = kernel
    if f_step is not None and bad == 0:
        out["roofline_fp64_valu"] = roofline_fp64(f_step, n, steps, kernel_ms, kernel)
    if bad and allow_failed:      # a rate over rollouts that partly left the integrator's domain is NOT a throughput: labelled, never `value`
        out["attempted_instance_steps_per_s_incl_failed"] = n * steps / dt
    return out


TREE14_PARENTS = [-1, 0, 1, 2, 3, 2, 5, 6, 1, 8, 8, 10, 0, 12]


def tree14_workload(pkg, n=32768, steps=300):
    """a BRANCHING mechanism (not a BASELINE config): 14 bodies, body i on body TREE14_PARENTS[i] by a revolute (bodies 0 and 5: prismatic)
    joint with random axis and anchors (examples.tree_mechanism, seed 4), every instance from the placed pose, random gains of scale 0.02 on joint 0"""
    ex = pkg.examples.tree_mechanism(TREE14_PARENTS, seed=4, prismatic=(0, 5))
    mech = ex["mech"]
    t, z00 = mech.tables(), mech.state()
    K = np.random.default_rng(0).normal(size=(steps + 5, 1, 12 * t.nb)) * 0.02
    octrl = dict(ctrl_joint=[0], K=K, N=steps + 6, zd=z00)
    return t, octrl, np.tile(z00[None], (n, 1, 1)), steps


def tree14_rate(pkg, capi, torch, dev, count=True):
    t, octrl, z0, steps = tree14_workload(pkg)
    mh = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mh, octrl["ctrl_joint"], K=octrl["K"], N=octrl["N"], zd=octrl["zd"])
    f = flops_per_instance_step(t, octrl, z0, steps) if count else None
    lanes, lds = mh.geometry()
    out = dict(_timed_rollout(capi, torch, dev, mh, ctrl, z0, steps, False, f_step=f, kernel=kernel_name(mh)), lanes_per_instance=lanes, lds_bytes_per_workgroup=lds,
               workload="branching tree of 14 bodies (parents %s, tools/gpu_tree_rate.py), %d instances x %d steps, record=false: the register-resident tree "
                        "kernel of round 4 (8.0 M inst-steps/s on the LDS-resident kernel of rounds 1-3)" % (TREE14_PARENTS, len(z0), steps))
    ctrl.close(); mh.close()
    return out


def deltabot_workload(pkg, capi, n=32768):
    """examples/lqr_deltabot.jl:25-53 as a batch (not a BASELINE config): five bodies, seven joints, three closed kinematic loops,
    feedback on the two actuated joints with a setpoint-holding torque scaled 0.97..1.03 per instance (tools/gpu_loop_rate.py)"""
    ex = pkg.examples.deltabot()
    mech_py = ex["mech"]
    t = mech_py.tables()
    cj = [mech_py.joint_index(e) for e in ex["eqcids"]]
    z00 = mech_py.state()
    rng = np.random.default_rng(0)
    K = rng.normal(size=(1, 2, 12 * t.nb)) * 0.05
    scale = rng.uniform(0.97, 1.03, n)
    mh = capi.MechHandle(t)
    ctrl = capi.CtrlHandle(mh, cj, K=np.repeat(K[None], n, 0), N=0, zd=np.repeat(z00[None, None], n, 0), Fd=scale[:, None, None] * ex["Fd"].reshape(1, 1, 2), n_ctrl=n)
    return mh, ctrl, np.tile(z00[None], (n, 1, 1)), "rollout_loop_kernel<%d>" % ((5 * t.ne + 7) // 8)


def deltabot_flops_per_instance_step(pkg, n_sample=2, steps=8):
    """algorithmic fp64 flops of one instance-step of the deltabot workload, counted on the dense-KKT checker (oracle/loops.py flops_per_step: the tree
    oracle's own bookkeeping for the evaluations, measured on liborc_flops.so, + the dense Schur assembly and LU a loop needs; iterations and line-search
    evaluations counted under the parity stopping rule) on n_sample instances x steps steps under the SAME law: u = scale_i Fd - K dz"""
    from oracle import loops
    ex = pkg.examples.deltabot()
    lm = loops.from_tables(ex["mech"].tables())
    z00 = ex["mech"].state()
    cj = [ex["mech"].joint_index(e) for e in ex["eqcids"]]
    rng = np.random.default_rng(0)
    K = rng.normal(size=(1, 2, 12 * lm.nb)) * 0.05
    scale = rng.uniform(0.97, 1.03, 32768)
    tot, cnt = 0.0, 0
    for i in range(n_sample):
        z, lam = z00.copy(), np.zeros(lm.nrows)
        for k in range(steps):
            u = np.zeros(len(lm.joints))
            u[cj] = scale[i] * ex["Fd"].reshape(2) - K[0] @ loops.state_error(z, z00)
            fl, z, lam, its = loops.flops_per_step(lm, z, lam, u, ctrl_flops=2.0 * 2 * 12 * lm.nb)
            tot += fl; cnt += 1
    return tot / cnt


def deltabot_rate(pkg, capi, torch, dev, n=32768, steps=200, count=True):
    """Roofline figure since round 5: the checker of this kernel (oracle/loops.py, numpy dense-KKT) got a flop counter -- see deltabot_flops_per_instance_step."""
    mh, ctrl, z0, kern = deltabot_workload(pkg, capi, n)
    lanes, lds = mh.geometry()
    f = deltabot_flops_per_instance_step(pkg) if count else None
    out = dict(_timed_rollout(capi, torch, dev, mh, ctrl, z0, steps, False, f_step=f, kernel=kern), lanes_per_instance=lanes, lds_bytes_per_workgroup=lds,
               workload="lqr_deltabot.jl as a batch: %d deltabots (5 bodies, 7 joints, 35 constraint rows of rank 28) x %d steps, feedback on the two actuated "
                        "joints, record=false: the closed-loop kernel with the register-resident Gauss-Jordan solve of round 4 (4.6 M inst-steps/s with the "
                        "LDS-resident complete-pivoting solve of rounds 2-3)" % (n, steps))
    ctrl.close(); mh.close()
    return out


def other_configs(pkg, capi, torch, dev, count=True):
    """BASELINE configs[3] and configs[4] at their full sizes through the same C-ABI (not the headline; one entry each, every entry names
    its workload).  lqr_sawyer.jl: 8192 seven-joint arms, horizon 20 s, (a) at SURVEY 8d's joint angles ~ U(-0.05, 0.05) about the zero
    pose -- about 30 % of these starts leave the script controller's region of attraction ("Currently somewhat broken",
    lqr_sawyer.jl:1) and come back flagged: failed_instances is in the line and no rate is claimed; (b) at U(-0.002, 0.002), where every
    start converges; (c) with a setpoint PER INSTANCE: 8192 distinct infinite-horizon LQRs built on the device in one call.
    trackingLQR_triple_cartpole.jl: 16384 instances, TrackingLQR about the swing-up of the script's own input U, friction + Philox cart noise."""
    out = {}
    n = 8192
    mech, lq, z_script, setup, rng = sawyer_cfg4_workload(pkg, 0.05, n)
    z_in = pkg.joint_position_states(mech, rng.uniform(-0.002, 0.002, (n, 7)))
    t = mech.tables()
    mh = mech._cclqr_handle
    ctrl = lq._ctrl_handle(mh)
    kern = kernel_name(mh)
    f4 = flops_per_instance_step(t, dict(ctrl_joint=lq.ctrl_joints, K=lq.K, N=lq.N, zd=lq.zd), z_in, 2000) if count else None
    out["sawyer_cfg4"] = dict(_timed_rollout(capi, torch, dev, mh, ctrl, z_script, 2000, False, allow_failed=True, kernel=kern), lqr_construct_s=setup,
                              riccati_kbreak=int(lq.kbreak),
                              workload="lqr_sawyer.jl (configs[3]): 8192 Sawyer arms, joint angles ~ U(-0.05, 0.05) rad about the zero pose (SURVEY 8d), "
                                       "Q=1000 I R=1 g=0 horizon 20 s, 2000 steps; starts outside the script controller's region of attraction are frozen "
                                       "and flagged (the oracle loses the same set: tests/test_gpu_fullsize.py)")
    out["sawyer_cfg4_inside_region_of_attraction"] = dict(_timed_rollout(capi, torch, dev, mh, ctrl, z_in, 2000, False, f_step=f4, kernel=kern),
                                                          workload="the same with joint angles ~ U(-0.002, 0.002) rad: every start converges")
    ctrl.close()
    # the same arm with a setpoint PER INSTANCE (SURVEY 8d configs[3]: "Riccati run per instance on distinct setpoints"): 8192 poses, ONE
    # batched infinite-horizon LQR construction on the device (linearsystem + dlqr of up to 999 backward steps each + per-instance tables;
    # only Ku[1] per setpoint is kept, lqr.jl:40-43: 39 MB of gains), one rollout of 2000 steps
    rng = np.random.default_rng(44)
    ang, off = rng.uniform(-0.8, 0.8, (n, 7)), rng.uniform(-0.002, 0.002, (n, 7))
    zdb, z0b = pkg.joint_position_states(mech, ang), pkg.joint_position_states(mech, ang + off)
    t0 = time.time()
    # iteration cap 2000 (the script's horizon 20 s) instead of LQR{T,Inf}'s default ceil(10/Δt) = 1000 (lqr.jl:26): with the script's
    # weights the recursion needs ~1590 backward steps to meet the 1e-5 test of lqr.jl:172
    ncap = 2000
    bl = capi.BatchLqrHandle(mh, zdb, list(range(7)), lq.Q, lq.R, ncap, infinite_horizon=True)
    setup_b = time.time() - t0
    kb = bl.kbreak
    steps_run = int((ncap - 1 - np.maximum(kb, 1) + 1).sum())
    m = 7 + 35
    f_ric = 4 * 84 ** 3 + 4 * 84 ** 2 * m + 2 * 84 * (35 ** 2 + m ** 2) + 2 / 3 * m ** 3 + 2 / 3 * 35 ** 3
    out["sawyer_cfg4_setpoint_per_instance"] = dict(
        _timed_rollout(capi, torch, dev, mh, bl, z0b, 2000, False, f_step=f4, kernel=kern), batched_lqr_construct_s=setup_b, gain_table_bytes_in_hbm=int(n) * 7 * 84 * 8,
        riccati_kbreak_min_max=[int(kb.min()), int(kb.max())], riccati_not_converged=int((kb <= 1).sum()), riccati_backward_steps_total=steps_run,
        riccati_tflops_lower_bound_incl_linearize_and_copies=f_ric * steps_run / setup_b / 1e12,
        riccati_gains_per_s=int(n) / setup_b,      # LQR constructions (linearsystem + dlqr to convergence) per second, whole call
        workload="8192 Sawyer arms, each regulated about its OWN pose (joint angles ~ U(-0.8, 0.8) rad) by its own LQR{T,Inf}: "
                 "cclqr_ctrl_create_lqr_batch(infinite_horizon) = 8192 linearsystem + 8192 dlqr (mx 84, mu 7, ml 35, <= 1999 steps, fp64 MFMA), "
                 "starts within 0.002 rad of the setpoint, 2000 steps")
    bl.close()
    mech, tl, ex, octrl5, setup, z00 = tracking_cfg5_workload(pkg)
    mh = mech._cclqr_handle
    ctrl = tl._ctrl_handle(mh, fric=ex["fric"], noise_scale=2.0, noise_seed=0xC0FFEE)
    f5 = flops_per_instance_step(mech.tables(), octrl5, np.tile(z00, (4, 1, 1)), 1000) if count else None
    out["triple_cartpole_tracking_cfg5"] = dict(_timed_rollout(capi, torch, dev, mh, ctrl, np.tile(z00, (16384, 1, 1)), 1000, True, f_step=f5, kernel=kernel_name(mh, 1)),
                                                swingup_plus_trackinglqr_construct_s=setup,
                                                workload="trackingLQR_triple_cartpole.jl (configs[4]): 16384 triple cartpoles from the zero pose, TrackingLQR about the "
                                                         "swing-up of the script's input U (999 knots linearised + time-varying dlqr on the device), friction 0.1 + "
                                                         "2 N(0,1) cart noise from Philox-4x32 per (instance, step), 1000 steps, record=true")
    # configs[4] also asks for a "hipGraph-captured step": the same 1000 steps as 1000 single-step launches (state and multipliers round-trip
    # HBM between them, Philox samples generated per launch into a workspace sized beforehand) captured ONCE into a hipGraph and replayed --
    # the shape an MPC loop has, with the caller free to look at every state
    try:
        out["triple_cartpole_tracking_cfg5"]["step_per_launch_in_a_hip_graph"] = _graph_captured_steps(capi, torch, dev, mh, ctrl, np.tile(z00, (16384, 1, 1)), 1000, mech.tables().ne)
    except Exception as e:
        out["triple_cartpole_tracking_cfg5"]["step_per_launch_in_a_hip_graph"] = {"error": repr(e)}
    ctrl.close()
    try:
        out["triple_cartpole_tracking_cfg5"]["host_closure_controlfunction"] = _host_closure_rate(pkg, capi, torch, mech, tl, ex, np.tile(z00, (16384, 1, 1)), 100)
    except Exception as e:
        out["triple_cartpole_tracking_cfg5"]["host_closure_controlfunction"] = {"error": repr(e)}
    try:
        out["triple_cartpole_tracking_cfg5"]["device_closure_controlfunction"] = _device_closure_rate(pkg, capi, torch, mech, tl, ex, np.tile(z00, (16384, 1, 1)), 1000)
    except Exception as e:
        out["triple_cartpole_tracking_cfg5"]["device_closure_controlfunction"] = {"error": repr(e)}
    return out


def _graph_captured_steps(capi, torch, dev, mh, ctrl, z0, steps, ne, branches=(1, 2, 4)):
    """configs[4]'s "hipGraph-captured step": `steps` single-step launches (state and multipliers round-trip HBM between them, the Philox sample
    generated inside the step kernel) captured once and replayed.  branches = 1: ONE chain of launches over the whole batch -- every step waits for
    the slowest wavefront of the step before (the Newton iteration counts of the noise-floor line searches are heavy-tailed), which a persistent
    launch never does.  branches = B: the batch cut into B independent sub-batches, each its own chain of single-step launches on its own captured
    stream (parallel branches of one graph): a straggler only holds up its own sub-batch while the other chains keep the device busy.  Same
    launches, same bits; what an MPC loop that looks at every state of a sub-batch would do."""
    n = z0.shape[0]
    nb13 = z0.shape[1] * 13 * 8
    z0_d = torch.from_numpy(np.ascontiguousarray(z0)).to(dev)
    ref, st = torch.empty_like(z0_d), torch.zeros(n, dtype=torch.int32, device=dev)
    capi.rollout_dev(mh, ctrl, n, steps, 1, z0_d.data_ptr(), 0, 0, 0, 0, ref.data_ptr(), st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    failed_fused = int((st <= 0).sum().item())      # the persistent launch's status covers the whole horizon (a lost instance stays flagged)
    za, zb = z0_d.clone(), torch.empty_like(z0_d)
    lam = torch.zeros((n, 5 * ne), dtype=torch.float64, device=dev)
    res = {}
    for B in branches:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        subs = [torch.cuda.Stream() for _ in range(B)] if B > 1 else [side]
        per = (n + B - 1) // B
        side.wait_stream(torch.cuda.current_stream())
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            graph.capture_begin()
            for b in range(B):
                lo, cnt = b * per, min(per, n - b * per)
                if cnt <= 0:
                    continue
                if B > 1:
                    subs[b].wait_stream(side)
                src, dst = za, zb
                for k in range(1, steps + 1):
                    capi.rollout_dev(mh, ctrl, cnt, 1, k, src.data_ptr() + lo * nb13, lam.data_ptr() + lo * 5 * ne * 8, 0, 0, 0, dst.data_ptr() + lo * nb13,
                                     st.data_ptr() + lo * 4, subs[b].cuda_stream, first_instance=lo, flags=capi.ROLLOUT_NO_ALLOC | capi.ROLLOUT_CARRY_STATUS)
                    src, dst = dst, src
            if B > 1:
                for b in range(B):
                    side.wait_stream(subs[b])
            graph.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        capture_s = time.perf_counter() - t0
        times = []
        for _ in range(3):
            za.copy_(z0_d)
            lam.zero_()
            st.zero_()                  # (CCLQR_ROLLOUT_CARRY_STATUS: the array carries every instance's status through the captured launches)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        dt = min(times[1:])
        # The captured launches carry the status array from step to step (CCLQR_ROLLOUT_CARRY_STATUS, round 5): what is left there speaks for the whole
        # horizon and a lost instance stays frozen, as in the persistent launch.  A rate is claimed only when nobody failed AND the graph's final state
        # equals the persistent launch's bit for bit.
        same = bool(torch.equal(src, ref))
        failed_graph = int((st <= 0).sum().item())
        ok = same and failed_fused == 0 and failed_graph == 0
        res[B] = {"value": (n * steps / dt) if ok else None, "ms_per_rollout": 1e3 * dt, "attempted_instance_steps_per_s": n * steps / dt, "capture_s": capture_s,
                  "graph_nodes": "%d chain(s) x %d launches of one step (Philox samples generated in the step kernel: no fill launches)" % (B, steps),
                  "same_bits_as_one_persistent_launch": same, "failed_instances_over_the_horizon": failed_graph}
        del graph
    best = max(res, key=lambda b: res[b]["attempted_instance_steps_per_s"])
    out = {"instances": n, "sim_steps": steps, "record": False, "unit": "instance-steps/s", "failed_instances_of_the_persistent_launch_of_the_same_horizon": failed_fused}
    out.update(res[best])
    out["independent_chains_in_the_graph"] = best
    out["by_independent_chains"] = {str(b): res[b] for b in res}
    return out


def _host_closure_rate(pkg, capi, torch, mech, tl, ex, z0, steps):
    """the `controlfunction` hook (lqr.jl:14, :56; examples/trackingLQR_triple_cartpole.jl:93-117) on the same workload: the script's law as a HOST
    closure -- control_trackinglqr! + friction on every joint -- stepped one launch per step, states to the host and inputs back every step
    (lqr.py::_simulate_hosted).  No noise: the closure owns the law, and a host RNG stream is not the device's."""
    import copy
    fric = np.asarray(ex["fric"], dtype=np.float64)
    t = mech.tables()
    parent, child = np.asarray(t.parent), np.asarray(t.child)

    def law(batch, ctrl, k):
        pkg.control_lqr(batch, ctrl, k)
        for j, e in enumerate(mech.eqconstraints):      # viscous friction -fric * relative joint velocity (the script's :99-101)
            if fric[j] == 0.0:
                continue
            if int(t.type[j]) == 1:
                rel = batch.v[:, child[j], 1] - (batch.v[:, parent[j], 1] if parent[j] >= 0 else 0.0)
            else:
                rel = batch.ω[:, child[j], 0] - (batch.ω[:, parent[j], 0] if parent[j] >= 0 else 0.0)
            batch.u[j] = batch.u.get(j, np.zeros(batch.n_inst)) - fric[j] * rel
    tc = copy.copy(tl)
    tc.controlfunction = law
    t0 = time.perf_counter()
    st = pkg.simulate(mech, steps * mech.Δt, tc, record=False, z0=z0)
    dt = time.perf_counter() - t0
    ok = bool((st.status > 0).all())
    return {"instances": int(z0.shape[0]), "sim_steps": steps, "value": (z0.shape[0] * steps / dt) if ok else None, "unit": "instance-steps/s", "s_per_run": dt,
            "what": "host closure controlfunction(batch, controller, k) = control_trackinglqr! + joint friction in numpy, one launch per step, states D2H and inputs H2D "
                    "every step through cclqr_ctrl_set_feedforward (PCIe and host arithmetic inside the time)"}


def _device_closure_rate(pkg, capi, torch, mech, tl, ex, z0, steps):
    """the same hook as a DEVICE closure (cclqr.on_device: lqr.py::_simulate_device_closure): the script's law -- control_trackinglqr! + friction on every
    joint -- written in torch on the batch's state tensor in HBM; single-step launches, the closure's inputs handed over device to device on the same
    stream, no host round trip per step.  No noise (the closure owns the law)."""
    import copy
    fric = torch.tensor(np.asarray(ex["fric"], dtype=np.float64), device="cuda")
    t = mech.tables()
    parent, child = [int(x) for x in t.parent], [int(x) for x in t.child]
    typ = [int(x) for x in t.type]
    fr = [float(x) for x in ex["fric"]]

    @pkg.on_device
    def law(batch, ctrl, k):
        pkg.control_lqr(batch, ctrl, k)
        for j in range(len(fr)):
            if fr[j] == 0.0:
                continue
            comp = (batch.v, 1) if typ[j] == 1 else (batch.ω, 0)
            rel = comp[0][:, child[j], comp[1]] - (comp[0][:, parent[j], comp[1]] if parent[j] >= 0 else 0.0)
            prev = batch.u.get(j)
            batch.u[j] = (prev if prev is not None else 0.0) - fr[j] * rel
    tc = copy.copy(tl)
    tc.controlfunction = law
    pkg.simulate(mech, 5 * mech.Δt, tc, record=False, z0=z0)          # warm-up: the tables go to the device once
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = pkg.simulate(mech, steps * mech.Δt, tc, record=False, z0=z0)
    dt = time.perf_counter() - t0
    ok = bool((st.status > 0).all())
    out = {"instances": int(z0.shape[0]), "sim_steps": steps, "value": (z0.shape[0] * steps / dt) if ok else None, "unit": "instance-steps/s", "s_per_run": dt,
           "what": "device closure (cclqr.on_device) controlfunction(batch, controller, k) = control_trackinglqr! + joint friction in torch on the state tensor in HBM, "
                   "one launch per step, inputs handed over device to device (wall time of simulate incl. the upload of the states and the final download)"}
    # the same closure with graph=True: the horizon captured by the first call, replayed by the second (what a sweep over initial states pays per batch)
    tg = copy.copy(tl)
    tg.__dict__.pop("_device_closure_runs", None)
    tg.controlfunction = pkg.on_device(lambda b, c, k: law(b, c, k), graph=True)
    t0 = time.perf_counter()
    s0 = pkg.simulate(mech, steps * mech.Δt, tg, record=False, z0=z0)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    s1 = pkg.simulate(mech, steps * mech.Δt, tg, record=False, z0=z0)
    t_replay = time.perf_counter() - t0
    same = bool(np.array_equal(s1.zT, st.zT) and np.array_equal(s0.zT, st.zT))
    out["captured_in_a_hip_graph"] = {"value": (z0.shape[0] * steps / t_replay) if (ok and same) else None, "unit": "instance-steps/s", "s_per_replayed_run": t_replay,
                                      "s_first_run_with_capture": t_first, "same_bits_as_the_eager_loop": same}
    for r in tg.__dict__.get("_device_closure_runs", {}).values():
        r.close()
    return out


def build_native_oracle():
    """the CPU baseline build of the oracle: -O3 -march=native, compiled HERE (on the box that runs the bench: a -march=native
    object built elsewhere may not run), next to the portable -O2 checker build that the tests use"""
    src = os.path.join(ROOT, "oracle", "cclqr_oracle.c")
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liborc_native.so")
    flags = ["-O3", "-march=native", "-fPIC", "-fopenmp", "-std=gnu11", "-ffp-contract=off"]
    subprocess.check_call(["gcc"] + flags + ["-shared", "-o", so, src, "-lm"])
    return so, " ".join(flags)


def count_flops(pkg, t, lqr, z0, T):
    """fp64 flops of one instance-step of the headline workload (flops_per_instance_step on its LQR)"""
    return flops_per_instance_step(t, dict(ctrl_joint=lqr.ctrl_joints, K=lqr.K, N=lqr.N, zd=lqr.zd), z0, T)


def cpu_baseline(pkg, t, lqr, z0, T):
    """oracle (CPU restatement, NOT ConstrainedControl.jl itself) on a bounded sample of the same workload on the host cores this
    process is allowed to use; reports the single-thread rate and the parallel efficiency next to the all-core rate"""
    from oracle import orc
    cores, affinity, quota = host_cpu_allowance()
    flags = "-O2 (oracle/Makefile)"
    try:
        so, flags = build_native_oracle()
        orc.use_library(so)
    except Exception as e:      # no compiler on the box: the portable build is the baseline
        sys.stderr.write("bench: native oracle build failed (%s); using oracle/liborc.so\n" % (e,))
    octrl = orc.ctrl_desc(t.nb, lqr.ctrl_joints, K=lqr.K, N=lqr.N, zd=lqr.zd)
    sample_steps = min(T, 200)
    # single thread: 2 instances x sample_steps
    t0 = time.time()
    orc.rollout(t, octrl, z0[:2], sample_steps, nthreads=1)
    v_one = 2 * sample_steps / (time.time() - t0)
    n_s = max(cores, min(len(z0), int(15.0 * cores * v_one / sample_steps)))      # ~15 s of work on all cores
    n_s = (n_s // cores) * cores
    t0 = time.time()
    orc.rollout(t, octrl, z0[:n_s], sample_steps, nthreads=cores)
    v_all = n_s * sample_steps / (time.time() - t0)
    orc.use_library(None)
    return {
        "cpu_baseline": {"value": v_all, "unit": "instance-steps/s", "cores": cores, "kind": "port",
                         "single_thread_value": v_one, "parallel_efficiency": v_all / (cores * v_one),
                         "affinity_cores": affinity, "cgroup_cpu_quota": quota, "build_flags": flags,
                         "sample": "%d instances x %d steps of the same workload, the oracle (OpenMP over instances, %d threads)" % (n_s, sample_steps, cores)},
    }


if __name__ == "__main__":
    main()
