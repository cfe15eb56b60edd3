"""Stage A of VERDICT r4 item 1 on the GPU box: tools/micro/eval_shapes.hip (chain_eval<JAC> + Schur rows of the 17-body headline chain in three lane
decompositions, each at the occupancy the whole kernel would have in that shape) -- checks that the three shapes compute the same norms and the same
Schur blocks, then times them with the device filled.  Usage: python tools/gpu_micro_eval_shapes.py [shape ...] [--reps N] [--out file.json]
With one shape given and --quiet it only launches (for rocprofv3 --pmc / --kernel-trace passes around it)."""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

SO = os.path.join(ROOT, "tools", "micro", "libeval_shapes.so")
SRC = os.path.join(ROOT, "tools", "micro", "eval_shapes.hip")
NAMES = {0: "A: lane = link, 2 instances / wavefront, 1 wavefront / SIMD (the shipped chain_eval<32, true>)",
         1: "B: 1 instance / wavefront, 3 lanes / link, <= 256 registers, 2 wavefronts / SIMD",
         2: "C: lane = link, 1 instance / wavefront, <= 256 registers, 2 wavefronts / SIMD"}


def build():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=fast", "--offload-arch=gfx950", "-shared", "-fPIC", SRC, "-o", SO])
    return C.CDLL(SO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="*", type=int, default=[0, 1, 2])
    ap.add_argument("--reps", type=int, default=400)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--instances", type=int, default=8192)
    ap.add_argument("--out", default=None)
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()
    pkg = graft.load_package()
    capi = pkg._capi
    L = build()
    L.micro_eval_run.restype = C.c_double
    ex = pkg.examples.cartpole_n(16)
    t = ex["mech"].tables()
    arrs = [capi.f64(t.mass), capi.f64(t.inertia), capi.i32(t.parent), capi.i32(t.child), capi.i32(t.type), capi.f64(t.p1), capi.f64(t.p2), capi.f64(t.axis), capi.f64(t.qoff)]
    d = capi.MechDesc(t.nb, t.ne, t.dt, t.g, capi._d(arrs[0]), capi._d(arrs[1]), capi._i(arrs[2]), capi._i(arrs[3]), capi._i(arrs[4]), capi._d(arrs[5]), capi._d(arrs[6]),
                      capi._d(arrs[7]), capi._d(arrs[8]))
    n = args.instances
    rng = np.random.default_rng(7)
    phi = rng.uniform(-0.2, 0.2, (n, 16))
    phi[:, 0] += np.pi
    z0 = pkg.examples.cartpole_states(16, rng.uniform(-0.5, 0.5, n), phi)      # the headline workload's starts (bench.py build_workload)
    z0[:, :, 7:] = rng.normal(size=(n, 17, 6)) * 0.05                             # + some velocity: a mid-rollout state
    z0 = np.ascontiguousarray(z0)
    res, norms, images = {}, {}, {}
    for sh in args.shapes:
        nrm = np.zeros((n, args.reps))
        img = np.zeros((n, 80 * 17))
        per_wg = 2 if sh == 0 else 1
        cyc = np.zeros((n + per_wg - 1) // per_wg, dtype=np.uint64)
        nwg, occ = C.c_int(0), C.c_int(0)
        ms = L.micro_eval_run(C.byref(d), capi._d(z0), C.c_longlong(n), C.c_int(args.reps), C.c_int(sh), C.c_int(args.launches), capi._d(nrm), capi._d(img),
                              cyc.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.byref(nwg), C.byref(occ))
        if ms < 0:
            raise SystemExit("micro_eval_run failed for shape %d" % sh)
        norms[sh], images[sh] = nrm, img
        evals = n * args.reps
        simds = 256 * 4
        # s_memtime ticks at 100 MHz on gfx950: wall time of a wavefront's loop, converted with the clock the kernel ran at is not known here;
        # cycles per evaluation are therefore derived from the kernel's duration: SIMD-slot time per evaluation of a PAIR of instances
        waves_per_simd = occ.value / 4.0
        res[sh] = {"shape": NAMES[sh], "kernel_ms": ms, "instances": n, "evaluations_per_instance": args.reps, "workgroups": nwg.value,
                   "workgroups_per_cu_by_the_occupancy_api": occ.value, "wavefronts_per_simd": waves_per_simd,
                   "instance_evaluations_per_s": evals / (ms * 1e-3),
                   "ns_of_one_simd_per_pair_of_instance_evaluations": ms * 1e6 * simds / (evals / 2.0),
                   "s_memtime_ticks_per_evaluation_median_wavefront": float(np.median(cyc)) / args.reps}
    ref = args.shapes[0]
    for sh in args.shapes[1:]:
        dn = float(np.abs(norms[sh] - norms[ref]).max() / np.abs(norms[ref]).max())
        di = float(np.abs(images[sh] - images[ref]).max() / np.abs(images[ref]).max())
        res[sh]["max_rel_diff_of_the_norms_vs_shape_%d" % ref] = dn
        res[sh]["max_rel_diff_of_the_schur_blocks_and_rhs_vs_shape_%d" % ref] = di
        assert dn < 1e-11 and di < 1e-11, (sh, dn, di)
    if 0 in res:
        for sh in res:
            res[sh]["throughput_vs_shape_A"] = res[sh]["instance_evaluations_per_s"] / res[0]["instance_evaluations_per_s"]
    if not args.quiet:
        print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
