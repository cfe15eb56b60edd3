"""turn the output of tools/profile_round.sh (gpurun_out/<dir>) into the committed evidence under profiles/<round>/ (diagnostic, not a test)"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

src, rnd = sys.argv[1], sys.argv[2]                     # e.g. gpurun_out/prof_r03 r03
tag = sys.argv[3] if len(sys.argv) > 3 else "bench"
# argv[4]: where the summary goes.  Default profiles/<round>/ (run here, on files merged back from the box); tools/profile_round.sh passes
# gpurun_out/<dir>/summary so that it runs ON THE BOX and only the summaries travel back (the raw counter CSVs are tens of MiB)
KERNEL = "rollout_chain_kernel<32, 17, 0, false, 1, 32>"
WAVES, STEPS, NINST = 4096.0, 1000.0, 8192.0
dst = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)

def counters(sub):
    tot = collections.defaultdict(float)
    for f in glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    return dict(tot)

for f in glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, tag + "_kernel_stats.csv"))
for sub in ("pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if "rollout_chain_kernel" in r["Kernel_Name"]]
        with open(os.path.join(dst, "%s_%s.csv" % (tag, sub)), "w", newline="") as o:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
            w.writeheader(); w.writerows(rows)
line = [x for x in open(os.path.join(src, "trace.log")) if x.startswith("{")][-1]
open(os.path.join(dst, tag + ".json"), "w").write(line)

fetch, write = counters("pmc_fetch").get("FETCH_SIZE", 0.0), counters("pmc_write").get("WRITE_SIZE", 0.0)
# MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half of the bytes of a streaming read
hbm = (2.0 * fetch + write) * 1024.0
traffic = {"config": {"links": 16, "instances_per_gpu": 8192, "sim_steps": 1000, "record": True}, "kernel": KERNEL,
           "kernel_source_sha": bench.kernel_source_sha(), "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": 6672.0 * NINST * STEPS,
           "note": "2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, one launch of the default bench workload; separate --pmc passes"}
if len(sys.argv) <= 4:
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)

tot = {}
for sub in ("pmcA", "pmcB", "pmcC"):
    tot.update(counters(sub))
per = {k: v / (WAVES * STEPS) for k, v in tot.items()}
wave_cycles = 4.0 * per.get("SQ_WAVE_CYCLES", 0.0)
f64 = per.get("SQ_INSTS_VALU_FMA_F64", 0) + per.get("SQ_INSTS_VALU_MUL_F64", 0) + per.get("SQ_INSTS_VALU_ADD_F64", 0) + per.get("SQ_INSTS_VALU_TRANS_F64", 0)
lanes = per.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * max(per.get("SQ_ACTIVE_INST_VALU", 1), 1e-9)) * 64.0
mix = {"command": "rocprofv3 --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (three passes; kernel %s, %d wavefronts of two instances x %d steps)" % (KERNEL, WAVES, STEPS),
       "totals": tot, "per_wavefront_step": per,
       "derived": {"cycles_per_wavefront_step": wave_cycles, "valu_issue_share": per.get("SQ_ACTIVE_INST_VALU", 0) / max(per.get("SQ_WAVE_CYCLES", 1), 1e-9),
                   "lds_issue_share": per.get("SQ_ACTIVE_INST_LDS", 0) / max(per.get("SQ_WAVE_CYCLES", 1), 1e-9),
                   "waiting_on_counters_share": per.get("SQ_WAIT_ANY", 0) / max(per.get("SQ_WAVE_CYCLES", 1), 1e-9),
                   "f64_share_of_valu": f64 / max(per.get("SQ_INSTS_VALU", 1), 1e-9), "active_lanes_per_valu_instruction": lanes,
                   "valu_instructions_per_instance_step": per.get("SQ_INSTS_VALU", 0) / 2.0,
                   "fp64_flops_per_instance_step_from_counters": (2 * per.get("SQ_INSTS_VALU_FMA_F64", 0) + per.get("SQ_INSTS_VALU_MUL_F64", 0) + per.get("SQ_INSTS_VALU_ADD_F64", 0)) * lanes / 2.0,
                   "lds_bank_conflict_share_of_lds_active": per.get("SQ_LDS_BANK_CONFLICT", 0) / max(per.get("SQ_LDS_IDX_ACTIVE", 1), 1e-9)},
       "note": "SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md)"}
json.dump(mix, open(os.path.join(dst, "rollout_pmc_mix.json"), "w"), indent=1)
print(json.dumps({"traffic": traffic, "derived": mix["derived"]}, indent=1))
