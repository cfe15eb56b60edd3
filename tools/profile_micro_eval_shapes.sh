#!/bin/bash
# Stage A of VERDICT r4 item 1 (run on the GPU box through gpurun): the three shapes of tools/micro/eval_shapes.hip un-profiled (timing + agreement of the results),
# then one kernel-trace pass and two counter passes (never together with a trace) whose per-kernel sums go to a small JSON made here on the box.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-micro_r05}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 200 python3 tools/gpu_micro_eval_shapes.py --out $OUT/micro_eval_shapes.json > $OUT/micro.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/gpu_micro_eval_shapes.py --quiet --launches 1 > $OUT/trace.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmcA -- python3 tools/gpu_micro_eval_shapes.py --quiet --launches 1 > $OUT/pmcA.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmcB -- python3 tools/gpu_micro_eval_shapes.py --quiet --launches 1 > $OUT/pmcB.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "micro_eval" not in k:
            continue
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        ndisp[k][row["Counter_Name"]] += 1
res = {}
for k in tot:
    c = dict(tot[k])
    nd = max(ndisp[k].values())
    r = {"dispatches_summed": nd, "counters": c}
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        r["of_the_wavefronts_cycles"] = {"issuing_any_instruction": c.get("SQ_ACTIVE_INST_ANY", 0) / wc, "issuing_a_vector_alu_instruction": c.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                                          "parked_on_a_counter_or_barrier (SQ_WAIT_ANY)": c.get("SQ_WAIT_ANY", 0) / wc, "issue_stalled (SQ_WAIT_INST_ANY)": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                                          "issuing_an_lds_instruction": c.get("SQ_ACTIVE_INST_LDS", 0) / wc}
    if c.get("SQ_INSTS_VALU"):
        r["lanes_per_vector_instruction"] = c.get("SQ_THREAD_CYCLES_VALU", 0) / c["SQ_INSTS_VALU"] if c.get("SQ_THREAD_CYCLES_VALU") else None
        r["vector_instructions_per_wavefront"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"] if c.get("SQ_WAVES") else None
    res[k] = r
json.dump(res, open(out + "/micro_eval_shapes_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/micro_eval_shapes_kernel_stats.csv 2>/dev/null || true
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/micro_eval_shapes.json
