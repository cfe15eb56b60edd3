#!/bin/bash
# rocprofv3 evidence for profiles/rNN (run on the GPU box through gpurun): kernel trace + stats of the default bench command,
# HBM traffic counters and the instruction-mix counters, each in its own pass (counters never together with a trace).
set -e
# (a pass that runs into its timeout stops the script: nothing is profiled after a GPU step that had to be killed)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof_r02}
mkdir -p $OUT
# Rule: the CPU suite -- which holds the register/spill guards of every kernel -- runs before anything is
# profiled or committed; a red suite aborts the round's profile.
if [ "${SKIP_CPU_SUITE:-0}" != "1" ]; then
  (cd $R && python3 -m pytest tests -x -q -m "not gpu" > $OUT/cpu_suite.log 2>&1) || { echo "CPU suite red: see $OUT/cpu_suite.log"; tail -5 $OUT/cpu_suite.log; exit 1; }
fi
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmcA -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/pmcA.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/pmcB -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/pmcB.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmcC -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/pmcC.log 2>&1
# summaries are made here, on the box; the raw per-dispatch counter files (tens of MiB) do not travel back
python3 tools/profile_summarize.py $OUT ${2:-r03} ${3:-bench} $OUT/summary > $OUT/summarize.log 2>&1 || tail -5 $OUT/summarize.log
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*kernel_stats.csv" | head -2
