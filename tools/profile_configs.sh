#!/bin/bash
# rocprofv3 evidence for the NON-headline BASELINE configs, run on the GPU box through gpurun: per config a kernel trace +
# two instruction-mix counter passes (counters never together with a trace), summarised ON THE BOX by tools/profile_configs_summarize.py.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof_cfg_r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
for cfg in ${CONFIGS:-cartpole_cfg2 cartpole_cfg2_filled sawyer_cfg4 tracking_cfg5 tree14}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg/trace -- python3 tools/gpu_config_rollout.py $cfg 3 > $OUT/$cfg.trace.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/$cfg/pmcA -- python3 tools/gpu_config_rollout.py $cfg 1 > $OUT/$cfg.pmcA.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/$cfg/pmcB -- python3 tools/gpu_config_rollout.py $cfg 1 > $OUT/$cfg.pmcB.log 2>&1
  echo "$cfg profiled"
done
python3 tools/profile_configs_summarize.py $OUT $OUT/summary > $OUT/summarize.log 2>&1 || tail -5 $OUT/summarize.log
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
ls $OUT/summary
