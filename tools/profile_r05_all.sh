#!/bin/bash
# round-5 evidence in one gpurun call: the headline bench under rocprofv3 (kernel stats, FETCH / WRITE, instruction mix), every other config's kernel,
# the Riccati MFMA-busy pass, and the un-profiled default bench line.  Summaries are made on the box; raw counter files do not travel back.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
set -x
SKIP_CPU_SUITE=1 bash tools/profile_round.sh prof_r05 r05 bench_r05 > gpurun_out/prof_r05.stdout 2>&1 || tail -5 gpurun_out/prof_r05.stdout
CONFIGS="cartpole_cfg2 cartpole_cfg2_filled sawyer_cfg4 tracking_cfg5 tree14 deltabot" bash tools/profile_configs.sh prof_cfg_r05 > gpurun_out/prof_cfg_r05.stdout 2>&1 || tail -5 gpurun_out/prof_cfg_r05.stdout
cd /tmp && export TMPDIR=/tmp && cd $R
mkdir -p gpurun_out/ric_pmc_r05
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/ric_pmc_r05/pmc -- python3 tools/gpu_riccati_resident.py 1024 200 > gpurun_out/ric_pmc_r05/pmc.log 2>&1
python3 tools/profile_riccati_pmc.py gpurun_out/ric_pmc_r05/pmc gpurun_out/ric_pmc_r05/riccati_resident_mfma_pmc.json 1024 200
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ric_pmc_r05/trace -- python3 tools/gpu_riccati_resident.py 1024 200 > gpurun_out/ric_pmc_r05/trace.log 2>&1
cp $(find gpurun_out/ric_pmc_r05/trace -name "*kernel_stats.csv" | head -1) gpurun_out/ric_pmc_r05/riccati_batched_sawyer1024_kernel_stats.csv
find gpurun_out/ric_pmc_r05 -name "*counter_collection.csv" -delete; find gpurun_out/ric_pmc_r05 -name "*kernel_trace.csv" -delete
timeout -k 10 600 python3 bench.py > gpurun_out/bench_r05_final_default_run.json 2> gpurun_out/bench_r05_final_default_run.err
tail -c 400 gpurun_out/bench_r05_final_default_run.json
