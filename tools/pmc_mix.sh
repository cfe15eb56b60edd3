set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d gpurun_out/r2_pmcA -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2_pmcA.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r2_pmcB -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2_pmcB.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/r2_pmcC -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2_pmcC.log 2>&1
ls gpurun_out/r2_pmcA/*/ | head
