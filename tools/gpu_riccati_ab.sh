#!/bin/bash
# A/B timing of Riccati kernel variants (GPU box): for each library name given (libcclqr_<name>.so next to the shipped one; "base" = the shipped library)
# the batched resident kernel on 1024 Sawyer problems x 199 steps under rocprofv3 --kernel-trace --stats, A/B/A/B; prints the kernel's average duration per run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ric_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
for rep in 1 2; do
  for v in "$@"; do
    rm -rf $OUT/t_$v
    CCLQR_LIB_VARIANT=$v timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$v -- python3 tools/gpu_riccati_resident.py 1024 200 > $OUT/$v.$rep.log 2>&1 || { echo "$v failed"; tail -3 $OUT/$v.$rep.log; continue; }
    f=$(find $OUT/t_$v -name "*kernel_stats.csv" | head -1)
    echo "$v run$rep: $(grep riccati_resident_kernel $f | awk -F, '{printf "%s calls, max %.3f ms", $(NF-6), $(NF-1)/1e6}')  err: $(grep -o '"gain_rel_err[^,]*' $OUT/$v.$rep.log)"
    find $OUT/t_$v -name "*kernel_trace.csv" -delete
  done
done
