#!/bin/bash
# kernel trace of the step-per-launch graph (configs[4]) with 1 and 2 independent chains: per-node durations and the gaps between nodes of a queue
# (tools/gpu_graph_trace_analyze.py) -> gpurun_out/graph_trace/summary.txt.  Run through gpurun.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $R
mkdir -p gpurun_out/graph_trace
for B in 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/graph_trace/b$B -- python3 tools/gpu_graph_chains.py $B > gpurun_out/graph_trace/b$B.log 2>&1 || exit 1
  echo "== $B independent chain(s)" >> gpurun_out/graph_trace/summary.txt
  grep -v amdgpu.ids gpurun_out/graph_trace/b$B.log | grep '"value"\|ms_per_rollout' >> gpurun_out/graph_trace/summary.txt
  python3 tools/gpu_graph_trace_analyze.py gpurun_out/graph_trace/b$B | grep -v "^columns" >> gpurun_out/graph_trace/summary.txt
  find gpurun_out/graph_trace/b$B -name "*kernel_trace.csv" -delete
done
cat gpurun_out/graph_trace/summary.txt
