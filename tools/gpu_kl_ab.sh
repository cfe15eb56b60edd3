#!/bin/bash
# A/B (GPU box): the BASELINE configs whose lane groups have lanes to spare, with several lanes per link (the shipped library) against one lane per link
# (libcclqr_kl1.so = make variant NAME=kl1 VFLAGS=-DCHAIN_ONE_LANE_PER_LINK: the kernels of round 4), A/B/A/B in one session.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
  for v in ${VARIANTS:-kl1 base}; do
    for cfg in ${CONFIGS:-cartpole_cfg2 cartpole_cfg2_filled sawyer_cfg4 tracking_cfg5}; do
      echo "$v run$rep $cfg: $(CCLQR_LIB_VARIANT=$v timeout -k 10 200 python3 tools/gpu_config_rollout.py $cfg 3 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.2f M inst-steps/s, kernel %.3f ms, failed %s" % (d["value"]/1e6 if d["value"] else -1, d["kernel_ms"], d.get("failed_instances")))')"
    done
  done
done
