"""per-config summaries of tools/profile_configs.sh (diagnostic): kernel time from the trace, instruction mix and lane / issue figures from the two
counter passes, per wavefront-step and per instance-step.  argv: <dir written by profile_configs.sh> <summary dir>"""
import collections, csv, glob, json, os, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
table = {}
for cfg in ("cartpole_cfg2", "cartpole_cfg2_filled", "sawyer_cfg4", "tracking_cfg5", "tree14", "deltabot"):
    if not os.path.exists(os.path.join(src, cfg + ".pmcA.log")):
        continue
    logs = [x for x in open(os.path.join(src, cfg + ".pmcA.log")) if x.startswith("{")]
    if not logs:
        continue
    info = json.loads(logs[-1])
    kern = info["kernel"].split("<")[0]
    launches = 2.0                      # gpu_config_rollout.py: one untimed + `reps` = 1 timed launch per counter pass
    tot = collections.defaultdict(float)
    for sub in ("pmcA", "pmcB"):
        for f in glob.glob(os.path.join(src, cfg, sub, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for f in glob.glob(os.path.join(src, cfg, "trace", "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, cfg + "_kernel_stats.csv"))
    ws = info["wavefronts"] * info["sim_steps"] * launches
    per = {k: v / ws for k, v in tot.items()}
    lanes = per.get("SQ_THREAD_CYCLES_VALU", 0) / max(per.get("SQ_ACTIVE_INST_VALU", 1e-9), 1e-9)
    f64 = sum(per.get(k, 0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
    ipw = info["instances_per_wavefront"]
    simds = 1024.0
    derived = {"kernel": info["kernel"], "kernel_ms_unprofiled_events": info["kernel_ms"], "instances": info["instances"], "sim_steps": info["sim_steps"],
               "instances_per_wavefront": ipw, "wavefronts": info["wavefronts"], "wavefronts_per_simd_launched": info["wavefronts"] / simds,
               "workgroups_per_cu_by_lds": info["workgroups_per_cu_by_lds"], "lds_bytes_per_workgroup": info["lds_bytes_per_workgroup"],
               "cycles_per_wavefront_step": 4.0 * per.get("SQ_WAVE_CYCLES", 0), "valu_instructions_per_instance_step": per.get("SQ_INSTS_VALU", 0) / ipw,
               "valu_issue_share": per.get("SQ_ACTIVE_INST_VALU", 0) / max(per.get("SQ_WAVE_CYCLES", 1e-9), 1e-9),
               "lds_issue_share": per.get("SQ_ACTIVE_INST_LDS", 0) / max(per.get("SQ_WAVE_CYCLES", 1e-9), 1e-9),
               "waiting_on_counters_share": per.get("SQ_WAIT_ANY", 0) / max(per.get("SQ_WAVE_CYCLES", 1e-9), 1e-9),
               "f64_share_of_valu": f64 / max(per.get("SQ_INSTS_VALU", 1e-9), 1e-9), "active_lanes_per_valu_instruction": lanes,
               "fp64_flops_per_instance_step_from_counters": (2 * per.get("SQ_INSTS_VALU_FMA_F64", 0) + per.get("SQ_INSTS_VALU_MUL_F64", 0) + per.get("SQ_INSTS_VALU_ADD_F64", 0)) * lanes / ipw,
               "lds_bank_conflict_share_of_lds_active": per.get("SQ_LDS_BANK_CONFLICT", 0) / max(per.get("SQ_LDS_IDX_ACTIVE", 1e-9), 1e-9)}
    table[cfg] = derived
    json.dump({"command": "rocprofv3 --pmc <counters> -- python3 tools/gpu_config_rollout.py %s 1 (two passes)" % cfg, "totals": dict(tot), "per_wavefront_step": per,
               "derived": derived, "note": "SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles (MI355X_MICROARCH.md)"},
              open(os.path.join(dst, cfg + "_pmc_mix.json"), "w"), indent=1)
json.dump(table, open(os.path.join(dst, "configs_table.json"), "w"), indent=1)
print(json.dumps(table, indent=1))
