"""summarise a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass over tools/gpu_riccati_resident.py: MFMA-busy share of the batched
LDS-resident Riccati kernel.  argv: <dir with *_counter_collection.csv> <out.json> <nprob> <N>"""
import csv, glob, json, os, sys
src, dst, nprob, N = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = {}
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "riccati_resident_kernel" not in k:
            continue
        d = rows.setdefault((k, r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["wg"] = int(r.get("Grid_Size", 0) or 0) // max(int(r.get("Workgroup_Size", 1) or 1), 1)
        d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
big = [v for (k, _), v in rows.items() if v.get("wg", 0) >= nprob]     # the timed batch (the warm-up calls are 2 and 64 problems)
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 tools/gpu_riccati_resident.py %d %d" % (nprob, N),
       "dispatches": len(big)}
if big:
    busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in big) / len(big)
    act = sum(v["GRBM_GUI_ACTIVE"] for v in big) / len(big)
    simds, xcds = 1024, 8
    act_xcd = act / xcds            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (8 x the dispatch's own cycle count)
    ns = sum(v["ns"] for v in big) / len(big)
    # per backward step and problem: MFMAs issued by the kernel (v_mfma_f64_16x16x4_f64 = 64 busy cycles each on one SIMD)
    out.update(avg_SQ_VALU_MFMA_BUSY_CYCLES=busy, avg_GRBM_GUI_ACTIVE=act, active_cycles_per_xcd=act_xcd, duration_ms_under_counters=ns / 1e6,
               mfma_busy_share_of_all_simds=busy / (act_xcd * simds), mfma_tflops_issued=busy / 64.0 * 2048 / (ns * 1e-9) / 1e12,
               mfma_instructions_per_problem_step=busy / 64.0 / (nprob * (N - 1)),
               note="SQ_VALU_MFMA_BUSY_CYCLES sums over the device's SIMDs; GRBM_GUI_ACTIVE = active cycles of the dispatch summed over the 8 XCDs; share = busy / (active / 8 x 1024 SIMDs); one v_mfma_f64_16x16x4_f64 = 64 busy cycles = 2048 flop. "
                    "The share counts every MFMA the kernel issues (16-row padding of 84 and the extra products of the projected form included), beside "
                    "the 'fraction of peak by the reference formulation's flop count' (4.136 MFLOP per problem-step) that DESIGN 4.3 quotes")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
