"""diagnostic: reads a rocprofv3 kernel_trace.csv of tools/gpu_graph_chains.py and prints, for the single-step kernels, durations / gaps / overlap statistics per queue"""
import csv, glob, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rollout_chain_kernel<8, 4, 3" in r["Kernel_Name"]]
print("columns", list(rows[0].keys()))
s = np.array([int(r["Start_Timestamp"]) for r in rows]); e = np.array([int(r["End_Timestamp"]) for r in rows]); q = np.array([int(r["Queue_Id"]) for r in rows])
g = np.array([int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)) for r in rows])
o = np.argsort(s); s, e, q, g = s[o], e[o], q[o], g[o]
print("kernels", len(s), "queues", collections.Counter(q.tolist()), "grids", collections.Counter(g.tolist()))
for grid in sorted(set(g.tolist())):
    m = g == grid
    ss, ee, qq = s[m], e[m], q[m]
    d = (ee - ss) / 1e3
    print("grid %d: n %d duration us: min %.1f median %.1f mean %.1f max %.1f" % (grid, m.sum(), d.min(), np.median(d), d.mean(), d.max()))
    # split into replays by large gaps
    span = (ee.max() - ss.min()) / 1e3
    busy = 0.0; cur_s, cur_e = ss[0], ee[0]
    for a, b in zip(ss[1:], ee[1:]):
        if a > cur_e: busy += cur_e - cur_s; cur_s, cur_e = a, b
        else: cur_e = max(cur_e, b)
    busy += cur_e - cur_s
    print("   union of kernel intervals %.1f us of span %.1f us" % (busy / 1e3, span))
    for qu in sorted(set(qq.tolist())):
        k = qq == qu
        gaps = (ss[k][1:] - ee[k][:-1]) / 1e3
        gaps = gaps[gaps < 1000]
        print("   queue %d: n %d gap between consecutive kernels us: median %.2f mean %.2f p90 %.2f" % (qu, k.sum(), np.median(gaps), gaps.mean(), np.percentile(gaps, 90)))
