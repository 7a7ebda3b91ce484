#!/usr/bin/env python3
"""tools/sq_quick.py <dir> -- per-launch means of the counters tools/sq_quick.sh collected, for the largest-grid dispatches of every
k_fused* / k_encode* kernel (the last 10 of each: the timed launches)."""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> values per dispatch
dur = defaultdict(list)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(lambda: defaultdict(float))  # (kernel, dispatch) -> counter -> value
    grid = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not ("k_fused" in k or "k_encode" in k):
            continue
        key = (k, int(r["Dispatch_Id"]))
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        grid[key] = int(r["Grid_Size"])
    by_k = defaultdict(list)
    for (k, d), c in per.items():
        by_k[k].append((d, grid[(k, d)], c))
    for k, lst in by_k.items():
        big = max(g for _, g, _ in lst)
        lst = sorted([x for x in lst if x[1] == big])[-10:]
        for _, _, c in lst:
            for name, v in c.items():
                rows[k][name].append(v)
for f in glob.glob(os.path.join(src, "a", "**", "*kernel_trace.csv"), recursive=True):
    by_k = defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_fused" in k or "k_encode" in k:
            by_k[k].append((int(r["Dispatch_Id"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, lst in by_k.items():
        big = max(g for _, g, _ in lst)
        dur[k] = [t for _, g, t in sorted(lst) if g == big][-10:]
for k in sorted(rows):
    c = {n: sum(v) / len(v) for n, v in rows[k].items()}
    print(k[:90])
    ms = sum(dur[k]) / len(dur[k]) / 1e6 if dur.get(k) else float("nan")
    print("  ms under counters %.4f" % ms)
    for n in sorted(c):
        print("  %-24s %.4g" % (n, c[n]))
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        print("  shader cycles/launch %.4g" % cyc)
        if "SQ_ACTIVE_INST_VALU" in c:
            print("  VALU busy %.3f   (4 x SQ_ACTIVE_INST_VALU / (cycles x 1024 SIMDs))" % (4 * c["SQ_ACTIVE_INST_VALU"] / (cyc * 1024)))
        if "SQ_INSTS_VALU" in c:
            print("  VALU issue slots used %.3f   (4 x SQ_INSTS_VALU / (cycles x 1024))" % (4 * c["SQ_INSTS_VALU"] / (cyc * 1024)))
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        print("  SQ_WAIT_ANY / SQ_WAVE_CYCLES %.3f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
