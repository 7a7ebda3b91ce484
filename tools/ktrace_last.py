"""Prints the kernels of the LAST decode of a rocprofv3 kernel trace (csv), in start order, with durations and the gaps between them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last k_es_cold starts the last walk
idx = max(i for i, r in enumerate(rows) if "k_es_cold" in r["Kernel_Name"])
prev_end = None
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[max(0, idx - 8):]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%9.1f us  +gap %7.1f  dur %8.1f  grid %s  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Kernel_Name"][:60]))
    prev_end = e
