"""Reads a rocprofv3 kernel trace of tools/bench_gpu_walk.py and reports, for the steady part of the run, the sum of kernel durations, the
time during which at least one kernel ran (union of the intervals) and the wall time: how busy the GPU is under the ring of batches."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-24:]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo = t0 + (t1 - t0) * 0.5  # second half of the run: the ring in steady state (the first half holds warm-up and the other depths)
rows = [r for r in rows if r[0] >= lo]
wall = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e, total = 0, rows[0][0], rows[0][1], 0
per = {}
for s, e, n in rows:
    total += e - s
    per[n] = per.get(n, 0) + e - s
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("wall %.2f ms  busy (union) %.2f ms = %.0f %%  sum of kernel durations %.2f ms = %.2f x wall" % (wall / 1e6, busy / 1e6, 100.0 * busy / wall, total / 1e6, total / wall))
for n, v in sorted(per.items(), key=lambda kv: -kv[1])[:10]:
    print("   %-26s %.2f ms  (%.0f %% of the wall time)" % (n, v / 1e6, 100.0 * v / wall))
