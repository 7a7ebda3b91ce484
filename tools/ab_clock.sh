#!/bin/bash
# tools/ab_clock.sh lib1.so lib2.so ... -- per build: kernel time, GRBM_GUI_ACTIVE cycles and the implied shader clock
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MIJ_BENCH_NOCHECK=1
for lib in "$@"; do
	tag=$(basename $(dirname $lib))
	OUT=gpurun_out/abclk_$tag
	rm -rf $OUT; mkdir -p $OUT
	MIJ_LIB=$(realpath $lib) timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 bench.py --images 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e >$OUT/stdout 2>$OUT/stderr
	python3 - $OUT $tag <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = []
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused420" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            rows.append((dur[r["Dispatch_Id"]], float(r["Counter_Value"]) / 8))
rows = rows[3:]
if rows:
    ms = sum(r[0] for r in rows) / len(rows) / 1e6
    cyc = sum(r[1] for r in rows) / len(rows)
    print("%-8s n=%d  %.3f ms  %.2f Mcycles  %.0f MHz" % (tag, len(rows), ms, cyc / 1e6, cyc / (ms * 1e3)))
PY
done
