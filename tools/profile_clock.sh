#!/bin/bash
# tools/profile_clock.sh -- GRBM_GUI_ACTIVE (busy core-clock cycles) per dispatch / dispatch duration = shader clock under the kernel
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_clock
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/grbm" -- python3 bench.py --images 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e >"$OUT/grbm.stdout" 2>"$OUT/grbm.stderr"
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/grbm/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))
for f in glob.glob(out + "/grbm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fused420" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            name, ns, grid = dur.get(r["Dispatch_Id"], ("?", 0, 0))
            if ns:
                print("dispatch %s grid %d: %.3f ms  GRBM_GUI_ACTIVE %.0f  -> %.0f MHz (if the counter is summed over 8 XCDs: %.0f MHz)" % (r["Dispatch_Id"], grid, ns / 1e6, float(r["Counter_Value"]), float(r["Counter_Value"]) / ns * 1e3, float(r["Counter_Value"]) / 8 / ns * 1e3))
PY
