#!/bin/bash
# tools/profile_encode.sh <tag> -- rocprofv3 passes over tools/bench_encode.py (kernel time split + issue counters)
set -u
TAG=${1:-enc}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
run() {
	name=$1
	shift
	echo "== $name: rocprofv3 $*" | tee -a "$OUT/log.txt"
	timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 tools/bench_encode.py >"$OUT/$name.stdout" 2>"$OUT/$name.stderr"
	echo "   rc=$?" | tee -a "$OUT/log.txt"
}
run trace --kernel-trace --stats
run sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run mem --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
find "$OUT" -name '*.csv' -size +4M -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("trace", "sq", "mem", "fetch", "write"):
    for f in glob.glob(out + "/" + name + "/**/*kernel_stats.csv", recursive=True):
        print("--", name, "kernel_stats")
        for row in list(csv.DictReader(open(f)))[:6]:
            print("  ", row.get("Name", "")[:60], row.get("Calls"), row.get("AverageNs"), row.get("Percentage"))
    for f in glob.glob(out + "/" + name + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:40]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        print("--", name, "counters (sum over dispatches)")
        for k, d in acc.items():
            if "encode" in k:
                print("  ", k, {c: int(v) for c, v in d.items()})
PY
