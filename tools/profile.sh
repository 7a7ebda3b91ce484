#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- rocprofv3 passes over bench.py on the GPU box.
#   pass 1: --kernel-trace --stats          (per-kernel time; the roofline's duration cross-check)
#   pass 2: --pmc FETCH_SIZE                (TCC read traffic; gfx950 reports 1/2 of wide reads)
#   pass 3: --pmc WRITE_SIZE                (TCC write traffic)
#   pass 4: --pmc SQ_* / LDS counters       (issue mix, LDS conflicts)
# Counter passes carry --kernel-trace only (never sys/runtime/hip traces: gpurun refuses that mix).
# Outputs land under gpurun_out/prof_<tag>/ ; summaries are copied into profiles/ by hand.
set -u
TAG=${1:-r01}
shift || true
ARGS=${@:---images 1024 --steps 30 --warmup 10 --no-cpu-baseline --no-e2e}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
run() {
	name=$1
	shift
	echo "== $name: rocprofv3 $*" | tee -a "$OUT/log.txt"
	timeout -k 10 400 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS >"$OUT/$name.stdout" 2>"$OUT/$name.stderr"
	echo "   rc=$?" | tee -a "$OUT/log.txt"
}
run trace --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run lds --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
# keep only the small CSVs
find "$OUT" -name '*.csv' -size +4M -delete
ls -R "$OUT" | head -50
