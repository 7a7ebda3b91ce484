#!/bin/bash
# tools/sq_quick.sh <tag> [bench args] -- issue counters of the headline kernel only: two small --pmc passes over
# `python3 bench.py --no-legs --no-e2e --no-cpu-baseline --steps 10 --warmup 5`, summarised by tools/sq_quick.py.
# Every run gets its own directory (never overwritten); counter passes carry --kernel-trace only.
set -u
TAG=${1:-sq}
shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
RUN=sq_${TAG}_$(date -u +%Y%m%dT%H%M%S)_$$
OUT=$REPO/gpurun_out/$RUN
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
ARGS="--no-legs --no-e2e --no-cpu-baseline --steps 10 --warmup 5 $*"
FAILED=0
run() {
	name=$1
	shift
	timeout -k 10 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS >"$OUT/$name.stdout" 2>"$OUT/$name.stderr"
	rc=$?
	echo "$name rc=$rc" >> "$OUT/log.txt"
	[ $rc -ne 0 ] && FAILED=1
}
run a --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY
run c --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
run d --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
python3 tools/sq_quick.py "$OUT" | tee "$OUT/summary.txt"
find "$OUT" -name '*.csv' -size +8M -delete
exit $FAILED
