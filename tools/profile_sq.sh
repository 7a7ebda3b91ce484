#!/bin/bash
# tools/profile_sq.sh <tag> -- issue counters of the bench's kernels in two small --pmc passes (SQ has 8 slots; the eight-counter
# pass of profile_r2.sh came back empty on this pool, four at a time work)
set -u
TAG=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
# every invocation gets its own directory: a failed pass is evidence, a rerun must never overwrite it (round 2 lost the record of a
# parity failure that way).  gpurun_out/prof_$TAG.latest names the newest one for tools/summarize_r2.py.
RUN=prof_${TAG}_$(date -u +%Y%m%dT%H%M%S)_$$
OUT=$REPO/gpurun_out/$RUN
mkdir -p "$OUT"
echo "$RUN" > "$REPO/gpurun_out/prof_$TAG.latest"
FAILED=0
export TMPDIR=/tmp
cd "$REPO"
ARGS="--no-cpu-baseline --no-e2e"
run() {
	name=$1
	shift
	echo "== $name: rocprofv3 $* -- python3 bench.py $ARGS" | tee -a "$OUT/log.txt"
	timeout -k 10 500 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS >"$OUT/$name.stdout" 2>"$OUT/$name.stderr"
	rc=$?
	echo "   rc=$rc" | tee -a "$OUT/log.txt"
	[ $rc -ne 0 ] && FAILED=1
	# a leg that failed its parity check reports inside the JSON line and bench.py still exits 0: count that as a failed pass too
	if grep -q '"error"' "$OUT/$name.stdout"; then
		echo "   $name: a leg reported an error (kept: $OUT/$name.stdout, gpurun_out/evidence/)" | tee -a "$OUT/log.txt"
		FAILED=1
	fi
}
run sq --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY
run grbm --kernel-trace --pmc GRBM_GUI_ACTIVE
find "$OUT" -name '*.csv' -size +8M -delete
ls -R "$OUT" | grep counter
exit $FAILED
