#!/bin/bash
# tools/profile_r2.sh <tag> -- rocprofv3 passes over the driver's own command (python3 bench.py, legs included) on the GPU box.
#   trace : --kernel-trace --stats            per-kernel time; the roofline's duration cross-check
#   fetch : --pmc FETCH_SIZE                  TCC read traffic (gfx950 reports 1/2 of wide reads: doubled by the summariser)
#   write : --pmc WRITE_SIZE                  TCC write traffic
#   sq    : --pmc SQ_* issue counters         VALU instruction count, wave cycles, waits
# Counter passes carry --kernel-trace only (never sys/runtime/hip traces: gpurun refuses that mix).  The program after
# "--" is python3 itself.  Outputs land under gpurun_out/prof_<tag>/; tools/summarize_r2.py turns them into profiles/.
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
run trace --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
# keep only the small CSVs
# the SQ pass of the full leg set writes a large counter file: keep the rows of the kernels the summary reads
for f in $(find "$OUT" -name "*counter_collection.csv" -size +8M); do { head -1 "$f"; grep -E "k_fused4|k_encode4|k_resample_fast|k_idct_planes" "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"; done
find "$OUT" -name "*.csv" -size +30M -delete
ls -R "$OUT" | head -40
exit $FAILED
