#!/bin/bash
# tools/ab_walk.sh lib1.so lib2.so ... -- interleaved A/B of library builds on ONE device in one job: the GPU-walk ring as bench.py's
# end_to_end runs it (tools/bench_gpu_walk.py: 2048 x 1080p, chunks of 256, four batches in flight) and one walk at a time (depth 1).
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-3}
for r in $(seq 1 $ROUNDS); do
	for lib in "$@"; do
		echo "round $r $(basename $(dirname $lib)):"
		MIJ_LIB=$(realpath $lib) BGW_N=${BGW_N:-2048} BGW_THREADS=16 BGW_CHUNKS=256 BGW_DEPTHS=${DEPTHS:-1,4} python tools/bench_gpu_walk.py 2>&1 | grep chunk
	done
done
