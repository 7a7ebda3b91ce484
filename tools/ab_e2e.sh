#!/bin/bash
# tools/ab_e2e.sh lib1.so lib2.so ... -- interleaved A/B of library builds on ONE device in one job: the GPU-walk ring of tools/bench_gpu_walk.py
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-3}
for r in $(seq 1 $ROUNDS); do
	for lib in "$@"; do
		echo -n "round $r $(basename $(dirname $lib)): "
		MIJ_LIB=$(realpath $lib) BGW_N=2048 BGW_THREADS=16 BGW_CHUNKS=128 BGW_DEPTHS=4 python tools/bench_gpu_walk.py 2>&1 | grep chunk
	done
done
