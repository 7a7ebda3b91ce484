#!/bin/bash
# tools/ab.sh lib1.so lib2.so ... -- interleaved A/B of library builds on ONE device in one job
# (rule: perf deltas come from interleaved rounds in one process/box, never across boxes).
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-3}
for r in $(seq 1 $ROUNDS); do
	for lib in "$@"; do
		echo -n "round $r $(basename $lib): "
		MIJ_LIB=$(realpath $lib) python bench.py --images 1024 --steps 30 --warmup 10 --no-cpu-baseline --no-e2e 2>/dev/null |
			python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'])"
	done
done
