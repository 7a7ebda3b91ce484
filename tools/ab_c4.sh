#!/bin/bash
# tools/ab_c4.sh lib1.so lib2.so ... -- interleaved A/B of library builds on ONE device: bench.py's config 4 leg (host stage of progressive streams)
cd "$(dirname "$0")/.."
for r in 1 2 3; do
	for lib in "$@"; do
		echo -n "round $r $(basename $(dirname $lib)): "
		MIJ_LIB=$(realpath $lib) python bench.py --no-cpu-baseline --no-e2e --legs config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['legs']['config4']; e=c.get('end_to_end',{})
print('e2e', e.get('mpix_s'), 'host stage', e.get('host_stage_only_mpix_s'), 'single thread', c.get('host_progressive_stage_mpix_s_single_thread'))"
	done
done
