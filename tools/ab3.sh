#!/bin/bash
# tools/ab3.sh lib1.so lib2.so ... -- interleaved A/B of library builds on ONE device in one job (perf deltas never come from
# different boxes): the headline kernel and the legs named in LEGS (default int16_planes,harsh_batch), ROUNDS rounds (default 3).
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-3}
LEGS=${LEGS:-int16_planes,harsh_batch}
for r in $(seq 1 $ROUNDS); do
	for lib in "$@"; do
		echo -n "round $r $(basename $(dirname $lib)): "
		MIJ_LIB=$(realpath $lib) python bench.py --no-cpu-baseline --no-e2e --legs $LEGS 2>/dev/null |
			python -c "
import json,sys
d=json.loads(sys.stdin.read()); L=d.get('legs',{})
print(d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'], ' '.join('| %s %s' % (k, v.get('kernel_ms_per_launch', v.get('error','?'))) for k,v in L.items()))"
	done
done
