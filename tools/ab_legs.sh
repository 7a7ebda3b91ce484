#!/bin/bash
# tools/ab_legs.sh lib1.so lib2.so ... -- interleaved A/B of library builds on ONE device in one job: the headline kernel and the
# kernel times of the legs (int16 planes, harsh batch, config 4, config 5, 4:2:2, two-pass)
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-2}
for r in $(seq 1 $ROUNDS); do
	for lib in "$@"; do
		echo -n "round $r $(basename $(dirname $lib)): "
		MIJ_LIB=$(realpath $lib) python bench.py --no-cpu-baseline --no-e2e 2>/dev/null |
			python -c "
import json,sys
d=json.loads(sys.stdin.read()); L=d['legs']
def ms(x):
    return x.get('kernel_ms_per_launch', x.get('error', '?'))
print(d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'], '| int16', ms(L['int16_planes']), '| harsh', ms(L['harsh_batch']), '| cfg4', ms(L['config4']), '| cfg5', ms(L['config5']), '| h2v1', ms(L['h2v1']), '| two_pass', {k: v['ms_per_launch'] for k, v in L.get('two_pass', {}).items() if isinstance(v, dict)})"
	done
done
