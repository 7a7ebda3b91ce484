#!/bin/bash
# tools/ab_enc.sh lib1.so lib2.so ... -- interleaved A/B of library builds on the two encoder legs of bench.py (config 5: k_encode420, q = 95: k_encode444)
cd "$(dirname "$0")/.."
ROUNDS=${ROUNDS:-3}
for r in $(seq 1 $ROUNDS); do
	for lib in "$@"; do
		echo -n "round $r $(basename $(dirname $lib)): "
		MIJ_LIB=$(realpath $lib) python bench.py --legs config5,config5_q95_444 2>/dev/null |
			python -c "
import json,sys
d=json.loads(sys.stdin.read()); L=d['legs']
print({k: (L[k].get('kernel_ms_per_launch'), L[k].get('frac'), L[k].get('parity'), L[k].get('error')) for k in ('config5', 'config5_q95_444')})"
	done
done
