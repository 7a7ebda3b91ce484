#!/bin/bash
# tools/gw_env_ab.sh -- kernel times of the GPU walk with the staged and the scatter write pass (MIJ_ES_SCATTER)
export TMPDIR=/tmp BGW_THREADS=16 BGW_CHUNKS=${BGW_CHUNKS:-256}
cd "$(dirname "$0")/.."
for sc in 0 1 0 1; do
	d=gpurun_out/gwenv_$sc
	rm -rf $d; mkdir -p $d
	MIJ_ES_SCATTER=$sc rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_gpu_walk.py >$d/out.txt 2>&1
	echo "== scatter=$sc"
	grep chunk $d/out.txt
	python3 - $d <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0])):
    if 'mij' in r['Name'] or 'fill' in r['Name']:
        print('  ', r['Name'].split('(')[0][-28:].ljust(28), r['Calls'].rjust(4), 'avg %.3f ms' % (float(r['AverageNs']) / 1e6), 'min %.3f' % (float(r['MinNs']) / 1e6))
PY
done
