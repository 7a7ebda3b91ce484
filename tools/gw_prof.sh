#!/bin/bash
# tools/gw_prof.sh -- kernel trace of tools/bench_gpu_walk.py with int16 and with byte-coefficient planes
export TMPDIR=/tmp BGW_THREADS=16
cd "$(dirname "$0")/.."
for fmt in 0 1; do
	for ch in 128 512; do
		d=gpurun_out/gw_${fmt}_${ch}
		mkdir -p $d
		MIJ_COEF_BYTES=$fmt BGW_CHUNKS=$ch rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_gpu_walk.py >$d/out.txt 2>&1
		echo "== bytes=$fmt chunk=$ch"
		grep chunk $d/out.txt
		python3 - $d <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0])):
    if 'mij' in r['Name']:
        print('  ', r['Name'].split('(')[0][-28:].ljust(28), r['Calls'].rjust(4), 'avg %.3f ms' % (float(r['AverageNs']) / 1e6), 'min %.3f' % (float(r['MinNs']) / 1e6))
PY
	done
done
