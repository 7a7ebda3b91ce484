#!/bin/bash
# tools/gw_ab.sh lib1.so lib2.so ... -- kernel times of the GPU walk (tools/bench_gpu_walk.py under rocprofv3) per library build
export TMPDIR=/tmp BGW_THREADS=16 BGW_CHUNKS=${BGW_CHUNKS:-256}
cd "$(dirname "$0")/.."
for lib in "$@"; do
	d=gpurun_out/gwab_$(basename $lib .so)
	mkdir -p $d
	MIJ_LIB=$(realpath $lib) rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_gpu_walk.py >$d/out.txt 2>&1
	echo "== $(basename $lib)"
	grep chunk $d/out.txt
	python3 - $d <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0])):
    if 'mij' in r['Name']:
        print('  ', r['Name'].split('(')[0][-28:].ljust(28), r['Calls'].rjust(4), 'avg %.3f ms' % (float(r['AverageNs']) / 1e6), 'min %.3f' % (float(r['MinNs']) / 1e6))
PY
done
