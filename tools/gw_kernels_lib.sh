#!/bin/bash
# tools/gw_kernels_lib.sh lib.so [tag] -- tools/gw_kernels.sh's first part (per-kernel times, one walk in flight) for the library named
export TMPDIR=/tmp BGW_THREADS=16
cd "$(dirname "$0")/.."
TAG=${2:-gwl}
d=gpurun_out/$TAG
mkdir -p $d
MIJ_LIB=$(realpath $1) BGW_CHUNKS=256 BGW_DEPTHS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_gpu_walk.py >$d/out.txt 2>&1
grep chunk $d/out.txt
python3 - $d <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0])):
    if 'k_es' in r['Name']:
        print('  ', r['Name'].split('(')[0][-34:].ljust(34), r['Calls'].rjust(5), 'avg %.3f ms' % (float(r['AverageNs']) / 1e6), 'min %.3f' % (float(r['MinNs']) / 1e6))
PY
