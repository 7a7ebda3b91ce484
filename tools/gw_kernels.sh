#!/bin/bash
# tools/gw_kernels.sh [tag] -- per-kernel times of the GPU Huffman walk: tools/bench_gpu_walk.py (1024 x 1080p, chunks of 256,
# one walk in flight so kernels do not overlap) under rocprofv3 --kernel-trace --stats; then the ring of four 128-image batches
# without the profiler (the end-to-end figure).
export TMPDIR=/tmp BGW_THREADS=16
cd "$(dirname "$0")/.."
TAG=${1:-gw}
d=gpurun_out/$TAG
mkdir -p $d
BGW_CHUNKS=256 BGW_DEPTHS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_gpu_walk.py >$d/out.txt 2>&1
grep chunk $d/out.txt
python3 - $d <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0])):
    if 'mij' in r['Name'] or 'fill' in r['Name'] or 'copy' in r['Name']:
        print('  ', r['Name'].split('(')[0][-34:].ljust(34), r['Calls'].rjust(5), 'avg %.3f ms' % (float(r['AverageNs']) / 1e6), 'min %.3f' % (float(r['MinNs']) / 1e6), 'total %.1f ms' % (float(r['TotalDurationNs']) / 1e6))
PY
BGW_CHUNKS=128 BGW_DEPTHS=2,4 python3 tools/bench_gpu_walk.py 2>&1 | grep chunk
