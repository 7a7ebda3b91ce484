#!/bin/bash
# tools/gw_ab.sh lib1.so lib2.so ... -- kernel times of the GPU walk (tools/bench_gpu_walk.py, chunks of 256, one walk in flight)
# for several library builds (ablation variants: image-codecs_amd/lib_v<bits>/, MIJ_VARIANT bits in mij_entropy_kernels.h) on one box
export TMPDIR=/tmp BGW_THREADS=16 BGW_CHUNKS=256 BGW_DEPTHS=1
cd "$(dirname "$0")/.."
for lib in "$@"; do
	d=gpurun_out/gwab_$(basename $(dirname $lib))
	mkdir -p $d
	MIJ_LIB=$(realpath $lib) rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_gpu_walk.py >$d/out.txt 2>&1
	echo "== $lib"
	python3 - $d <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0])):
    if 'k_es_' in r['Name']:
        print('  ', r['Name'].split('(')[0][-20:].ljust(20), r['Calls'].rjust(5), 'avg %.3f ms' % (float(r['AverageNs']) / 1e6), 'min %.3f' % (float(r['MinNs']) / 1e6))
PY
done
