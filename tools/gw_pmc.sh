#!/bin/bash
# tools/gw_pmc.sh -- SQ counters of the GPU-walk kernels (tools/bench_gpu_walk.py, one chunk size, --pmc passes only)
export TMPDIR=/tmp BGW_THREADS=16 BGW_CHUNKS=256 BGW_N=1024 BGW_DEPTHS=2
cd "$(dirname "$0")/.."
run() {
	d=gpurun_out/gwpmc_$1
	shift
	rm -rf $d; mkdir -p $d
	rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d -- python3 tools/bench_gpu_walk.py >$d/out.txt 2>&1
	python3 - $d <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'k_es_' not in k:
        continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[(k, r['Counter_Name'])] += 1
for k in sorted(acc):
    print(k.ljust(28), '  '.join('%s=%.3g' % (c, v / n[(k, c)]) for c, v in sorted(acc[k].items())))
PY
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_WAVES SQ_ACTIVE_INST_LDS
run c SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
