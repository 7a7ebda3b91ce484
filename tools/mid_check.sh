#!/bin/bash
# tools/mid_check.sh [tag] -- the whole GPU suite, then the default bench.py line, summarised (one gpurun call between milestones)
cd "$(dirname "$0")/.."
TAG=${1:-mid}
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.txt 2>&1
rc=$?
tail -3 gpurun_out/${TAG}_tests.txt
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python tools/bench_brief.py gpurun_out/${TAG}_bench.json
