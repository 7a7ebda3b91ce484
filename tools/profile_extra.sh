#!/bin/bash
# kernel-trace --stats of the secondary kernels (4:4:4 decode, encode transform)
set -u
TAG=${1:-r01}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
B444_N=16 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/k444" -- python3 tools/bench_444.py > "$OUT/k444.stdout" 2> "$OUT/k444.stderr"
BENC_N=256 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kenc" -- python3 tools/bench_encode.py > "$OUT/kenc.stdout" 2> "$OUT/kenc.stderr"
B422_N=256 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/k422" -- python3 tools/bench_422.py > "$OUT/k422.stdout" 2> "$OUT/k422.stderr"
BENT_N=256 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kes" -- python3 tools/bench_entropy.py > "$OUT/kes.stdout" 2> "$OUT/kes.stderr"
find "$OUT" -name '*.csv' -size +4M -delete
cat "$OUT"/k444/*/*_kernel_stats.csv "$OUT"/kenc/*/*_kernel_stats.csv "$OUT"/k422/*/*_kernel_stats.csv "$OUT"/kes/*/*_kernel_stats.csv
cat "$OUT/k444.stdout" "$OUT/kenc.stdout" "$OUT/k422.stdout" "$OUT/kes.stdout"
