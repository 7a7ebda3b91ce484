#!/bin/bash
# tools/seg_sweep.sh -- column segments of the band kernel (MIJ_SEG_MIN / MIJ_SEG_COLS, mij_runtime.hip) against the whole-row forms, by picture size
cd "$(dirname "$0")/.."
L=image-codecs_amd
SIZES="2560x1440 3840x2160 5120x2880 6000x4000 8192x5464"
echo "== whole rows (w / x forms; two-pass beyond 5840)"; MIJ_LIB=$PWD/$L/lib_q/libimagecodecs_mi355x.so python tools/bench_sizes.py $SIZES
echo "== segments <= 119 columns, 256 threads"; MIJ_SEG_MIN=121 MIJ_SEG_COLS=119 MIJ_LIB=$PWD/$L/lib/libimagecodecs_mi355x.so python tools/bench_sizes.py $SIZES
echo "== segments <= 100 columns, 256 threads"; MIJ_SEG_MIN=121 MIJ_SEG_COLS=100 MIJ_LIB=$PWD/$L/lib/libimagecodecs_mi355x.so python tools/bench_sizes.py $SIZES
echo "== segments <= 180 columns, 512 threads"; MIJ_SEG_MIN=121 MIJ_SEG_COLS=180 MIJ_LIB=$PWD/$L/lib_c512/libimagecodecs_mi355x.so python tools/bench_sizes.py $SIZES
echo "== segments <= 119 columns, 512 threads"; MIJ_SEG_MIN=121 MIJ_SEG_COLS=119 MIJ_LIB=$PWD/$L/lib_c512/libimagecodecs_mi355x.so python tools/bench_sizes.py $SIZES
