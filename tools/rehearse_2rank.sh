#!/bin/bash
# tools/rehearse_2rank.sh [total_images] -- bench.py exactly as the driver launches it at N = 2 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
# the environment, one process per rank), both ranks on the one GPU of a gpurun box (MIJ_BENCH_SHARE_DEVICE=1): slices, per-rank
# verification and kernel times, the end-to-end GPU-walk ring on every rank at the same time with its share of the host cores, the CPU
# baseline on rank 0.  Rank 0's line goes to gpurun_out/bench_2rank.json.
cd "$(dirname "$0")/.."
TOTAL=${1:-2048}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 WORLD_SIZE=2 LOCAL_WORLD_SIZE=2 MIJ_BENCH_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
RANK=1 LOCAL_RANK=1 python3 bench.py --gpus 2 --total-images $TOTAL > gpurun_out/bench_2rank_r1.out 2> gpurun_out/bench_2rank_r1.err &
P1=$!
RANK=0 LOCAL_RANK=0 python3 bench.py --gpus 2 --total-images $TOTAL > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
RC0=$?
wait $P1
RC1=$?
echo "rank 0 rc=$RC0, rank 1 rc=$RC1"
[ $RC0 -eq 0 ] && [ $RC1 -eq 0 ]
