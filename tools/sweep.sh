#!/bin/bash
# tools/sweep.sh -- quick GPU sweep of the fused kernel's band height (MCU rows per workgroup)
cd "$(dirname "$0")/.."
for r in 4 8 12 17 23 34 68; do
  echo -n "MIJ_BAND_ROWS=$r  "
  MIJ_BAND_ROWS=$r python bench.py --images 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_per_launch'], d['roofline']['frac'])"
done
