#!/bin/bash
# ablation: scan-kernel time when it stops after stage k (1 convert + comparisons, 2 + edge patterns / queue, 3 + gate, 4 + DF, 5 + frame slicing = all)
for k in 1 2 3 4 5; do
  RSB_STAGE_LIMIT=$k timeout -k 10 120 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage_limit $k scan_ms %.3f'%d['stage_ms']['scan'])" || exit 1
done
