#!/bin/bash
for f in 0 1 2 3 4 7; do
  RSB_ABL=$f timeout -k 10 120 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abl $f scan_ms %.3f'%d['stage_ms']['scan'])" || exit 1
done
