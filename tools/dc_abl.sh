#!/bin/bash
# k_convert_dc with parts switched off (RSB_DC_ABL: 1 no block scan, 2 no stores, 4 no sqrt): step time of the --dcfilter bench
for abl in 0 1 2 4 7; do
  RSB_DC_ABL=$abl timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --format SC16 --dcfilter 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abl $abl step_ms %.3f'%d['ms_per_step'])" || exit 1
done
