#!/bin/bash
# k_modeac stage by stage (RSB_ABL bits: 8 = stop after the block sums, 16 = skip the sums, 32 = stop after the tile loads, 64 = after F1, 128 = after F2)
for f in 8 32 64 128 0 16; do
  RSB_ABL=$f timeout -k 10 120 python -u bench.py --modeac --steps 3 --warmup 1 --no-cpu-baseline --steps-in-flight 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abl $f modeac_ms %.3f scan_ms %.3f step %.3f'%(d['stage_ms']['modeac'], d['stage_ms']['scan'], d['ms_per_step']))" || exit 1
done
