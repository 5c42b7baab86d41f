#!/bin/bash
# A/B on one box: k_scan time and step time of each prebuilt variant (build/variants/<name>.so), interleaved
for rep in 1 2 3; do
  for v in "$@"; do
    cp build/variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
    timeout -k 10 120 python -u bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v scan_ms %.4f prune_ms %.4f step_ms %.3f %s'%(d['stage_ms']['scan'], d['stage_ms']['prune'], d['ms_per_step'], d['config']['full_size_check']))" || exit 1
  done
done
