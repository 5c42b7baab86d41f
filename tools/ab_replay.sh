#!/bin/bash
# A/B on one box: device replay time and step time of each prebuilt variant (build/variants/<name>.so), interleaved
for rep in 1 2 3; do
  for v in "$@"; do
    cp build/variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
    timeout -k 10 120 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v replay_ms %.3f step_ms %.3f'%(d['stage_ms']['device_replay'], d['ms_per_step']))" || exit 1
  done
done
