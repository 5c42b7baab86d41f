#!/bin/bash
for cfg in "256 1 0" "256 1 1" "240 2 0" "240 4 0" "224 4 0" "240 8 0"; do
  set -- $cfg
  RSB_EMIT_DIRECT=$3 RSB_SCAN_GRID=$1 timeout -k 10 200 python bench.py --steps 5 --batches $2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['stage_ms']
print('grid $1 batches $2 direct $3 step_ms %.3f' % d['ms_per_step'], {k: round(v,2) for k,v in st.items() if k in ('scan','prune','sort','device_replay','copy_out')}, d['config']['full_size_check'], d['config']['oracle_check'])" || exit 1
done
