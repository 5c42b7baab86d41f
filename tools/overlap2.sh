#!/bin/bash
# batches in flight: does the tail of batch k hide behind the scan of batch k + 1 when the scan leaves CUs free?
for cfg in "256 1" "240 2" "240 4" "224 4" "240 8" "224 8"; do
  set -- $cfg
  RSB_SCAN_GRID=$1 timeout -k 10 200 python bench.py --steps 5 --batches $2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['stage_ms']
print('grid $1 batches $2 step_ms %.3f' % d['ms_per_step'], {k: round(v,2) for k,v in st.items() if k in ('scan','prune','sort','device_replay')}, d['config']['full_size_check'], d['config']['oracle_check'])" || exit 1
done
