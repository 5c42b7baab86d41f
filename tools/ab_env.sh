#!/bin/bash
# A/B of environment settings on one box, interleaved: bash tools/ab_env.sh "VAR=1" "VAR=2" ...  (k_scan time and step time)
for rep in 1 2 3; do
  for v in "$@"; do
    env $v timeout -k 10 120 python -u bench.py --steps 6 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v scan_ms %.4f prune_ms %.4f replay_ms %.3f step_ms %.3f %s'%(d['stage_ms']['scan'], d['stage_ms']['prune'], d['stage_ms']['device_replay'], d['ms_per_step'], d['config']['oracle_check']))" || exit 1
  done
done
