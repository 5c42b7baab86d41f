#!/bin/bash
# two steps in flight against one, and the scan grid that leaves room for the tail: bash tools/overlap_steps.sh
for cfg in "2 -1"; do
  set -- $cfg
  timeout -k 10 150 python -u bench.py --steps 8 --warmup 1 --no-cpu-baseline --steps-in-flight $1 --scan-workgroups $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps-in-flight $1 grid $2: step_ms %.3f (one at a time %s) scan %.3f prune %.3f replay %.3f copy %.3f frac %.4f %s %s'%(d['ms_per_step'], d['config'].get('ms_per_step_one_at_a_time'), d['stage_ms']['scan'], d['stage_ms']['prune'], d['stage_ms']['device_replay'], d['stage_ms']['copy_out'], d['roofline']['frac'], d['config']['oracle_check'], d['config']['full_size_check']))" || exit 1
done
