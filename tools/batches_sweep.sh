#!/bin/bash
# step time against the number of batches a 1-Gsample step is submitted in (pipelined tail): bash tools/batches_sweep.sh
for cfg in "1 3 -1" "4 4 224" "8 4 224" "8 4 232" "16 4 224" "4 4 -1"; do
  set -- $cfg
  timeout -k 10 150 python -u bench.py --steps 6 --warmup 1 --no-cpu-baseline --batches $1 --in-flight $2 --scan-workgroups $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batches $1 in-flight $2 grid $3: step_ms %.3f scan %.3f prune %.3f replay %.3f frac %.4f %s'%(d['ms_per_step'], d['stage_ms']['scan'], d['stage_ms']['prune'], d['stage_ms']['device_replay'], d['roofline']['frac'], d['config']['oracle_check']))" || exit 1
done
