#!/bin/bash
# where a step's wall time goes beyond its stage times (profiles/r3_step_modes.txt): with / without the device-side fence, the next step
# submitted ahead (on all CUs, on fewer), --dcfilter
cd $GRAFT_REPO_ROOT
for flags in "--fence 0" "--fence 1" "--submit-ahead 1" "--submit-ahead 1 --scan-workgroups 224" "--submit-ahead 1 --scan-workgroups 192" \
             "--format SC16 --dcfilter" "--format SC16 --dcfilter --submit-ahead 1" "--format UC8 --dcfilter" "--format SC16"; do
  timeout -k 10 200 python bench.py $flags --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('%-45s step %.3f  scan %.3f prune %.3f replay %.3f copy %.3f sum %.3f %s'%('$flags', d['ms_per_step'], s['scan'], s['prune'], s['device_replay'], s['copy_out'], s['scan']+s['prune']+s['device_replay'], d['config']['oracle_check']))" || exit 1
done
