#!/bin/bash
# where a step's wall time goes beyond its stage times: with / without the device-side fence, --dcfilter with / without submit-ahead
cd $GRAFT_REPO_ROOT
for flags in "--fence 0" "--fence 1" "--format SC16 --dcfilter --submit-ahead 0" "--format SC16 --dcfilter --submit-ahead 1" "--format SC16 --dcfilter --fence 1" "--format UC8 --dcfilter --submit-ahead 0" "--format UC8 --dcfilter --submit-ahead 1" "--format SC16"; do
  timeout -k 10 200 python bench.py $flags --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('%-45s step %.3f  scan %.3f prune %.3f replay %.3f copy %.3f sum %.3f'%('$flags', d['ms_per_step'], s['scan'], s['prune'], s['device_replay'], s['copy_out'], s['scan']+s['prune']+s['device_replay']))" || exit 1
done
