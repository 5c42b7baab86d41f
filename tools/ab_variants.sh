#!/bin/bash
# A/B on one box: device replay time and step time of each prebuilt variant (variants/<name>.so, tools/mkvariant.sh), interleaved;
# the tree's own library is put back at the end.   bash tools/ab_variants.sh "<bench flags>" name1 name2 ...
flags=$1; shift
cp readsb_amd/libreadsb_amd.so /tmp/tree_lib.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
    timeout -k 10 160 python -u bench.py $flags --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v prune_ms %.3f replay_ms %.3f step_ms %.3f'%(d['stage_ms']['prune'], d['stage_ms']['device_replay'], d['ms_per_step']))" || exit 1
  done
done
cp /tmp/tree_lib.so readsb_amd/libreadsb_amd.so
