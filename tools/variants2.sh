#!/bin/bash
# A/B of prebuilt library variants (variants/*.so), interleaved: [BENCH_ARGS="--format SC16"] bash tools/variants2.sh a1d1 a2d2 ...   (scan and step time, one step at a time)
cp readsb_amd/libreadsb_amd.so /tmp/keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
    timeout -k 10 120 python -u bench.py --steps 6 --warmup 1 --no-cpu-baseline --steps-in-flight 1 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v scan_ms %.4f replay_ms %.4f step_ms %.3f %s'%(d['stage_ms']['scan'], d['stage_ms']['device_replay'], d['ms_per_step'], d['config']['oracle_check']))" || exit 1
  done
done
cp /tmp/keep.so readsb_amd/libreadsb_amd.so
