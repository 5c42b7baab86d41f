#!/bin/bash
# A/B of prebuilt library variants on any bench configuration, whole-step time: BENCH_ARGS="--dcfilter" bash tools/ab_cfg.sh base new
cp readsb_amd/libreadsb_amd.so /tmp/keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
    timeout -k 10 150 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline --steps-in-flight 1 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v [$BENCH_ARGS] step_ms %.3f stages %s'%(d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()}))" || exit 1
  done
done
cp /tmp/keep.so readsb_amd/libreadsb_amd.so
