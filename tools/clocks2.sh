#!/bin/bash
# per-stage wave clocks of the scan kernel from RSB_STAGE_CLOCKS variants (variants/<name>.so)
cp readsb_amd/libreadsb_amd.so /tmp/keep.so
for v in "$@"; do
  cp variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
  echo "== $v"
  timeout -k 10 120 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline --steps-in-flight 1 2>&1 | grep "stage clocks" | tail -1 | cut -c1-700 || exit 1
done
cp /tmp/keep.so readsb_amd/libreadsb_amd.so
