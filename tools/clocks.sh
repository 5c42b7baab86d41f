#!/bin/bash
# per-stage wave clocks of the scan kernel from an RSB_STAGE_CLOCKS build (build/variants/<name>.so)
for v in "$@"; do
  cp build/variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
  echo "== $v"
  timeout -k 10 120 python -u bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "stage clocks\|scan" | tail -2 | cut -c1-700 || exit 1
done
