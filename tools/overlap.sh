#!/bin/bash
# two batches in flight: does the replay of batch k overlap the scan of batch k + 1?  (scan grid below 256 leaves CUs free)
set -o pipefail
for cfg in "256 1" "256 2" "240 2" "224 2" "240 4" "224 4" "192 4"; do
  set -- $cfg
  RSB_SCAN_GRID=$1 timeout -k 10 200 python bench.py --steps 5 --batches $2 > gpurun_out/ov_$1_$2.json 2> gpurun_out/ov_$1_$2.err || exit 1
done
python - <<PY
import json
for g,b in ((256,1),(256,2),(240,2),(224,2),(240,4),(224,4),(192,4)):
    d=json.loads(open(f"gpurun_out/ov_{g}_{b}.json").read().strip().splitlines()[-1])
    st=d["stage_ms"]
    print(g, b, round(d["ms_per_step"],3), {k: round(v,2) for k,v in st.items() if k in ("scan","sort","device_replay")}, d["config"]["full_size_check"])
PY
