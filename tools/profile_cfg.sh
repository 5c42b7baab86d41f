#!/bin/bash
# kernel trace + stats of one bench configuration: bash tools/profile_cfg.sh <tag> <bench flags...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
rm -rf "$out" && mkdir -p "$out"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/trace.log" 2>&1 || exit 1
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'][:90]:90s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:10.1f} {r['Percentage']}%")
PY
