#!/bin/bash
# kernel trace + stats of a bench command: bash tools/trace_bench.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/trace_$tag
rm -rf "$out" && mkdir -p "$out"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 bench.py --warmup 1 --steps 3 --no-cpu-baseline "$@" > "$out/run.log" 2>&1 || exit 1
python3 tools/timeline.py "$out/t_kernel_trace.csv" > "$out/timeline.txt"
python3 - "$out/t_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    n = n.split("(")[0][-40:] if "rocprim" not in n else "rocprim:" + n[n.find("detail::") + 8:][:32]
    print(f"{n:42s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:9.1f} {r['Percentage']}%")
PY
