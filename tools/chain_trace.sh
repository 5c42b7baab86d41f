#!/bin/bash
# Per-kernel times of the chain for one or more prebuilt variants (variants/<name>.so): rocprofv3 kernel trace of the bench, one step at a time.
#   bash tools/chain_trace.sh name1 name2 ...   -> gpurun_out/chain_trace_<name>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cp readsb_amd/libreadsb_amd.so /tmp/tree_lib.so
for v in "$@"; do
  cp variants/$v.so readsb_amd/libreadsb_amd.so || exit 1
  out=gpurun_out/ct_$v
  rm -rf "$out"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 bench.py --warmup 1 --no-cpu-baseline --steps-in-flight 1 --steps 5 > "$out.log" 2>&1 || exit 1
  python3 - "$out" > gpurun_out/chain_trace_$v.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'k_prune' in n or 'k_replay' in n:
        k = n[n.index("k_"):].split("(")[0]
        print('%-40s %3s %9.1f us' % (k, r['Calls'], float(r['AverageNs']) / 1e3))
        tot += float(r['AverageNs']) / 1e3
    elif 'k_scan' in n:
        print('%-40s %3s %9.1f us' % ('k_scan', r['Calls'], float(r['AverageNs']) / 1e3))
print('chain total %.1f us' % tot)
PY
  cat gpurun_out/chain_trace_$v.txt
done
cp /tmp/tree_lib.so readsb_amd/libreadsb_amd.so
