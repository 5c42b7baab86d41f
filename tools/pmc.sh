#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>   -- one PMC pass over tools/one_pass.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/one_pass.py frames > gpurun_out/pmc_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, collections, glob, sys
tag = sys.argv[1]
for f in glob.glob(f'gpurun_out/pmc_{tag}/*/*counter_collection.csv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-30:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, v in agg.items():
        if 'k_scan' in k or 'k_' in k:
            print(k, {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
