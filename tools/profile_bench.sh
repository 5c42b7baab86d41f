#!/bin/bash
# The round's profile of the default bench command: kernel trace + stats, then HBM traffic counters in
# two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md "HBM").
# Run on the GPU box from the repo root: bash tools/profile_bench.sh <tag>
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
rm -rf "$out" && mkdir -p "$out"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$out/trace.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/fetch.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/write.log" 2>&1 || exit 1
find "$out" -name "*.csv" | head -20
