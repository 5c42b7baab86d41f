#!/bin/bash
# the --dcfilter tests and bench lines (fused: the kernels filter for themselves; RSB_DC_UNFUSED=1: converter + magnitude buffer)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "dcfilter" > gpurun_out/dc_tests.log 2>&1 || { tail -20 gpurun_out/dc_tests.log; exit 1; }
tail -1 gpurun_out/dc_tests.log
for fmt in SC16 UC8; do
timeout -k 10 300 python bench.py --format $fmt --dcfilter --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${fmt}_dc.json 2> gpurun_out/bench_dc.err
RSB_DC_UNFUSED=1 timeout -k 10 300 python bench.py --format $fmt --dcfilter --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_${fmt}_dc_unfused.json 2>> gpurun_out/bench_dc.err
done
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_default_now.json 2>> gpurun_out/bench_dc.err
