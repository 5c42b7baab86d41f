#!/bin/bash
# the round's bench lines on the final tree: default command, every configuration, --dcfilter (fused / converter), 32 channels
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > gpurun_out/final_default.json 2> gpurun_out/final.err || exit 1
timeout -k 10 900 bash tools/bench_configs.sh > gpurun_out/final_configs.jsonl || exit 1
: > gpurun_out/final_dcfilter.jsonl
for fmt in SC16 UC8; do
  timeout -k 10 300 python bench.py --format $fmt --dcfilter --steps 10 --warmup 2 --no-cpu-baseline 2>> gpurun_out/final.err | tail -1 >> gpurun_out/final_dcfilter.jsonl || exit 1
  RSB_DC_UNFUSED=1 timeout -k 10 300 python bench.py --format $fmt --dcfilter --steps 10 --warmup 2 --no-cpu-baseline 2>> gpurun_out/final.err | tail -1 >> gpurun_out/final_dcfilter.jsonl || exit 1
done
timeout -k 10 300 python bench.py --channels-per-gpu 32 --steps-in-flight 1 --no-cpu-baseline 2>> gpurun_out/final.err | tail -1 > gpurun_out/final_ch32_one.json || exit 1
timeout -k 10 300 python bench.py --channels-per-gpu 32 --no-cpu-baseline 2>> gpurun_out/final.err | tail -1 > gpurun_out/final_ch32_two.json || exit 1
wc -l gpurun_out/final_*.json*
