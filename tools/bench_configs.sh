#!/bin/bash
# one bench line per BASELINE configuration (no CPU baseline): bash tools/bench_configs.sh > gpurun_out/bench_configs.jsonl
run() { timeout -k 10 300 python -u bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 || exit 1; }
run
run --steps-in-flight 1
run --output beast
run --format SC16
run --format SC16Q11
run --format SC16 --dcfilter
run --dcfilter
run --modeac
run --modeac --fix-bits 2 --dense
run --modeac --format SC16
run --channels-per-gpu 32
run --channels-per-gpu 32 --steps-in-flight 1
run --shard-recording 134217728 --shard-contexts 4
run --batches 4 --in-flight 4
