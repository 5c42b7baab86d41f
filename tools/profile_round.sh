#!/bin/bash
# The round's evidence for the default bench command (configs[1]), every pass a run of its own:
#   trace   rocprofv3 --kernel-trace --stats
#   sq      SQ counters of k_scan (VALU / LDS instruction counts, active cycles, bank conflicts, wave and busy cycles)
#   fetch / write / tcc   HBM-side counters (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md "HBM")
# usage (GPU box, repo root): bash tools/profile_round.sh <tag>      then here: python tools/profile_summary.py <tag>
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag
rm -rf "$out" && mkdir -p "$out"
T="python3 bench.py --warmup 1 --no-cpu-baseline"
# counters are collected device-wide while the kernel runs: one step at a time, so that no record copy of the step before runs beside k_scan
B="$T --steps-in-flight 1"
# kernel trace, twice: one step at a time (what the kernel costs; this is the summary bench.py's roofline must agree with) and the
# default command (two steps in flight).  Under rocprofv3 the device -> host record copy is a blit KERNEL that shares the CUs with
# the next step's scan (2.2 instead of 1.6 ms per scan launch); without the profiler bench.py's own HIP-event timing of the scan
# is the same in both modes, so the default trace is kept for the record, not for the roofline.
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- $B --steps 5 > "$out/trace.log" 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_default" -o bench -- $T --steps 5 > "$out/trace_default.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d "$out/sq" -o bench -- $B --steps 2 > "$out/sq.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES --output-format csv -d "$out/sq2" -o bench -- $B --steps 2 > "$out/sq2.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o bench -- $B --steps 2 > "$out/fetch.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o bench -- $B --steps 2 > "$out/write.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/tcc" -o bench -- $B --steps 2 > "$out/tcc.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/grbm" -o bench -- $B --steps 2 > "$out/grbm.log" 2>&1 || exit 1
find "$out" -name "*.csv" | wc -l
