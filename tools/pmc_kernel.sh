#!/bin/bash
# SQ / TCC counters of one kernel (name fragment) over the default bench command: bash tools/pmc_kernel.sh <tag> <kernel fragment> [bench args]
tag=$1; frag=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmck_$tag
rm -rf "$out" && mkdir -p "$out"
B="python3 bench.py --warmup 1 --steps 2 --no-cpu-baseline $*"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d "$out/a" -o b -- $B > "$out/a.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d "$out/b" -o b -- $B > "$out/b.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$out/c" -o b -- $B > "$out/c.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$out/d" -o b -- $B > "$out/d.log" 2>&1
python3 - "$out" "$frag" <<'PY'
import csv, glob, sys, collections
out, frag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if frag in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
print("vgpr sgpr lds grid wg:", meta)
for k in sorted(agg): print(f"{k:36s} {sum(agg[k]) / len(agg[k]):16.0f}  (n={len(agg[k])})")
PY
