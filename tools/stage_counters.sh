#!/bin/bash
# k_scan's counters when it stops after stage k (1 convert + comparisons, 2 + edge patterns / queue, 3 + gate, 4 + DF, 5 = all):
# per-stage VALU / LDS instruction and LDS cycle budget.  bash tools/stage_counters.sh <tag>
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/stages_$tag
rm -rf "$out" && mkdir -p "$out"
for k in 1 2 3 4 5; do
  RSB_STAGE_LIMIT=$k timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$out/s$k" -o b -- python3 bench.py --warmup 1 --steps 2 --no-cpu-baseline > "$out/s$k.log" 2>&1 || exit 1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for k in range(1, 6):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/s{k}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "::k_scan<" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    res[k] = {c: sum(v) / len(v) for c, v in agg.items()}
n = float(1 << 30)
prev = collections.defaultdict(float)
names = {1: "convert + comparisons", 2: "edge patterns + queue", 3: "gate", 4: "DF of 5 phases + frame list", 5: "slicing + CRC + classify"}
table = []
for k in range(1, 6):
    r = res[k]
    row = {"stage": names[k], "cumulative_ms": r["_ns"] / 1e6,
           "valu_wave_instr_per_sample": (r["SQ_INSTS_VALU"] - prev["SQ_INSTS_VALU"]) / n,
           "lds_wave_instr_per_sample": (r["SQ_INSTS_LDS"] - prev["SQ_INSTS_LDS"]) / n,
           "lds_cycles_per_sample": (r["SQ_LDS_IDX_ACTIVE"] - prev["SQ_LDS_IDX_ACTIVE"]) / n,
           "lds_conflict_cycles_per_sample": (r["SQ_LDS_BANK_CONFLICT"] - prev["SQ_LDS_BANK_CONFLICT"]) / n,
           "valu_wave_instr_per_strip": (r["SQ_INSTS_VALU"] - prev["SQ_INSTS_VALU"]) / n * 2768,
           "lds_wave_instr_per_strip": (r["SQ_INSTS_LDS"] - prev["SQ_INSTS_LDS"]) / n * 2768}
    table.append(row)
    prev = r
    print(json.dumps(row))
json.dump({"note": "k_scan<UC8> stopped after stage k (RSB_STAGE_LIMIT), 2^30 samples per launch; rows are DIFFERENCES between consecutive limits; profiled durations",
           "stages": table}, open(f"{out}/stage_counters.json", "w"), indent=1)
PY
