#!/usr/bin/env python3
"""gpurun_out/prof_<tag> (tools/profile_round.sh) -> the committed summaries:
  profiles/<tag>_kernel_stats.csv   per-kernel calls / average / total of the default bench command
  profiles/<tag>_pmc_scan.json      k_scan's counters per launch, VALU / LDS utilisation derived, HBM-side bytes
  profiles/hbm_traffic.json         what bench.py reports as roofline.traffic (tied to the kernel source by its SHA-256)
usage: python tools/profile_summary.py <tag> [samples_per_launch]"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"prof_{tag}"), os.path.join(root, "profiles")

scan_avg_ns = None
for f in glob.glob(os.path.join(src, "trace_default", "**", "*kernel_stats.csv"), recursive=True)[:1]:
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_default_command.csv"))  # two steps in flight, profiler's blit copy beside the scan
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:9.1f} {r['Percentage']}%")
        if "::k_scan<" in r["Name"]:
            scan_avg_ns = float(r["AverageNs"])

cnt = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "::k_scan<" in r["Kernel_Name"]:
            cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
per = {k: sum(v) / len(v) for k, v in cnt.items()}  # average per k_scan launch
sha = hashlib.sha256()
for f in ("kernels.hip", "kernels.hpp"):
    sha.update(open(os.path.join(root, "readsb_amd", "csrc", f), "rb").read())
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
out = {"kernel": "k_scan<UC8>", "command": "python3 bench.py --warmup 1 --no-cpu-baseline --steps-in-flight 1 --steps 2 (one rocprofv3 --pmc pass per counter group; kernel trace: the same, --steps 5; the default command's trace is <tag>_kernel_stats_default_command.csv)",
       "git_head_at_summary": head, "kernel_src_sha256": sha.hexdigest(), "samples_per_launch": samples,
       "launches_averaged": {k: len(v) for k, v in cnt.items()}, "per_launch": per}
d = {}
if scan_avg_ns:
    d["avg_launch_ms_kernel_trace"] = scan_avg_ns / 1e6
    d["algorithmic_GBps"] = 2.0 * samples / scan_avg_ns
    d["frac_of_8TBps"] = d["algorithmic_GBps"] / 8000.0
if "SQ_INSTS_VALU" in per:
    d["valu_wave_instr_per_sample"] = per["SQ_INSTS_VALU"] / samples
    d["lds_wave_instr_per_sample"] = per.get("SQ_INSTS_LDS", 0) / samples
if "SQ_ACTIVE_INST_VALU" in per and "SQ_BUSY_CYCLES" in per:
    # SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, cycle constants); SQ_BUSY_CYCLES is summed over
    # the shader engines' SQs.  Utilisation against the kernel's own duration: active quad-cycles x 4 / (SIMDs x cycles of the launch)
    if scan_avg_ns and "GRBM_GUI_ACTIVE" in per:
        cyc = per["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        d["launch_cycles_grbm"] = cyc
        d["effective_clock_GHz"] = cyc / scan_avg_ns
        # issue cost measured with tools/ubench_ops.hip (profiles/r2_valu_issue_cost.txt): 64-bit encodings (VOP3, VOP3P, SDWA, DPP) 4 cycles
        # per wave-instruction and SIMD, 32-bit encodings (VOP1/VOP2) ~2.4: the busy share of the VALU lies between the two
        d["valu_busy_if_every_instruction_costs_4_cycles"] = per["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc)
        d["valu_busy_if_every_instruction_costs_2.4_cycles"] = per["SQ_INSTS_VALU"] * 2.4 / (1024.0 * cyc)
        d["lds_busy_fraction"] = per.get("SQ_LDS_IDX_ACTIVE", 0) / (256.0 * cyc)
    if per.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share_of_lds_cycles"] = per.get("SQ_LDS_BANK_CONFLICT", 0) / per["SQ_LDS_IDX_ACTIVE"]
if "FETCH_SIZE" in per:
    # rocprofv3 reports KB; gfx950 tallies the 128-B requests of wide streaming reads at 64 B: x2 (MI355X_MICROARCH.md "HBM")
    d["fetch_bytes_x2"] = per["FETCH_SIZE"] * 1024 * 2
    d["write_bytes"] = per.get("WRITE_SIZE", 0) * 1024
    d["hbm_bytes_per_launch"] = d["fetch_bytes_x2"] + d["write_bytes"]
    d["ratio_to_algorithmic"] = d["hbm_bytes_per_launch"] / (2.0 * samples)
if "TCC_EA0_RDREQ_sum" in per:
    r32, rall = per.get("TCC_EA0_RDREQ_32B_sum", 0), per["TCC_EA0_RDREQ_sum"]
    d["ea_read_requests"], d["ea_read_requests_32B"] = rall, r32
    d["ea_bytes_if_other_requests_are_64B"] = (rall - r32) * 64 + r32 * 32
    d["ea_bytes_if_other_requests_are_128B"] = (rall - r32) * 128 + r32 * 32
    if per.get("TCC_HIT_sum") is not None:
        d["l2_hit_rate"] = per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per.get("TCC_MISS_sum", 0))
if all(k in per for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")):
    wc = per["SQ_WAVE_CYCLES"]
    d["wave_time_share"] = {"issuing": per["SQ_ACTIVE_INST_ANY"] / wc, "ready_but_not_issued": per["SQ_WAIT_INST_ANY"] / wc,
                            "waiting_on_s_waitcnt": per["SQ_WAIT_ANY"] / wc}
out["derived"] = d
with open(os.path.join(dst, f"{tag}_pmc_scan.json"), "w") as f:
    json.dump(out, f, indent=1)
if "hbm_bytes_per_launch" in d:
    with open(os.path.join(dst, "hbm_traffic.json"), "w") as f:
        json.dump({"kernel": "k_scan<UC8>", "samples": samples, "kernel_src_sha256": sha.hexdigest(), "git_head_at_summary": head,
                   "bytes_per_launch": d["hbm_bytes_per_launch"], "algorithmic_bytes": 2 * samples, "ratio_to_algorithmic": d["ratio_to_algorithmic"],
                   "source": f"profiles/{tag}_pmc_scan.json"}, f, indent=1)
print(json.dumps(out["derived"], indent=1))
