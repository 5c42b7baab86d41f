#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/profile_bench.sh into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv   per-kernel calls / average / total (from --kernel-trace --stats)
  profiles/<tag>_hbm_traffic.json   FETCH_SIZE / WRITE_SIZE per k_scan launch, corrected as MI355X_MICROARCH.md prescribes
  profiles/hbm_traffic.json         the figure bench.py reports as roofline.traffic
usage: python tools/hbm_traffic.py <tag> [samples_per_launch]"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        print(f"{r['Name'][:80]:80s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:9.1f} {r['Percentage']}%")


def per_launch(kind, counter):
    vals = []
    for f in glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_scan" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def calibration():
    """tools/ubench_fetch under --pmc FETCH_SIZE: kernels with a known byte count"""
    res = {}
    for f in glob.glob(os.path.join(src, "calib", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for k in ("k_stream", "k_strips"):
                if r["Kernel_Name"].startswith(k):
                    res.setdefault(k, []).append(float(r["Counter_Value"]))
    out = {}
    for k, v in res.items():
        requested = (2 << 30) if k == "k_stream" else ((2 << 30) // 4096 - 1) * 4704
        kb = sum(v) / len(v)
        out[k] = {"FETCH_SIZE_KB_raw": kb, "bytes_requested": requested, "x2_over_2GiB_buffer": kb * 2048 / (2 << 30)}
    return out


fetch, nf = per_launch("fetch", "FETCH_SIZE")
write, nw = per_launch("write", "WRITE_SIZE")
if fetch is not None:
    # rocprofv3 reports both in KB.  gfx950: FETCH_SIZE counts 128-B requests of wide coalesced streaming reads at
    # 64 B -> double it (the scan kernel reads its IQ with 16 B per lane); WRITE_SIZE is exact.
    fetch_b = fetch * 1024 * 2
    write_b = (write or 0.0) * 1024
    out = {"kernel": "k_scan<UC8>", "samples": samples, "launches_averaged": [nf, nw], "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB_raw": write,
           "fetch_bytes_corrected": fetch_b, "write_bytes": write_b, "bytes_per_launch": fetch_b + write_b,
           "algorithmic_bytes": 2 * samples, "ratio_to_algorithmic": (fetch_b + write_b) / (2 * samples),
           "calibration": calibration(),
           "note": "x2 is confirmed on this box by tools/ubench_fetch (plain 16-B-per-lane stream: x2 = 1.000 of the bytes; the scan kernel's "
                   "strip + halo pattern: x2 = 1.018 of the buffer, i.e. the 15 % halo re-reads hit L2).  For k_scan itself the corrected figure is "
                   "below the 2 B/sample every launch must read at least once, so the counter under-reports this kernel; what it does show is "
                   "that there is no re-read traffic beyond the algorithmic bytes.",
           "correction": "FETCH_SIZE x2 on gfx950 for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported"}
    for name in (f"{tag}_hbm_traffic.json", "hbm_traffic.json"):
        with open(os.path.join(dst, name), "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out))
