#!/usr/bin/env python3
"""Kernel timeline of the last bench step in a rocprofv3 --kernel-trace CSV: python tools/timeline.py <kernel_trace.csv> [memory_copy_trace.csv]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["_k"] = "K"
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        r["_k"] = "C"; r["Kernel_Name"] = "memcpy " + r.get("Direction", "")
        rows.append(r)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_scan<" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:i0 + 70]:
    n = r["Kernel_Name"]
    n = n.split("(")[0][-34:] if "rocprim" not in n else "rocprim:" + n[n.find("detail::") + 8:][:40]
    if "k_scan<" in n and r is not rows[i0]:
        break
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n}")
    prev_end = max(prev_end, e)
