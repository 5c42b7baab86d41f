#!/usr/bin/env python3
"""All kernels of the last bench step with start / end / queue: python tools/timeline2.py <kernel_trace.csv> [n_scans_per_step]"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
idx = [i for i, r in enumerate(rows) if "k_scan<" in r["Kernel_Name"]]
i0 = idx[-per]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    n = r["Kernel_Name"]
    m = re.search(r"(k_\w+)", n)
    n = m.group(1) if m else n[:36]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - t0) / 1e3 > 20000: break
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  +{(e - s) / 1e3:8.1f}  q{r.get('Queue_Id','?'):>3s}  grid {r.get('Grid_Size','?'):>9s} wg {r.get('Workgroup_Size','?'):>5s}  {n}")
