#!/usr/bin/env python3
"""Basic-block instruction statistics of one kernel in a hipcc -S listing.
usage: python tools/isa_blocks.py build/kernels.s k_scanILi0E [min_instructions]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", collections.Counter(), start]
blocks.append(cur)
for i in range(start + 1, end):
    t = lines[i].strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        cur = [m.group(1), collections.Counter(), i + 1]
        blocks.append(cur)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
           else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
    cur[1][cls] += 1
    cur[1]["op:" + op] += 1
tot = collections.Counter()
for b in blocks:
    c = b[1]
    n = sum(v for k, v in c.items() if not k.startswith("op:"))
    for k, v in c.items():
        tot[k] += v
    if n >= minsz:
        ops = ", ".join(f"{k[3:]} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if k.startswith("op:"))[:400]
        print(f"{b[0]:12s} line {b[2]:6d}  " + " ".join(f"{k}={c[k]}" for k in ("valu", "salu", "lds", "vmem") if c[k]) + "\n      " + ops)
print("TOTAL", {k: v for k, v in tot.items() if not k.startswith("op:")})
