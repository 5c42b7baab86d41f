#!/usr/bin/env python3
"""VGPRs, spills, scratch and LDS of the library's kernels, read from the code object inside the .so:
   python tools/kernel_resources.py [library.so] [name filter, default k_]"""
import re
import subprocess
import sys
import tempfile

lib = sys.argv[1] if len(sys.argv) > 1 else "readsb_amd/libreadsb_amd.so"
flt = sys.argv[2] if len(sys.argv) > 2 else "rsb"
data = open(lib, "rb").read()
for start in [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data)]:
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(data[start:])
        f.flush()
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    for block in notes.split("- .agpr_count:")[1:]:
        get = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, block) or [None, "?"])[1]
        name = get("name")
        if flt not in name:
            continue
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = short.replace("(anonymous namespace)::", "").replace("void ", "").replace("rsb::", "")
        short = re.sub(r"\(.*", "", short)
        print(f"{short:44s} vgpr {get('vgpr_count'):>4s} spill {get('vgpr_spill_count'):>3s} scratch {get('private_segment_fixed_size'):>4s} B  lds {get('group_segment_fixed_size'):>6s} B  sgpr {get('sgpr_count'):>4s}")
