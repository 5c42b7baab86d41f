"""PCIe-inclusive rate of the live hand-over (rsb_stream): a host-resident recording pushed block by block.
Usage: python tools/stream_rate.py [blocks_per_batch] [Msamples]  -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import readsb_amd
from bench import make_segment

BLOCK = 131072
k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
total = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
seg = make_segment(1 << 23, seed=1, fmt="UC8", dense=False)
ch = readsb_amd.Channel(max_batch_samples=k * BLOCK)
for rep in range(2):  # first pass warms up
    ch.reset()
    st = readsb_amd.Stream(ch, k)
    t0 = time.perf_counter()
    for pos in range(0, total, BLOCK):
        o = pos % seg.shape[0]
        st.push(seg[o:o + BLOCK])
    st.close()
    dt = time.perf_counter() - t0
print(json.dumps({"what": "rsb_stream, UC8, host blocks of 131072 samples -> messages on the host", "blocks_per_batch": k,
                  "samples": total, "seconds": dt, "Msamples_per_s": total / dt / 1e6, "messages": int(st.messages().size)}))
