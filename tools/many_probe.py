"""Where the 32-channel step goes: host time to enqueue the 32 replay chains against the time the GPU needs for them.
   python tools/many_probe.py   (GPU box)"""
import time

import numpy as np
import torch

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import readsb_amd
from readsb_amd.synth import Scenario, generate

C, B = 32, 131072
n = 256 * B  # 32 Msamples per channel
seg = generate(Scenario(8 << 20, seed=1, mix={"df17": 0.7, "df11": 0.3}, collide_every=0, df11_iid_every=0))
iq = torch.from_numpy(np.tile(seg, (n // seg.shape[0], 1))).cuda()
chans = [readsb_amd.Channel(fmt="UC8", fix_bits=1, max_batch_samples=n, queue_div=32) for _ in range(C)]
outs = [torch.empty((1 << 16) * 56, dtype=torch.uint8, pin_memory=True).numpy().view(readsb_amd.MSG_DTYPE) for _ in range(C)]
for rep in range(3):
    for c in chans:
        c.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    readsb_amd.submit_many(chans, [iq.data_ptr()] * C, [n] * C)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for c in chans:
        c.collect_prepare()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    got = readsb_amd.collect_many(chans, outs)
    t5 = time.perf_counter()
    print(f"submit_many host {1e3 * (t1 - t0):.2f} ms, scans done after {1e3 * (t2 - t0):.2f}; 32 x collect_prepare host {1e3 * (t3 - t2):.2f} ms, chains done after {1e3 * (t4 - t2):.2f}; "
          f"collect_many (results ready) {1e3 * (t5 - t4):.2f} ms; messages {sum(g.size for g in got)}")
