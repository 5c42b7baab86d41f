#!/usr/bin/env python3
"""where the time of one --shard-recording step goes on the host (cProfile of simulate_parallel, one context)"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import readsb_amd  # noqa: E402
from readsb_amd import dist as rd  # noqa: E402
from readsb_amd.synth import Scenario, generate  # noqa: E402

dev = torch.device("cuda", 0)
seg = generate(Scenario(1 << 23, seed=1000))
iq = torch.from_numpy(seg).to(dev).repeat(128, 1).contiguous()
n, S = iq.shape[0], 1 << 27
ch = readsb_amd.Channel(max_batch_samples=S, defer_prune=True)
carry = lambda a: np.ascontiguousarray(seg[np.arange(a - 326, a) % seg.shape[0]]) if a else None
run = lambda: rd.simulate_parallel([ch], n, S, lambda a, b: iq.data_ptr() + 2 * a, carry, cap=1 << 19)
for _ in range(2):
    ch.reset()
    run()
ch.reset()
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
