import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, readsb_amd
from readsb_amd.synth import Scenario, generate
seg = generate(Scenario(1 << 23, seed=1000))
iq = torch.from_numpy(seg).cuda().repeat(16, 1).contiguous()
n = iq.shape[0]
ch = readsb_amd.Channel(max_batch_samples=n)
ch.submit(iq.data_ptr(), n, 0); m = ch.collect(1 << 20)
st = ch.stats(); print({k: st[k] for k in ("samples_processed","gated_offsets","live_candidates","survivors")}, m.size)
