import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, readsb_amd
from readsb_amd.synth import Scenario, generate
seg = generate(Scenario(1 << 23, seed=1000))
iq = torch.from_numpy(seg).cuda().repeat(16, 1).contiguous()
n = iq.shape[0]
ch = readsb_amd.Channel(max_batch_samples=n)
ch.submit(iq.data_ptr(), n, 0); m = ch.collect(1 << 20)
st = ch.stats(); print({k: v for k, v in st.items()}, 'msgs', m.size, 'timing', ch.timing())
print('per 2048-offset strip: gated %.1f items %.1f survivors %.1f' % (st['gated_offsets'] / (n / 2048), st['live_candidates'] / (n / 2048), st['survivors'] / (n / 2048)))
noise = generate(Scenario(1 << 23, seed=5, frames_per_sec=0, noise_lsb=3.0))
iq = torch.from_numpy(noise).cuda().repeat(16, 1).contiguous()
ch.reset(); ch.submit(iq.data_ptr(), n, 0); m = ch.collect(1 << 20)
st = ch.stats(); print('noise', st, ch.timing())
