import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, readsb_amd
from readsb_amd.synth import Scenario, generate
noise = generate(Scenario(1 << 22, seed=5, frames_per_sec=2353 if len(sys.argv) > 1 else 0, noise_lsb=3.0))
iq = torch.from_numpy(noise).cuda().repeat(64, 1).contiguous()
n = iq.shape[0]
ch = readsb_amd.Channel(max_batch_samples=n)
for _ in range(3):
    ch.reset(); ch.submit(iq.data_ptr(), n, 0); m = ch.collect(1 << 20)
print(ch.stats(), ch.timing())
