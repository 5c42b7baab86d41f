"""Does a 41 MB device -> pinned-host copy on its own stream run beside the scan kernel without slowing it (SDMA), and how fast?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, readsb_amd
from readsb_amd.synth import Scenario, generate
seg = generate(Scenario(1 << 23, seed=1000))
iq = torch.from_numpy(seg).cuda().repeat(64, 1).contiguous()   # 512 Mi samples
n = iq.shape[0]
ch = readsb_amd.Channel(max_batch_samples=n)
src = torch.empty(41 << 20, dtype=torch.uint8, device="cuda")
dst = torch.empty(41 << 20, dtype=torch.uint8, pin_memory=True)
cs = torch.cuda.Stream()
def scan():
    ch.reset(); ch.submit(iq.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
def finish():
    ch.collect(1 << 20)
for mode in ("scan only", "copy only", "scan + copy"):
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1, c0, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        t0 = time.perf_counter()
        if mode != "copy only":
            e0.record(); scan(); e1.record()
        if mode != "scan only":
            with torch.cuda.stream(cs):
                c0.record(); dst.copy_(src, non_blocking=True); c1.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        msg = f"{mode:12s} wall {dt*1e3:6.2f} ms"
        if mode != "copy only": msg += f"  scan+prune {e0.elapsed_time(e1):6.3f} ms"
        if mode != "scan only": msg += f"  copy {c0.elapsed_time(c1):6.3f} ms = {41*1.048576/c0.elapsed_time(c1):5.1f} GB/s"
        print(msg)
        if mode != "copy only": finish()
