"""Which kernels can run beside the persistent scan kernel when its grid leaves CUs free?  usage: RSB_SCAN_GRID=240 python tools/coexec_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, readsb_amd
from readsb_amd.synth import Scenario, generate
seg = generate(Scenario(1 << 23, seed=1000))
iq = torch.from_numpy(seg).cuda().repeat(64, 1).contiguous()   # 512 Mi samples: ~0.9 ms of scan
n = iq.shape[0]
ch = readsb_amd.Channel(max_batch_samples=n)
x = torch.rand(1 << 16, device="cuda"); y = torch.rand(1 << 20, device="cuda")
side = torch.cuda.Stream()
def probes():
    out = []
    for name, fn in (("add 64K (no LDS)", lambda: x.add_(1.0)), ("cumsum 64K (LDS)", lambda: torch.cumsum(x, 0)), ("sort 1M (LDS, many WGs)", lambda: torch.sort(y)),
                     ("sum 1M", lambda: y.sum())):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); out.append((name, a, b))
    return out
for mode in ("alone", "beside the scan"):
    for rep in range(2):
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if mode != "alone":
            ch.reset(); s0.record(); ch.submit(iq.data_ptr(), n, torch.cuda.current_stream().cuda_stream); s1.record()
            time.sleep(0.0002)
        with torch.cuda.stream(side):
            ev = probes()
        torch.cuda.synchronize()
        line = f"grid {os.environ.get('RSB_SCAN_GRID','256')} {mode:16s}" + (f" scan {s0.elapsed_time(s1):.3f} ms |" if mode != "alone" else " |")
        print(line, "  ".join(f"{nm}: {a.elapsed_time(b)*1e3:.0f} us" for nm, a, b in ev))
        if mode != "alone": ch.collect(1 << 20)
