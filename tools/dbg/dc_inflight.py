import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import readsb_amd
from readsb_amd.synth import generate
from scenarios import SCENARIOS
name = sys.argv[1] if len(sys.argv) > 1 else "uc8_dc"
sc, kw = SCENARIOS[name]
iq = generate(sc)
n, nb = sc.n_samples, 2 * 131072
bps = 2 if sc.fmt == "UC8" else 4
dev = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
cuts = list(range(0, n, nb)) + [n]
def seq(mags):
    ch = readsb_amd.Channel(fmt=sc.fmt, fix_bits=kw["nfix"], mode_ac=False, dc_filter=True, max_batch_samples=nb)
    parts = []
    for a, b in zip(cuts, cuts[1:]):
        ch.submit(dev.data_ptr() + bps * a, b - a)
        parts.append(ch.collect(cap=1 << 14).copy())
        if mags: ch.batch_magnitudes(b - a)
    ch.close()
    return parts
def fl():
    ch = readsb_amd.Channel(fmt=sc.fmt, fix_bits=kw["nfix"], mode_ac=False, dc_filter=True, max_batch_samples=nb)
    again = []
    ch.submit(dev.data_ptr(), cuts[1])
    for i in range(1, len(cuts)):
        if i + 1 < len(cuts):
            ch.submit(dev.data_ptr() + bps * cuts[i], cuts[i + 1] - cuts[i])
        again.append(ch.collect(cap=1 << 14).copy())
    ch.close()
    return again
runs = {"seq_mags": seq(True), "seq": seq(False), "fl1": fl(), "fl2": fl()}
base = runs["seq_mags"]
for k, v in runs.items():
    print(k, [p.size for p in v], [p.tobytes() == q.tobytes() for p, q in zip(v, base)])
    for bi, (p, q) in enumerate(zip(v, base)):
        if p.tobytes() != q.tobytes():
            m = min(p.size, q.size)
            d = [i for i in range(m) if p[i].tobytes() != q[i].tobytes()]
            print("  batch", bi, "first diffs", d[:5], "of", len(d))
            i = d[0] if d else m - 1
            print("   got ", p[i]); print("   want", q[i])
            break
