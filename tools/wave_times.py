#!/usr/bin/env python3
"""Start / end stamps of every k_scan wavefront (RSB_WAVE_TIMES builds the stamps into the library's debug buffer):
is the persistent scan balanced?  Two recordings of 2^30 samples: the benchmark's (128 copies of one 8 Mi segment) and a
non-repeating one (16 different segments in random order).  Writes gpurun_out/wave_times.json (copied to profiles/).

  python3 tools/wave_times.py [out.json]
"""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(tempfile.gettempdir(), "rsb_wave_times.bin")
os.environ["RSB_WAVE_TIMES"] = path

import numpy as np  # noqa: E402
import torch  # noqa: E402

import readsb_amd  # noqa: E402
from readsb_amd.synth import Scenario, generate  # noqa: E402


def measure(iq, label):
    n = iq.shape[0]
    ch = readsb_amd.Channel(max_batch_samples=n, queue_div=32)
    for _ in range(3):
        ch.reset()
        ch.submit(iq.data_ptr(), n, 0)
        m = ch.collect(1 << 22)
    t = ch.timing()
    w = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    start, end, hw, strips = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64), w[:, 2], w[:, 3]
    t0 = start.min()
    span = float(end.max() - t0)
    dur = (end - start).astype(np.float64)
    rel_end = (end - t0) / span
    rel_start = (start - t0) / span
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
    hist, edges = np.histogram(rel_end, bins=20, range=(0.0, 1.0))
    per_xcc = {}
    for x in sorted(set(xcc.tolist())):
        sel = xcc == x
        per_xcc[str(x)] = {"waves": int(sel.sum()), "mean_end": float(rel_end[sel].mean()), "min_end": float(rel_end[sel].min()),
                           "max_end": float(rel_end[sel].max()), "mean_duration_us": float(dur[sel].mean() / 100.0)}
    # per position of the wavefront's run inside the repeating segment (32 runs of 262144 samples per 8 Mi segment)
    k = np.arange(w.shape[0])
    by_phase = [float(dur[k % 32 == p].mean() / 100.0) for p in range(32)]
    res = {"recording": label, "samples": int(n), "messages": int(m.size), "scan_ms_events": float(t["scan_ms"]),
           "waves": int(w.shape[0]), "strips_per_wave": int(strips.max()), "tick_ns": 10,
           "span_us": span / 100.0, "mean_wave_duration_us": float(dur.mean() / 100.0), "min_wave_duration_us": float(dur.min() / 100.0),
           "max_wave_duration_us": float(dur.max() / 100.0), "mean_resident_fraction": float(dur.mean() / span),
           "latest_start_fraction": float(rel_start.max()),
           "end_time_quantiles": {q: float(np.quantile(rel_end, float(q))) for q in ("0.01", "0.1", "0.25", "0.5", "0.75", "0.9", "0.99")},
           "end_time_histogram_20_bins": hist.tolist(), "per_xcc": per_xcc, "mean_duration_us_by_run_position_mod_32": by_phase}
    print(json.dumps(res))
    return res


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wave_times.json"
    dev = torch.device("cuda", 0)
    seg = generate(Scenario(1 << 23, seed=1))
    iq = torch.from_numpy(seg).to(dev).repeat(128, 1).contiguous()
    results = [measure(iq, "bench: 128 copies of one 8 Mi segment")]
    del iq
    rng = np.random.default_rng(7)
    segs = [torch.from_numpy(generate(Scenario(1 << 23, seed=100 + s))).to(dev) for s in range(16)]
    iq = torch.cat([segs[i] for i in rng.integers(0, 16, 128)]).contiguous()
    del segs
    results.append(measure(iq, "16 different 8 Mi segments, 128 draws in random order"))
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    with open(out, "w") as f:
        json.dump({"kernel": "k_scan<UC8>", "how": "wall_clock64() at wavefront start and end, lane 0 (tools/wave_times.py)", "results": results}, f, indent=1)


if __name__ == "__main__":
    main()
