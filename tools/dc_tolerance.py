#!/usr/bin/env python3
"""What the --dcfilter path achieves against the reference, per scenario (GPU box): python tools/dc_tolerance.py > gpurun_out/dc_tolerance.json
max |magnitude - exact double recurrence|, max |magnitude - reference's float32 recurrence| (the oracle's converter is pinned to
the reference's by tests/golden/convert_*_dc.npz), messages in only one of the two lists, largest relative signal-level
difference among the common ones.  tests/test_gpu_parity.py::test_dcfilter_stream holds 2x these numbers."""
import json
import os
import sys

import numpy as np
from scipy.signal import lfilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import readsb_amd  # noqa: E402
from oracle.oracle import Oracle, REC_DTYPE  # noqa: E402
from readsb_amd.synth import generate  # noqa: E402
from scenarios import SCENARIOS  # noqa: E402


def measure(name):
    sc, kw = SCENARIOS[name]
    iq = generate(sc)
    n = sc.n_samples
    ref, _ = Oracle(fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"], dc_filter=True).run_stream(iq)
    ch = readsb_amd.Channel(fmt=sc.fmt, fix_bits=kw["nfix"], mode_ac=kw["modeac"], dc_filter=True, max_batch_samples=1 << 22)
    got = ch.process_host(iq).view(REC_DTYPE)
    mag, _ = ch.batch_magnitudes(n, want_means=kw["modeac"])
    ch.close()
    raw = np.ascontiguousarray(iq).reshape(n, 2)
    if sc.fmt == "UC8":
        x = ((raw.astype(np.float32) - np.float32(127.5)) / np.float32(127.5)).astype(np.float64)
    else:
        x = (raw.astype(np.float32) / np.float32(32768.0 if sc.fmt == "SC16" else 2048.0)).astype(np.float64)
    dc_b = float(np.float32(np.exp(-2.0 * np.pi / 2.4e6)))
    dc_a = float(np.float32(1.0 - dc_b))
    y = x - lfilter([dc_a], [1.0, -dc_b], x, axis=0)
    exact = np.floor(np.sqrt(np.minimum(1.0, (y * y).sum(axis=1))) * 65535.0 + 0.5).astype(np.int64)
    refmag = Oracle(fmt=sc.fmt, dc_filter=True).convert(iq)[0].astype(np.int64)
    m = mag.astype(np.int64)
    key = lambda r: (int(r["ts"]), bytes(r["msg"]), int(r["msgbits"]))
    rd, gd = {key(r): r for r in ref}, {key(r): r for r in got}
    common = rd.keys() & gd.keys()
    rel = max((abs(rd[k]["signal"] - gd[k]["signal"]) / abs(rd[k]["signal"]) for k in common if rd[k]["signal"] != 0), default=0.0)
    fields_differ = sum(any(rd[k][f] != gd[k][f] for f in ("msgtype", "crc", "correctedbits", "addr", "score", "source", "iid")) for k in common)
    return {"samples": n, "format": sc.fmt,
            "max_abs_diff_vs_exact_recurrence_lsb": int(np.abs(m - exact).max()),
            "max_abs_diff_vs_reference_lsb": int(np.abs(m - refmag).max()),
            "p999_abs_diff_vs_reference_lsb": float(np.quantile(np.abs(m - refmag), 0.999)),
            "reference_vs_exact_recurrence_lsb": int(np.abs(refmag - exact).max()),
            "messages_reference": len(rd), "messages_device": len(gd), "only_in_reference": len(rd.keys() - gd.keys()),
            "only_on_device": len(gd.keys() - rd.keys()), "common_with_other_fields_differing": int(fields_differ),
            "max_relative_signal_level_difference": float(rel)}


if __name__ == "__main__":
    out = {name: measure(name) for name, (sc, kw) in SCENARIOS.items() if kw.get("dc_filter")}
    print(json.dumps(out, indent=1))
