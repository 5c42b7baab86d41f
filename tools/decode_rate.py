#!/usr/bin/env python3
"""SURVEY.md 8 f-4, the measurement behind it: what does the per-message field decode cost on the host (the reference's own
decodeModesMessage, one core, through oracle/_ref/ref_harness timedecode) and on the device (k_decode_fields on the message list of
a benchmark step, records and structs resident in HBM)?  Writes gpurun_out/decode_rate.json (copied to profiles/).

  python3 tools/decode_rate.py [out.json]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import readsb_amd  # noqa: E402
from oracle.oracle import RefHarness  # noqa: E402
from readsb_amd.synth import Scenario, generate  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/decode_rate.json"
    dev = torch.device("cuda", 0)
    torch.cuda.init()
    seg = generate(Scenario(1 << 23, seed=1))  # the benchmark's segment
    iq = torch.from_numpy(seg).to(dev).repeat(128, 1).contiguous()
    n = iq.shape[0]
    ch = readsb_amd.Channel(max_batch_samples=n, queue_div=32)
    ch.submit(iq.data_ptr(), n, 0)
    recs = ch.collect(1 << 21).copy()
    del iq
    d_in = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev)
    d_out = torch.empty(recs.size * 416, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream()
    times = []
    for rep in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        ch._check(ch.l.rsb_decode_fields_device(ch.h, d_in.data_ptr(), recs.size, 0, 131072, d_out.data_ptr(), None, s.cuda_stream), "rsb_decode_fields_device")
        e1.record(s)
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = float(np.median(times[2:]))
    res = {"messages": int(recs.size), "by_downlink_format": {str(int(k)): int(v) for k, v in zip(*np.unique(recs["msgtype"], return_counts=True))},
           "device": {"kernel": "k_decode_fields", "ms": ms, "messages_per_s": recs.size / ms * 1e3,
                      "bytes_moved": int(recs.size) * (56 + 416), "GB_per_s": recs.size * (56 + 416) / ms / 1e6,
                      "how": "HIP events on the launch stream, median of 10 launches, records and structs in HBM"}}
    if RefHarness.available():
        # the frames as the reference's slicer had them are not needed for timing: the repaired frames decode to the same fields
        frames = recs["msg"]
        addrs = np.unique(recs["addr"][(recs["msgtype"] != 17) & (recs["msgtype"] != 18) & (recs["msgtype"] != 11)])
        t = RefHarness.timedecode(frames, addrs[:1500], reps=4)
        res["reference_host"] = {"function": "decodeModesMessage (mode_s.c:424-734, decodeCommB included), one thread", "seconds": t["seconds"], "messages": t["messages"],
                                 "accepted": t["accepted"], "messages_per_s_per_core": t["messages"] / t["seconds"]}
        res["ratio_device_to_one_core"] = res["device"]["messages_per_s"] / res["reference_host"]["messages_per_s_per_core"]
    print(json.dumps(res))
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
