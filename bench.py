#!/usr/bin/env python3
"""bench.py -- Msamples/s of demod+CRC (UC8 @ 2.4 MS/s) on N MI355X GPUs.

One "step" = one pass of the whole hot path (IQ -> magnitude -> preamble gate -> 5-phase slicing
-> CRC/syndrome repair -> filter-superset prune -> device replay -> message list in host memory) over one
1-Gsample synthetic UC8 recording that is already resident in HBM (BASELINE.json configs[1]).
With N > 1 every rank owns one independent recording/channel of the same size (weak scaling,
SURVEY.md section 8e): the data path has no collective; RCCL is used for the barrier, the
max-over-ranks time and the message-count gather only.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- the dominant kernel (k_scan) against the HBM roofline, timed with HIP events on
                  the stream it is launched on (inside the library, rsb_get_timing);
  cpu_baseline -- the unmodified reference (oracle/_ref/ref_harness, kind "reference") or, where
                  that binary is absent, our C restatement (kind "port"), single thread, on a
                  bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_segment(seg_samples: int, seed: int, fmt: str = "UC8", dense: bool = False):
    from readsb_amd.synth import Scenario, generate

    # mixed SNR (amplitudes 0.05..0.8 FS over sigma = 2 LSB noise), DF11/DF17-dominated traffic with
    # every 7th frame carrying a 1-bit error, a few Address/Parity DFs and near-collisions
    if not dense:
        return generate(Scenario(seg_samples, seed=seed, fmt=fmt))
    # BASELINE configs[4]: dense overlapping traffic at low SNR with 1- and 2-bit errors and Mode A/C replies
    return generate(Scenario(seg_samples, seed=seed, fmt=fmt, frames_per_sec=4000, err1_every=7, err2_every=10, modeac_per_sec=300,
                             amp_lo=0.04, amp_hi=0.54))


def cpu_baseline(seg: np.ndarray, budget_s: float = 12.0, fmt: str = "UC8", nfix: int = 1, modeac: bool = False, dc: bool = False) -> dict:
    from oracle.oracle import Oracle, RefHarness

    n = seg.shape[0]
    what = f"convert ({fmt}{', --dcfilter' if dc else ''}) + demodulate2400{' + demodulate2400AC' if modeac else ''}, " + \
        {0: "--no-fix", 1: "--fix", 2: "--aggressive"}[nfix]
    if RefHarness.available():
        with tempfile.NamedTemporaryFile(suffix=".iq", delete=False) as f:
            seg.tofile(f)
            path = f.name
        try:
            probe = RefHarness.time(path, fmt=fmt, nfix=nfix, modeac=modeac, dc_filter=dc, reps=1)
            reps = max(1, int(budget_s / max(probe["seconds"], 1e-3)))
            r = RefHarness.time(path, fmt=fmt, nfix=nfix, modeac=modeac, dc_filter=dc, reps=reps)
        finally:
            os.unlink(path)
        return {"value": r["samples"] / r["seconds"] / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "reference",
                "sample": f"{n}-sample segment of the same recording x {reps} passes ({what})"}
    o = Oracle(fmt=fmt, nfix=nfix, modeac=modeac, dc_filter=dc)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < budget_s:
        o.run_stream(seg)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": n * reps / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{n}-sample segment of the same recording x {reps} passes (oracle/modes_oracle.c)"}


def cpu_baseline_sharded(seg: np.ndarray, single: dict, budget_s: float = 6.0, fmt: str = "UC8", nfix: int = 1, modeac: bool = False,
                         dc: bool = False):
    """The reference's path sharded the way the GPUs are: one independent channel per core, every core running the
    unmodified reference on its own copy of the segment at the same time.  value = all cores' samples / wall time."""
    import subprocess

    from oracle.oracle import REF_BIN, RefHarness

    if not RefHarness.available():
        return None
    cores = max(1, min(16, len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16"))))  # the box's CPU share
    n = seg.shape[0]
    reps = max(1, int(budget_s * single["value"] * 1e6 / n))
    with tempfile.NamedTemporaryFile(suffix=".iq", delete=False) as f:
        seg.tofile(f)
        path = f.name
    try:
        cmd = [REF_BIN, "time", fmt, str(int(nfix)), str(int(modeac)), str(int(dc)), path, str(reps)]
        t0 = time.perf_counter()
        procs = [subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(cores)]
        ok = all(p.wait() == 0 for p in procs)
        dt = time.perf_counter() - t0
    finally:
        os.unlink(path)
    if not ok:
        return None
    return {"value": n * reps * cores / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "reference",
            "sample": f"{cores} processes at once (one channel per core), each {reps} passes over the {n}-sample segment; wall time incl. process start"}


def kernel_src_sha() -> str:
    """identifies the kernel build a committed counter file belongs to (the GPU box has no .git)"""
    import hashlib

    h = hashlib.sha256()
    for f in ("kernels.hip", "kernels.hpp"):
        with open(os.path.join(ROOT, "readsb_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: N fresh child processes, one rank per GPU, rendezvous on 127.0.0.1.
    The parent never initialises the GPU (torch.cuda.device_count() does not) and only relays rank 0's JSON line."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} HIP device(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def shard_recording_bench(args, rank: int, world: int, local_rank: int, dev) -> int:
    """--shard-recording: one recording decoded by all ranks together (BASELINE configs[3], "one 64 Gsample recording halo-sharded")"""
    import torch

    import readsb_amd
    from readsb_amd import dist as rd

    if args.format != "UC8" or args.dcfilter or args.modeac:
        raise SystemExit("--shard-recording: UC8 without --dcfilter / --modeac (the filter coupling is what this mode is about)")
    S = args.shard_recording
    if S % 131072 or S <= 0:
        raise SystemExit("--shard-recording: a multiple of 131072 samples")
    seg_n = min(args.segment, args.samples)
    reps = args.samples // seg_n
    n = reps * seg_n
    seg = make_segment(seg_n, seed=rd.channel_seed(0), fmt="UC8", dense=args.dense)  # every rank holds the SAME recording
    iq = torch.from_numpy(seg).to(dev).repeat(reps, 1).contiguous()
    torch.cuda.synchronize()
    base = iq.data_ptr()

    def carry(start):  # the 326 samples before a cut, from the host's copy of the tiled segment
        if start == 0:
            return None
        idx = np.arange(start - 326, start) % seg_n
        return np.ascontiguousarray(seg[idx])

    K = max(1, args.shard_contexts) if world == 1 else 1
    chans = [readsb_amd.Channel(fmt="UC8", fix_bits=args.fix_bits, device=local_rank, max_batch_samples=S, queue_div=16 if args.dense else 32, defer_prune=True)
             for _ in range(K)]
    cap = max(1 << 16, S // (256 if args.dense else 512))
    nseg = -(-n // S)
    per_ctx = -(-nseg // (K * world))  # segments per context and step
    # every context's records of a step in one pinned buffer: the lists a step returns are views into it, nothing is copied on the host
    outs = [torch.empty(cap * (per_ctx + 1) * readsb_amd.MSG_DTYPE.itemsize, dtype=torch.uint8, pin_memory=True).numpy().view(readsb_amd.MSG_DTYPE) for _ in range(K)]

    def step(st=None):
        for c in chans:
            c.reset()
        if world == 1:
            return rd.simulate_parallel(chans, n, S, lambda a, b: base + 2 * a, carry, cap=cap, stats=st, outs=outs, join=False)
        return [m for _, m in rd.decode_recording_parallel(chans[0], n, S, lambda a, b: base + 2 * a, carry, dev, cap=cap, stats=st, out=outs[0])]

    for _ in range(args.warmup):
        step()
    st = {}
    rd.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        parts = step(st)
    rd.barrier(dev)
    dt = rd.max_over_ranks(time.perf_counter() - t0, dev)
    msgs = np.concatenate(parts) if parts else np.zeros(0, dtype=readsb_amd.MSG_DTYPE)
    # outside the timed region: all ranks' lists merged by timestamp = the unsharded decode of the recording by one channel (rank 0)
    merged = rd.merge_channels(msgs, dev) if world > 1 else msgs
    check = "n/a"
    if rank == 0:
        one = readsb_amd.Channel(fmt="UC8", fix_bits=args.fix_bits, device=local_rank, max_batch_samples=n, queue_div=16 if args.dense else 32)
        one.submit(base, n, torch.cuda.current_stream().cuda_stream)
        want = one.collect(max(1 << 16, n // (256 if args.dense else 512)))
        check = "sharded-equals-unsharded" if want.size == merged.size and want.tobytes() == merged.tobytes() else f"SHARD-MISMATCH {want.size} != {merged.size}"
        print(json.dumps({
            "metric": "Msamples/s demod+CRC (UC8 @2.4MS/s)", "value": n * args.steps / dt / 1e6, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u16",
            "data": "synthetic",
            "config": {"workload": "configs[3], second form: ONE synthetic UC8 recording cut at multiples of 131072 samples, segments round-robin over the ranks "
                                   "with 326-sample halos, ICAO filter coupling as a fixed point over the shards",
                       "recording_samples": n, "segment_samples": S, "segments": -(-n // S), "ranks": world, "contexts_on_one_gpu": K, "fix_bits": args.fix_bits,
                       "messages_per_step": int(merged.size), "rounds_per_step": st.get("rounds"), "check_passes_last_step": st.get("passes"),
                       "shards_decoded_again_last_step_rank0": st.get("redone"), "assumptions_checked_against_the_kept_items_rank0": st.get("checked"), "shard_check": check}}))
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=1 << 30, help="samples per GPU per step")
    ap.add_argument("--segment", type=int, default=1 << 23, help="synthetic segment that is tiled to --samples")
    ap.add_argument("--batches", type=int, default=1, help="submit the recording as this many batches")
    ap.add_argument("--in-flight", type=int, default=3, help="batches submitted ahead of the one being collected (the library holds up to 4)")
    ap.add_argument("--steps-in-flight", type=int, default=2, choices=(1, 2),
                    help="2: step k + 1 (its own receiver channel) is submitted before step k is collected, so the replay / copy tail of one "
                         "recording runs beside the scan of the next; 1: strictly one step at a time")
    ap.add_argument("--fence", type=int, default=0, choices=(0, 1),
                    help="with two steps in flight: queue the next step's scan behind this step's chain on the device (rsb_collect_fence); "
                         "default off: measured 2.34 against 1.97 ms per step without it (profiles/r3_step_modes.txt)")
    ap.add_argument("--scan-workgroups", type=int, default=-1, help="rsb_config.scan_workgroups: CUs the persistent scan kernel takes; -1 = all with one batch, "
                    "224 with several (one CU per shader engine stays free for the replay kernels of the batch before)")
    ap.add_argument("--output", choices=("records", "beast", "raw"), default="records",
                    help="what a step delivers to the host: rsb_message records (default, the metric's message list) or the Beast / raw "
                         "byte stream encoded on the device (rsb_collect_wire)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # the other BASELINE.json configurations (the default line stays configs[1]); parity for each is covered in tests/
    ap.add_argument("--format", choices=("UC8", "SC16", "SC16Q11"), default="UC8", help="configs[2]: 4-byte samples (bladeRF / Pluto)")
    ap.add_argument("--chain-on-submit-stream", type=int, default=-1, choices=(-1, 0, 1),
                    help="with two steps in flight: a step's prune / replay / emit chain on the stream its scan was submitted on "
                         "(rsb_collect_prepare_on), the next step's scan behind it.  Default: on, except with --modeac (the Mode A/C merge "
                         "is enqueued at collect time, i.e. behind the next scan: 4.3 instead of 3.6 ms per step)")
    ap.add_argument("--submit-ahead", type=int, default=-1, choices=(-1, 0, 1),
                    help="two steps in flight: submit step k + 1 before waiting for step k's chain (default off: measured slower)")
    ap.add_argument("--dcfilter", action="store_true", help="configs[2]: the DC-block IIR as a chunked scan")
    ap.add_argument("--modeac", action="store_true", help="configs[4]: also demodulate2400AC")
    ap.add_argument("--fix-bits", type=int, default=1, choices=(0, 1, 2), help="0 --no-fix, 1 --fix (default), 2 --aggressive (configs[4])")
    ap.add_argument("--dense", action="store_true", help="configs[4] traffic: dense, low SNR, 1- and 2-bit errors, Mode A/C replies")
    ap.add_argument("--channels-per-gpu", type=int, default=1,
                    help="configs[3]: split each GPU's --samples over this many independent receiver channels (own filter, own stream position), "
                         "all submitted before the first is collected")
    ap.add_argument("--shard-recording", type=int, default=0, metavar="SEGMENT_SAMPLES",
                    help="configs[3], second form: ONE recording of --samples (the same on every rank) cut into segments of this many samples "
                         "(a multiple of 131072), segment k decoded by rank k mod N with its 326-sample halo; the ICAO filter coupling is "
                         "resolved as a fixed point over the shards (readsb_amd/dist.py: decode_recording_parallel).  Strong scaling: the "
                         "line's value is the recording's samples over the job's time")
    ap.add_argument("--shard-contexts", type=int, default=1, help="with --shard-recording on ONE GPU: this many contexts take the ranks' roles in turn")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)  # no launcher around us: start one rank per GPU ourselves (nothing has touched the GPU yet)

    import torch
    import torch.distributed as dist

    import readsb_amd
    from readsb_amd import dist as rd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    if args.shard_recording:
        rc = shard_recording_bench(args, rank, world, local_rank, dev)
        if world > 1:
            dist.destroy_process_group()
        return rc

    n = args.samples
    seg_n = min(args.segment, n)
    reps = n // seg_n
    n = reps * seg_n
    seg = make_segment(seg_n, seed=rd.channel_seed(rank), fmt=args.format, dense=args.dense)  # every rank = its own receiver channel
    bps = 2 if args.format == "UC8" else 4                                                   # bytes per sample
    seg_dev = torch.from_numpy(seg).to(dev)
    iq = seg_dev.repeat(reps, 1).contiguous()  # (n, 2) uint8 / int16 = the resident recording
    del seg_dev
    torch.cuda.synchronize()

    # channels: the GPU's samples split over C independent receiver channels (configs[3]); every channel = whole copies of the segment
    C = max(1, args.channels_per_gpu)
    if C > 1 and (reps % C or args.batches > 1 or args.output != "records"):
        raise SystemExit("--channels-per-gpu: must divide samples / segment, with --batches 1 and --output records")
    nc = n // C                                 # samples per channel
    # batches: whole numbers of 131072-sample blocks (only the last one may be ragged)
    nb = max(1, args.batches)
    bs = -(-nc // nb) // 131072 * 131072 or nc
    cuts = list(range(0, nc, bs)) + [nc]
    # two steps in flight: two receiver channels take the steps in turn (one channel, one batch per step, records out)
    # --dcfilter: without Mode A/C the kernels filter for themselves (no magnitude buffer, batches overlap as without the filter); Mode A/C
    # contexts keep the converter and its one magnitude buffer: one step at a time
    dc_fused = args.dcfilter and not args.modeac and not os.environ.get("RSB_DC_UNFUSED")
    if args.chain_on_submit_stream < 0:
        # (many channels: measured 2.88 against 2.55 ms -- the 32 record copies then start beside a running scan instead of on an idle chip)
        args.chain_on_submit_stream = 0 if args.modeac or args.channels_per_gpu > 1 else 1
    if args.submit_ahead < 0:
        args.submit_ahead = 0  # (measured: the next step's scan takes every CU and starves this step's chain -- 4.7 instead of 4.1 ms with --dcfilter)
    overlap = args.steps_in_flight == 2 and C == 1 and nb == 1 and args.output == "records" and (not args.dcfilter or dc_fused)
    scan_wgs = args.scan_workgroups if args.scan_workgroups >= 0 else (0 if nb == 1 else 224)
    chans = [readsb_amd.Channel(fmt=args.format, fix_bits=args.fix_bits, mode_ac=args.modeac, dc_filter=args.dcfilter, device=local_rank,
                                max_batch_samples=max(b - a for a, b in zip(cuts, cuts[1:])), queue_div=16 if args.dense else 32,
                                scan_workgroups=scan_wgs)
             for _ in range(2 if overlap else C)]
    # ... and with several channels: two sets of C channels take the steps in turn (rsb_collect_many_begin: a step's records cross PCIe
    # while the other set's scans run)
    overlap_many = args.steps_in_flight == 2 and C > 1 and not args.dcfilter
    chans2 = [readsb_amd.Channel(fmt=args.format, fix_bits=args.fix_bits, mode_ac=args.modeac, dc_filter=args.dcfilter, device=local_rank,
                                 max_batch_samples=max(b - a for a, b in zip(cuts, cuts[1:])), queue_div=16 if args.dense else 32,
                                 scan_workgroups=scan_wgs) for _ in range(C if overlap_many else 0)]
    ch = chans[0]
    stream = torch.cuda.current_stream().cuda_stream
    if os.environ.get("RSB_BENCH_OWN_STREAM"):  # (measurement: submits on a stream of their own instead of the default stream)
        own_stream = torch.cuda.Stream()
        stream = own_stream.cuda_stream
    cap = max(1 << 16, n // (256 if args.dense else 512))
    # the caller's result buffer: pinned, so the device -> host copy of the message list runs at link speed
    outbuf = torch.empty(cap * readsb_amd.MSG_DTYPE.itemsize, dtype=torch.uint8, pin_memory=True).numpy().view(readsb_amd.MSG_DTYPE)
    wirebuf = torch.empty(cap * readsb_amd.WIRE_MAX_BYTES if args.output != "records" else 1, dtype=torch.uint8, pin_memory=True).numpy()
    keys = ("scan_ms", "prune_ms", "modeac_ms", "sort_ms", "device_replay_ms", "copy_ms", "replay_ms", "total_ms", "replay_iterations", "replay_on_device")
    base = iq.data_ptr()
    chan_msgs = [0, 0]  # [start, end) of channel 0's records in outbuf after a step

    def step(tm=None):
        """One pass over the recording: batch i + 1 is submitted before batch i is collected; with several channels every
        channel's scan is submitted before the first channel is collected."""
        pos = 0
        acc = {k: 0.0 for k in keys}
        acc["replay_on_device"] = 1

        def collect(c):
            nonlocal pos
            if args.output == "records":
                m = c.collect(out=outbuf[pos:])
                pos += m.size
            else:
                b, _ = c.collect_wire(args.output, mlat=True, out=wirebuf[pos:])
                pos += b.size
            t = c.timing()
            for k in keys:
                if k == "replay_on_device":
                    acc[k] = min(acc[k], t[k])
                elif k == "replay_iterations":
                    acc[k] = max(acc[k], t[k])
                else:
                    acc[k] += t[k]

        if C == 1:
            ch.reset()
            nbat, nxt, done = len(cuts) - 1, 0, 0
            while done < nbat:  # up to args.in_flight batches submitted ahead of the one being collected
                while nxt < nbat and nxt - done < args.in_flight:
                    ch.submit(base + bps * cuts[nxt], cuts[nxt + 1] - cuts[nxt], stream)
                    nxt += 1
                collect(ch)
                done += 1
            chan_msgs[1] = pos
        else:
            # configs[3]: N blocks x C channels through the batched entry points (rsb_submit_many / rsb_collect_many: all scans
            # queued, all replay chains enqueued before the first wait, records leaving while the next channel is looked at)
            for c in chans:
                c.reset()
            readsb_amd.submit_many(chans, [base + bps * nc * i for i in range(C)], [nc] * C, stream)
            per = cap // C
            views = readsb_amd.collect_many(chans, [outbuf[i * per:(i + 1) * per] for i in range(C)])
            chan_msgs[1] = views[0].size
            for c in chans:
                t = c.timing()
                for k in keys:
                    acc[k] = min(acc[k], t[k]) if k == "replay_on_device" else \
                        max(acc[k], t[k]) if k in ("replay_iterations", "prune_ms", "device_replay_ms") else acc[k] + t[k]
        if tm is not None:
            for k in keys:
                tm[k].append(acc[k])
        if C > 1:
            return views  # one list per channel, each in its own part of the pinned buffer (joined after the timed region)
        return outbuf[:pos] if args.output == "records" else wirebuf[:pos]

    def barrier():
        rd.barrier(dev)

    outbuf2 = torch.empty(cap * readsb_amd.MSG_DTYPE.itemsize if overlap or overlap_many else 1, dtype=torch.uint8, pin_memory=True).numpy().view(
        readsb_amd.MSG_DTYPE if overlap or overlap_many else np.uint8)

    def steps_overlapped_many(count, tm=None):
        """count steps of C channels each, every step all C recordings from cold filters, on the set of channels whose turn it is.  A
        step's record copies (rsb_collect_many_begin .. rsb_collect_wait) run beside the next step's scans; everything else as in
        step().  Returns the last step's lists (complete: the last waits are inside)."""
        sets, bufs, per = (chans, chans2), (outbuf, outbuf2), cap // C
        ptrs, lens = [base + bps * nc * i for i in range(C)], [nc] * C
        for c in sets[0]:
            c.reset()
        readsb_amd.submit_many(sets[0], ptrs, lens, stream)
        views = None
        for k in range(count):
            mine, other = sets[k % 2], sets[(k + 1) % 2]
            if k > 0:
                for c in other:
                    c.collect_wait()               # step k - 1's records are in host memory
            if args.chain_on_submit_stream:
                # the shared chain of this step on the submit stream, the next step's scans behind it, then the host's part of the collect
                readsb_amd.collect_many_prepare(mine, stream)
                if k + 1 < count:
                    for c in other:
                        c.reset()
                    readsb_amd.submit_many(other, ptrs, lens, stream)
                views = readsb_amd.collect_many(mine, [bufs[k % 2][i * per:(i + 1) * per] for i in range(C)], wait=False)
            else:
                if k + 1 < count:
                    for c in other:
                        c.reset()
                views = readsb_amd.collect_many(mine, [bufs[k % 2][i * per:(i + 1) * per] for i in range(C)], wait=False)
                if k + 1 < count:
                    readsb_amd.submit_many(other, ptrs, lens, stream)
            if tm is not None:
                acc = {key: 0.0 for key in keys}
                acc["replay_on_device"] = 1
                for c in mine:
                    t = c.timing()
                    for key in keys:
                        # (prune / replay: the one chain all channels share reports the same times on every channel)
                        acc[key] = min(acc[key], t[key]) if key == "replay_on_device" else \
                            max(acc[key], t[key]) if key in ("replay_iterations", "prune_ms", "device_replay_ms") else acc[key] + t[key]
                for key in keys:
                    tm[key].append(acc[key])
        for c in sets[(count - 1) % 2]:
            c.collect_wait()
        chan_msgs[1] = views[0].size
        return views

    def steps_overlapped(count, tm=None):
        """count steps, every one the whole recording from a cold filter on the receiver channel whose turn it is (two channels, two
        pinned result buffers).  A step's records cross PCIe (rsb_collect_begin .. rsb_collect_wait) while the next step's scan
        runs; everything else of a step -- scan, prune, replay, emit -- runs one after the other as before.  Returns the last
        step's records (complete: the last wait is inside)."""
        bufs = (outbuf, outbuf2)
        out = None
        chans[0].reset()
        chans[0].submit(base, nc, stream)
        trace = os.environ.get("RSB_BENCH_TRACE")  # (measurement: where the host spends a step, averaged, to stderr)
        spent, mark = {}, time.perf_counter()

        def lap(what):
            nonlocal mark
            if trace:
                now = time.perf_counter()
                spent[what] = spent.get(what, 0.0) + now - mark
                mark = now

        for k in range(count):
            c, o = chans[k % 2], chans[(k + 1) % 2]
            lap("loop")
            if k > 0:
                o.collect_wait()               # step k - 1's records are in host memory (their copy ran beside step k's scan)
            lap("collect_wait")
            if args.chain_on_submit_stream:
                # step k's chain goes onto the submit stream, right behind its scan, and step k + 1's scan behind the chain: scan, chain,
                # scan, chain in one queue, no hop to another stream and no host turn-around between them
                c.collect_prepare(on_stream=stream)
                lap("collect_prepare")
                if k + 1 < count:
                    o.reset()
                    o.submit(base, nc, stream)
                lap("reset+submit")
                out = c.collect_begin(bufs[k % 2])
                lap("collect_begin")
            elif args.fence:
                # step k's chain goes onto its stream now, and step k + 1's scan is queued behind it ON THE DEVICE: the GPU goes
                # from emit k to scan k + 1 without waiting for the host to notice that step k is done
                c.collect_prepare()
                lap("collect_prepare")
                if k + 1 < count:
                    o.reset()
                    c.collect_fence(stream)
                    o.submit(base, nc, stream)
                lap("reset+fence+submit")
                out = c.collect_begin(bufs[k % 2])
                lap("collect_begin")
            else:
                if k + 1 < count:
                    o.reset()                      # the other channel's next recording starts cold; done while step k's scan runs
                lap("reset")
                if args.submit_ahead and k + 1 < count:
                    o.submit(base, nc, stream)     # step k + 1 queued behind step k's scan: its first kernels run beside step k's chain
                out = c.collect_begin(bufs[k % 2])  # waits for step k's scan, prune, replay and emit; its copy is under way
                lap("collect_begin")
                if not args.submit_ahead and k + 1 < count:
                    o.submit(base, nc, stream)
                lap("submit")
            if tm is not None:
                t = c.timing()
                for key in keys:
                    tm[key].append(t[key])
        chans[(count - 1) % 2].collect_wait()
        if trace and tm is not None:
            print("host ms per step: " + ", ".join(f"{k} {v / count * 1e3:.3f}" for k, v in spent.items()), file=sys.stderr)
        chan_msgs[1] = out.size
        return out

    for _ in range(args.warmup):
        msgs = step()
        if overlap:
            msgs = steps_overlapped(2)
        if overlap_many:
            msgs = steps_overlapped_many(2)
    tm = {k: [] for k in keys}
    scan_ms = tm["scan_ms"]
    barrier()
    t0 = time.perf_counter()
    if overlap:
        msgs = steps_overlapped(args.steps, tm)
    elif overlap_many:
        msgs = steps_overlapped_many(args.steps, tm)
    else:
        for _ in range(args.steps):
            msgs = step(tm)
    barrier()
    dt = time.perf_counter() - t0
    if isinstance(msgs, list):
        msgs = np.concatenate(msgs)
    serial_ms = None
    if overlap or overlap_many:  # for the record, outside the timed region: the same steps one at a time (submit -> collect latency of one recording)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        serial_ms = (time.perf_counter() - t1) / 2 * 1e3
    dt = rd.max_over_ranks(dt, dev)                      # the slowest rank's time
    wire_check = None
    if args.output != "records":
        # full-size check of the device encoder: the same recording once more as records, encoded by the host encoder
        wire_bytes = msgs.copy()
        out_kind, args.output = args.output, "records"
        msgs = step()
        args.output = out_kind
        wire_check = "ok" if np.array_equal(readsb_amd.wire_encode(msgs, out_kind, mlat=True), wire_bytes) else "WIRE-MISMATCH"
    nmsgs = sum(rd.gather_counts(int(msgs.size), dev))   # the result-count gather: the only data the ranks exchange

    # Full-size check, outside the timed region.  The recording is `reps` copies of one segment whose length is a multiple of
    # 131072, so (a) copy 0 (cold filter, zero overlap) must be what the ORACLE decodes from the segment followed by the
    # start of its next copy, record for record, and (b) from the second copy on (filter warm, real overlap) EVERY copy must
    # decode to the same list as copy 1, shifted by seg_n * 5 ticks (the last copy loses the frames in its final 326 samples).
    parity, oracle_check = "n/a", "n/a"
    if args.dcfilter and nc // seg_n >= 2 and seg_n % 131072 == 0:
        # The DC block: the reference's float recurrence cannot be followed bit for bit by any parallel evaluation (tests/test_gpu_parity.py:
        # test_dcfilter_stream has the tolerance oracle in full).  Here, copy 0 against the oracle's list: the messages in both lists
        # must agree in every field (signal level to 0.5 %), and at most 0.5 % of the messages may be in one list only (frames at the
        # slicer's decision margin; 1 of 377 observed on UC8, none on SC16).
        from oracle.oracle import Oracle, REC_DTYPE

        period = seg_n * 5
        m0 = msgs[:chan_msgs[1]]
        m0 = m0[m0["ts"] < period].view(REC_DTYPE)
        want, _ = Oracle(fmt=args.format, nfix=args.fix_bits, modeac=args.modeac, dc_filter=True).run_stream(np.concatenate([seg, seg[:131072]]))
        want = want[want["ts"] < period]
        key = lambda r: (int(r["ts"]), bytes(r["msg"]), int(r["msgbits"]))
        wd, gd = {key(r): r for r in want}, {key(r): r for r in m0}
        common = wd.keys() & gd.keys()
        one_sided = len(wd.keys() ^ gd.keys())
        fields_ok = all(wd[k][f] == gd[k][f] for k in common for f in ("msgtype", "crc", "correctedbits", "addr", "score", "source", "iid")) and \
            all(abs(wd[k]["signal"] - gd[k]["signal"]) <= 0.005 * abs(wd[k]["signal"]) + 1e-12 for k in common)
        ok = fields_ok and len(common) > 0 and one_sided <= max(2, len(common) // 200)
        oracle_check = f"copy0-within-dc-tolerance ({len(common)} common, {one_sided} one-sided)" if ok else f"ORACLE-MISMATCH ({len(common)} common, {one_sided} one-sided, fields {fields_ok})"
    if nc // seg_n >= 4 and seg_n % 131072 == 0 and not args.dcfilter:  # (the DC filter's state keeps settling: its copies are not identical)
        from oracle.oracle import Oracle, REC_DTYPE, first_mismatch

        period = seg_n * 5
        all_msgs, msgs = msgs, msgs[:chan_msgs[1]]  # channel 0 (the whole list with one channel)
        reps = nc // seg_n
        k = msgs["ts"] // period
        want, _ = Oracle(fmt=args.format, nfix=args.fix_bits, modeac=args.modeac).run_stream(np.concatenate([seg, seg[:131072]]))
        bad = first_mismatch(want[want["ts"] < period], msgs[k == 0].view(REC_DTYPE))
        oracle_check = "copy0-equals-oracle" if bad is None else f"ORACLE-MISMATCH {bad}"
        a = msgs[k == 1]
        fields = ("msg", "msgbits", "score", "correctedbits", "crc", "addr", "signal")
        same = a.size > 0
        for j in range(2, reps - 1):
            b = msgs[k == j]
            same = same and a.size == b.size and np.array_equal(a["ts"] - period, b["ts"] - period * j) \
                and all(np.array_equal(a[f], b[f]) for f in fields)
        parity = f"copies-1..{reps - 2}-identical" if same else "PERIODICITY-MISMATCH"
        msgs = all_msgs
    checks_ok = sum(rd.gather_counts(int(not ("MISMATCH" in parity or "MISMATCH" in oracle_check)), dev))

    if rank == 0:
        if args.format == "UC8" and args.fix_bits == 1 and not (args.modeac or args.dcfilter or args.dense):
            workload = "configs[1]: 1 Gsample synthetic UC8 recording, mixed SNR, demodulate2400 + 1-bit CRC correction"
        elif args.modeac or args.dense or args.fix_bits == 2:
            workload = f"configs[4]: {args.format}, {'dense low-SNR traffic' if args.dense else 'mixed SNR'}, fix_bits {args.fix_bits}" + \
                (", demodulate2400AC" if args.modeac else "")
        else:
            workload = f"configs[2]: {args.format} ({bps}-byte samples)" + (" with the DC-block IIR as a chunked scan" if args.dcfilter else ", no DC filter")
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")  # written by tools/hbm_traffic.py from rocprofv3 --pmc passes
        if os.path.exists(tfile):
            with open(tfile) as f:
                t = json.load(f)
            # ... of this very build: the file names the kernel source it was measured on
            # ... and on this workload: one launch over the whole recording, configs[1]
            same_workload = C == 1 and len(cuts) == 2 and args.format == "UC8" and args.fix_bits == 1 and not (args.dcfilter or args.modeac or args.dense)
            if t.get("samples") == n and same_workload and t.get("kernel_src_sha256") == kernel_src_sha():
                traffic = t.get("bytes_per_launch")
        scan_s = float(np.mean(scan_ms)) / 1e3
        scan_bps = 2 if args.dcfilter and not dc_fused else bps  # --dcfilter with the converter: the scan kernel reads its 16-bit magnitudes
        achieved = float(scan_bps) * n / scan_s / 1e9
        st = ch.stats()
        out = {
            "metric": "Msamples/s demod+CRC (UC8 @2.4MS/s)",
            "value": n * world * args.steps / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {"workload": workload,
                       "samples_per_gpu": n, "segment_samples": seg_n, "format": args.format, "fix_bits": args.fix_bits,
                       "mode_ac": bool(args.modeac), "dc_filter": bool(args.dcfilter),
                       "batches_per_step": len(cuts) - 1, "batches_in_flight": min(args.in_flight, len(cuts) - 1), "scan_workgroups": scan_wgs or "all",
                       "steps_in_flight": 2 if overlap or overlap_many else 1, **({"chain_on_submit_stream": bool(args.chain_on_submit_stream)} if overlap or overlap_many else {}),
                       **({"ms_per_step_one_at_a_time": serial_ms} if serial_ms is not None else {}),
                       "channels": world * C, "channels_per_gpu": C, "messages_per_step": nmsgs, "survivors_per_step": st["survivors"],
                       "full_size_check": parity, "oracle_check": oracle_check, "ranks_with_checks_ok": checks_ok, "output": args.output,
                       **({"wire_bytes_per_step": int(wire_bytes.size), "wire_check": wire_check} if wire_check else {})},
            "roofline": {"bound": "hbm", "kernel": f"k_scan<{'Mag16' if args.dcfilter and not dc_fused else args.format}{', DC' if dc_fused else ''}>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": scan_bps * n // (C * (len(cuts) - 1)), "avg_launch_ms": scan_s * 1e3 / (C * (len(cuts) - 1)),
                         "note": "per k_scan launch (one batch); the launches of a step together read the whole recording once"},
            "stage_ms": {"scan": float(np.mean(tm["scan_ms"])), "prune": float(np.mean(tm["prune_ms"])), "modeac": float(np.mean(tm["modeac_ms"])),
                         "sort": float(np.mean(tm["sort_ms"])),
                         "device_replay": float(np.mean(tm["device_replay_ms"])), "copy_out": float(np.mean(tm["copy_ms"])),
                         "host": float(np.mean(tm["replay_ms"])), "all_kernels": float(np.mean(tm["total_ms"])),
                         "replay_iterations": float(np.mean(tm["replay_iterations"])), "replay_on_device": int(min(tm["replay_on_device"]))},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seg, fmt=args.format, nfix=args.fix_bits, modeac=args.modeac, dc=args.dcfilter)
            sharded = cpu_baseline_sharded(seg, out["cpu_baseline"], fmt=args.format, nfix=args.fix_bits, modeac=args.modeac, dc=args.dcfilter)
            if sharded:
                out["cpu_baseline_sharded"] = sharded  # north_star: "single-threaded and per-core-sharded reference CPU path ... core count stated"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
