"""Multi-GPU plumbing: one process per GPU, one independent receiver channel (recording) per rank.

The receive path has no cross-channel data dependency: ranks never exchange
samples or candidates.  What the job needs from torch.distributed is a barrier, the slowest rank's
time, the per-rank message counts, and -- when one consumer wants every channel's output -- a
gather of the variable-length message lists.  Works on any backend ("nccl" = RCCL on the GPUs,
"gloo" in the CPU tests); tensors live on `device`.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import MSG_DTYPE


def world() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def channel_seed(rank: int, base: int = 1000) -> int:
    """Every rank decodes its own channel: the synthetic recording's seed is a function of the rank only."""
    return base + rank


def barrier(device: torch.device | None = None) -> None:
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if world()[1] > 1:
        dist.barrier()


def max_over_ranks(value: float, device: torch.device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if world()[1] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(count: int, device: torch.device) -> list[int]:
    """Per-rank message counts, on every rank."""
    _, n = world()
    mine = torch.tensor([count], dtype=torch.int64, device=device)
    if n == 1:
        return [int(count)]
    out = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(n)]
    dist.all_gather(out, mine)
    return [int(x.item()) for x in out]


def gather_messages(msgs: np.ndarray, device: torch.device) -> list[np.ndarray]:
    """Every rank's message list (rsb_message records), in rank (= channel) order, on every rank.
    Lists differ in length: sizes first, then one all_gather of byte tensors padded to the longest."""
    msgs = np.ascontiguousarray(msgs, dtype=MSG_DTYPE)
    counts = gather_counts(int(msgs.size), device)
    _, n = world()
    if n == 1:
        return [msgs]
    longest = max(counts)
    buf = torch.zeros(max(1, longest) * MSG_DTYPE.itemsize, dtype=torch.uint8, device=device)
    if msgs.size:
        buf[:msgs.nbytes] = torch.from_numpy(msgs.view(np.uint8).reshape(-1).copy()).to(device)
    out = [torch.zeros_like(buf) for _ in range(n)]
    dist.all_gather(out, buf)
    return [o.cpu().numpy()[:c * MSG_DTYPE.itemsize].view(MSG_DTYPE).copy() for o, c in zip(out, counts)]


def merge_channels(msgs: np.ndarray, device: torch.device) -> np.ndarray:
    """Every rank's message list gathered and merged on the host in timestamp order (BASELINE north star; the merge itself is the
    library's rsb_merge_by_timestamp: stable, ties in rank order).  The same merged list on every rank."""
    from . import merge_by_timestamp

    return merge_by_timestamp(gather_messages(msgs, device))


# ---- one recording decoded in shards (include/readsb_amd.h, "one recording in shards") ----------------
#
# Segment k = samples [k * seg, (k + 1) * seg) of the recording, seg a multiple of 131072.  In round j rank r scans segment
# j * world + r.  The scans of a round run in parallel; what couples the segments is the ICAO filter:
#   * every rank's adder list is all-gathered (the path's one collective) and added to every rank's prune superset;
#   * the sequential half then runs in stream order: a rank receives the filter's contents from the rank that holds the segment
#     before its own, collects, and sends its filter on (point to point: a ring over the ranks, wrapping into the next round).

def shard_scan(ch, start: int, nsamples: int, dev_ptr: int, carry) -> np.ndarray:
    """step 1 on one rank: seek, scan, the shard's adder list"""
    ch.seek(start, carry)
    ch.submit(dev_ptr, nsamples)
    return ch.export_adders()


def shard_finish(ch, all_adders: np.ndarray, filter_in: np.ndarray | None, cap: int, forward=None):
    """steps 2 and 3 on one rank: widen the superset, take over the filter, collect -> (messages, filter to pass on).
    forward(filter): called as soon as the filter is final -- the next rank in the ring waits for nothing else -- which with a
    channel that has collect_begin / collect_wait is BEFORE this shard's records have crossed PCIe (rsb_collect_begin: the
    replay has converged and folded its new members into the filter; the record copy is what remains)."""
    ch.superset_add(all_adders)
    if filter_in is not None:
        ch.filter_add_many(np.unique(filter_in))  # one call: the scan of this shard is pending, every add would wait and copy
    if hasattr(ch, "collect_begin"):
        buf = getattr(ch, "_shard_out", None)
        if buf is None or buf.size < cap:  # pinned, or the copy would block inside collect_begin
            buf = torch.empty(cap * MSG_DTYPE.itemsize, dtype=torch.uint8, pin_memory=torch.cuda.is_available()).numpy().view(MSG_DTYPE)
            ch._shard_out = buf
        view = ch.collect_begin(buf[:cap])
        flt = ch.filter_export()
        if forward is not None:
            forward(flt)
        ch.collect_wait()
        return view.copy(), flt
    msgs = ch.collect(cap).copy()
    flt = ch.filter_export()
    if forward is not None:
        forward(flt)
    return msgs, flt


def _all_gather_u32(mine: np.ndarray, device: torch.device) -> np.ndarray:
    """variable-length all-gather of uint32 lists -> their concatenation (on every rank)"""
    _, n = world()
    if n == 1:
        return mine
    counts = gather_counts(int(mine.size), device)
    longest = max(1, max(counts))
    buf = torch.zeros(longest, dtype=torch.int64, device=device)
    if mine.size:
        buf[:mine.size] = torch.from_numpy(mine.astype(np.int64)).to(device)
    out = [torch.zeros_like(buf) for _ in range(n)]
    dist.all_gather(out, buf)
    return np.concatenate([o.cpu().numpy()[:c].astype(np.uint32) for o, c in zip(out, counts)])


def _isend_u32(values: np.ndarray, dst: int, device: torch.device) -> list:
    """Non-blocking (the receiver may first have to pass a collective the sender has already left: the hand-over from the
    last rank of a round to rank 0 of the next).  -> (request, tensor) pairs to wait on / keep alive."""
    size = torch.tensor([values.size], dtype=torch.int64, device=device)
    pending = [(dist.isend(size, dst), size)]
    if values.size:
        data = torch.from_numpy(values.astype(np.int64)).to(device)
        pending.append((dist.isend(data, dst), data))
    return pending


def _recv_u32(src: int, device: torch.device) -> np.ndarray:
    n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.recv(n, src)
    if int(n.item()) == 0:
        return np.zeros(0, dtype=np.uint32)
    buf = torch.zeros(int(n.item()), dtype=torch.int64, device=device)
    dist.recv(buf, src)
    return buf.cpu().numpy().astype(np.uint32)


def decode_recording_sharded(ch, total_samples: int, segment_samples: int, get_segment, get_carry, device: torch.device, cap: int = 1 << 20):
    """Decode ONE recording on all ranks.  ch: this rank's Channel, created with defer_prune=True (or anything with its
    seek / submit / export_adders / superset_add / filter_add / collect / filter_export methods).
    get_segment(start, stop) -> device pointer of samples [start, stop) resident on this rank's GPU;
    get_carry(start) -> host array of the 326 samples before start (None at 0).
    -> list of (segment index, messages) of this rank's segments; all ranks' lists in segment order = the reference's output."""
    from . import BLOCK

    assert segment_samples % BLOCK == 0 and segment_samples > 0
    rank, n = world()
    nseg = (total_samples + segment_samples - 1) // segment_samples
    mine, pending = [], []
    for j in range((nseg + n - 1) // n):
        k = j * n + rank
        adders = np.zeros(0, dtype=np.uint32)
        if k < nseg:
            start, stop = k * segment_samples, min(total_samples, (k + 1) * segment_samples)
            adders = shard_scan(ch, start, stop - start, get_segment(start, stop), get_carry(start))
        union = _all_gather_u32(adders, device)
        if k < nseg:
            filter_in = _recv_u32((rank - 1) % n, device) if (k >= 1 and n > 1) else None
            def forward(flt, k=k):  # the serial part of the ring ends here: the filter goes on before the records have left the GPU
                nonlocal pending
                if k + 1 < nseg and n > 1:
                    for req, _ in pending:
                        req.wait()
                    pending = _isend_u32(flt, (rank + 1) % n, device)

            msgs, _ = shard_finish(ch, union, filter_in, cap, forward)
            mine.append((k, msgs))
    for req, _ in pending:
        req.wait()
    return mine


def simulate_sharded(chs, total_samples: int, segment_samples: int, get_segment, get_carry, cap: int = 1 << 20) -> np.ndarray:
    """The same schedule with len(chs) channels in one process and no torch.distributed (tests on one GPU): 'rank' r is chs[r]."""
    n = len(chs)
    nseg = (total_samples + segment_samples - 1) // segment_samples
    out, flt = [], None
    for j in range((nseg + n - 1) // n):
        ks = [k for k in range(j * n, (j + 1) * n) if k < nseg]
        adders = []
        for k in ks:
            start, stop = k * segment_samples, min(total_samples, (k + 1) * segment_samples)
            adders.append(shard_scan(chs[k % n], start, stop - start, get_segment(start, stop), get_carry(start)))
        union = np.concatenate(adders) if adders else np.zeros(0, dtype=np.uint32)
        for k in ks:
            msgs, flt = shard_finish(chs[k % n], union, flt if k >= 1 else None, cap)
            out.append(msgs)
    return np.concatenate(out) if out else np.zeros(0, dtype=MSG_DTYPE)


# ---- one recording in shards WITHOUT the serial ring: the filter coupling as a fixed point over the shards ----------------------
#
# The only thing a shard needs from the shards before it is the SET of addresses they taught the filter (mode_s.c:717-726).  Every
# rank therefore replays its shard at once, in parallel with all others, under an assumption about that set -- the exact filter at
# the start of the round plus every address the earlier shards of the round COULD add (their scans' adder lists: a clean DF17 or
# DF11 nearly always is accepted) -- and the assumption is checked afterwards: the ranks all-gather what their shards really added,
# and a rank whose true starting filter differs from the assumed one decodes its shard again with the true one.  By induction over
# the shards (shard 0's starting filter is always exact) the loop ends after at most `world` passes with exactly the serial result;
# with the optimistic start it normally ends after the first.  Prune, replay, emit and the record copy all run in parallel; what is
# serial is two small all-gathers per round.
# A wrong assumption is the rule, not the exception, on long shards (noise alone produces a dozen clean-looking DF11 per Gsample, and
# one in six of them lies inside another message's skip window and is never looked at): so a rank first asks whether the difference
# can matter at all -- rsb_batch_references: does any kept item of its shard involve one of the addresses by which the assumed and
# the true filter differ? -- and decodes again only if one does.

def _gather_lists_u32(mine: np.ndarray, device: torch.device) -> list[np.ndarray]:
    """variable-length all-gather of uint32 lists -> one list per rank (on every rank)"""
    _, n = world()
    mine = np.ascontiguousarray(mine, dtype=np.uint32)
    if n == 1:
        return [mine]
    counts = gather_counts(int(mine.size), device)
    longest = max(1, max(counts))
    buf = torch.zeros(longest, dtype=torch.int64, device=device)
    if mine.size:
        buf[:mine.size] = torch.from_numpy(mine.astype(np.int64)).to(device)
    out = [torch.zeros_like(buf) for _ in range(n)]
    dist.all_gather(out, buf)
    return [o.cpu().numpy()[:c].astype(np.uint32) for o, c in zip(out, counts)]


class _Shard:
    """one rank's side of the fixed point (also driven directly by simulate_parallel)"""

    def __init__(self, ch, cap, out=None):
        # out (optional): a caller-owned record buffer (pinned memory) for ALL of this rank's segments of the job; the message lists
        # returned are then views into it, and nothing is copied on the host
        self.ch, self.cap, self.out, self.pos, self.last = ch, cap, out, 0, (0, 0)
        self.seen = np.zeros(0, dtype=np.uint32)  # the adder list of this rank's context so far (it only grows)
        self.redone = self.checked = 0
        self.pre = np.zeros(0, dtype=np.uint32)

    def scan(self, start, nsamples, dev_ptr, carry):
        self.args = (start, nsamples, dev_ptr, carry)
        now = shard_scan(self.ch, start, nsamples, dev_ptr, carry)
        new = np.setdiff1d(now, self.seen)
        self.seen = now
        return new

    def replay(self, assumed: np.ndarray, superset: np.ndarray):
        """collect with the filter holding `assumed` (and whatever an earlier unconfirmed assumption left in this context)
        -> (messages, addresses this shard added); self.pre = the filter's contents the replay really started from"""
        self.ch.superset_add(superset)
        self.ch.filter_add_many(np.setdiff1d(assumed, self.ch.filter_export()))
        self.pre = self.ch.filter_export()
        msgs = self._collect()
        return msgs, np.setdiff1d(self.ch.filter_export(), self.pre)

    def _collect(self, again=False):
        if self.out is None:
            return self.ch.collect(self.cap).copy()
        if again:
            self.pos = self.last[0]  # the shard decoded again: over its first list
        msgs = self.ch.collect(out=self.out[self.pos:self.pos + self.cap])
        self.last = (self.pos, msgs.size)
        self.pos += msgs.size
        return msgs

    def must_redo(self, true_in: np.ndarray) -> bool:
        """could the shard's result differ had it started from true_in instead of self.pre?"""
        diff = np.setxor1d(self.pre, true_in)
        if diff.size == 0:
            return False
        self.checked += 1
        refs = self.ch.batch_references(diff) if diff.size <= 4096 and hasattr(self.ch, "batch_references") else -1
        return refs != 0

    def redo(self, true_filter: np.ndarray, superset: np.ndarray):
        """the assumption was wrong: the shard again from a context that holds the true filter (rsb_reset forgets the optimistic one)"""
        self.redone += 1
        self.ch.reset()
        self.ch.filter_add_many(true_filter)
        self.seen = shard_scan(self.ch, *self.args)
        self.ch.superset_add(superset)
        self.pre = self.ch.filter_export()
        msgs = self._collect(again=True)
        return msgs, np.setdiff1d(self.ch.filter_export(), self.pre)


def decode_recording_parallel(ch, total_samples: int, segment_samples: int, get_segment, get_carry, device: torch.device, cap: int = 1 << 20,
                              optimistic: bool = True, stats: dict | None = None, out=None):
    """decode_recording_sharded's result without its serial ring (see above).  ch: this rank's Channel, created with defer_prune=True.
    optimistic=False starts from "the earlier shards of the round add nothing" (tests: forces the correction passes).
    -> list of (segment index, messages) of this rank's segments."""
    from . import BLOCK

    assert segment_samples % BLOCK == 0 and segment_samples > 0
    rank, n = world()
    nseg = (total_samples + segment_samples - 1) // segment_samples
    me = _Shard(ch, cap, out)
    f_in = np.zeros(0, dtype=np.uint32)  # the exact filter at the start of the round, the same on every rank
    mine, passes = [], 0
    for j in range((nseg + n - 1) // n):
        k = j * n + rank
        active = k < nseg
        new = np.zeros(0, dtype=np.uint32)
        if active:
            start, stop = k * segment_samples, min(total_samples, (k + 1) * segment_samples)
            new = me.scan(start, stop - start, get_segment(start, stop), get_carry(start))
        potential = _gather_lists_u32(new, device)
        superset = np.unique(np.concatenate(potential)) if potential else new
        assumed = np.unique(np.concatenate([f_in] + (potential[:rank] if optimistic else [])))
        msgs, added = me.replay(assumed, superset) if active else (None, np.zeros(0, dtype=np.uint32))
        while True:
            passes += 1
            adds = _gather_lists_u32(added, device)
            true_in = np.unique(np.concatenate([f_in] + adds[:rank]))
            wrong = active and me.must_redo(true_in)
            if wrong:
                msgs, added = me.redo(true_in, superset)
            if max_over_ranks(1.0 if wrong else 0.0, device) == 0.0:
                break
        f_in = np.unique(np.concatenate([f_in] + adds))
        if active:
            mine.append((k, msgs))
    if stats is not None:
        stats.update(redone=me.redone, checked=me.checked, passes=passes, rounds=(nseg + n - 1) // n)
    return mine


def simulate_parallel(chs, total_samples: int, segment_samples: int, get_segment, get_carry, cap: int = 1 << 20, optimistic: bool = True,
                      stats: dict | None = None, outs=None, join: bool = True):
    """decode_recording_parallel's schedule with len(chs) channels in one process and no torch.distributed: 'rank' r is chs[r]."""
    n = len(chs)
    nseg = (total_samples + segment_samples - 1) // segment_samples
    shards = [_Shard(ch, cap, outs[r] if outs else None) for r, ch in enumerate(chs)]
    f_in = np.zeros(0, dtype=np.uint32)
    out, passes = [], 0
    for j in range((nseg + n - 1) // n):
        ks = [k for k in range(j * n, (j + 1) * n) if k < nseg]
        potential = []
        for k in ks:
            start, stop = k * segment_samples, min(total_samples, (k + 1) * segment_samples)
            potential.append(shards[k % n].scan(start, stop - start, get_segment(start, stop), get_carry(start)))
        superset = np.unique(np.concatenate(potential))
        assumed = [np.unique(np.concatenate([f_in] + (potential[:r] if optimistic else []))) for r in range(len(ks))]
        results = [shards[r].replay(assumed[r], superset) for r in range(len(ks))]
        while True:
            passes += 1
            adds = [a for _, a in results]
            any_wrong = False
            for r in range(len(ks)):  # (every 'rank' decides from the same gathered lists, as the ranks of a job do)
                true_in = np.unique(np.concatenate([f_in] + adds[:r]))
                if shards[r].must_redo(true_in):
                    results[r] = shards[r].redo(true_in, superset)
                    any_wrong = True
            if not any_wrong:
                break
        f_in = np.unique(np.concatenate([f_in] + [a for _, a in results]))
        out.extend(m for m, _ in results)
    if stats is not None:
        stats.update(redone=sum(s.redone for s in shards), checked=sum(s.checked for s in shards), passes=passes, rounds=(nseg + n - 1) // n)
    if not join:
        return out  # the segments' lists in stream order (views into `outs` when given)
    return np.concatenate(out) if out else np.zeros(0, dtype=MSG_DTYPE)
