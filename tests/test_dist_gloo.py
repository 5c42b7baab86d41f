"""The N > 1 path on CPU: two processes, gloo, one channel per rank (weak scaling, no data-path collective).
The channel work itself is stood in for by the oracle here (no GPU); what is under test is the rank -> channel
mapping and the count / time / message gathers that bench.py and a multi-channel consumer use."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import readsb_amd
    from oracle.oracle import Oracle
    from readsb_amd import dist as rd
    from readsb_amd.synth import Scenario, generate

    dev = torch.device("cpu")
    assert rd.world() == (rank, world)
    iq = generate(Scenario(300_000 + 50_000 * rank, seed=rd.channel_seed(rank)))  # ragged: channels differ in length
    recs, _ = Oracle(fmt="UC8", nfix=1).run_stream(iq)
    msgs = recs.view(readsb_amd.MSG_DTYPE)
    rd.barrier(dev)
    counts = rd.gather_counts(msgs.size, dev)
    slowest = rd.max_over_ranks(1.0 + rank, dev)
    everyone = rd.gather_messages(msgs, dev)
    empty = rd.gather_messages(msgs[:0] if rank == 0 else msgs[:3], dev)  # a rank with nothing to report
    merged = rd.merge_channels(msgs, dev)  # the host merge by timestamp (rsb_merge_by_timestamp)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), counts=np.array(counts), slowest=slowest, mine=msgs.view(np.uint8), merged=merged.view(np.uint8),
             **{f"ch{i}": m.view(np.uint8) for i, m in enumerate(everyone)}, empty_sizes=np.array([m.size for m in empty]))
    rd.barrier(dev)
    dist.destroy_process_group()


def test_two_ranks_two_channels(tmp_path, built):
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    sizes = [z[r]["mine"].size // 56 for r in range(world)]
    assert sizes[0] > 0 and sizes[1] > 0 and sizes[0] != sizes[1]
    for r in range(world):
        assert list(z[r]["counts"]) == sizes                       # the count gather, identical on every rank
        assert float(z[r]["slowest"]) == float(world)              # max over ranks of (1 + rank)
        for ch in range(world):                                    # every rank holds every channel's list, bit for bit
            assert np.array_equal(z[r][f"ch{ch}"], z[ch]["mine"])
        assert list(z[r]["empty_sizes"]) == [0, 3]
    assert not np.array_equal(z[0]["mine"][:56], z[1]["mine"][:56])  # channels are different recordings
    # the merged list: every channel's messages, ascending timestamps, ties in channel order -- the same on both ranks
    import readsb_amd

    both = np.concatenate([z[ch]["mine"].view(readsb_amd.MSG_DTYPE) for ch in range(world)])
    want = both[np.argsort(both["ts"], kind="stable")]
    for r in range(world):
        got = z[r]["merged"].view(readsb_amd.MSG_DTYPE)
        assert got.size == sum(sizes) and np.all(np.diff(got["ts"].astype(np.int64)) >= 0)
        assert got.tobytes() == want.tobytes()


class _FakeChannel:
    """Stands in for readsb_amd.Channel (no GPU here): one 'message' per segment that records what the filter and the prune
    superset held when the segment was replayed; every segment teaches the filter one address."""

    def __init__(self):
        self.filter, self.superset, self.pending, self.start = set(), set(), None, None

    def seek(self, start, carry):
        assert (carry is None) == (start == 0)
        assert carry is None or int(carry[0]) == start % 251  # the carry of THIS position
        self.start = start

    def submit(self, ptr, n):
        assert ptr == 7_000_000 + self.start
        self.pending = (self.start, n)

    def export_adders(self):
        return np.array([1000 + self.start // 131072], dtype=np.uint32)

    def superset_add(self, addrs):
        self.superset |= set(int(a) for a in addrs)

    def filter_add(self, a):
        self.filter.add(int(a))

    def filter_add_many(self, addrs):
        for a in addrs:
            self.filter.add(int(a))

    def collect(self, cap):
        import readsb_amd

        start, n = self.pending
        rec = np.zeros(1, dtype=readsb_amd.MSG_DTYPE)
        rec["ts"], rec["addr"], rec["crc"], rec["score"] = start, len(self.filter), len(self.superset), n
        self.filter.add(1000 + start // 131072)
        return rec

    def filter_export(self):
        return np.array(sorted(self.filter), dtype=np.uint32)

    def reset(self):
        self.filter, self.superset, self.pending = set(), set(), None
        self.resets = getattr(self, "resets", 0) + 1


class _BoastingChannel(_FakeChannel):
    """its scans announce an address (5000 + segment) that the replay then never adds: every later shard of the round assumed it"""

    def export_adders(self):
        k = self.start // 131072
        self.announced = getattr(self, "announced", set()) | {1000 + k, 5000 + k}
        return np.array(sorted(self.announced), dtype=np.uint32)


def _shard_rank_main(rank, world, port, outdir, nseg, total):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from readsb_amd import dist as rd

    mine = rd.decode_recording_sharded(_FakeChannel(), total, 131072, get_segment=lambda a, b: 7_000_000 + a,
                                       get_carry=lambda a: np.full(326, a % 251, dtype=np.uint8) if a else None,
                                       device=torch.device("cpu"))
    np.savez(os.path.join(outdir, f"shard{rank}.npz"), k=np.array([k for k, _ in mine]),
             msgs=np.concatenate([m for _, m in mine]).view(np.uint8) if mine else np.zeros(0, np.uint8))
    rd.barrier(torch.device("cpu"))
    dist.destroy_process_group()


def _parallel_rank_main(rank, world, port, outdir, nseg, total, boasting, optimistic):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from readsb_amd import dist as rd

    ch, st = (_BoastingChannel() if boasting else _FakeChannel()), {}
    mine = rd.decode_recording_parallel(ch, total, 131072, get_segment=lambda a, b: 7_000_000 + a,
                                        get_carry=lambda a: np.full(326, a % 251, dtype=np.uint8) if a else None,
                                        device=torch.device("cpu"), optimistic=optimistic, stats=st)
    np.savez(os.path.join(outdir, f"shard{rank}.npz"), k=np.array([k for k, _ in mine]), redone=st["redone"], passes=st["passes"], rounds=st["rounds"],
             msgs=np.concatenate([m for _, m in mine]).view(np.uint8) if mine else np.zeros(0, np.uint8))
    rd.barrier(torch.device("cpu"))
    dist.destroy_process_group()


@pytest.mark.parametrize("boasting,optimistic", [(False, True), (True, True), (False, False)])
def test_one_recording_parallel_fixed_point(tmp_path, built, boasting, optimistic):
    """decode_recording_parallel over gloo: every rank replays its shard at once under an assumed starting filter, the ranks compare
    what was really added and correct.  With honest adder lists the optimistic start is right at once (no shard decoded twice);
    with lists that announce addresses nobody adds, or from the pessimistic start, shards are decoded again -- and every shard
    still ends up having seen exactly the earlier shards' addresses."""
    import readsb_amd

    for world, nseg in ((2, 5), (3, 4)):
        total = (nseg - 1) * 131072 + 1000
        out = tmp_path / f"w{world}"
        out.mkdir()
        mp.spawn(_parallel_rank_main, args=(world, _free_port(), str(out), nseg, total, boasting, optimistic), nprocs=world, join=True)
        seen, redone = {}, 0
        for r in range(world):
            z = np.load(out / f"shard{r}.npz")
            msgs = z["msgs"].view(readsb_amd.MSG_DTYPE)
            assert list(z["k"]) == list(range(r, nseg, world))
            redone += int(z["redone"])
            for k, m in zip(z["k"], msgs):
                seen[int(k)] = m
            if not boasting and optimistic:
                assert int(z["passes"]) == int(z["rounds"])  # one check per round, nothing to correct
        assert sorted(seen) == list(range(nseg))
        for k in range(nseg):
            assert int(seen[k]["ts"]) == k * 131072
            assert int(seen[k]["addr"]) == k  # the filter held exactly the earlier segments' addresses when the shard was (last) replayed
        assert (redone == 0) == (not boasting and optimistic)


def test_one_recording_sharded_over_ranks(tmp_path, built):
    """the schedule of readsb_amd/dist.py: decode_recording_sharded over gloo: segments round-robin over the ranks, adder lists
    all-gathered per round, the filter handed on in stream order (ring, wrapping into the next round), a last round with an
    idle rank and a ragged last segment"""
    import readsb_amd

    for world, nseg in ((2, 5), (3, 4)):
        total = (nseg - 1) * 131072 + 1000
        out = tmp_path / f"w{world}"
        out.mkdir()
        mp.spawn(_shard_rank_main, args=(world, _free_port(), str(out), nseg, total), nprocs=world, join=True)
        seen = {}
        for r in range(world):
            z = np.load(out / f"shard{r}.npz")
            msgs = z["msgs"].view(readsb_amd.MSG_DTYPE)
            assert list(z["k"]) == list(range(r, nseg, world))      # round-robin
            for k, m in zip(z["k"], msgs):
                seen[int(k)] = m
        assert sorted(seen) == list(range(nseg))
        for k in range(nseg):
            m = seen[k]
            assert int(m["ts"]) == k * 131072
            assert int(m["addr"]) == k                               # the filter held exactly the earlier segments' addresses
            assert int(m["crc"]) == min(nseg, (k // world + 1) * world)  # the superset: every segment of the rounds so far
            assert int(m["score"]) == (1000 if k == nseg - 1 else 131072)
