"""The N > 1 path on CPU: two processes, gloo, one channel per rank (weak scaling, no data-path collective).
The channel work itself is stood in for by the oracle here (no GPU); what is under test is the rank -> channel
mapping and the count / time / message gathers that bench.py and a multi-channel consumer use."""
import os
import socket
import sys

import numpy as np
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
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), counts=np.array(counts), slowest=slowest, mine=msgs.view(np.uint8),
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
