"""One recording decoded by two processes (ranks) through readsb_amd/dist.py: decode_recording_sharded -- the real HIP channels
on the GPU, torch.distributed for the adder all-gather and the filter hand-over.  The one-GPU box has a single device, which
RCCL cannot share between two ranks, so the transport here is gloo (CPU tensors); on an N-GPU node the same code runs over
"nccl" with one device per rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

N_SAMPLES, SEG_BLOCKS, SEED = 9 * 131072 + 4321, 2, 41


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, outdir, parallel=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import readsb_amd
    from readsb_amd import dist as rd
    from readsb_amd.synth import Scenario, generate

    iq = generate(Scenario(N_SAMPLES, seed=SEED, frames_per_sec=3000))  # every rank reads "the same file"
    dev_iq = torch.from_numpy(iq).cuda()
    ch = readsb_amd.Channel(fmt="UC8", fix_bits=1, max_batch_samples=SEG_BLOCKS * 131072, defer_prune=True)
    decode = rd.decode_recording_parallel if parallel else rd.decode_recording_sharded
    mine = decode(ch, N_SAMPLES, SEG_BLOCKS * 131072, get_segment=lambda a, b: dev_iq.data_ptr() + 2 * a,
                  get_carry=lambda a: iq[a - 326:a] if a else None, device=torch.device("cpu"))
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), k=np.array([k for k, _ in mine]), sizes=np.array([m.size for _, m in mine]),
             msgs=np.concatenate([m for _, m in mine]).view(np.uint8))
    ch.close()
    rd.barrier(torch.device("cpu"))
    dist.destroy_process_group()


@pytest.mark.parametrize("parallel", [False, True], ids=["filter-ring", "fixed-point"])
def test_two_ranks_one_recording(tmp_path, built, parallel):
    import readsb_amd
    from oracle.oracle import Oracle, REC_DTYPE, first_mismatch
    from readsb_amd.synth import Scenario, generate

    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path), parallel), nprocs=world, join=True)
    parts = {}
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        msgs, pos = z["msgs"].view(readsb_amd.MSG_DTYPE), 0
        for k, size in zip(z["k"], z["sizes"]):
            parts[int(k)] = msgs[pos:pos + size]
            pos += size
    got = np.concatenate([parts[k] for k in sorted(parts)])
    want, _ = Oracle(fmt="UC8", nfix=1).run_stream(generate(Scenario(N_SAMPLES, seed=SEED, frames_per_sec=3000)))
    assert sorted(parts) == list(range(5))
    assert first_mismatch(want, got.view(REC_DTYPE)) is None
