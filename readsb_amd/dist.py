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
