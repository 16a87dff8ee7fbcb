"""Multi-GPU plumbing (one process per GPU, ``torch.distributed``; backend "nccl" is RCCL over xGMI).

The hot path shards at the instance level:

* **replica mode** — independent graphs / frames are dealt to ranks; there is *no* data-path
  collective, only the scalars of the timing protocol are reduced (``max_over_ranks``);
* **edge-sharded mode** — one graph, its edge list split contiguously across ranks; every rank
  assembles the partial normal equations of its edges and ``allreduce_normal_equations`` sums the
  concatenated ``[H values || b]`` array (5.4 MB of doubles for the 5000/1000 graph), after which
  every rank runs the identical solve.

The functions take / return tensors on whatever device the process group works on, so the same code
runs under ``gloo`` on CPU (tests, world_size 2) and under RCCL on MI355X.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_indices(n_items: int, rank: int, world: int) -> np.ndarray:
    """Round-robin deal of independent work items (graphs, frames, boxes): item i -> rank i mod world."""
    return np.arange(rank, n_items, world)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split (edge-sharded mode): sizes differ by at most one, earlier ranks get the extra."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(seconds: float, device=None) -> float:
    """Timing protocol of bench.py: barrier-bracketed interval, MAX over ranks."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_normal_equations(h_and_b):
    """Sum the partial ``[H values || b]`` arrays of all ranks in place (edge-sharded mode)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(h_and_b, op=dist.ReduceOp.SUM)
    return h_and_b


def init_edge_sharded(batch, device=None) -> None:
    """Edge-sharded mode on the GPUs: rank 0 creates the RCCL unique id (``sslam_comm_unique_id``), ``torch.distributed``
    carries its 128 bytes to the other ranks, and every rank joins the communicator that the library's LM loop uses for the
    all-reduce of the normal equations (``ncclAllReduce`` issued from C++ on the batch's own stream)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from ._lib import load_library
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = C.create_string_buffer(128)
    if rank == 0:
        rc = load_library().sslam_comm_unique_id(buf)
        if rc < 0:
            raise RuntimeError(load_library().sslam_last_error().decode())
    t = torch.tensor(list(buf.raw), dtype=torch.uint8, device=device)
    dist.broadcast(t, src=0)
    batch.comm_init(bytes(t.cpu().tolist()), rank, world)


def aggregate_throughput(units_this_rank: float, seconds_max: float, device=None) -> float:
    """Whole-job value = units all ranks processed / max-over-ranks time."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return units_this_rank / seconds_max
    t = torch.tensor([units_this_rank], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / seconds_max
