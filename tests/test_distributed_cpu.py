"""world_size-2 `gloo` tests of the multi-GPU plumbing (semantic_slam_amd/distributed.py): the sharding
plan, the timing reduction, and the edge-sharded all-reduce of the normal equations (SURVEY §8e mode E),
with the CPU oracle standing in for the per-rank Jacobian build."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp
    from semantic_slam_amd import distributed as D
    from semantic_slam_amd.synth import make_graph
    from oracle.oracle import GraphProblem
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # --- replica mode: deal 7 graphs, no data-path collective, whole-job throughput + max time
        mine = D.shard_indices(7, rank, world)
        t = D.max_over_ranks(0.5 + 0.25 * rank)
        total = D.aggregate_throughput(float(len(mine) * 10), t)
        # --- edge-sharded mode: partial normal equations of this rank's edges, summed by all-reduce
        g = make_graph(30, 6, seed=5)
        full = GraphProblem.from_synth(g)
        U, b = full.linearize()
        lo, hi = D.shard_range(full.ne, rank, world)
        part = GraphProblem(full.vtype, full.vfixed, full.est, full.etype[lo:hi], full.evi[lo:hi], full.evj[lo:hi],
                            full.meas[lo:hi], full.info[lo:hi])
        # embed the partial system in the full graph's ordering (vertices without edges in the shard have no
        # hessian index there), exactly what a rank holding the full structure would produce
        hf, n = full.hessian_index()
        hp, _ = part.hessian_index()
        Up, bp = part.linearize()
        Up = (Up + sp.triu(Up, 1).T).tocoo()
        inv = {}
        for v in range(full.nv):
            if hp[v] >= 0:
                for k in range(6 if full.vtype[v] == 0 else 3):
                    inv[hp[v] + k] = hf[v] + k
        H = np.zeros((n, n)); bb = np.zeros(n)
        for r, c, v in zip(Up.row, Up.col, Up.data):
            H[inv[r], inv[c]] += v
        for r, v in enumerate(bp):
            bb[inv[r]] += v
        packed = torch.from_numpy(np.concatenate([H.ravel(), bb]))
        D.allreduce_normal_equations(packed)
        Hs = packed[:n * n].numpy().reshape(n, n); bs = packed[n * n:].numpy()
        Hfull = (U + sp.triu(U, 1).T).toarray()
        q.put((rank, list(map(int, mine)), t, total, float(np.abs(Hs - Hfull).max() / np.abs(Hfull).max()),
               float(np.abs(bs - b).max() / np.abs(b).max())))
    finally:
        dist.destroy_process_group()


def test_world_size_two_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, m0, t0, tot0, eh0, eb0), (r1, m1, t1, tot1, eh1, eb1) = res
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5]                     # round-robin deal, every item exactly once
    assert t0 == t1 == 0.75                                           # MAX over ranks
    assert tot0 == tot1 == pytest.approx(70 / 0.75)                   # units of ALL ranks / max time
    assert max(eh0, eh1) < 1e-13 and max(eb0, eb1) < 1e-13            # sum of edge-shard partials == full H, b


def test_shard_range_is_a_partition():
    from semantic_slam_amd.distributed import shard_range, shard_indices
    for n in (0, 1, 7, 20099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
            assert sorted(np.concatenate([shard_indices(n, r, world) for r in range(world)]).tolist()) == list(range(n))
