"""Probe: one batch of B distinct L graphs as a stream group of K parts (sslam_batch_create_streams) against the single-stream batch.
usage: python tools/multistream_probe.py <B> <K> [<K> ...]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from semantic_slam_amd import GraphBatch

B = int(sys.argv[1]); Ks = [int(k) for k in sys.argv[2:]]
paths = bench.generate_graphs("point", 5000, 1000, range(B), os.path.join("/tmp", "sslam_bench_cache"))
base = bench.build_batch(paths, B, 0, -1)
graphs = base.graphs
del base
for K in Ks:
    bt = GraphBatch(graphs, streams=K)
    bt.optimize(5)
    best = None
    reps = []
    for rep in range(int(os.environ.get('REPS', '2'))):
        bt.upload()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = bt.optimize(20)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        reps.append(round(dt * 1e3, 1))
    its = sum(int(s.iterations) for s in st)
    chi = sum(float(s.chi2_after) for s in st)
    print(f"K={K}: reps {reps} best {best*1e3:.1f} ms, {its} graph-iterations, {its/best:.0f} iters/s, chi2 sum {chi:.9e}", flush=True)
    del bt
