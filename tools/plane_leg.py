"""the plane-landmark leg of bench.py alone: python tools/plane_leg.py [graphs] [streams]"""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from semantic_slam_amd import GraphBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 4
paths = bench.generate_graphs("plane", 5000, 1000, range(n), os.path.join("/tmp", "sslam_bench_cache"))
pb = bench.build_batch(paths, n, 0, -1)
if streams > 1:
    pb = GraphBatch(pb.graphs, streams=streams)
ps, pdt = bench.timed_optimize(pb, 20, 2, lambda: None)
pit = [int(s.iterations) for s in ps]
print(json.dumps({"plane_iters_per_sec": round(sum(pit) / pdt, 1), "iters_min": min(pit), "iters_max": max(pit), "trials": int(sum(int(s.trials) for s in ps)), "seconds": round(pdt, 4)}))
