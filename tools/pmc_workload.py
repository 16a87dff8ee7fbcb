"""PMC workload: ONE full-batch Jacobian build or ONE full-batch factorisation (+ warm-up) of B distinct L graphs (bench.py's graphs).
usage: python tools/pmc_workload.py <B> factor|jacobian      (run under rocprofv3 --pmc ...)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
B = int(sys.argv[1]); what = sys.argv[2]
paths = bench.generate_graphs("point", 5000, 1000, range(B), os.path.join("/tmp", "sslam_bench_cache"))
b = bench.build_batch(paths, B, 0, -1)
if what == "factor":
    print("factor / solve ms", b.time_solver(3))      # warm-up + one timed factorisation: two dispatches of every kernel
else:
    print("jacobian ms", b.time_linearize(1))         # warm-up + one timed build
