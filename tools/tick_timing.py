"""Where a tick of the orchestrator spends its time (host timers inside sslam_graph_optimize: SSLAM_TIMING=1).
usage (GPU box): SSLAM_TIMING=1 python tools/tick_timing.py 2> timing.txt"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
r = bench.bench_tick(0, n_samples=600, cpu_baseline=False)
print({k: v for k, v in r.items() if k not in ("workload", "cpu_baseline")})
