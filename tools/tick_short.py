"""the short tick replay (110 keyframes) alone, for a kernel trace: rocprofv3 --kernel-trace --stats -- python tools/tick_short.py"""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
t = bench.bench_tick(0, cpu_baseline=False)
print(json.dumps({k: t[k] for k in ("keyframes", "landmarks", "ms_per_tick", "ms_per_tick_optimize", "lm_iterations_per_tick")}))
