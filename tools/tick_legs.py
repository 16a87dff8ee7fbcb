"""the two tick replays of bench.py alone (no CPU baseline): python tools/tick_legs.py"""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
KEYS = ("keyframes", "landmarks", "ms_per_tick", "ms_per_tick_association", "ms_per_tick_optimize", "ms_per_tick_marginals", "lm_iterations_per_tick")
for kw in ({}, {"n_samples": 2400, "n_landmarks": 160}):
    t = bench.bench_tick(0, cpu_baseline=False, **kw)
    print(json.dumps({k: t[k] for k in KEYS}))
