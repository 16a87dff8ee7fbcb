"""single-graph latency and the short tick replay (bench.py legs) alone: python tools/small_legs.py"""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
paths = bench.generate_graphs("point", 5000, 1000, range(1), os.path.join("/tmp", "sslam_bench_cache"))
b1 = bench.build_batch(paths[:1], 1, 0, -1)
s1, d1 = bench.timed_optimize(b1, 20, 1, lambda: None)
n1 = max(int(s1[0].iterations), 1)
f, v = b1.time_solver(5)
print(json.dumps({"single_graph_iters_per_sec": round(n1 / d1, 1), "ms_per_iter": round(1e3 * d1 / n1, 3), "factor_ms": round(f, 3), "solve_ms": round(v, 3)}))
t = bench.bench_tick(0, cpu_baseline=False)
KEYS = ("keyframes", "landmarks", "ms_per_tick", "ms_per_tick_association", "ms_per_tick_optimize", "ms_per_tick_marginals", "lm_iterations_per_tick")
print(json.dumps({k: t[k] for k in KEYS}))
t = bench.bench_tick(0, n_samples=2400, cpu_baseline=False, n_landmarks=160)
print(json.dumps({k: t[k] for k in KEYS}))
