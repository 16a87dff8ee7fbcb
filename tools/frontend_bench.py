"""frontend leg of bench.py alone: python tools/frontend_bench.py"""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
print(json.dumps(bench.bench_frontend(0, cpu_baseline=False), indent=1))
