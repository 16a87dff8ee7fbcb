#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/t_solver.py <<PY
import sys, os, time
sys.path.insert(0, "$R")
import bench
from semantic_slam_amd import GraphSLAM, GraphBatch
B = int(sys.argv[1]); solver = int(sys.argv[2])
paths = bench.generate_graphs("point", 5000, 1000, range(B), "/tmp/sslam_bench_cache")
b = bench.build_batch(paths, B, 0, solver)
f, s = b.time_solver(3)
print("solver", solver, "batch", B, "factor ms", round(f, 3), "backward ms", round(s, 3), "valu" if os.environ.get("SSLAM_WCHOL_VALU") else "mfma")
st = b.optimize(5)
print("  chi2", st[0].chi2_after, "iters", st[0].iterations)
PY
SSLAM_WCHOL_VALU=1 timeout 600 python /tmp/t_solver.py 512 3 2>&1 | tail -2
timeout 600 python /tmp/t_solver.py 512 3 2>&1 | tail -2
timeout 300 python /tmp/t_solver.py 1 3 2>&1 | tail -2
SSLAM_WCHOL_VALU=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python /tmp/t_solver.py 512 3 > $O/prof.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats_valu.csv; head -6 $O/kernel_stats_valu.csv | cut -c1-200; rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python /tmp/t_solver.py 512 3 > $O/prof.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats_mfma.csv; head -6 $O/kernel_stats_mfma.csv | cut -c1-200; rm -rf $O/stats
