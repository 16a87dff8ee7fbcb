#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_graph_gpu.py -m gpu -x -q -k "point_point or window" > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
echo "--- tick, piece plan"; python $R/tools/tick_timing.py 2>/dev/null | tail -1
echo "--- tick, window plan MFMA"; SSLAM_WCHOL=1 python $R/tools/tick_timing.py 2>/dev/null | tail -1
echo "--- tick, window plan VALU"; SSLAM_WCHOL=1 SSLAM_WCHOL_VALU=1 python $R/tools/tick_timing.py 2>/dev/null | tail -1
