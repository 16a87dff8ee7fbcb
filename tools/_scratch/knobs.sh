#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_graph_gpu.py -m gpu -x -q -k shim 2>&1 | tail -8
for V1 in 1 0; do echo "LIN_V1=$V1"; SSLAM_LIN_V1=$V1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-frontend --plane-batch 0 --no-single 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['graphs_terminated'], d['kernel_ms'])"; done
