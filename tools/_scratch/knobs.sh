#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-frontend --plane-batch 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernel_ms'], d.get('single_graph'))"
SSLAM_CHOL_STAMPS=1 python tools/prof_opt.py 1 3 2>&1 | grep -E "stamps"
