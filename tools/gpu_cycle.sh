#!/bin/bash
# One GPU round trip of the development loop: backend parity tests, then per-launch profiles at batch 512 and batch 1.
# usage (on the GPU box): bash tools/gpu_cycle.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=${1:-cyc}
cd /tmp && export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_graph_gpu.py -m gpu -x -q > $O/${T}_tests.txt 2>&1; tail -3 $O/${T}_tests.txt
for B in 512 1; do
  rm -rf $O/_p$B
  rocprofv3 --kernel-trace --output-format csv -d $O/_p$B -- python $R/tools/prof_opt.py $B 3 > $O/${T}_p$B.log 2>&1
  python $R/tools/level_profile.py $O/_p$B > $O/${T}_lv$B.txt; rm -rf $O/_p$B
done
head -8 $O/${T}_lv512.txt; grep -E "k_chol" $O/${T}_lv512.txt | tail -18; head -6 $O/${T}_lv1.txt
SSLAM_CHOL_STAMPS=1 python $R/tools/prof_opt.py 1 3 2>&1 | grep stamps
