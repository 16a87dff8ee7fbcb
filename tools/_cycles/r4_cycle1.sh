#!/bin/bash
# round 4, cycle 1: new elimination order (multiple minimum degree over independent sets) -- parity, then factor / solve time under plan knobs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout 900 python -m pytest $R/tests/test_graph_gpu.py -m gpu -x -q > $O/r4c1_tests.txt 2>&1; tail -3 $O/r4c1_tests.txt
bash $R/tools/knob_sweep.sh "SSLAM_CHOL_ORDER=mindeg" "SSLAM_CHOL_ORDER=mmd" "SSLAM_CHOL_ORDER_SLACK=1.5,2" "SSLAM_CHOL_CAP_LEAF=1400" "SSLAM_CHOL_CAP_LEAF=2000" \
  "SSLAM_CHOL_TAIL_WIDTH=3" "SSLAM_CHOL_TAIL_WIDTH=12" "SSLAM_CHOL_CAP_LEAF=1400 SSLAM_CHOL_TAIL_WIDTH=12" "SSLAM_CHOL_CAP_LEAF=600" > $O/r4c1_sweep.txt 2>&1
cat $O/r4c1_sweep.txt
for o in mindeg mmd; do echo "== small legs $o"; SSLAM_CHOL_ORDER=$o timeout 600 python $R/tools/small_legs.py 2>&1 | tail -3; done > $O/r4c1_small.txt 2>&1
cat $O/r4c1_small.txt
SSLAM_CHOL_STAMPS=1 python $R/tools/prof_opt.py 1 3 2>&1 | grep stamps > $O/r4c1_stamps.txt; cat $O/r4c1_stamps.txt
