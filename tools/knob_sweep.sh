#!/bin/bash
# full-batch factor / backward-solve time of 512 distinct L graphs under plan options: bash tools/knob_sweep.sh "mid_width=60,cap_mid=1800" "group_cap=1000" ...
# (each argument is one SSLAM_CHOL_OPTS string; "-" = the defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
for e in "$@"; do echo "== $e"; o="$e"; [ "$e" = "-" ] && o=""; SSLAM_CHOL_OPTS="$o" timeout 300 python $R/tools/pmc_workload.py 512 factor 2>&1 | grep -E "factor|chol-dump.*tail" | tail -2; done
