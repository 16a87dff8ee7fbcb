#!/bin/bash
# full-batch factor / backward-solve time of 512 distinct L graphs under plan knobs: bash tools/knob_sweep.sh "ENV=V ENV2=V" "ENV=V" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp
for e in "$@"; do echo "== $e"; env $e timeout 300 python $R/tools/pmc_workload.py 512 factor 2>&1 | grep -E "factor|chol-dump.*tail" | tail -2; done
