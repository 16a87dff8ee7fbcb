#!/bin/bash
# kernel times of the batched frontend call under the timing switches of SSLAM_SEG_DBG
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for d in ${DBGS:-0 1 2 4}; do
  rm -rf /tmp/fp; SSLAM_SEG_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/tools/frontend_kernels.py > /dev/null 2>&1
  echo "== SSLAM_SEG_DBG=$d"; python - <<PY
import csv,glob
f=glob.glob('/tmp/fp/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'seg::' in n: print(f"  {n.split('seg::')[1][:28]:28s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
done
