#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for F in 0 1 2 3 4 8 12 15; do echo -n "LIN_DBG=$F "; SSLAM_LIN_DBG=$F python tools/lin_only.py 512 2>&1 | tail -1; done
