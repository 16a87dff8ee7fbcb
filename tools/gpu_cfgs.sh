#!/bin/bash
# Compare factorisation configurations at batch 512 (and batch 1): per-launch kernel times from rocprofv3.
# usage (GPU box): bash tools/gpu_cfgs.sh "<NT_LEAF> <CAP_LEAF> <PCAP_LEAF> <CAP_TAIL> <GROUP_CAP> <USTAGE>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  export SSLAM_CHOL_NT_LEAF=$1 SSLAM_CHOL_CAP_LEAF=$2 SSLAM_CHOL_PCAP_LEAF=$3 SSLAM_CHOL_CAP_TAIL=${4:-4608} SSLAM_CHOL_GROUP_CAP=${5:-0} SSLAM_CHOL_USTAGE=${6:-1}
  T="cfg_$1_$2_$3_${4:-4608}_${5:-0}_${6:-1}"
  for B in 512 1; do
    rm -rf $O/_p
    rocprofv3 --kernel-trace --output-format csv -d $O/_p -- python $R/tools/prof_opt.py $B 2 > $O/${T}_$B.log 2>&1
    python $R/tools/level_profile.py $O/_p > $O/${T}_lv$B.txt; rm -rf $O/_p
    echo "== $T batch $B: $(tail -1 $O/${T}_$B.log)"; grep -E "k_chol" $O/${T}_lv$B.txt | head -4 | cut -c1-110
  done
done
