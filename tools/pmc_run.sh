#!/bin/bash
# SQ instruction / occupancy counters per kernel (separate rocprofv3 --pmc passes; no trace domains combined with them).
# usage (GPU box): bash tools/pmc_run.sh <tag> <batch> ; SSLAM_CHOL_OPTS selects the factorisation configuration
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=${1:-pmc}; B=${2:-512}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $O/_pmc
  rocprofv3 --pmc $set --output-format csv -d $O/_pmc -- python $R/tools/prof_opt.py $B 1 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $O/_pmc 2>/dev/null | grep -E "k_chol|linearize" >> $O/${T}_pmc.txt
  rm -rf $O/_pmc
done
sort $O/${T}_pmc.txt | cut -c1-260
