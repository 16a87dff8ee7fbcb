#!/bin/bash
# instruction / wait / LDS-conflict counters of the window Cholesky kernels (separate rocprofv3 --pmc passes)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r3c; mkdir -p $O; B=${1:-128}
cd /tmp && export TMPDIR=/tmp
rm -f $O/pmc.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS" "MeanOccupancyPerCU GRBM_GUI_ACTIVE"; do
  rm -rf $O/_pmc
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/_pmc -- python $R/tools/prof_opt.py $B 1 > $O/_pmc.log 2>&1 || { echo "set failed: $set"; tail -3 $O/_pmc.log; }
  python $R/tools/pmc_summary.py $O/_pmc 2>/dev/null | grep -E "k_wchol" >> $O/pmc.txt
  rm -rf $O/_pmc
done
sort $O/pmc.txt | cut -c1-330
