#!/bin/bash
# Occupancy / dispatch-stall / memory-pipeline counters per kernel (separate rocprofv3 --pmc passes; no trace domains combined with them).
# usage (GPU box): bash tools/pmc_mem.sh <tag> <batch>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=${1:-mem}; B=${2:-512}
cd /tmp && export TMPDIR=/tmp
rm -f $O/${T}_pmc.txt
for set in "MeanOccupancyPerCU OccupancyPercent MemUnitStalled" \
           "SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_RES_STALL_CSN SPI_CSN_BUSY" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_SPI_BUSY"; do
  rm -rf $O/_pmc
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/_pmc -- python $R/tools/prof_opt.py $B 1 > $O/_pmc.log 2>&1 || { echo "set failed: $set"; tail -3 $O/_pmc.log; }
  python $R/tools/pmc_summary.py $O/_pmc 2>/dev/null | grep -E "k_chol|linearize" >> $O/${T}_pmc.txt
  rm -rf $O/_pmc
done
sort $O/${T}_pmc.txt | cut -c1-330
