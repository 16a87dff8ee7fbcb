#!/bin/bash
# PMC evidence for the frontend kernels (csrc/sslam_seg.hip): HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and where a wave's cycles go
# (SQ busy / wait / active per instruction class), per kernel of one batched frontend call (32 frames x 32 boxes, tools/frontend_kernels.py).
# usage (GPU box): bash tools/pmc_frontend.sh   -> gpurun_out/r6_pmc_frontend.json   (copy into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  rm -rf $O/_pmcf$i
  rocprofv3 --pmc $set --output-format csv -d $O/_pmcf$i -- python $R/tools/frontend_kernels.py > /dev/null 2>&1
  i=$((i+1))
done
python $R/tools/pmc_frontend_summary.py $O/r6_pmc_frontend.json $O/_pmcf0 $O/_pmcf1 $O/_pmcf2 $O/_pmcf3 $O/_pmcf4
rm -rf $O/_pmcf*
