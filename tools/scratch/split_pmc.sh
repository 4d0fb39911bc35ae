#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_split/$n -- python3 $R/tools/scratch/split_pmc.py > $R/gpurun_out/pmc_split_$n.log 2>&1 || echo "pass $set failed"
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_split | grep -A12 "linear_split" > gpurun_out/r03_split_pmc.txt; cat gpurun_out/r03_split_pmc.txt
