#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_split16/$n -- python3 $R/tools/scratch/split16_pmc.py > $R/gpurun_out/pmc_split16_$n.log 2>&1 || echo "pass $set failed"
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_split16 | grep -A22 "linear_split16" > gpurun_out/r03_split16_pmc.txt; cat gpurun_out/r03_split16_pmc.txt
rm -rf gpurun_out/pmc_split16
