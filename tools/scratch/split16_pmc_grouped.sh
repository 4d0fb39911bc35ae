#!/bin/bash
# PMC passes over the two-plane layer kernel at the grouped MARL shape (20 networks x 4096 x 512 x 512: five 256 x 128 tiles per CU in the
# persistent loop) beside the PPO shape -- L2 hit rate, reads beyond L2 and the wave-time split.  Output: gpurun_out/r04_split16_pmc_grouped.txt
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PMC_G=20 PMC_M=4096 PMC_K=512 PMC_N=512
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_split16g/$n -- python3 $R/tools/scratch/split16_pmc.py > $R/gpurun_out/pmc_split16g_$n.log 2>&1 || echo "pass $set failed"
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_split16g | grep -A16 "linear_split16" > gpurun_out/r04_split16_pmc_grouped.txt; cat gpurun_out/r04_split16_pmc_grouped.txt
rm -rf gpurun_out/pmc_split16g gpurun_out/pmc_split16g_*.log
