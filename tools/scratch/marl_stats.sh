#!/bin/bash
# per-kernel averages of the grouped MARL policy pass (tools/bench_marl_policy.py) -> gpurun_out/r04_marl_policy_kernel_stats.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/marl_stats -- python3 $R/tools/bench_marl_policy.py > /dev/null 2>&1
cp $(ls $R/gpurun_out/marl_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/r04_marl_policy_kernel_stats.csv
rm -rf $R/gpurun_out/marl_stats
python3 -c "
import csv
for r in list(csv.DictReader(open('$R/gpurun_out/r04_marl_policy_kernel_stats.csv')))[:22]:
    print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3,2), r['Percentage'])"
