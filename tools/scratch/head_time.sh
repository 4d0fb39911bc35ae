#!/bin/bash
# average duration of the policy kernels inside the bench's rollout graph (rocprofv3 --stats)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/head_time -- python3 $R/bench.py --steps 64 --warmup 16 --no-cpu-baseline > /dev/null 2>&1
python3 -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('$R/gpurun_out/head_time/*/*kernel_stats.csv')[0])):
    if 'ppo_head_act' in r['Name'] or 'split16_kernel' in r['Name'] or 'ant_step' in r['Name'] or 'marl_heads' in r['Name']: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2))"
rm -rf $R/gpurun_out/head_time
