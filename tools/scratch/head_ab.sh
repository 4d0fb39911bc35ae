#!/bin/bash
# duration of ppo_head_act_kernel inside the rollout graph for A/B builds of the library (MMS_LIB)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" _head_NOSAMPLE _head_NOLOOP; do
  export MMS_LIB=$R/massive_marl_benchmark_amd/lib/libmms$v.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/head_ab$v -- python3 $R/bench.py --steps 64 --warmup 16 --no-cpu-baseline > /dev/null 2>&1
  echo "variant '$v'"; python3 -c "
import csv,glob,sys
for r in csv.DictReader(open(glob.glob('$R/gpurun_out/head_ab$v/*/*kernel_stats.csv')[0])):
    if 'ppo_head_act' in r['Name'] or 'split16_kernel<2, 0' in r['Name'] or 'ant_step' in r['Name']: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))"
  rm -rf $R/gpurun_out/head_ab$v
done
