#!/bin/bash
# One measurement pass of round 3 on a GPU box: PMC traffic of the step kernel, the bench (K = 512 and the driver's K = 20), the
# rocprofv3 kernel trace + stats of the same bench command, the side benches, the GPU test suite.  Outputs under gpurun_out/r03_*.
R=$GRAFT_REPO_ROOT
cd $R
set -e          # a step that fails or times out ends the pass: no further GPU step behind it
T="timeout -k 10"
$T 600 python tools/pmc_traffic.py > gpurun_out/r03_pmc_traffic.log 2>&1 && cp gpurun_out/step_kernel_traffic.json profiles/step_kernel_traffic.json
cp profiles/step_kernel_traffic.json gpurun_out/r03_step_kernel_traffic.json
$T 400 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench_err.log
$T 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_k20.json 2>> gpurun_out/r03_bench_err.log
$T 300 python bench.py --exact-fp32-layers --no-cpu-baseline > gpurun_out/r03_bench_exact_fp32_layers.json 2>> gpurun_out/r03_bench_err.log
$T 300 python bench.py --no-obs-planes --no-cpu-baseline > gpurun_out/r03_bench_no_obs_planes.json 2>> gpurun_out/r03_bench_err.log
(cd /tmp && export TMPDIR=/tmp && $T 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r03_bench_under_rocprof.json 2>> $R/gpurun_out/r03_bench_err.log)
cp $(ls gpurun_out/r03_prof/*/*kernel_stats.csv | head -1) gpurun_out/r03_bench_kernel_stats.csv
rm -rf gpurun_out/r03_prof
$T 300 python tools/bench_tasks.py > gpurun_out/r03_bench_tasks.jsonl 2>> gpurun_out/r03_bench_err.log
$T 300 python tools/bench_marl_datapath.py > gpurun_out/r03_marl_datapath.json 2>> gpurun_out/r03_bench_err.log
$T 300 python tools/bench_offpolicy_collect.py > gpurun_out/r03_offpolicy_collect.json 2>> gpurun_out/r03_bench_err.log
$T 300 python tools/bench_marl_policy.py > gpurun_out/r03_marl_policy.json 2>> gpurun_out/r03_bench_err.log
$T 300 python tools/bench_mappo_rollout.py > gpurun_out/r03_mappo_rollout.json 2>> gpurun_out/r03_bench_err.log
$T 400 python tools/bench_mappo_rollout.py --agents 100 --num-envs 2048 --iters 4 > gpurun_out/r03_mappo_rollout_swarm.json 2>> gpurun_out/r03_bench_err.log
$T 300 python tools/scratch/split16_group_cost.py > gpurun_out/r03_split16_group_cost.txt 2>&1
$T 300 python tools/scratch/split_fixed_cost.py > gpurun_out/r03_split_fixed_cost.txt 2>&1
$T 300 python tools/scratch/split16_fixed_cost.py > gpurun_out/r03_split16_fixed_cost.txt 2>&1
$T 300 python tools/scratch/split16_probe.py > gpurun_out/r03_split16_probe.txt 2>&1
$T 300 python tools/bench_mappo_rollout.py --split-format bf16x3 > gpurun_out/r03_mappo_rollout_bf16x3.json 2>> gpurun_out/r03_bench_err.log
tools/scratch/trace_gaps.sh > /dev/null 2>&1 && cp gpurun_out/trace_gaps.txt gpurun_out/r03_trace_rollout.txt
tools/scratch/marl_stats.sh > /dev/null 2>&1
$T 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest.log 2>&1
cp gpurun_out/parity_margins.json gpurun_out/r03_parity_margins_gpu.json
tail -3 gpurun_out/r03_gputest.log; head -c 600 gpurun_out/r03_bench.json; echo; tail -5 gpurun_out/r03_bench_err.log
