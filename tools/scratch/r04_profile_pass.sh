#!/bin/bash
# One measurement pass of round 4 on a GPU box: PMC traffic of the step kernel, the bench (K = 512 and the driver's K = 20) with its
# eager series and without the fused policy head, the layer kernel's time against K, the rocprofv3 kernel trace + stats of the same bench command, the rollout timeline, the side benches (both
# MultiIngenuity configurations), the sizing probe of the policy's launch shapes, the GPU test suite.  Outputs under gpurun_out/r04_*.
R=$GRAFT_REPO_ROOT
cd $R
set -e          # a step that fails or times out ends the pass: no further GPU step behind it
T="timeout -k 10"
$T 600 python tools/pmc_traffic.py > gpurun_out/r04_pmc_traffic.log 2>&1 && cp gpurun_out/step_kernel_traffic.json profiles/step_kernel_traffic.json
cp profiles/step_kernel_traffic.json gpurun_out/r04_step_kernel_traffic.json
$T 400 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench_err.log
$T 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_k20.json 2>> gpurun_out/r04_bench_err.log
$T 300 python bench.py --no-graph --no-cpu-baseline > gpurun_out/r04_bench_no_graph.json 2>> gpurun_out/r04_bench_err.log
$T 300 python bench.py --no-head-fusion --no-cpu-baseline > gpurun_out/r04_bench_no_head_fusion.json 2>> gpurun_out/r04_bench_err.log
$T 200 python tools/scratch/split16_fixed_cost.py > gpurun_out/r04_split16_fixed_cost.txt 2>> gpurun_out/r04_bench_err.log
(cd /tmp && export TMPDIR=/tmp && $T 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r04_bench_under_rocprof.json 2>> $R/gpurun_out/r04_bench_err.log)
cp $(ls gpurun_out/r04_prof/*/*kernel_stats.csv | head -1) gpurun_out/r04_bench_kernel_stats.csv
rm -rf gpurun_out/r04_prof
tools/scratch/trace_gaps.sh > /dev/null 2>&1 && cp gpurun_out/trace_gaps.txt gpurun_out/r04_trace_rollout.txt
$T 300 python tools/bench_tasks.py > gpurun_out/r04_bench_tasks.jsonl 2>> gpurun_out/r04_bench_err.log
$T 300 python tools/bench_marl_datapath.py > gpurun_out/r04_marl_datapath.json 2>> gpurun_out/r04_bench_err.log
$T 300 python tools/bench_offpolicy_collect.py > gpurun_out/r04_offpolicy_collect.json 2>> gpurun_out/r04_bench_err.log
$T 300 python tools/bench_offpolicy_collect.py --env-spacing 0 > gpurun_out/r04_offpolicy_collect_in_flight.json 2>> gpurun_out/r04_bench_err.log
$T 300 python tools/bench_marl_policy.py > gpurun_out/r04_marl_policy.json 2>> gpurun_out/r04_bench_err.log
tools/scratch/marl_stats.sh > /dev/null 2>&1 || true
$T 300 python tools/bench_mappo_rollout.py > gpurun_out/r04_mappo_rollout.json 2>> gpurun_out/r04_bench_err.log
$T 400 python tools/bench_mappo_rollout.py --agents 100 --num-envs 2048 --iters 4 > gpurun_out/r04_mappo_rollout_swarm.json 2>> gpurun_out/r04_bench_err.log
for v in "" "MMS_HEAD_RT=2" "MMS_SPLIT_MT=4" "MMS_SPLIT_MT=2"; do env $v $T 200 python tools/probe_policy_split.py; done > gpurun_out/r04_policy_split_probe.jsonl 2>> gpurun_out/r04_bench_err.log
$T 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest.log 2>&1
cp gpurun_out/parity_margins.json gpurun_out/r04_parity_margins_gpu.json
tail -3 gpurun_out/r04_gputest.log; head -c 600 gpurun_out/r04_bench.json; echo; tail -5 gpurun_out/r04_bench_err.log
