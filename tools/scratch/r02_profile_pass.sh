set -e
R=$GRAFT_REPO_ROOT
cd $R
python tools/pmc_traffic.py > gpurun_out/pmc_traffic.log 2>&1
cp gpurun_out/step_kernel_traffic.json profiles/step_kernel_traffic.json
python bench.py > gpurun_out/r02_v8_bench.json 2> gpurun_out/bench_err.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_v8_bench_k20.json 2>> gpurun_out/bench_err.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r02_v8_bench_under_rocprof.json 2>> $R/gpurun_out/bench_err.log
cd $R
python tools/bench_tasks.py > gpurun_out/r02_v8_bench_tasks.jsonl 2>> gpurun_out/bench_err.log
python tools/bench_marl_datapath.py > gpurun_out/r02_v8_marl_datapath.json 2>> gpurun_out/bench_err.log
python tools/bench_offpolicy_collect.py > gpurun_out/r02_v8_offpolicy_collect.json 2>> gpurun_out/bench_err.log
python tools/bench_marl_policy.py > gpurun_out/r02_v8_marl_policy.json 2>> gpurun_out/bench_err.log
python tools/bench_mappo_rollout.py > gpurun_out/r02_v8_mappo_rollout.json 2>> gpurun_out/bench_err.log
python tools/bench_mappo_rollout.py --agents 100 --num-envs 2048 --iters 4 > gpurun_out/r02_v8_mappo_rollout_swarm.json 2>> gpurun_out/bench_err.log
python -m pytest tests -m gpu -x -q > gpurun_out/r02_v8_gputest.log 2>&1
cp gpurun_out/parity_margins.json gpurun_out/r02_v8_parity_margins_gpu.json
ls gpurun_out/r02_prof/*/ | head
