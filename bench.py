#!/usr/bin/env python3
"""bench.py -- env-steps/sec, TenAnt, 4096 envs per GPU, PPO rollout (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one PPO rollout step over this rank's 4096 envs: actor + critic inference (the reference's
ActorCritic, agents/algorithms/rl/ppo/module.py: two MLPs [1024,1024,512], ELU, fp32 -- each hidden layer of both networks one
launch with bias + ELU fused: mms_linear_group_act_split16, the fp32 product evaluated on the 16-bit matrix pipe from operands carried
as two fp16 planes under a power-of-two scale per row (operands kept to 2^-22, three plane products, fp32 accumulation; its error
against float64 is measured in this run next to the exact-fp32 MFMA kernel's and the three-bf16-plane kernel's,
`policy_layers_error_vs_f64`, and the rollout with each of those two layer kernels is timed beside it:
`rollout_exact_fp32_layers`, `rollout_bf16x3_layers`), both last layers + Gaussian action sample + log-prob + the
add_transitions stores (mms_ppo_heads_act, one HIP launch), the fused VecTask
step (mms_step: physics substeps + reset + obs + reward in one HIP launch, writing observation / reward / done
straight into the rollout slots), and every nsteps=8 steps the GAE scan + advantage normalisation
(the span timed as collection_time in agents/algorithms/rl/ppo/ppo.py:123-161 plus compute_returns).
Envs are sharded over ranks with no data-path collective (weak scaling); `value` = all ranks' env-steps
divided by the slowest rank's time.  Inputs are synthetic and resident in HBM before the timed region.

The JSON line also carries
  sim_only      the same engine stepped with pre-drawn actions (no policy), the series the step kernel's
                roofline is computed from,
  roofline      step kernel: algorithmic bytes (4660 B per env-step) / measured launch
                duration against the 8 TB/s HBM peak,
  cpu_baseline  the CPU oracle (oracle/mms_oracle.c, OpenMP over envs) on a bounded sample of the same
                workload -- rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 4660
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec
NSTEPS = 8                              # cfg/ppo/config.yaml:23
GAMMA, LAM = 0.96, 0.95                 # cfg/ppo/config.yaml:30-31


def step_kernel_source_hash():
    """sha256 over what the step kernel is compiled from: the kernel file, the lane header, the launch-argument header and the
    Makefile (flags).  profiles/step_kernel_traffic.json records the hash of the build its PMC passes profiled."""
    import hashlib
    h = hashlib.sha256()
    for name in ("step_kernels.hip", "mms_lane.h", "step_args.h", "Makefile"):
        with open(os.path.join(ROOT, "massive_marl_benchmark_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic():
    """HBM bytes per step-kernel launch from the committed rocprofv3 PMC passes (profiles/step_kernel_traffic.json, written by
    tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of tools/profile_step.py, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot run the profiler on itself.  None when the file is
    absent OR was measured on another build of the kernel (its source hash differs from the sources next to this file)."""
    path = os.path.join(ROOT, "profiles", "step_kernel_traffic.json")
    try:
        with open(path) as f:
            tr = json.load(f)
    except Exception:
        return None
    if tr.get("source_hash") != step_kernel_source_hash():
        return None
    return tr


def shard_for_rank(rank, world, envs_per_gpu):
    """(env_offset, total_envs) of rank's shard: rank r owns global envs [r*n, (r+1)*n)."""
    return rank * envs_per_gpu, world * envs_per_gpu


POLICY_CFG = {"pi_hid_sizes": [1024, 1024, 512], "vf_hid_sizes": [1024, 1024, 512], "activation": "elu"}   # cfg/ppo/config.yaml:6-9
INIT_NOISE_STD = 0.8                                                                                        # cfg/ppo/config.yaml:33


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--num-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--policy-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--no-graph", action="store_true", help="launch the rollout step eagerly instead of replaying a hipGraph")
    ap.add_argument("--library-gemms", action="store_true",
                    help="A/B: the policy's layers as library GEMMs + separate ELU passes (critic on a second stream) instead of mms_linear2_act / mms_ppo_heads_act")
    ap.add_argument("--exact-fp32-layers", action="store_true",
                    help="A/B: headline with the hidden layers on the exact-fp32 MFMA kernel (mms_linear2_act) instead of a split kernel")
    ap.add_argument("--split-format", default="f16x2", choices=["f16x2", "bf16x3"],
                    help="planes of the split layer kernel: two scaled fp16 planes (mms_linear_group_act_split16, default) or three exact bf16 planes (mms_linear_group_act_split)")
    ap.add_argument("--no-obs-planes", action="store_true",
                    help="A/B: the policy splits the observation rows itself (mms_split_planes16_group) instead of reading the operand planes the step kernel writes beside them (mms_bind_obs_planes16)")
    ap.add_argument("--rollouts-per-graph", type=int, default=1,
                    help="A/B: whole rollouts (8 steps + GAE) per hipGraph replay (default 1: one PPO iteration's collection phase)")
    ap.add_argument("--one-stream", action="store_true", help="A/B: actor and critic MLPs on one stream")
    ap.add_argument("--defer-critic", action="store_true",
                    help="A/B: let the critic pass overlap the sampling kernel and the env step (joined before the GAE); faster, but "
                         "the step kernel then shares the GPU with GEMMs and its rocprof average no longer is its stand-alone duration")
    ap.add_argument("--all-obs-rows", action="store_true", help="A/B: the step kernel also writes the engine's raw and clamped observation buffers in the rollout")
    ap.add_argument("--unfused", action="store_true", help="A/B: torch sampling + add_transitions copies instead of mms_ppo_act and bound rollout slots")
    ap.add_argument("--no-head-fusion", action="store_true",
                    help="A/B: the policy's output heads + sampling as their own launch (mms_ppo_heads_act) instead of the step kernel's prologue (mms_bind_policy_head)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="oracle steps for the CPU baseline (0 = sized for ~15 s)")
    args = ap.parse_args()

    import torch
    from massive_marl_benchmark_amd.algorithms.rl.ppo.storage import RolloutStorage
    from massive_marl_benchmark_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (the engine has no CPU path)")
    dist = None
    # under torch.distributed.run (RANK / MASTER_ADDR in the environment) the process group is initialised at ANY world size, 1 included:
    # the RCCL init, barrier and all_reduce(MAX) path of the multi-GPU runs is then the one a single-GPU box can rehearse
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ and "TORCHELASTIC_RUN_ID" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MMS_BENCH_BACKEND=gloo is a rehearsal of the multi-rank control flow on a box with fewer GPUs than ranks (ranks share
        # the devices round robin; the timing it prints is not a scaling figure).  The driver's runs use RCCL, one rank per GPU.
        backend = os.environ.get("MMS_BENCH_BACKEND", "nccl")
        if backend == "gloo":
            local_rank = local_rank % torch.cuda.device_count()
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    N = args.num_envs
    env_offset, total_envs = shard_for_rank(rank, world, N)
    eng = Engine("TenAnt", num_envs=N, device=local_rank, seed=0, env_offset=env_offset, total_envs=total_envs, clip_obs=5.0)
    obs_dim, act_dim = eng.obs_dim, eng.num_actions
    pdtype = torch.float32 if args.policy_dtype == "fp32" else torch.bfloat16
    torch.manual_seed(1234 + rank)
    from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic
    ac = ActorCritic((obs_dim,), (0,), (act_dim,), INIT_NOISE_STD, POLICY_CFG, seed=1234, row_offset=env_offset).to(device)
    storage = RolloutStorage(N, NSTEPS, (obs_dim,), (0,), (act_dim,), device=str(device))
    states = torch.zeros(N, 0, device=device)
    actions_buf, rew, reset = eng.tensor("actions"), eng.tensor("rew"), eng.tensor("reset")
    obs_clipped = eng.tensor("obs_clipped")
    half_log_2pi = 0.9189385332046727

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- sim-only series: pre-drawn U(-1,1) actions, ring of 16 ---------------------
    g = torch.Generator(device="cpu").manual_seed(1234)
    ring = [(torch.rand(N, act_dim, generator=g) * 2 - 1).to(device) for _ in range(16)]

    def sim_step(i):
        eng.bind_actions(ring[i % 16])              # the engine reads the caller's tensor in place (mms_bind_actions): no copy kernel
        eng.step()

    for i in range(64):                         # reset-all + settle
        sim_step(i)
    barrier()
    sim_steps = max((args.steps // 64) * 64, 256)
    sim_graph = None
    if not args.no_graph:                       # 16 steps (one pass over the ring) per hipGraph replay
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            sim_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(sim_graph, stream=side):
                for i in range(16):
                    sim_step(i)
        torch.cuda.current_stream(device).wait_stream(side)
        for _ in range(8):                      # the first replays of a fresh graph can carry a one-off runtime stall (tens of ms)
            sim_graph.replay()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # timed in four equal batches; the series' figure is the MEDIAN batch: early in a process a replay occasionally carries a
    # one-off runtime stall of tens of ms (seen with every kernel layout, tools/scratch/sim_graph_probe.py), which would otherwise
    # decide this informational series.  All four batch times are reported.
    sim_batches = []
    per_batch = sim_steps // 4
    for b in range(4):
        t0 = time.perf_counter()
        if sim_graph is None:
            for i in range(per_batch):
                sim_step(b * per_batch + i)
        else:
            for _ in range(per_batch // 16):
                sim_graph.replay()
        torch.cuda.synchronize()
        sim_batches.append(time.perf_counter() - t0)
    eng.bind_actions(None)
    sim_wall = 4.0 * sorted(sim_batches)[1]            # (lower) median batch x 4; the mean and every batch are reported beside it
    sim_wall_mean = sum(sim_batches)
    # step-kernel launch duration: back-to-back launches with nothing else on the stream, HIP events around them.
    # Measured twice -- here, after the sim-only series, and again right after the GEMM-heavy rollout series -- because the
    # kernel is VALU-issue bound and so follows the core clock, which the rollout's matrix-core bursts pull down.
    step_kernel_batches = []

    def time_step_kernel(n=256, batches=8):
        # eight batches of 32 launches, the median batch: early in a process a launch occasionally carries a one-off runtime stall of tens
        # of ms (the same stall the sim-only series reports as outlier batches; profiles/r02_graph_stall_probe.txt) -- one of them inside a
        # single 256-launch interval once put 170 us into this figure (round 4, K = 20 run).  Every batch time is reported.
        per = n // batches
        times = []
        for _ in range(batches):
            ev0.record()
            for _ in range(per):
                eng.step()
            ev1.record()
            torch.cuda.synchronize()
            times.append(ev0.elapsed_time(ev1) / per)
        step_kernel_batches.append(times)
        return sorted(times)[(batches - 1) // 2]
    kernel_ms_pre = time_step_kernel()

    # ---- PPO rollout series -----------------------------------------------------------------------------------
    PLANES_SCALE = 2048.0                                                   # clip_obs 5 x 2^11 <= 2^14
    planes_buf = torch.empty(N * ((obs_dim + 31) // 32) * 128, dtype=torch.uint8, device=device)
    planes_state = {"used": False}
    head_state = {"fused": False}

    def measure_rollout(pdtype, K_req, W_req, split=True, fmt=None, use_graph=True):
        ac_ = ac if pdtype == torch.float32 else ac_bf16
        ac_.split_layers = bool(split)
        ac_.split_format = fmt or args.split_format
        # the output heads + sampling tail run in the step kernel's prologue where the engine has the layout for it (mms_bind_policy_head)
        ac_.bind_rollout(None if args.unfused else storage, None if args.unfused else actions_buf,
                         step_engine=None if (args.unfused or args.no_head_fusion or args.library_gemms) else eng)
        head_state["fused"] = ac_._step_engine is not None
        ac_.two_streams = not args.one_stream
        ac_.fuse_head = not args.library_gemms
        ac_.fuse_layers = not args.library_gemms
        ac_.defer_value = args.defer_critic and not args.one_stream

        # The step kernel also leaves the clamped row as the layers' operand planes (two fp16 planes, constant scale 2^11: the row is
        # bounded by clip_obs = 5): the policy then has no split pass over the observation.  Valid from the first step on.
        use_planes = (pdtype == torch.float32 and split and ac_.split_format == "f16x2" and not args.no_obs_planes and not args.unfused
                      and not args.library_gemms and ac_._split_applies(N, [m for m in ac_.actor if isinstance(m, torch.nn.Linear)][:-1]))
        planes_state["used"] = use_planes
        planes_valid = [False]
        eng.bind_obs_planes(planes_buf if use_planes else None, PLANES_SCALE)

        def rollout_step_fused():
            # Zero-copy rollout: the engine writes observation t+1, reward t and done t into the storage slots, mms_ppo_act
            # writes the action into the engine and action / log-prob / value / mu / sigma into slot t; add_transitions
            # recognises the slots by address and has nothing left to copy.
            t = storage.step
            obs_t = storage.observations[t]
            if t == 0:
                obs_t.copy_(obs_clipped)                                     # the first slot of a rollout: the current observation
            pl = (planes_buf, PLANES_SCALE) if (use_planes and planes_valid[0]) else None
            act, logp, value, mu, sigma = ac_.act(obs_t, states, obs_planes=pl)   # module.py:73-87
            last = t + 1 == NSTEPS
            eng.bind_obs_out(None if last else storage.observations[t + 1])
            if not args.all_obs_rows:
                eng.set_obs_outputs(raw=False, clipped=last)                 # one observation row per env-step, not three
            eng.bind_rollout_out(storage.rewards[t].view(-1), storage.dones[t].view(-1))
            eng.step()
            planes_valid[0] = True
            storage.add_transitions(obs_t, states, act, storage.rewards[t], storage.dones[t], value, logp, mu, sigma)
            if storage.step == NSTEPS:
                ac_.join()                                                   # deferred critic passes: values are read from here on
                with torch.no_grad():
                    last_values = (ac_.critic(obs_clipped.to(pdtype)) if args.library_gemms else
                                   ac_.value(obs_clipped.to(pdtype), obs_planes=(planes_buf, PLANES_SCALE) if use_planes else None)).float()
                storage.compute_returns(last_values, GAMMA, LAM)
                storage.clear()

        def rollout_step_unfused():
            t = storage.step
            cur_obs = obs_clipped                                            # current observation (clamped +-5, vec_task.py:131)
            with torch.no_grad():
                x = cur_obs.to(pdtype)
                mean = ac_.actor(x).float()
                value = ac_.critic(x).float()
                log_std = ac_.log_std.detach().float()
                std = log_std.exp()
                noise = torch.randn_like(mean)
                act = mean + std * noise
                logp = (-0.5 * noise * noise - log_std - half_log_2pi).sum(-1)
            storage.observations[t].copy_(cur_obs)                           # the obs the action was computed from
            actions_buf.copy_(act)
            eng.step()
            storage.add_transitions(storage.observations[t], states, act, rew, reset, value, logp, mean, std.repeat(N, 1))
            if storage.step == NSTEPS:
                with torch.no_grad():
                    last_values = ac_.critic(obs_clipped.to(pdtype)).float()
                storage.compute_returns(last_values, GAMMA, LAM)
                storage.clear()

        rollout_step = rollout_step_unfused if args.unfused else rollout_step_fused
        eng.bind_obs_out(None)
        eng.bind_rollout_out(None, None)
        eng.set_obs_outputs(True, True)
        storage.clear()

        graph, graph_big = None, None
        RPG = max(1, args.rollouts_per_graph)
        for _ in range(NSTEPS):                      # eager warm-up (allocator, rocBLAS handles)
            rollout_step()
        torch.cuda.synchronize()
        if use_graph and not args.no_graph:
            # one graph = one full PPO iteration's rollout (ActorCritic.refresh + 8 steps + GAE): launch-bound inner loop -> hipGraph.
            # The first act of a rollout (storage.step == 0) re-derives the layers' operand planes, scales and bound chain from the
            # parameters on the device (refresh: device work at stable addresses), so it is part of the captured graph and of the
            # timed region: a replay after an optimizer step computes with the updated parameters.
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(NSTEPS):
                        rollout_step()
                if RPG > 1:
                    graph_big = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph_big, stream=side):
                        for _ in range(RPG * NSTEPS):
                            rollout_step()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize()

        K = max(1, K_req)                                   # timed: exactly K steps
        tail_graph = None
        if graph is not None and K % NSTEPS:
            # the K mod 8 steps of the last, unfinished rollout of the timed region as a second graph, so that the figure does not
            # depend on whether K is a multiple of 8 (round 1: 2 replays + 4 eager steps at the driver's --steps 20)
            storage.clear()
            with torch.cuda.stream(side):
                tail_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(tail_graph, stream=side):
                    for _ in range(K % NSTEPS):
                        rollout_step()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize()
            storage.clear()

        def run_steps(k):
            # whole rollouts (8 steps + GAE) as graph replays, then the k mod 8 steps of an unfinished rollout (tail graph when k is
            # the timed K, eagerly otherwise; after a replay the engine's clamped observation row is current, which is what step 0
            # of a rollout starts from)
            if graph is not None:
                if graph_big is not None:
                    for _ in range(k // (RPG * NSTEPS)):
                        graph_big.replay()
                    k_done = (k // (RPG * NSTEPS)) * RPG * NSTEPS
                else:
                    k_done = 0
                for _ in range((k - k_done) // NSTEPS):
                    graph.replay()
                if k % NSTEPS and k == K and tail_graph is not None:
                    tail_graph.replay()
                    storage.step = k % NSTEPS                # (host-side cursor: a replay does not run the Python that advances it)
                    return
                k = k % NSTEPS
            for _ in range(k):
                rollout_step()

        W = W_req if graph is None else ((W_req + NSTEPS - 1) // NSTEPS) * NSTEPS    # warm-up: whole rollouts, at least W steps
        storage.clear()
        run_steps(max(64, NSTEPS))                          # always: 64 untimed steps before the W warm-up steps the caller asked for
        storage.clear()                                     # (clocks, allocator pools, first replays), outside the timed region
        run_steps(W)
        barrier()
        t0 = time.perf_counter()
        run_steps(K)
        barrier()
        elapsed_ = time.perf_counter() - t0
        storage.clear()
        # a second, untimed-for-the-headline pass over the same K steps: `value` is the FIRST timing, as the contract says; the repeat is
        # reported beside it so that a one-off runtime stall inside either (tens of ms, rare: see sim_only.outlier_batches) can be told
        t1 = time.perf_counter()
        run_steps(K)
        barrier()
        repeat_ms.append(1e3 * (time.perf_counter() - t1) / K)
        storage.clear()
        # the step kernel inside the rollout (right after the policy GEMMs the core clock is lowered for ~100 us and the VALU-bound
        # kernel follows it): HIP events around each of 4 x NSTEPS eager rollout steps, outside the timed region
        pairs = []
        plain_step = eng.step

        def probed_step():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            plain_step()
            b.record()
            pairs.append((a, b))
        def robust_mean():
            per_launch = sorted(a.elapsed_time(b) for a, b in pairs)
            typical = per_launch[len(per_launch) // 2]
            kept = [x for x in per_launch if x <= 3.0 * typical]     # (a launch that carries a one-off runtime stall is not the kernel's duration)
            return sum(kept) / len(kept)
        fused_engine = ac_._step_engine
        eng.step = probed_step
        try:
            if fused_engine is not None:
                # with the policy head in the step kernel's prologue the rollout launches a DIFFERENT instantiation (more work per launch):
                # timed here for the record, then the probe proper runs with the head as its own launch so that `roofline` keeps
                # describing the step kernel alone (the instantiation the back-to-back launches and the PMC passes measure)
                for _ in range(2 * NSTEPS):
                    rollout_step()
                torch.cuda.synchronize()
                in_rollout_head_ms.append(robust_mean())
                pairs.clear()
                ac_._step_engine = None
            for _ in range(4 * NSTEPS):
                rollout_step()
        finally:
            eng.step = plain_step
            ac_._step_engine = fused_engine
        torch.cuda.synchronize()
        in_rollout_ms.append(robust_mean())
        # launches of the plain step kernel inside rollout steps of this series: all of them, or -- head fused -- the probe's only
        rollout_counts.append(4 * NSTEPS if fused_engine is not None else NSTEPS + max(64, NSTEPS) + W + 2 * K + 4 * NSTEPS)
        return elapsed_, K, W, graph is not None

    in_rollout_ms, rollout_counts, repeat_ms, in_rollout_head_ms = [], [], [], []

    ac_bf16 = None
    bf_elapsed, bf_K = 0.0, 0
    if args.policy_dtype == "fp32":
        import copy
        ac_bf16 = copy.deepcopy(ac).to(torch.bfloat16)
        bf_elapsed, bf_K, _, _ = measure_rollout(torch.bfloat16, min(args.steps, 128), 16)     # informational series
    else:
        ac_bf16 = ac.to(torch.bfloat16)
    # the same rollout with the hidden layers on the exact-fp32 MFMA kernel (round 2's headline path), beside the split kernel
    ex_elapsed, ex_K, b3_elapsed, b3_K = 0.0, 0, 0.0, 0
    if args.policy_dtype == "fp32" and not args.exact_fp32_layers and not args.library_gemms:
        ex_elapsed, ex_K, _, _ = measure_rollout(torch.float32, min(args.steps, 128), 16, split=False)
        if args.split_format != "bf16x3":
            b3_elapsed, b3_K, _, _ = measure_rollout(torch.float32, min(args.steps, 128), 16, fmt="bf16x3")
    # the same K steps launched eagerly from Python, the way the reference's loop drives the path (ppo.py:126-161: act -> step ->
    # add_transitions per step) -- what a swapped-import PPO.run obtains without capturing anything
    eg_elapsed, eg_K = 0.0, 0
    if not args.no_graph:
        eg_elapsed, eg_K, _, _ = measure_rollout(pdtype, min(args.steps, 512), 16, split=not args.exact_fp32_layers, use_graph=False)
    elapsed, K, W, graphed = measure_rollout(pdtype, args.steps, args.warmup, split=not args.exact_fp32_layers)
    graph = graphed or None
    refresh_us = None
    if pdtype == torch.float32 and not args.library_gemms:
        for _ in range(4):
            ac.refresh()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(32):
            ac.refresh()
        ev1.record()
        torch.cuda.synchronize()
        refresh_us = 1e3 * ev0.elapsed_time(ev1) / 32
    ac.bind_rollout(None, None)                 # (a bound storage at step 0 would make every probe call below refresh the derived buffers first)
    layer_err = policy_layer_errors(torch, ac, obs_clipped) if (args.policy_dtype == "fp32" and not args.library_gemms) else None
    layer_roof = policy_layer_roofline(torch, ac, obs_clipped, args) if (args.policy_dtype == "fp32" and not args.library_gemms) else None
    eng.bind_obs_out(None)
    eng.bind_obs_planes(None)
    eng.bind_rollout_out(None, None)
    eng.set_obs_outputs(True, True)
    kernel_ms_post = time_step_kernel()
    kernel_ms_b2b = 0.5 * (kernel_ms_pre + kernel_ms_post)
    kernel_ms_roll = in_rollout_ms[-1]
    # average over the step-kernel launches of this run (what a kernel trace of the same command averages): back-to-back
    # launches (sim-only series, its warm-up, the timing loops) and launches inside rollout steps (all series)
    n_b2b = 64 + 8 * 16 + sim_steps + 2 * 256
    n_roll = sum(rollout_counts)
    kernel_ms = (n_b2b * kernel_ms_b2b + n_roll * kernel_ms_roll) / (n_b2b + n_roll)
    tmax = torch.tensor([elapsed, sim_wall, kernel_ms, bf_elapsed, kernel_ms_b2b, kernel_ms_roll, ex_elapsed], dtype=torch.float64, device=device)
    if dist is not None:
        if dist.get_backend() == "gloo":
            tmax = tmax.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed, sim_wall, kernel_ms, bf_elapsed, kernel_ms_b2b, kernel_ms_roll, ex_elapsed = [float(x) for x in tmax.tolist()]
    finite = bool(torch.isfinite(obs_clipped).all().item()) and bool(torch.isfinite(rew).all().item())
    resets_seen = int(eng.tensor("reset_count").sum().item())

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(N, args.cpu_steps)

    # which packed layout mms_step launches (csrc/step_kernels.hip: launch_step): 16 envs per 768-thread block once every CU gets one
    force16 = os.environ.get("MMS_STEP_BLOCK16")
    block16 = (force16[0] != "0") if force16 else N >= 16 * torch.cuda.get_device_properties(device).multi_processor_count
    step_kernel_name = "mms::ant_step_kernel<TEN_ANT, 768, 16, 10>" if block16 else "mms::ant_step_kernel<TEN_ANT, 192, 4, 10>"
    if rank == 0:
        value = world * N * K / elapsed
        sim_value = world * N * sim_steps / sim_wall
        achieved = ALGO_BYTES_PER_ENV_STEP * N / (kernel_ms * 1e-3) / 1e9
        tr = measured_traffic()
        traffic = tr["traffic_bytes_per_launch"] if (tr and tr.get("num_envs") == N) else None
        line = {
            "metric": "env-steps/sec (whole node), TenAnt 4096 envs/GPU, PPO rollout",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup, "warmup_effective": W,
            "ms_per_step": 1e3 * elapsed / K, "ms_per_step_repeat": repeat_ms[-1], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (("f32 (policy-layer products: fp32 operands as 2 row-scaled fp16 planes (kept to 2^-22), 3 f16 MFMA products, fp32 accumulation; "
                       if args.split_format == "f16x2" else
                       "f32 (policy-layer products: fp32 operands as 3 exact bf16 planes, 6 bf16 MFMA products, fp32 accumulation; ") +
                      "error vs float64 <= the exact-fp32 MFMA kernel's, see policy_layers_error_vs_f64)") if (not args.exact_fp32_layers and not args.library_gemms and args.policy_dtype == "fp32") else "f32",
            "data": "synthetic",
            "config": {"workload": "TenAnt num_envs=%d per GPU, PPO rollout: ActorCritic MLP [1024,1024,512]x2 (%s) + fused sim step "
                                   "(dt 0.0166, 2 substeps) + RolloutStorage + GAE every %d steps" % (N, args.policy_dtype, NSTEPS),
                       "envs_per_gpu": N, "global_envs": world * N, "parallelism": "env-sharded x%d, no data-path collective" % world,
                       "hipgraph": bool(graph), "refresh_in_graph": bool(graph) and not args.unfused, "rollouts_per_graph": max(1, args.rollouts_per_graph), "fused_act_and_bound_slots": not args.unfused, "obs_planes_from_step_kernel": bool(planes_state["used"]), "policy_head_in_step_kernel": bool(head_state["fused"]), "critic_stream": not args.one_stream, "policy_layers": "library GEMMs" if args.library_gemms else ((("mms_linear_group_act_split16 (2 x fp16 planes, row scales, fp32 accumulate)" if args.split_format == "f16x2" else "mms_linear_group_act_split (3 x bf16 planes, fp32 accumulate)") if (not args.exact_fp32_layers and not args.library_gemms and args.policy_dtype == "fp32") else "mms_linear2_act (exact fp32 MFMA)") + " + mms_ppo_heads_act"),
                       "friction": {"rule": "average" if abs(eng.config.model.boxgnd_mu) > 0 else "min", "gnd_mu": eng.config.model.gnd_mu,
                                    "boxgnd_mu": eng.config.model.boxgnd_mu, "antbox_mu": eng.config.model.antbox_mu,
                                    "note": "this build's modelling choice (PhysX default combine rule), not reference-pinned: DESIGN.md section 4"},
                       "critic_deferred": bool(args.defer_critic), "finite": finite, "resets_total": resets_seen,
                       "distributed": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size()}},
            "sim_only": {"value": sim_value, "unit": "env-steps/s", "steps": sim_steps, "ms_per_step": 1e3 * sim_wall / sim_steps,
                         "value_mean": world * N * sim_steps / sim_wall_mean, "ms_per_step_mean": 1e3 * sim_wall_mean / sim_steps,
                         "note": "engine step with pre-drawn actions (ring of 16); four equal batches: `value` from the median batch, "
                                 "`value_mean` from all four; a batch more than 1.5 x the median is listed in `outlier_batches`",
                         "batch_ms": [1e3 * b for b in sim_batches],
                         "outlier_batches": [i for i, b in enumerate(sim_batches) if b > 1.5 * sorted(sim_batches)[1]],
                         "hipgraph": sim_graph is not None},
            "roofline": {"bound": "hbm", "kernel": step_kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": (tr or {}).get("source"),
                         "traffic_note": "launches of the rollout step (one observation row per env-step); the stand-alone engine "
                                         "configuration of the back-to-back launches also writes its raw + clamped observation "
                                         "buffers: %s bytes per launch" % ((tr or {}).get("engine_buffers_configuration") or {}).get("traffic_bytes_per_launch"),
                         "bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * N, "launch_ms": kernel_ms,
                         "obs_planes_bytes_per_launch": (((obs_dim + 31) // 32) * 128 * N) if planes_state["used"] else 0,
                         "obs_planes_note": "in the rollout the launch also stores the clamped row as the policy layers' operand planes "
                                            "(mms_bind_obs_planes16): work moved INTO this kernel from the policy's split pass, not counted in the "
                                            "algorithmic bytes -- the in-rollout duration includes it",
                         "launch_ms_back_to_back": kernel_ms_b2b, "launch_ms_in_rollout": kernel_ms_roll,
                         "launch_ms_in_rollout_with_policy_head": (in_rollout_head_ms[-1] if (in_rollout_head_ms and head_state["fused"]) else None),
                         "policy_head_note": "with config.policy_head_in_step_kernel the rollout's launches are the <..., false, true> instantiation of the step kernel: "
                                             "the policy's output heads + sampling tail (mms_ppo_heads_act's arithmetic: ~5.4 KB more algorithmic bytes per env-step) in "
                                             "its prologue -- launch_ms_in_rollout_with_policy_head; every other figure of this object is the step kernel alone",
                         "back_to_back_batches_ms": step_kernel_batches,
                         "launches": {"back_to_back": n_b2b, "in_rollout": n_roll},
                         "frac_back_to_back": ALGO_BYTES_PER_ENV_STEP * N / (kernel_ms_b2b * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "launch_note": "HIP events on the launch stream: 2 x 256 back-to-back launches (each as eight batches of 32, median batch; all batch times in back_to_back_batches_ms) and 32 launches inside eager rollout steps "
                                        "(right after the policy GEMMs, when the core clock is lowered: the kernel is VALU-issue bound and follows "
                                        "it); launch_ms = average over all step-kernel launches of this run, the figure a kernel trace of the same "
                                        "command averages"},
            "cpu_baseline": cpu,
        }
        if ex_K:
            line["rollout_exact_fp32_layers"] = {"value": world * N * ex_K / ex_elapsed, "unit": "env-steps/s", "steps": ex_K, "ms_per_step": 1e3 * ex_elapsed / ex_K,
                                                 "note": "the same rollout with the hidden layers on the exact-fp32 MFMA kernel (mms_linear2_act, v_mfma_f32_32x32x2_f32): "
                                                         "round 2's headline path, kept as the A/B of the split kernel"}
        if b3_K:
            line["rollout_bf16x3_layers"] = {"value": world * N * b3_K / b3_elapsed, "unit": "env-steps/s", "steps": b3_K, "ms_per_step": 1e3 * b3_elapsed / b3_K,
                                             "note": "the same rollout with the hidden layers on the three-bf16-plane kernel (mms_linear_group_act_split: every operand exact, "
                                                     "six products): the headline path of this round's first half, kept as an A/B"}
        if eg_K:
            line["rollout_eager"] = {"value": world * N * eg_K / eg_elapsed, "unit": "env-steps/s", "steps": eg_K, "ms_per_step": 1e3 * eg_elapsed / eg_K,
                                     "ratio_to_graph": (eg_elapsed / eg_K) / (elapsed / K),
                                     "note": "the same rollout step launched eagerly from Python, no hipGraph: what the reference's own loop (ppo.py:126-161) "
                                             "obtains with the imports swapped; `value` above is the hipGraph replay of the same launches (INTEGRATION.md: "
                                             "how to capture it)"}
        if refresh_us is not None:
            line["policy_refresh"] = {"us_per_refresh": refresh_us, "per_step_share_us": refresh_us / NSTEPS,
                                      "note": "ActorCritic.refresh(): the hidden layers' operand planes, row scales and bound chain re-derived from the "
                                              "fp32 parameters on the device; runs at the first act of every rollout, INSIDE the captured graph and the "
                                              "timed region (eager back-to-back figure here)"}
        if layer_err is not None:
            line["policy_layers_error_vs_f64"] = layer_err
        if layer_roof is not None:
            line["roofline_policy_layers"] = layer_roof
        if bf_K:
            line["rollout_bf16_policy"] = {"value": world * N * bf_K / bf_elapsed, "unit": "env-steps/s", "steps": bf_K,
                                           "note": "same rollout with the policy MLPs in bf16 (fp32 accumulate); informational, not the headline"}
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()
    eng.close()


def policy_layer_roofline(torch, ac, obs, args):
    """The hidden layers of both networks (the kernels that take two thirds of a rollout step), timed live with HIP events: the
    launches of ActorCritic._fused_hidden back to back on the bench's observation rows -- the observation's split + the three layer
    launches.  achieved = plane-product FLOPs (2 M N K per network and layer x the products the format forms per fp32 product) per
    second, peak = the dense 16-bit MFMA peak of MI355X_MICROARCH.md; the fp32-equivalent rate beside it."""
    obs = obs.detach().clone()
    lins = [m for m in ac.actor if isinstance(m, torch.nn.Linear)][:-1]
    M = obs.shape[0]
    flops32 = 2.0 * 2.0 * M * sum(l.in_features * l.out_features for l in lins)
    split = ac.split_layers and ac._split_applies(M, lins)
    products = (3 if ac.split_format == "f16x2" else 6) if split else 1
    with torch.no_grad():
        for _ in range(8):
            if ac._fused_hidden(obs, obs) is None:
                return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 64
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            ac._fused_hidden(obs, obs)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    # the clock the chip holds inside the layer launches, measured live (mms_layer_clock_probe: workgroup 0 of every two-plane layer
    # launch stores its life in shader cycles and in 100-MHz ticks), in the last of eight back-to-back passes, eight times; and the matrix pipe's share of
    # those cycles: a 32-wide k-step is 48 v_mfma_f32_16x16x32_f16 of 16 cycles for each of a SIMD's two waves on 256-row tiles, 24 on
    # 128-row tiles (one tile per CU at these shapes)
    clock = None
    if split and ac.split_format == "f16x2":
        import ctypes
        from massive_marl_benchmark_amd import _lib
        L = _lib.lib()
        nl = len(lins)
        probe = torch.zeros(nl, 2, dtype=torch.int64, device=obs.device)
        acc = torch.zeros(nl, 2, dtype=torch.float64)
        reps = 8
        with torch.no_grad():
            _lib.check(L.mms_layer_clock_probe(obs.device.index or 0, ctypes.c_void_p(probe.data_ptr()), nl), None, "mms_layer_clock_probe")
            try:
                for _ in range(reps):
                    _lib.check(L.mms_layer_clock_probe(obs.device.index or 0, ctypes.c_void_p(probe.data_ptr()), nl), None, "mms_layer_clock_probe")   # slot 0 = first layer
                    for _ in range(8):                                   # back to back: the slots keep the last pass (the clock of a busy chip)
                        ac._fused_hidden(obs, obs)
                    torch.cuda.synchronize()
                    acc += probe.cpu().double()
            finally:
                L.mms_layer_clock_probe(obs.device.index or 0, None, 0)
        tiles256 = 2 * (M // 256) if M % 256 == 0 else 0
        per_layer = []
        for i, l in enumerate(lins):
            cyc, ticks = float(acc[i, 0]) / reps, float(acc[i, 1]) / reps
            big = tiles256 * (l.out_features // 128) >= 256                      # (launch_linear_split16's choice on a 256-CU part)
            mfma_cycles = ((l.in_features + 31) // 32) * (2 * 48 if big else 2 * 24) * 16
            per_layer.append({"layer": "%d -> %d" % (l.in_features, l.out_features), "clock_ghz": cyc / max(ticks, 1.0) * 0.1, "workgroup0_us": ticks * 0.01,
                              "matrix_pipe_busy_of_cycles": mfma_cycles / max(cyc, 1.0)})
        clock = {"per_layer": per_layer, "note": "workgroup 0 of each launch: s_memtime / s_memrealtime deltas over its life (prologue, k-loop, epilogue), the last of 8 back-to-back passes, %d such runs averaged; "
                                                 "matrix_pipe_busy_of_cycles = the launch's MFMA cycles per SIMD / the workgroup's cycles" % reps}
    peak = 2500.0 if split else 157.3
    achieved = flops32 * products / (ms * 1e-3) / 1e12
    kernel = ("mms::linear_split16_kernel (2 fp16 planes)" if ac.split_format == "f16x2" else "mms::linear_split_kernel (3 bf16 planes)") if split else "mms::linear_act_fast_kernel (fp32 MFMA)"
    return {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "fp32_equivalent_tflops": flops32 / (ms * 1e-3) / 1e12, "frac_of_fp32_mfma_peak": flops32 / (ms * 1e-3) / 1e12 / 157.3,
            "ms_per_pass": ms, "launches_per_pass": len(lins) + (1 if split else 0), "plane_products_per_fp32_product": products, "traffic": None,
            "note": "both networks' hidden layers [%s] at %d rows: observation split + one launch per layer, eager, back to back, HIP events; "
                    "achieved counts every plane product the matrix pipe executes; `peak` is the dense 16-bit figure at 2.4 GHz -- the two-plane "
                    "kernel runs power-limited: `clock_in_kernel` is the clock the chip holds in each launch of this run and the matrix pipe's share "
                    "of the launch's cycles (85 %% of the k-loop's: profiles/r04_split16_kloop_experiments.txt; DESIGN.md 5.10)" % (", ".join(str(l.out_features) for l in lins), M),
            "clock_in_kernel": clock}


def policy_layer_errors(torch, ac, obs):
    """Both networks' last hidden activations (three layers deep) on the bench's own observation rows and weights: the two split
    kernels, the exact-fp32 MFMA kernel and torch's fp32 nn.Linear + nn.ELU (the reference's arithmetic), each against the same layers
    evaluated in float64."""
    import copy
    obs = obs.detach().clone()
    out = {}
    with torch.no_grad():
        ref = [copy.deepcopy(net[:-1]).double()(obs.double()) for net in (ac.actor, ac.critic)]
        scale = float(torch.cat(ref).pow(2).mean().sqrt())
        for name, split, fmt in (("split_2xf16", True, "f16x2"), ("split_3xbf16", True, "bf16x3"), ("exact_fp32_mfma", False, None)):
            keep = (ac.split_layers, ac.split_format)
            ac.split_layers, ac.split_format = split, (fmt or ac.split_format)
            try:
                hid = ac._fused_hidden(obs, obs)
            finally:
                ac.split_layers, ac.split_format = keep
            if hid is None:
                return None
            e = torch.cat([hid[g].double() - ref[g] for g in range(2)])
            out[name] = {"rms": float(e.pow(2).mean().sqrt()) / scale, "max": float(e.abs().max()) / scale, "mean": float(e.mean()) / scale}
        # the reference's own arithmetic on this box: torch's fp32 nn.Linear (F.linear -> the library's fp32 GEMM) + nn.ELU, the modules as
        # the reference builds them (agents/algorithms/rl/ppo/module.py:27-52), on the same rows and weights
        hid = [net[:-1](obs) for net in (ac.actor, ac.critic)]
        e = torch.cat([hid[g].double() - ref[g] for g in range(2)])
        out["torch_fp32_linear"] = {"rms": float(e.pow(2).mean().sqrt()) / scale, "max": float(e.abs().max()) / scale, "mean": float(e.mean()) / scale}
        out["split_2xf16_over_torch_fp32"] = {"rms": out["split_2xf16"]["rms"] / out["torch_fp32_linear"]["rms"],
                                              "max": out["split_2xf16"]["max"] / out["torch_fp32_linear"]["max"]}
    out["unit"] = "rms of the float64 activations (%.4f)" % scale
    out["note"] = ("error of the last hidden layer's activations [2 x %d x %d], three layers deep, against the same layers in float64, on the "
                   "observation rows and weights of this run" % (obs.shape[0], ref[0].shape[1]))
    return out


def cpu_baseline(n_envs, steps):
    """The CPU oracle (a port: the reference's own CPU pipeline is Isaac Gym, absent here) on the same workload:
    TenAnt N envs, sim step with pre-drawn actions, OpenMP over envs on all host cores."""
    import numpy as np
    from oracle.oracle import OracleEngine
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("MMS_CPU_THREADS", min(avail, 16)))   # a 1-GPU box's CPU share is 16 cores
    os.environ["OMP_NUM_THREADS"] = str(cores)                       # read by libgomp when the oracle library loads
    ora = OracleEngine("TenAnt", num_envs=n_envs, seed=0)
    rng = np.random.default_rng(1234)
    ring = [rng.uniform(-1, 1, (n_envs, 80)).astype(np.float32) for _ in range(16)]
    for i in range(4):
        ora.step(ring[i])
    t0 = time.perf_counter()
    ora.step(ring[4])
    one = time.perf_counter() - t0
    if steps <= 0:
        steps = int(max(8, min(400, 15.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for i in range(steps):
        ora.step(ring[i % 16])
    dt = time.perf_counter() - t0
    out = {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
           "sample": "oracle engine (C, OpenMP over envs), TenAnt %d envs x %d sim steps, %.1f s" % (n_envs, steps, dt)}
    # BASELINE configs[0]'s shape beside it ("TenAnt N=4096 and OneAnt N=64"): ~1 s of the same oracle
    one_ant = OracleEngine("OneAnt", num_envs=64, seed=0)
    acts = [rng.uniform(-1, 1, (64, 8)).astype(np.float32) for _ in range(16)]
    for i in range(16):
        one_ant.step(acts[i])
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < 1.0:
        for i in range(16):
            one_ant.step(acts[i])
        n1 += 16
    out["one_ant_64_envs"] = {"value": 64 * n1 / (time.perf_counter() - t0), "unit": "env-steps/s", "steps": n1}
    out["note"] = ("host-dependent: the same 400-step sample took 10.2 s and 13.6 s on two boxes of round 2 (160 K / 120 K env-steps/s) with "
                   "identical code -- the pool's hosts differ and share their cores; compare within one run only")
    # the PRODUCT's CPU build (lib/libmms_cpu.so = the reference's `--sim_device cpu` pipeline of this engine, csrc/cpu/: the kernels'
    # own lane math compiled for the host, OpenMP over envs) on the same workload, beside the oracle port
    try:
        import torch
        from massive_marl_benchmark_amd.engine import Engine
        ce = Engine("TenAnt", num_envs=n_envs, device="cpu", seed=0, clip_obs=5.0)
        acts = [torch.from_numpy(r) for r in ring]
        for i in range(4):
            ce.bind_actions(acts[i])
            ce.step()
        t0 = time.perf_counter()
        ce.bind_actions(acts[4])
        ce.step()
        one = time.perf_counter() - t0
        psteps = int(max(8, min(400, 10.0 / max(one, 1e-3))))
        t0 = time.perf_counter()
        for i in range(psteps):
            ce.bind_actions(acts[i % 16])
            ce.step()
        pdt = time.perf_counter() - t0
        ce.bind_actions(None)
        ce.close()
        out["product_cpu_build"] = {"value": n_envs * psteps / pdt, "unit": "env-steps/s", "cores": cores, "kind": "port",
                                    "sample": "libmms_cpu.so (the engine's own CPU build: device_type=cpu), TenAnt %d envs x %d sim steps, %.1f s"
                                              % (n_envs, psteps, pdt)}
    except Exception as e:                                  # the CPU build is optional for the bench; say why it is missing
        out["product_cpu_build"] = {"value": None, "error": repr(e)}
    return out


if __name__ == "__main__":
    main()
