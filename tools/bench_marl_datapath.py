#!/usr/bin/env python3
"""The MAPPO / HAPPO data path around the env step, TenAnt at 4096 envs, without the
ten agents' networks (callers): pre-drawn actions / values / log-probs, T = 8 env steps, then compute_returns + after_update.

  separated   the reference's way (runner.py:128-255): MultiVecTaskPython.step materialises obs_all / state_all, then ten
              SeparatedReplayBuffer.insert calls copy the same 388-wide row ten times, ten compute_returns
  shared      SharedRolloutBuffers: the step kernel writes share_obs[t+1] in place (once), one gather kernel for obs[t+1],
              one GAE launch for all agents

    python tools/bench_marl_datapath.py [--num-envs 4096] [--iters 32]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=32, help="timed rollouts of T = 8 steps")
    args = ap.parse_args()
    import torch
    from massive_marl_benchmark_amd.algorithms.marl.utils.separated_buffer import SeparatedReplayBuffer
    from massive_marl_benchmark_amd.algorithms.marl.utils.shared_buffer import SharedRolloutBuffers
    from massive_marl_benchmark_amd.model import default_cfg
    from massive_marl_benchmark_amd.tasks.agent_base.multi_vec_task import MultiVecTaskPython
    from massive_marl_benchmark_amd.tasks.ten_ant import TenAnt

    n, T, A = args.num_envs, 8, 10
    conf = dict(episode_length=T, n_rollout_threads=n, hidden_size=64, recurrent_N=1, gamma=0.99, gae_lambda=0.95, use_gae=True,
                use_popart=True, use_valuenorm=False, use_proper_time_limits=False)       # cfg/mappo/config.yaml

    def make_env():
        cfg = default_cfg("TenAnt")
        cfg["env"]["numEnvs"] = n
        cfg["clip_observations"] = 7.0
        cfg["seed"] = 3
        return MultiVecTaskPython(TenAnt(cfg, None, "physx", "cuda", 0, True, is_multi_agent=True), "cuda:0")

    class Norm:
        def __init__(self, k):
            self.m, self.v = torch.tensor([0.3 * k], device="cuda"), torch.tensor([1.0 + 0.5 * k], device="cuda")

        def running_mean_var(self):
            return self.m, self.v

    norms = [Norm(k) for k in range(A)]
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = [[torch.rand(n, 8, generator=g, device="cuda") * 2 - 1 for _ in range(A)] for _ in range(T)]
    vals = [torch.randn(n, A, generator=g, device="cuda") for _ in range(T + 1)]
    logp = [[torch.randn(n, 8, generator=g, device="cuda") for _ in range(A)] for _ in range(T)]
    rnn = torch.zeros(n, 1, 64, device="cuda")
    out = {"task": "TenAnt", "num_envs": n, "agents": A, "T": T}

    def timed(rollout):
        for _ in range(3):
            rollout()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            rollout()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (args.iters * T)
        return {"ms_per_env_step": ms, "env_steps_per_s": n / (ms * 1e-3)}

    # ---- the reference's way ----
    env = make_env()
    bufs = [SeparatedReplayBuffer(conf, env.observation_space[k], env.share_observation_space[k], env.action_space[k], "cuda:0") for k in range(A)]
    obs, share, _ = env.reset()
    for k in range(A):
        bufs[k].share_obs[0].copy_(share[:, k]); bufs[k].obs[0].copy_(obs[:, k])

    def separated():
        for t in range(T):
            obs, share, rew, dones, _, _ = env.step(acts[t])
            dones_env = torch.all(dones != 0, dim=1)
            masks = torch.ones(n, A, 1, device="cuda")
            masks[dones_env] = 0
            for k in range(A):
                bufs[k].insert(share[:, k], obs[:, k], rnn, rnn, acts[t][k], logp[t][k], vals[t][:, k:k + 1], rew[:, k], masks[:, k])
        for k in range(A):
            bufs[k].compute_returns(vals[T][:, k:k + 1], norms[k])
            bufs[k].after_update()
    out["separated"] = timed(separated)
    out["separated"]["buffer_bytes"] = sum(t.numel() * t.element_size() for b in bufs for t in vars(b).values() if torch.is_tensor(t))
    env.task.engine.close()
    del bufs

    # ---- rollout-buffer fusion ----
    env = make_env()
    sh = SharedRolloutBuffers(conf, env, "cuda:0")
    sh.warmup()

    def shared():
        for t in range(T):
            rew, dones = sh.env_step(acts[t])
            sh.insert_step(rew, dones, vals[t], acts[t], logp[t])
        sh.compute_returns(vals[T], norms)
        sh.after_update()
    out["shared"] = timed(shared)
    out["shared"]["buffer_bytes"] = sum(t.numel() * t.element_size() for t in vars(sh).values() if torch.is_tensor(t))
    env.task.engine.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
