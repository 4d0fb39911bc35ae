#!/usr/bin/env python3
"""BASELINE configs[2]: MultiIngenuity, 8192 envs, the DDPG / TD3 collection loop (ddpg.py:151-159) --
MLPActorCritic.act (cfg/ddpg/config.yaml: 3 x 256, ReLU, tanh output; exploration noise module.py:54-61, drawn on the device) +
engine step + ReplayBuffer.add_transitions.  The learner's update is a caller and not part of the measurement.

Two variants of the same loop: `copies` (the wrapper's return values copied into the ring, as the reference does) and
`bound` (the step kernel writes next_obs / reward / done into the ring row `ReplayBuffer.slot()` names).  Each is timed
eagerly and as a replayed hipGraph of one pass over a 16-row window of the ring.

    python tools/bench_offpolicy_collect.py [--num-envs 8192] [--steps 512]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="MultiIngenuity")
    ap.add_argument("--num-envs", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--replay-size", type=int, default=1024)
    ap.add_argument("--library-actor", action="store_true", help="A/B: the actor's layers as library GEMMs + activation passes")
    ap.add_argument("--env-spacing", type=float, default=None,
                    help="override env.envSpacing (0: every env at the origin, the helicopters fly; default: the reference's grid, where every env away "
                         "from the origin resets on every step -- positions and goals are global-frame, multi_ingenuity.py:381-453)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from massive_marl_benchmark_amd import spaces
    from massive_marl_benchmark_amd.algorithms.rl.ddpg.module import MLPActorCritic
    from massive_marl_benchmark_amd.algorithms.rl.ddpg.storage import ReplayBuffer
    from massive_marl_benchmark_amd.engine import Engine

    N = args.num_envs
    torch.manual_seed(0)
    out = {"task": args.task, "num_envs": N, "replay_size": args.replay_size, "actor": "library" if args.library_actor else "mms_linear2_act",
           "env_spacing": "reference default" if args.env_spacing is None else args.env_spacing}
    from massive_marl_benchmark_amd.model import default_cfg
    cfg = None
    if args.env_spacing is not None:
        cfg = default_cfg(args.task)
        cfg["env"]["envSpacing"] = float(args.env_spacing)
    for variant in ("copies", "bound"):
        eng = Engine(args.task, cfg=cfg, num_envs=N, device=0, seed=0, clip_obs=5.0)
        W, AD = eng.obs_dim, eng.num_actions
        ac = MLPActorCritic(spaces.Box(-np.inf * np.ones(W), np.inf * np.ones(W)), spaces.Box(-np.ones(AD), np.ones(AD)), 0.1, "cuda:0",
                            hidden_sizes=[256, 256, 256]).cuda()                       # cfg/ddpg/config.yaml: hidden_nodes 256 x 3
        if args.library_actor:
            ac.pi.forward = lambda obs, _pi=ac.pi: _pi.act_limit * _pi.pi(obs)
        buf = ReplayBuffer(N, args.replay_size, 64, 8, (W,), (0,), (AD,), "cuda:0")
        states = torch.zeros(N, 0, device="cuda")
        act_buf, rew, done, obs_c = eng.tensor("actions"), eng.tensor("rew"), eng.tensor("reset"), eng.tensor("obs_clipped")
        eng.reset_all()
        eng.step()
        cur = obs_c.clone()
        if variant == "bound":
            eng.set_obs_outputs(raw=False, clipped=False)

        def step():
            a = ac.act(cur, deterministic=False)                                   # act_noise 0.1, act_limit 1
            k = buf.slot()
            if variant == "bound":
                eng.bind_obs_out(buf.next_observations[k])
                eng.bind_rollout_out(buf.rewards[k].view(-1), buf.dones[k].view(-1))
                act_buf.copy_(a)
                eng.step()
                buf.add_transitions(cur, states, a, buf.rewards[k], buf.next_observations[k], buf.dones[k])
                cur.copy_(buf.next_observations[k])
            else:
                act_buf.copy_(a)
                eng.step()
                buf.add_transitions(cur, states, a, rew, obs_c, done)
                cur.copy_(obs_c)

        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(64):
                step()
            s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(args.steps):
                step()
            e1.record(s); s.synchronize()
            eager_ms = e0.elapsed_time(e1) / args.steps
            # one pass over a 16-row window as a graph (the window's addresses are baked in; a learner that wants the whole ring
            # captures replay_size / 16 such graphs or replays eagerly)
            buf.step = 16
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                for _ in range(16):
                    step()
            buf.step = 16
            graph.replay(); s.synchronize()
            reps = max(1, args.steps // 16)
            e0.record(s)
            for _ in range(reps):
                graph.replay()
            e1.record(s); s.synchronize()
            graph_ms = e0.elapsed_time(e1) / (reps * 16)
        finite = bool(torch.isfinite(buf.next_observations[:32]).all())
        n_steps = 1 + 64 + args.steps + 16 + 16 + reps * 16
        out[variant] = {"resets_per_env_step": int(eng.tensor("reset_count").sum()) / float(N * n_steps),"eager_ms_per_step": eager_ms, "eager_env_steps_per_s": N / (eager_ms * 1e-3),
                        "graph_ms_per_step": graph_ms, "graph_env_steps_per_s": N / (graph_ms * 1e-3), "finite": finite}
        eng.bind_obs_out(None); eng.bind_rollout_out(None, None)
        del graph
        eng.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
