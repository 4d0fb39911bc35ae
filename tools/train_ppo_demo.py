#!/usr/bin/env python3
"""End-to-end sanity run: PPO (the update rule of agents/algorithms/rl/ppo/ppo.py:244-320 with cfg/ppo/config.yaml's
hyper-parameters) on the engine through the drop-in pieces -- VecTaskPython, ActorCritic, RolloutStorage.

Not part of the product path (the learner is a caller, out of scope); it answers two questions about the engine's own
physics model that no fixture can: does a policy learn on it, and does it stay finite when a policy searches for exploits?

    python tools/train_ppo_demo.py --task OneAnt --num-envs 4096 --iterations 300
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="OneAnt", choices=["OneAnt", "TenAnt"])
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--iterations", type=int, default=300)
    ap.add_argument("--hidden", type=int, nargs="+", default=[256, 128, 64])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--nsteps", type=int, default=8)          # cfg/ppo/config.yaml: nsteps 8
    ap.add_argument("--gamma", type=float, default=0.96)      # cfg/ppo/config.yaml: gamma 0.96
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--fixed-lr", action="store_true", help="no adaptive KL schedule")
    ap.add_argument("--value-coef", type=float, default=1.0)
    ap.add_argument("--exact-fp32-layers", action="store_true",
                    help="hidden layers on the exact-fp32 MFMA kernel instead of the split kernel (which applies when batch and hidden widths are multiples of 128)")
    ap.add_argument("--split-format", default="f16x2", choices=["f16x2", "bf16x3"], help="planes of the split layer kernel (module default: f16x2)")
    ap.add_argument("--obs-planes", action="store_true",
                    help="the step kernel also writes the observation's operand planes (Engine.bind_obs_planes) and act() reads them instead of splitting the rows")
    ap.add_argument("--friction-combine", default="average", choices=["average", "min"],
                    help="cfg env.frictionCombine: PhysX's average rule (default) or min = a box that is frictionless against everything")
    ap.add_argument("--split-min-tiles", type=int, default=None, help="ActorCritic.split_min_tiles (0: the split layers whatever the batch size)")
    return ap.parse_args(argv)


def train(args, log=print):
    """The loop; returns {"reward_per_step": [one mean per iteration], "episodes": [(sum of returns, count) of the episodes that ended in
    each iteration], "value_error": [per iteration, E(return - value)^2 / var(return) of the rollout as collected], "ac", "env", "obs", "states"} (tests/test_gpu_parity.py drives it)."""
    import torch
    from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic
    from massive_marl_benchmark_amd.algorithms.rl.ppo.storage import RolloutStorage
    from massive_marl_benchmark_amd.model import default_cfg
    from massive_marl_benchmark_amd.tasks.agent_base.vec_task import VecTaskPython
    from massive_marl_benchmark_amd.tasks.one_ant import OneAnt
    from massive_marl_benchmark_amd.tasks.ten_ant import TenAnt

    torch.manual_seed(args.seed)
    cfg = default_cfg(args.task)
    cfg["env"]["numEnvs"] = args.num_envs
    cfg["seed"] = args.seed
    cfg["clip_observations"] = 5.0
    cfg["env"]["frictionCombine"] = args.friction_combine
    task = {"OneAnt": OneAnt, "TenAnt": TenAnt}[args.task](cfg, None, "physx", "cuda", 0, True)
    env = VecTaskPython(task, "cuda:0", 5.0, 1.0)
    N, obs_dim, act_dim = env.num_envs, env.observation_space.shape[0], env.action_space.shape[0]
    dev = torch.device("cuda:0")
    # cfg/ppo/config.yaml: nsteps 8, 5 epochs x 4 minibatches, clip 0.2, lr 3e-4 adaptive on KL 0.016, gamma 0.96, lam 0.95
    T, EPOCHS, MINIB, CLIP, GAMMA, LAM, DESIRED_KL, MAX_GRAD = args.nsteps, 5, 4, 0.2, args.gamma, 0.95, 0.016, 1.0
    ac = ActorCritic((obs_dim,), (0,), (act_dim,), 0.8, {"pi_hid_sizes": args.hidden, "vf_hid_sizes": args.hidden, "activation": "elu"},
                     seed=args.seed).to(dev)
    ac.split_layers = not args.exact_fp32_layers
    ac.split_format = args.split_format
    if args.split_min_tiles is not None:
        ac.split_min_tiles = args.split_min_tiles
    history, episodes, vloss_hist = [], [], []
    storage = RolloutStorage(N, T, (obs_dim,), (0,), (act_dim,), device=str(dev))
    opt = torch.optim.Adam(ac.parameters(), lr=args.lr)
    lr = args.lr
    states = torch.zeros(N, 0, device=dev)
    obs = env.reset().clone()
    ep_ret = torch.zeros(N, device=dev)
    ep_len = torch.zeros(N, device=dev)
    done_ret, done_len, done_cnt = 0.0, 0.0, 0
    run_ret, run_cnt = 0.0, 0
    t0 = time.time()
    log("task %s, %d envs, obs %d, actions %d, hidden %s" % (args.task, N, obs_dim, act_dim, args.hidden))
    planes, planes_ok = None, False
    if args.obs_planes:
        planes = torch.empty(N * ((obs_dim + 31) // 32) * 128, dtype=torch.uint8, device=dev)
        env.task.engine.bind_obs_planes(planes, 2048.0)                 # clip_observations 5 x 2^11 <= 2^14
    for it in range(args.iterations):
        for _ in range(T):
            # (`obs` holds a copy of the row the engine clamped in its last step: the planes it wrote beside that row are its planes)
            actions, logp, values, mu, sigma = ac.act(obs, states, obs_planes=(planes, 2048.0) if planes_ok else None)
            next_obs, rew, dones, _ = env.step(actions)
            planes_ok = planes is not None
            storage.add_transitions(obs, states, actions, rew, dones, values, logp, mu, sigma)
            obs.copy_(next_obs)
            ep_ret += rew
            ep_len += 1
            fin = dones > 0
            n_fin = int(fin.sum())
            if n_fin:
                done_ret += float(ep_ret[fin].sum()); done_len += float(ep_len[fin].sum()); done_cnt += n_fin
                run_ret += float(ep_ret[fin].sum()); run_cnt += n_fin
                ep_ret[fin] = 0
                ep_len[fin] = 0
        with torch.no_grad():
            last_values = ac.critic(obs)
        mean_step_reward = float(storage.rewards.mean())
        history.append(mean_step_reward)
        episodes.append((run_ret, run_cnt))
        run_ret, run_cnt = 0.0, 0
        storage.compute_returns(last_values, GAMMA, LAM)
        flat = lambda x: x.view(-1, *x.shape[2:])
        B = N * T
        with torch.no_grad():                                  # how well the critic the rollout ran with explains the returns it produced
            ret_ = storage.returns.view(-1)
            vloss_hist.append(float((ret_ - storage.values.view(-1)).pow(2).mean() / ret_.var().clamp_min(1e-8)))
        for _ in range(EPOCHS):
            perm = torch.arange(B, device=dev)                 # 'sequential' sampler (cfg/ppo/config.yaml sampler)
            for idx in perm.chunk(MINIB):
                lp, _, v, mu_b, sg_b = ac.evaluate(flat(storage.observations)[idx], None, flat(storage.actions)[idx])
                old_mu, old_sg = flat(storage.mu)[idx], flat(storage.sigma)[idx]
                with torch.no_grad():                          # ppo.py:267-279 (sigma = log_std, as the reference stores it)
                    kl = torch.sum(sg_b - old_sg + (torch.square(old_sg.exp()) + torch.square(old_mu - mu_b)) /
                                   (2.0 * torch.square(sg_b.exp())) - 0.5, dim=-1).mean()
                    if args.fixed_lr:
                        pass
                    elif kl > DESIRED_KL * 2.0:
                        lr = max(1e-5, lr / 1.5)
                    elif DESIRED_KL / 2.0 > kl > 0.0:
                        lr = min(1e-2, lr * 1.5)
                    for g in opt.param_groups:
                        g["lr"] = lr
                adv = flat(storage.advantages)[idx].squeeze(-1)
                ratio = torch.exp(lp - flat(storage.actions_log_prob)[idx].squeeze(-1))
                surrogate = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - CLIP, 1.0 + CLIP)).mean()
                value_loss = (flat(storage.returns)[idx] - v).pow(2).mean()
                loss = surrogate + args.value_coef * value_loss
                opt.zero_grad()
                loss.backward()
                torch.nn.utils.clip_grad_norm_(ac.parameters(), MAX_GRAD)
                opt.step()
        storage.clear()
        if (it + 1) % args.log_every == 0 or it == 0:
            torch.cuda.synchronize()
            finite = bool(torch.isfinite(obs).all())
            root = env.task.engine.tensor("root_states")
            log("it %4d  reward/step %8.3f  episodes %6d  mean return %9.2f  mean length %6.1f  lr %.1e  std %.2f  max|v| %.1f  finite %s  %.0f env-steps/s"
                % (it + 1, mean_step_reward, done_cnt, done_ret / max(done_cnt, 1), done_len / max(done_cnt, 1), lr,
                   float(ac.log_std.detach().exp().mean()), float(root[:, 7:10].abs().max()), finite, (it + 1) * T * N / (time.time() - t0)))
            done_ret, done_len, done_cnt = 0.0, 0.0, 0
            if not finite:
                raise RuntimeError("non-finite observation")
    return {"reward_per_step": history, "episodes": episodes, "value_error": vloss_hist, "ac": ac, "env": env, "obs": obs, "states": states}


def main():
    train(parse(), log=lambda m: print(m, flush=True))


if __name__ == "__main__":
    main()
