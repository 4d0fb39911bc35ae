#!/usr/bin/env python3
"""The MAPPO rollout of BASELINE configs[3] on ONE GPU's shard: TenAnt, ten agents, 4096 envs, episode_length 8
(cfg/mappo/config.yaml: hidden 512, layer_N 2), policy inference INCLUDED -- what Runner.run does between two updates
(agents/algorithms/marl/runner.py:128-151: collect, envs.step, insert; then compute): env-steps/s of the whole collection loop.

  reference_way   per-agent torch modules (runner.py:198-227 over actor_critic.py), MultiVecTaskPython.step materialising obs_all /
                  state_all, ten SeparatedReplayBuffer.insert, per-agent bootstrap values, ten compute_returns -- on this engine
  fused           GroupedPolicyInference.collect_into (all twenty networks per stage, reads and writes the rollout slots in place),
                  SharedRolloutBuffers (share_obs once, written by the step kernel), one GAE launch; eager and as ONE hipGraph

    python tools/bench_mappo_rollout.py [--num-envs 4096] [--iters 16]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=16, help="timed rollouts of T = 8 steps")
    ap.add_argument("--agents", type=int, default=10, help="ants per env: 100 with --num-envs 2048 is BASELINE configs[4]'s per-GPU shard")
    ap.add_argument("--skip-reference-way", action="store_true")
    ap.add_argument("--exact-fp32-layers", action="store_true", help="A/B: the layers on the exact-fp32 MFMA kernel instead of a split kernel")
    ap.add_argument("--split-format", default="f16x2", choices=["f16x2", "bf16x3"], help="planes of the split layers: two scaled fp16 (default) or three exact bf16")
    args = ap.parse_args()
    import torch
    import marl_modules as mm
    from massive_marl_benchmark_amd.algorithms.marl.policy_inference import GroupedPolicyInference
    from massive_marl_benchmark_amd.algorithms.marl.utils.separated_buffer import SeparatedReplayBuffer
    from massive_marl_benchmark_amd.algorithms.marl.utils.shared_buffer import SharedRolloutBuffers
    from massive_marl_benchmark_amd.model import default_cfg
    from massive_marl_benchmark_amd.tasks.agent_base.multi_vec_task import MultiVecTaskPython
    from massive_marl_benchmark_amd.tasks.ten_ant import TenAnt

    n, T, A = args.num_envs, 8, args.agents
    OBS, SHARE = 46, 38 * A + 8
    conf = dict(episode_length=T, n_rollout_threads=n, hidden_size=512, recurrent_N=1, gamma=0.99, gae_lambda=0.95, use_gae=True,
                use_popart=False, use_valuenorm=False, use_proper_time_limits=False)

    def make_env():
        cfg = default_cfg("TenAnt")
        cfg["env"]["numEnvs"] = n
        cfg["clip_observations"] = 7.0
        cfg["seed"] = 3
        return MultiVecTaskPython(TenAnt(cfg, None, "physx", "cuda", 0, True, is_multi_agent=True, num_ants=A), "cuda:0")

    gen = torch.Generator().manual_seed(5)
    actors, critics = [], []
    for i in range(A):
        torch.manual_seed(i)
        a, c = mm.Actor(OBS, 8), mm.Critic(SHARE)
        mm.randomize(a, gen, 0.05)
        mm.randomize(c, gen, 0.05)
        actors.append(a.cuda())
        critics.append(c.cuda())

    def timed(rollout, iters):
        for _ in range(3):
            rollout()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            rollout()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (iters * T)
        return {"ms_per_env_step": ms, "env_steps_per_s": n / (ms * 1e-3), "agent_steps_per_s": n * A / (ms * 1e-3)}

    out = {"task": "TenAnt", "algo": "mappo", "num_envs": n, "agents": A, "T": T, "hidden": 512, "layer_N": 2, "obs": OBS, "share_obs": SHARE}

    # ---- the reference's way, on this engine ------------------------------------------------------------------------------------
    if not args.skip_reference_way:
      env = make_env()
      bufs = [SeparatedReplayBuffer(conf, env.observation_space[k], env.share_observation_space[k], env.action_space[k], "cuda:0") for k in range(A)]
      obs, share, _ = env.reset()
      for k in range(A):
          bufs[k].share_obs[0].copy_(share[:, k]); bufs[k].obs[0].copy_(obs[:, k])
      rnn = torch.zeros(n, 1, 512, device="cuda")

      def reference_way():
          with torch.no_grad():
              for t in range(T):
                  vals, acts, lps = [], [], []
                  for k in range(A):                                                    # runner.py:198-227
                      mean, std, value = mm.torch_forward(actors[k], critics[k], bufs[k].obs[t], bufs[k].share_obs[t])
                      dist = torch.distributions.Normal(mean, std)
                      act = dist.sample()
                      vals.append(value); acts.append(act); lps.append(dist.log_prob(act))
                  obs, share, rew, dones, _, _ = env.step(acts)
                  dones_env = torch.all(dones != 0, dim=1)
                  masks = torch.ones(n, A, 1, device="cuda")
                  masks[dones_env] = 0
                  for k in range(A):
                      bufs[k].insert(share[:, k], obs[:, k], rnn, rnn, acts[k], lps[k], vals[k], rew[:, k], masks[:, k])
              for k in range(A):                                                        # runner.py:229-241 (compute)
                  _, _, nxt = mm.torch_forward(actors[k], critics[k], bufs[k].obs[-1], bufs[k].share_obs[-1])
                  bufs[k].compute_returns(nxt, None)
                  bufs[k].after_update()
      out["reference_way_eager"] = timed(reference_way, max(2, args.iters // 4))
      env.task.engine.close()
      del bufs

    # ---- fused --------------------------------------------------------------------------------------------------------------------
    env = make_env()
    sh = SharedRolloutBuffers(conf, env, "cuda:0")
    sh.warmup()
    inf = GroupedPolicyInference(actors, critics, seed=3, split_layers=not args.exact_fp32_layers, split_format=args.split_format)
    out["policy_layers"] = "mms_linear_group_act (exact fp32 MFMA)" if args.exact_fp32_layers else (
        "mms_linear_group_act_split16 (2 x fp16 planes, row scales, fp32 accumulate)" if args.split_format == "f16x2" else "mms_linear_group_act_split (3 x bf16 planes, fp32 accumulate)")
    nxt = torch.zeros(n, A, device="cuda")

    def fused():
        for t in range(T):
            s = sh.step
            actions = inf.collect_into(sh)
            rew, dones = sh.env_step(actions)
            sh.insert_step(rew, dones, sh.value_preds[s], sh.actions[s], sh.action_log_probs[s])
        inf.values_into(sh, nxt)
        sh.compute_returns(nxt, None)
        sh.after_update()
    out["fused_eager"] = timed(fused, args.iters)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fused()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            fused()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    out["fused_graph"] = timed(graph.replay, args.iters)
    # A/B: without the unconditional refresh of the derived copies at step 0 of every rollout (GroupedPolicyInference.refresh_every_rollout:
    # what follows a trainer that updates through `.data`, hatrpo_trainer.py:122) -- round 3's figure was measured this way
    inf.refresh_every_rollout = False
    with torch.cuda.stream(side):
        fused()
        graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph2, stream=side):
            fused()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    out["fused_graph_no_refresh"] = timed(graph2.replay, args.iters)
    out["fused_eager_no_refresh"] = timed(fused, args.iters)
    inf.refresh_every_rollout = True
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    inf.refresh(); torch.cuda.synchronize()
    e0.record()
    for _ in range(4):
        inf.refresh()
    e1.record(); torch.cuda.synchronize()
    out["refresh_ms_eager"] = e0.elapsed_time(e1) / 4
    out["finite"] = bool(torch.isfinite(sh.returns).all()) and bool(torch.isfinite(sh.share_obs).all())
    # roofline of the collection step's dominant work, the policy layers: fp32-equivalent FLOPs per env step (T collect passes + one
    # critics-only bootstrap pass per rollout) over the step time, against the pipe they run on
    H, obs_w, sobs_w = 512, 46, 38 * A + 8
    f_act = 2.0 * n * A * (obs_w * H + 2 * H * H + H * 8)
    f_cri = 2.0 * n * A * (sobs_w * H + 2 * H * H + H)
    flops_step = f_act + f_cri + f_cri / T
    tf = flops_step / (out["fused_graph"]["ms_per_env_step"] * 1e-3) / 1e12
    peak = 157.3 if args.exact_fp32_layers else 2500.0 / (3.0 if args.split_format == "f16x2" else 6.0)
    out["roofline"] = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "frac_of_fp32_mfma_peak": tf / 157.3,
                       "note": "fp32-equivalent layer FLOPs per env step / whole env-step time (inference + env step + buffers + GAE); peak = the fp32 MFMA "
                               "peak for the exact kernel, the 16-bit dense peak / 3 (fp16 planes) or / 6 (bf16 planes) plane products for the split kernels"}
    if "reference_way_eager" in out:
        out["speedup_eager"] = out["reference_way_eager"]["ms_per_env_step"] / out["fused_eager"]["ms_per_env_step"]
        out["speedup_graph_vs_reference_eager"] = out["reference_way_eager"]["ms_per_env_step"] / out["fused_graph"]["ms_per_env_step"]
    env.task.engine.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
