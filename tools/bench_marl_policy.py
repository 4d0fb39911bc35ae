#!/usr/bin/env python3
"""MAPPO policy inference of one collect step, TenAnt shapes (ten agents, obs 46 / share_obs 388 / 8 actions, hidden 512, layer_N 2,
cfg/mappo/config.yaml) at 4096 envs: the grouped operators (algorithms/marl/policy_inference.py: ten launches for all twenty
networks) against the reference's way -- agent by agent, torch modules (runner.py:186-216 -> actor_critic.py:43-69, 137-155; the
fp32 torch statement of tests/marl_modules.py: LayerNorm, Linear, ELU, ..., Normal sample, log_prob), eager and as a hipGraph.

    python tools/bench_marl_policy.py [--num-envs 4096] [--iters 50]
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
    ap.add_argument("--agents", type=int, default=10)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    import torch
    import marl_modules as mm
    from massive_marl_benchmark_amd.algorithms.marl.policy_inference import GroupedPolicyInference
    n, M = args.agents, args.num_envs
    gen = torch.Generator().manual_seed(5)
    actors, critics = [], []
    for i in range(n):
        torch.manual_seed(i)
        a, c = mm.Actor(46, 8), mm.Critic(388)
        mm.randomize(a, gen)
        mm.randomize(c, gen)
        actors.append(a.cuda())
        critics.append(c.cuda())
    obs = [(torch.randn(M, 46, generator=gen) * 2).cuda() for _ in range(n)]
    sobs = [(torch.randn(M, 388, generator=gen) * 2).cuda() for _ in range(n)]
    inf = GroupedPolicyInference(actors, critics, seed=3)

    def grouped():
        return inf.get_actions(sobs, obs)

    def per_agent(validate=True):
        out = []
        for i in range(n):
            mean, std, value = mm.torch_forward(actors[i], critics[i], obs[i], sobs[i])
            dist = torch.distributions.Normal(mean, std, validate_args=validate)
            act = dist.sample() if validate else mean + std * torch.randn_like(mean)      # (torch.normal is not capturable here either)
            out.append((value, act, dist.log_prob(act)))
        return out

    def timeit(f, iters):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def graphed(f):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                f()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        return g.replay

    res = {"agents": n, "num_envs": M, "hidden": 512, "layer_N": 2}
    res["grouped_eager_ms"] = timeit(grouped, args.iters)
    res["per_agent_torch_eager_ms"] = timeit(per_agent, args.iters)
    res["grouped_graph_ms"] = timeit(graphed(grouped), args.iters)
    res["per_agent_torch_graph_ms"] = timeit(graphed(lambda: per_agent(False)), args.iters)     # (Normal's argument validation synchronises: not capturable as the reference writes it)
    flops = 2.0 * M * n * ((48 * 512 + 2 * 512 * 512 + 512 * 8) + (388 * 512 + 2 * 512 * 512 + 512))
    res["gemm_flops"] = flops
    res["grouped_graph_tflops"] = flops / (res["grouped_graph_ms"] * 1e-3) / 1e12
    res["speedup_graph"] = res["per_agent_torch_graph_ms"] / res["grouped_graph_ms"]
    res["speedup_eager"] = res["per_agent_torch_eager_ms"] / res["grouped_eager_ms"]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
