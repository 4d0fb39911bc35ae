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
    inf = GroupedPolicyInference(actors, critics, seed=3)                                  # split-operand layers (two scaled fp16 planes), all folds
    inf3 = GroupedPolicyInference(actors, critics, seed=3, split_format="bf16x3")          # ... on three exact bf16 planes
    inf32 = GroupedPolicyInference(actors, critics, seed=3, split_layers=False)           # the same folds on the exact-fp32 MFMA layer kernel

    def grouped():
        return inf.get_actions(sobs, obs)

    def grouped_fp32():
        return inf32.get_actions(sobs, obs)

    def grouped_bf16x3():
        return inf3.get_actions(sobs, obs)

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
    res["grouped_bf16x3_graph_ms"] = timeit(graphed(grouped_bf16x3), args.iters)
    res["grouped_exact_fp32_graph_ms"] = timeit(graphed(grouped_fp32), args.iters)
    res["per_agent_torch_graph_ms"] = timeit(graphed(lambda: per_agent(False)), args.iters)     # (Normal's argument validation synchronises: not capturable as the reference writes it)
    flops = 2.0 * M * n * ((48 * 512 + 2 * 512 * 512 + 512 * 8) + (388 * 512 + 2 * 512 * 512 + 512))
    res["gemm_flops"] = flops
    res["grouped_graph_tflops"] = flops / (res["grouped_graph_ms"] * 1e-3) / 1e12
    # roofline of the pass: the layers' FLOPs (fp32 products) against the matrix pipe they run on -- the 16-bit pipe's dense 2.5 PFLOP/s
    # carries three plane products per fp32 product (two fp16 planes; six with three bf16 planes): 833.3 (416.7) TFLOP/s of
    # fp32-equivalent work; the exact-fp32 MFMA peak (157.3) beside it
    res["roofline"] = {"bound": "mfma", "achieved": res["grouped_graph_tflops"], "peak": 2500.0 / 3.0, "unit": "TFLOP/s", "frac": res["grouped_graph_tflops"] / (2500.0 / 3.0),
                       "frac_of_fp32_mfma_peak": res["grouped_graph_tflops"] / 157.3,
                       "bf16x3_series": {"achieved": flops / (res["grouped_bf16x3_graph_ms"] * 1e-3) / 1e12, "peak": 2500.0 / 6.0,
                                         "frac": flops / (res["grouped_bf16x3_graph_ms"] * 1e-3) / 1e12 / (2500.0 / 6.0)},
                       "exact_fp32_series": {"achieved": flops / (res["grouped_exact_fp32_graph_ms"] * 1e-3) / 1e12, "peak": 157.3,
                                             "frac": flops / (res["grouped_exact_fp32_graph_ms"] * 1e-3) / 1e12 / 157.3},
                       "note": "fp32-equivalent FLOPs of the 60 hidden-layer GEMMs + heads; split layers = 2 row-scaled fp16 planes, 3 f16 MFMA products, fp32 accumulate "
                               "(peak at 2.4 GHz; the two-plane kernel runs power-limited at ~1.55 GHz: DESIGN.md 5.10, profiles/r04_split16_kloop_experiments.txt)"}
    # error of both paths against float64 (deterministic means / values of agent 0)
    import copy
    with torch.no_grad():
        a64, c64 = copy.deepcopy(actors[0]).double(), copy.deepcopy(critics[0]).double()
        mean64, _, value64 = mm.torch_forward(a64, c64, obs[0].double(), sobs[0].double())
        err = {}
        for name, obj in (("split_2xf16", inf), ("split_3xbf16", inf3), ("exact_fp32_mfma", inf32)):
            v, m, _ = obj.get_actions(sobs, obs, deterministic=True)
            err[name] = {"mean_max": float((m[0].double() - mean64).abs().max()), "value_max": float((v[0].double() - value64).abs().max())}
    res["error_vs_f64"] = err
    res["speedup_graph"] = res["per_agent_torch_graph_ms"] / res["grouped_graph_ms"]
    res["speedup_eager"] = res["per_agent_torch_eager_ms"] / res["grouped_eager_ms"]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
