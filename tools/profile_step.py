#!/usr/bin/env python3
"""Profiling driver: the sim-only loop of bench.py (TenAnt, 4096 envs, pre-drawn actions) and nothing else, so that
rocprofv3 traces / PMC passes see the step kernel in isolation.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/profile_step.py
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/profile_step.py --steps 64
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="TenAnt")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--substeps", type=int, default=0, help="override cfg sim.substeps (dt unchanged); 0 = the task's default")
    ap.add_argument("--rollout-outputs", action="store_true",
                    help="write what bench.py's rollout step writes: ONE clamped observation row per env-step (into a bound "
                         "rollout slot) plus the reward / done slots, instead of the engine's raw + clamped observation buffers")
    args = ap.parse_args()
    import torch
    from massive_marl_benchmark_amd.engine import Engine
    cfg = None
    if args.substeps:
        from massive_marl_benchmark_amd.model import default_cfg
        cfg = default_cfg(args.task)
        cfg["sim"]["substeps"] = args.substeps
    eng = Engine(args.task, cfg=cfg, num_envs=args.num_envs, device=0, seed=0)
    g = torch.Generator().manual_seed(1234)
    ring = [(torch.rand(args.num_envs, eng.num_actions, generator=g) * 2 - 1).cuda() for _ in range(16)]
    act = eng.tensor("actions")
    if args.rollout_outputs:
        slots = torch.empty(8, args.num_envs, eng.obs_dim, device="cuda")
        rew, done = torch.empty(8, args.num_envs, device="cuda"), torch.empty(8, args.num_envs, dtype=torch.uint8, device="cuda")
        eng.set_obs_outputs(raw=False, clipped=False)
    for i in range(args.warmup + args.steps):
        act.copy_(ring[i % 16])
        if args.rollout_outputs:
            eng.bind_obs_out(slots[i % 8])
            eng.bind_rollout_out(rew[i % 8], done[i % 8])
        eng.step()
    torch.cuda.synchronize()
    print("resets", int(eng.tensor("reset_count").sum()), "finite", bool(torch.isfinite(eng.tensor("obs")).all()))
    eng.close()


if __name__ == "__main__":
    main()
