#!/usr/bin/env python3
"""Sim-only series for the BASELINE.json configurations other than the headline one: fixed pre-drawn U(-1,1)
actions from a ring of 16 tensors, 64 warm-up steps, then K steps replayed from a hipGraph of 16 steps and timed with HIP
events; the step kernel's duration is taken from eager back-to-back launches.  Prints one JSON line per configuration with
the algorithmic bytes per env-step of SURVEY.md 8(d) and the resulting fraction of the 8 TB/s HBM peak.

    python tools/bench_tasks.py [--steps 512]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK = 8.0e12


def algorithmic_bytes(task, A):
    """actions R, persistent state R+W, previous-step caches R+W, obs row W, reward W, reset W, progress R+W."""
    if task == "MultiIngenuity":
        return 24 * 4 + 2 * 4 * (13 + 8) * 4 + 52 * 4 + 4 + 24
    if task == "OneAnt":
        return 32 + 2 * (13 + 16 + 13) * 4 + 2 * (2 + 2 + 1 + 1) * 4 + 60 * 4 + 4 + 8 + 16
    if task == "MultiAntCircle":       # two ants, no box in the reference's scene: actions R, state R+W, caches R+W, 76-wide row W, ...
        return 16 * 4 + 2 * 2 * (13 + 16) * 4 + 2 * 4 * 4 + 76 * 4 + 4 + 8 + 16
    state = A * (13 + 16) + 13
    caches = 4 * A + 2
    return 8 * A * 4 + 2 * state * 4 + 2 * caches * 4 + (38 * A + 8) * 4 + 4 + 8 + 16


def measure(task, N, A, steps, seed=0, env_spacing=None):
    import torch
    from massive_marl_benchmark_amd.engine import Engine
    from massive_marl_benchmark_amd.model import default_cfg

    cfg = None
    if env_spacing is not None:
        cfg = default_cfg(task)
        cfg["env"]["envSpacing"] = float(env_spacing)
    eng = Engine(task, cfg=cfg, num_envs=N, num_agents=A, device=0, seed=seed)
    g = torch.Generator().manual_seed(1234)
    ring = [(torch.rand(N, eng.num_actions, generator=g) * 2 - 1).cuda() for _ in range(16)]
    act = eng.tensor("actions")
    eng.reset_all()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(64):
            act.copy_(ring[i % 16]); eng.step()
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for i in range(16):
                act.copy_(ring[i]); eng.step()
        graph.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(1, steps // 16)
        e0.record(s)
        for _ in range(reps):
            graph.replay()
        e1.record(s); s.synchronize()
        ms_step = e0.elapsed_time(e1) / (reps * 16)
        # the kernel alone: eager launches back to back, no action copies in between
        for _ in range(8):
            eng.step()
        # (eight batches of 32, the median batch: a one-off ~35-ms runtime stall -- DESIGN.md section 6 -- inside a single 256-launch
        #  interval once made this read 319 us "per launch")
        batches = []
        for _ in range(8):
            e0.record(s)
            for _ in range(32):
                eng.step()
            e1.record(s); s.synchronize()
            batches.append(e0.elapsed_time(e1) / 32)
        ms_kernel = sorted(batches)[len(batches) // 2]
    resets = int(eng.tensor("reset_count").sum())
    total_steps = 64 + 16 + reps * 16 + 8 + 256
    finite = bool(torch.isfinite(eng.tensor("obs")).all())
    eng.close()
    b = algorithmic_bytes(task, eng.num_agents)
    return {"task": task, "num_envs": N, "env_spacing": "reference default" if env_spacing is None else env_spacing,
            "resets_per_env_step": resets / float(N * total_steps), "num_agents": eng.num_agents, "obs_dim": eng.obs_dim, "steps": reps * 16,
            "env_steps_per_s": N / (ms_step * 1e-3), "ms_per_step": ms_step, "step_kernel_ms_back_to_back": ms_kernel,
            "algorithmic_bytes_per_env_step": b, "achieved_GBps": N * b / (ms_kernel * 1e-3) / 1e9,
            "frac_of_hbm_peak": N * b / (ms_kernel * 1e-3) / HBM_PEAK, "resets_total": resets, "obs_finite": finite}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    cases = [("MultiIngenuity", 8192, None),        # BASELINE configs[2], the reference's behaviour: goals and positions in the GLOBAL frame
             #                                         (multi_ingenuity.py:381-453), so every env away from the origin resets on every step
             ("MultiIngenuity", 8192, None, 0.0),   # the same with envSpacing 0 (every env at the origin: the helicopters fly; resets by
             #                                         the reference's own rule only) -- VERDICT r3 item 6
             ("OneAnt", 64, None),                  # configs[0]'s shape on the GPU engine
             ("OneAnt", 4096, None),
             ("TenAnt", 4096, None),                # configs[1] (sim-only series, for reference beside bench.py)
             ("MultiAntCircle", 8192, None),        # the task of SURVEY 8(f)3 (intended semantics): generic one-env-per-wave layout
             ("TenAnt", 2048, 100)]                 # configs[4]: 100-ant swarm, 16384 envs over 8 GPUs = 2048 per GPU
    for case in cases:
        task, N, A = case[:3]
        if args.only and args.only != task:
            continue
        print(json.dumps(measure(task, N, A, args.steps, env_spacing=case[3] if len(case) > 3 else None)), flush=True)


if __name__ == "__main__":
    main()
