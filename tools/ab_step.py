#!/usr/bin/env python3
"""A/B timing of the step kernel across builds of the engine library (MMS_LIB) and model options: back-to-back launches of
TenAnt at 4096 envs (the bench's roofline kernel) timed with HIP events, each case in its own process.

    python tools/ab_step.py --libs massive_marl_benchmark_amd/lib/libmms.so /tmp/libmms_b.so [--combine average min] [--task TenAnt]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(task, n, agents, combine, launches, box_mu=None):
    import torch
    from massive_marl_benchmark_amd.engine import Engine
    from massive_marl_benchmark_amd.model import default_cfg
    cfg = default_cfg(task)
    cfg["env"]["frictionCombine"] = combine
    if box_mu is not None:
        cfg["env"]["boxGroundFriction"] = box_mu
    eng = Engine(task, cfg, num_envs=n, num_agents=agents, device=0, seed=0)
    g = torch.Generator().manual_seed(1234)
    ring = [(torch.rand(n, eng.num_actions, generator=g) * 2 - 1).cuda() for _ in range(16)]
    act = eng.tensor("actions")
    for i in range(96):                                   # past the first-step reset, into ordinary walking / falling states
        act.copy_(ring[i % 16]); eng.step()
    torch.cuda.synchronize()
    best, times = None, []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            eng.step()
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / launches * 1e3)
    times.sort()
    print(json.dumps({"lib": os.environ.get("MMS_LIB", "default"), "task": task, "num_envs": n, "combine": combine, "box_ground_mu": float(eng.config.model.boxgnd_mu), "ant_box_mu": float(eng.config.model.antbox_mu),
                      "us_median": round(times[2], 2), "us_min": round(times[0], 2), "us_max": round(times[-1], 2),
                      "finite": bool(torch.isfinite(eng.tensor("obs")).all())}), flush=True)
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=[None])
    ap.add_argument("--combine", nargs="*", default=["average"])
    ap.add_argument("--task", default="TenAnt")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--launches", type=int, default=256)
    ap.add_argument("--box-mu", type=float, default=None, help="cfg env.boxGroundFriction override")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a.task, a.num_envs, a.agents, a.combine[0], a.launches, a.box_mu)
    for lib in a.libs:
        for comb in a.combine:
            env = dict(os.environ)
            if lib:
                env["MMS_LIB"] = os.path.abspath(lib)
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--task", a.task, "--num-envs", str(a.num_envs),
                   "--combine", comb, "--launches", str(a.launches)] + (["--agents", str(a.agents)] if a.agents else []) + \
                  (["--box-mu", str(a.box_mu)] if a.box_mu is not None else [])
            subprocess.run(cmd, env=env, check=False)


if __name__ == "__main__":
    main()
