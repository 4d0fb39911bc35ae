#!/usr/bin/env python3
"""A/B of two builds of the two-plane layer kernel (MMS_LIB selects the library; one process per build): SHA-256 of what the PPO policy
(`ActorCritic.act` hidden layers + value path, 4096 rows, 388 -> 1024 -> 1024 -> 512) and the grouped MAPPO pass (ten agents, 4096 envs)
return on fixed inputs -- a change of the k-loop's schedule must leave every bit alone -- and HIP-event timings of the same calls.

    MMS_LIB=.../libmms_noroll.so python tools/scratch/split16_roll_ab.py ; python tools/scratch/split16_roll_ab.py
"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import marl_modules as mm  # noqa: E402
from massive_marl_benchmark_amd.algorithms.marl.policy_inference import GroupedPolicyInference  # noqa: E402
from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic  # noqa: E402

CFG = {"pi_hid_sizes": [1024, 1024, 512], "vf_hid_sizes": [1024, 1024, 512], "activation": "elu"}


def digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().float().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:16]


def timed(fn, n=64, warm=8):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


def main():
    out = {"lib": os.environ.get("MMS_LIB", "default")}
    N = 4096
    torch.manual_seed(0)
    ac = ActorCritic((388,), (0,), (80,), 0.8, CFG, seed=1).cuda()
    ac.split_min_tiles = 0
    obs = torch.randn(N, 388, device="cuda").clamp(-5, 5)
    small = torch.randn(384, 388, device="cuda").clamp(-5, 5)          # 128-row tiles, three tiles per network and column panel
    with torch.no_grad():
        ha, hc = ac._fused_hidden(obs, obs)
        out["ppo_hidden_sha"] = digest([ha, hc])
        out["ppo_value_sha"] = digest([ac.value(obs)])
        out["ppo_value_small_sha"] = digest([ac.value(small)])
        out["ppo_both_us"] = timed(lambda: ac._fused_hidden(obs, obs))
        out["ppo_value_us"] = timed(lambda: ac.value(obs))
    n = 10
    gen = torch.Generator().manual_seed(5)
    actors, critics = [], []
    for i in range(n):
        torch.manual_seed(i)
        a, c = mm.Actor(46, 8), mm.Critic(388)
        mm.randomize(a, gen)
        mm.randomize(c, gen)
        actors.append(a.cuda())
        critics.append(c.cuda())
    o = [(torch.randn(N, 46, generator=gen) * 2).cuda() for _ in range(n)]
    so = [(torch.randn(N, 388, generator=gen) * 2).cuda() for _ in range(n)]
    inf = GroupedPolicyInference(actors, critics, seed=3)
    with torch.no_grad():
        res = inf.get_actions(so, o, deterministic=True)
        flat = []
        for r in res:
            flat.extend(list(r) if isinstance(r, (list, tuple)) else [r])
        out["marl_sha"] = digest([t for t in flat if torch.is_tensor(t)])
        out["marl_pass_us"] = timed(lambda: inf.get_actions(so, o), n=32, warm=4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
