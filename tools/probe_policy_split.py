#!/usr/bin/env python3
"""Where the PPO policy's time goes, by launch shape (sizing of "actor per step + critic once per rollout"):
HIP-event timings, back to back, of

  both        the hidden layers of BOTH networks on 4096 rows (what `act` launches per step: one launch per layer, 256 tiles)
  actor_only  the actor's hidden layers alone on 4096 rows (design (b)'s per-step part)
  value       the critic alone on 4096 rows + its 1-wide head (the rollout's bootstrap pass)
  critic_9x   the critic alone on 9 x 4096 rows + head (design (b)'s per-rollout part: 8 stored observations + the bootstrap row)
  heads       mms_ppo_heads_act (both last layers + sampling + stores)

Environment A/Bs are read by the library at its first launch, so each variant is its own process:
  MMS_SPLIT_MT=2|4   force the 128- / 256-row tiling of the split layers        MMS_HEAD_RT=1|2   16 / 32 rows per heads block
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic  # noqa: E402
from massive_marl_benchmark_amd.algorithms.rl.ppo.storage import RolloutStorage  # noqa: E402

CFG = {"pi_hid_sizes": [1024, 1024, 512], "vf_hid_sizes": [1024, 1024, 512], "activation": "elu"}


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
    N = 4096
    torch.manual_seed(0)
    ac = ActorCritic((388,), (0,), (80,), 0.8, CFG, seed=1).cuda()
    ac.split_min_tiles = 0
    obs = torch.randn(N, 388, device="cuda").clamp(-5, 5)
    obs9 = torch.randn(9 * N, 388, device="cuda").clamp(-5, 5)
    states = torch.zeros(N, 0, device="cuda")
    nets = ac._networks()
    out = {"env": {k: os.environ.get(k) for k in ("MMS_SPLIT_MT", "MMS_HEAD_RT")}}
    with torch.no_grad():
        out["both_us"] = timed(lambda: ac._fused_hidden(obs, obs))
        out["actor_only_us"] = timed(lambda: ac._split_hidden([nets[0]], [obs], "probe_actor"))
        out["value_us"] = timed(lambda: ac.value(obs))
        out["critic_9x_us"] = timed(lambda: ac.value(obs9), n=16, warm=4)
        storage = RolloutStorage(N, 8, (388,), (0,), (80,), device="cuda")
        actions = torch.zeros(N, 80, device="cuda")
        ac.bind_rollout(storage, actions)
        ha, hc = ac._fused_hidden(obs, obs)
        storage.step = 1                                            # (not 0: no refresh inside the timed call)
        out["heads_us"] = timed(lambda: ac._sample(None, None, hidden=ha, vhidden=hc))
        storage.step = 0
        ac.bind_rollout(None, None)
        # per step: (b) = actor_only + critic_9x / 8 against (now) = both + value / 8
        out["per_step_now_us"] = out["both_us"] + out["value_us"] / 8
        out["per_step_design_b_us"] = out["actor_only_us"] + out["critic_9x_us"] / 8
    print(json.dumps(out))


if __name__ == "__main__":
    main()
