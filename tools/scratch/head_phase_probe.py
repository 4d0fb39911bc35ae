#!/usr/bin/env python3
"""Phases of the PPO heads + sampling body (csrc/head_block.h) for block 0 / wave 0, a -DMMS_HEAD_STAMP=1 build: us from entry until
the critic's dot products are done / operands loaded + MFMAs issued / first barrier passed / reduction + second barrier / sampling done;
stand-alone kernel (mms_ppo_heads_act) and fused into the step kernel (mms_bind_policy_head)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic
from massive_marl_benchmark_amd.algorithms.rl.ppo.storage import RolloutStorage
from massive_marl_benchmark_amd.engine import Engine
CFG = {"pi_hid_sizes": [1024, 1024, 512], "vf_hid_sizes": [1024, 1024, 512], "activation": "elu"}
N = 4096
torch.manual_seed(0)
ac = ActorCritic((388,), (0,), (80,), 0.8, CFG, seed=1).cuda()
obs = torch.randn(N, 388, device="cuda").clamp(-5, 5)
states = torch.zeros(N, 0, device="cuda")
storage = RolloutStorage(N, 8, (388,), (0,), (80,), device="cuda")
with torch.no_grad():
    for fused in (False, True):
        eng = Engine("TenAnt", num_envs=N, device=0, seed=0)
        actions = eng.tensor("actions")
        ac.bind_rollout(storage, actions, step_engine=eng if fused else None)
        res = []
        for it in range(12):
            storage.step = 1 + it % 6
            ac.act(obs, states)
            eng.step()
            torch.cuda.synchronize()
            res.append([round(float(v), 2) for v in storage.sigma[storage.step][0, 1:6].tolist()])
        print("fused" if fused else "stand-alone", res[-3:])
        ac.bind_rollout(None, None)
        eng.close()
