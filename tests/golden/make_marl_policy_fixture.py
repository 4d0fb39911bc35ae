#!/usr/bin/env python3
"""Golden vectors for the grouped MAPPO / HAPPO policy inference: the REFERENCE's own Actor and Critic
(agents/algorithms/marl/actor_critic.py, imported in place from /root/reference, CPU) with perturbed parameters, three agents of
TenAnt's shapes (obs 46, share_obs 388, 8 actions; hidden 64, layer_N 2), twelve rows.  Stored per agent: both state_dicts, the
inputs, the deterministic action (= the mean) with its log-probability, the value, and evaluate_actions' log-probability of given
actions.  Runs in the build container only; writes tests/golden/marl_policy_fixture.npz (plain arrays).

    python tests/golden/make_marl_policy_fixture.py
"""
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import run_reference_learners as rrl          # the import scaffolding (name-only gym / isaacgym / tensorboard stand-ins)
from marl_modules import randomize


def main():
    if not os.path.isdir(rrl.REF):
        sys.exit("reference tree not present")
    rrl.setup_imports()
    ac = rrl.load("agents.algorithms.marl.actor_critic", "agents/algorithms/marl/actor_critic.py")
    from gym import spaces                   # the name-only stand-in: Box with .shape
    conf = yaml.safe_load(open(os.path.join(rrl.REF, "cfg", "mappo", "config.yaml")))
    conf.update(hidden_size=64, algorithm_name="mappo")
    obs_space = spaces.Box(low=-np.inf, high=np.inf, shape=(46,))
    sobs_space = spaces.Box(low=-np.inf, high=np.inf, shape=(388,))
    act_space = spaces.Box(low=-1.0, high=1.0, shape=(8,))
    gen = torch.Generator().manual_seed(20261004)
    out = {"agents": np.array(3), "hidden": np.array(64), "layer_N": np.array(conf["layer_N"])}
    M = 12
    for i in range(3):
        torch.manual_seed(100 + i)
        actor = ac.Actor(conf, obs_space, act_space, torch.device("cpu"))
        critic = ac.Critic(conf, sobs_space, torch.device("cpu"))
        randomize(actor, gen)
        randomize(critic, gen)
        obs = torch.randn(M, 46, generator=gen) * 2.0
        sobs = torch.randn(M, 388, generator=gen) * 2.0
        given = torch.randn(M, 8, generator=gen)
        rnn = torch.zeros(M, 1, 64)
        masks = torch.ones(M, 1)
        with torch.no_grad():
            mean, mean_logp, _ = actor(obs, rnn, masks, None, deterministic=True)
            value, _ = critic(sobs, rnn, masks)
            given_logp, entropy = actor.evaluate_actions(obs, rnn, given, masks)
            dist = actor.act.action_out(actor.base(obs))
        for k, v in actor.state_dict().items():
            out["a%d.actor.%s" % (i, k)] = v.numpy()
        for k, v in critic.state_dict().items():
            out["a%d.critic.%s" % (i, k)] = v.numpy()
        out.update({"a%d.obs" % i: obs.numpy(), "a%d.share_obs" % i: sobs.numpy(), "a%d.mean" % i: mean.numpy(), "a%d.mean_logp" % i: mean_logp.numpy(),
                    "a%d.std" % i: dist.stddev[0].numpy(), "a%d.value" % i: value.numpy(), "a%d.given" % i: given.numpy(),
                    "a%d.given_logp" % i: given_logp.numpy()})
    np.savez_compressed(os.path.join(HERE, "marl_policy_fixture.npz"), **out)
    print("wrote marl_policy_fixture.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
