#!/usr/bin/env python3
"""Drop-in proof: the REFERENCE's own learners, imported in place from /root/reference and left unmodified, driven over this
build's VecTaskPython / MultiVecTaskPython on the CPU build of the engine (device_type="cpu" -- the reference's `--sim_device cpu`
pipeline, agents/tasks/agent_base/base_task.py:27-32; there is no GPU in the build container).

  1. agents/algorithms/rl/ppo/ppo.py `PPO.run` (:99-161 rollout, :163-175 returns + update) for 2 iterations on OneAnt, 64 envs
     (BASELINE configs[0]) -- with the reference's own RolloutStorage / ActorCritic, then again with this build's drop-in classes
     (massive_marl_benchmark_amd.algorithms.rl.ppo) patched into the reference module: same loop, zero edits.
  2. agents/algorithms/marl/runner.py `Runner.run` (:114-151) with algorithm_name = mappo for 2 episodes on TenAnt, 16 envs, ten
     agents -- the reference's policies, trainers and SeparatedReplayBuffers; then with this build's SeparatedReplayBuffer patched in;
     then with GroupedPolicyInference (algorithms/marl/policy_inference.py) as the Runner's collect step on top of that.
  3. the same `Runner.run` with cfg/happo and cfg/hatrpo (BASELINE configs[4]'s learner): the sequential update of
     runner.py:266-316 -- `update_factor`, the per-agent probability-ratio product -- with the reference's buffers, then with
     this build's.
  4. agents/algorithms/rl/ddpg/ddpg.py `DDPG.run` (:116-204) and agents/algorithms/rl/td3/td3.py `TD3.run` on MultiIngenuity, 64 envs
     (BASELINE configs[2]'s learners) over VecTaskPython -- with the reference's ReplayBuffer / MLPActorCritic, then with this build's.

Runs only where the reference tree exists.  Nothing of the reference is copied: modules are imported from where they lie, with the
name-only stand-ins of tests/golden/_isaacgym_stub for `gym` / `isaacgym` and a name-only `torch.utils.tensorboard.SummaryWriter`
(tensorboard is not installed).  Writes tests/golden/reference_learners_dropin.log.

    python tests/golden/run_reference_learners.py
"""
import contextlib
import importlib.util
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MMS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
LOG = []


def say(*a):
    line = " ".join(str(x) for x in a)
    LOG.append(line)
    print(line, flush=True)


def setup_imports():
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    sys.path.insert(0, os.path.join(HERE, "_isaacgym_stub"))             # name-only gym.spaces / isaacgym
    for name, sub in (("agents", "agents"), ("agents.algorithms", "agents/algorithms"), ("agents.algorithms.rl", "agents/algorithms/rl"),
                      ("agents.algorithms.rl.ppo", "agents/algorithms/rl/ppo"), ("agents.algorithms.rl.ddpg", "agents/algorithms/rl/ddpg"),
                      ("agents.algorithms.rl.td3", "agents/algorithms/rl/td3"), ("agents.algorithms.marl", "agents/algorithms/marl"),
                      ("agents.algorithms.marl.utils", "agents/algorithms/marl/utils"), ("agents.algorithms.utils", "agents/algorithms/utils"),
                      ("agents.utils", "agents/utils")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]                            # namespace only: the packages' __init__ (isaacgym, tensorboard) never run
        sys.modules[name] = m
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:                                                 # name-only stand-in: swallows every call
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    torch.utils.tensorboard = tb


def load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def run_ppo(tmp):
    from massive_marl_benchmark_amd.model import default_cfg
    from massive_marl_benchmark_amd.tasks.agent_base.vec_task import VecTaskPython
    from massive_marl_benchmark_amd.tasks.one_ant import OneAnt
    pkg = sys.modules["agents.algorithms.rl.ppo"]
    ref_storage = load("agents.algorithms.rl.ppo.storage", "agents/algorithms/rl/ppo/storage.py")
    ref_module = load("agents.algorithms.rl.ppo.module", "agents/algorithms/rl/ppo/module.py")
    pkg.RolloutStorage, pkg.ActorCritic = ref_storage.RolloutStorage, ref_module.ActorCritic
    ppo = load("agents.algorithms.rl.ppo.ppo", "agents/algorithms/rl/ppo/ppo.py")
    cfg_train = yaml.safe_load(open(os.path.join(REF, "cfg", "ppo", "config.yaml")))
    cfg_train["policy"]["pi_hid_sizes"] = cfg_train["policy"]["vf_hid_sizes"] = [64, 64]      # small networks: a plumbing run
    import massive_marl_benchmark_amd.algorithms.rl.ppo.module as our_module
    import massive_marl_benchmark_amd.algorithms.rl.ppo.storage as our_storage
    for label, storage_cls, module_cls in (("reference RolloutStorage + ActorCritic", ref_storage.RolloutStorage, ref_module.ActorCritic),
                                           ("this build's RolloutStorage + ActorCritic", our_storage.RolloutStorage, our_module.ActorCritic)):
        ppo.RolloutStorage, ppo.ActorCritic = storage_cls, module_cls
        cfg = default_cfg("OneAnt")
        cfg["env"]["numEnvs"] = 64
        cfg["seed"] = 1
        task = OneAnt(cfg, None, "physx", "cpu", 0, True)
        env = VecTaskPython(task, "cpu", 5.0, 1.0)
        torch.manual_seed(1)
        os.makedirs(os.path.join(tmp, "ppo"), exist_ok=True)            # (the real SummaryWriter creates its log_dir; PPO.save writes there)
        learner = ppo.PPO(vec_env=env, cfg_train=cfg_train, device="cpu", sampler=cfg_train["learn"].get("sampler", "sequential"),
                          log_dir=os.path.join(tmp, "ppo"), is_testing=False, print_log=True, apply_reset=False, asymmetric=False)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            learner.run(num_learning_iterations=2, log_interval=1)
        out = buf.getvalue()
        lines = [l.strip() for l in out.splitlines() if any(k in l for k in ("Learning iteration", "Mean reward", "Value function loss", "Mean episode length"))]
        assert "Learning iteration 1/2" in out, out[-2000:]
        assert bool(torch.isfinite(learner.storage.returns).all()) and bool(torch.isfinite(task.obs_buf).all())
        say("PPO.run (reference agents/algorithms/rl/ppo/ppo.py, unmodified) x 2 iterations on OneAnt 64 envs, CPU build, %s: ok" % label)
        for l in lines[-4:]:
            say("    " + " ".join(l.split()))
        task.engine.close()


def run_marl(tmp, algo="mappo"):
    from massive_marl_benchmark_amd.model import default_cfg
    from massive_marl_benchmark_amd.tasks.agent_base.multi_vec_task import MultiVecTaskPython
    from massive_marl_benchmark_amd.tasks.ten_ant import TenAnt
    runner = sys.modules.get("agents.algorithms.marl.runner") or load("agents.algorithms.marl.runner", "agents/algorithms/marl/runner.py")
    ref_buffer = sys.modules["agents.algorithms.marl.utils.separated_buffer"].SeparatedReplayBuffer
    from massive_marl_benchmark_amd.algorithms.marl.utils.separated_buffer import SeparatedReplayBuffer as OurBuffer
    conf = yaml.safe_load(open(os.path.join(REF, "cfg", algo, "config.yaml")))
    assert conf["algorithm_name"] == algo
    n = 16
    conf.update(n_rollout_threads=n, num_env_steps=2 * conf["episode_length"] * n, hidden_size=64, run_dir=os.path.join(tmp, "marl"),
                log_interval=1, save_interval=1000)
    from massive_marl_benchmark_amd.algorithms.marl.policy_inference import GroupedPolicyInference
    cases = [("reference SeparatedReplayBuffer", ref_buffer, False), ("this build's SeparatedReplayBuffer", OurBuffer, False)]
    if algo == "mappo":
        cases.append(("this build's SeparatedReplayBuffer + GroupedPolicyInference as Runner.collect", OurBuffer, True))
    for label, buf_cls, grouped in cases:
        runner.SeparatedReplayBuffer = buf_cls
        cfg = default_cfg("TenAnt")
        cfg["env"]["numEnvs"] = n
        cfg["clip_observations"] = 7.0
        cfg["seed"] = 1
        task = TenAnt(cfg, None, "physx", "cpu", 0, True, is_multi_agent=True)
        env = MultiVecTaskPython(task, "cpu")
        torch.manual_seed(1)
        r = runner.Runner(vec_env=env, config=dict(conf), model_dir="")
        if grouped:       # all twenty networks of a collect step in grouped launches, reading the reference's own Actor / Critic modules in place
            inf = GroupedPolicyInference.from_trainers(r.trainer, seed=1)
            r.collect = lambda step, inf=inf, r=r: inf.collect(r.buffer, step)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            r.run()
        out = buf.getvalue()
        assert "updates 1/2 episodes" in out, out[-2000:]
        assert all(bool(torch.isfinite(b.returns).all()) for b in r.buffer)
        if algo != "mappo":     # the sequential update went through update_factor: after the first agent the factor is a ratio product, not ones
            assert all(b.factor is not None and bool(torch.isfinite(torch.as_tensor(b.factor)).all()) for b in r.buffer)
        say("Runner.run (reference agents/algorithms/marl/runner.py, %s, unmodified) x 2 episodes on TenAnt %d envs x 10 agents, CPU build, %s: ok"
            % (algo, n, label))
        for l in [l.strip() for l in out.splitlines() if "updates" in l][-2:]:
            say("    " + l)
        task.engine.close()


def run_offpolicy(tmp, algo):
    """DDPG.run / TD3.run (agents/algorithms/rl/ddpg/ddpg.py:116-204, td3/td3.py:117-) on MultiIngenuity over VecTaskPython."""
    from massive_marl_benchmark_amd.model import default_cfg
    from massive_marl_benchmark_amd.tasks.agent_base.vec_task import VecTaskPython
    from massive_marl_benchmark_amd.tasks.multi_ingenuity import MultiIngenuity
    import importlib
    pkg = sys.modules["agents.algorithms.rl.%s" % algo]
    ref_storage = load("agents.algorithms.rl.%s.storage" % algo, "agents/algorithms/rl/%s/storage.py" % algo)
    ref_module = load("agents.algorithms.rl.%s.module" % algo, "agents/algorithms/rl/%s/module.py" % algo)
    pkg.ReplayBuffer, pkg.MLPActorCritic = ref_storage.ReplayBuffer, ref_module.MLPActorCritic     # what the package __init__ would export
    learner_mod = load("agents.algorithms.rl.%s.%s" % (algo, algo), "agents/algorithms/rl/%s/%s.py" % (algo, algo))
    our_storage = importlib.import_module("massive_marl_benchmark_amd.algorithms.rl.%s.storage" % algo)
    our_module = importlib.import_module("massive_marl_benchmark_amd.algorithms.rl.%s.module" % algo)
    cfg_train = yaml.safe_load(open(os.path.join(REF, "cfg", algo, "config.yaml")))
    cfg_train["learn"].update(hidden_nodes=64, replay_size=64, batch_size=8)          # small: a plumbing run that does reach update()
    cls = getattr(learner_mod, algo.upper())
    for label, storage_cls, module_cls in (("reference ReplayBuffer + MLPActorCritic", ref_storage.ReplayBuffer, ref_module.MLPActorCritic),
                                           ("this build's ReplayBuffer + MLPActorCritic", our_storage.ReplayBuffer, our_module.MLPActorCritic)):
        learner_mod.ReplayBuffer, learner_mod.MLPActorCritic = storage_cls, module_cls
        cfg = default_cfg("MultiIngenuity")
        cfg["env"]["numEnvs"] = 64
        cfg["seed"] = 1
        task = MultiIngenuity(cfg, None, "physx", "cpu", 0, True)
        env = VecTaskPython(task, "cpu", cfg_train["clip_observations"], cfg_train["clip_actions"])
        torch.manual_seed(1)
        os.makedirs(os.path.join(tmp, algo), exist_ok=True)
        learner = cls(vec_env=env, cfg_train=cfg_train, device="cpu", sampler="random", log_dir=os.path.join(tmp, algo),
                      is_testing=False, print_log=True, apply_reset=False, asymmetric=False)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            learner.run(num_learning_iterations=3, log_interval=1)
        out = buf.getvalue()
        assert "Learning iteration 2/3" in out, out[-2000:]
        assert learner.warm_up is False and bool(torch.isfinite(task.obs_buf).all())
        assert all(bool(torch.isfinite(p).all()) for p in learner.actor_critic.parameters())
        say("%s.run (reference agents/algorithms/rl/%s/%s.py, unmodified) x 3 iterations on MultiIngenuity 64 envs, CPU build, %s: ok"
            % (algo.upper(), algo, algo, label))
        lines = [l.strip() for l in out.splitlines() if any(k in l for k in ("Learning iteration", "Value function loss", "Surrogate loss", "Mean reward/step"))]
        for l in lines[-4:]:
            say("    " + " ".join(l.split()))
        task.engine.close()


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: this script runs in the build container only")
    setup_imports()
    say("# generated by tests/golden/run_reference_learners.py; torch %s" % torch.__version__)
    with tempfile.TemporaryDirectory() as tmp:
        run_ppo(tmp)
        run_marl(tmp, "mappo")
        run_marl(tmp, "happo")
        run_marl(tmp, "hatrpo")
        run_offpolicy(tmp, "ddpg")
        run_offpolicy(tmp, "td3")
    with open(os.environ.get("MMS_DROPIN_LOG", os.path.join(HERE, "reference_learners_dropin.log")), "w") as f:
        f.write("\n".join(LOG) + "\n")


if __name__ == "__main__":
    main()
