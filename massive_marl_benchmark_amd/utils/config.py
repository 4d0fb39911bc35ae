"""Isaac-Gym-free equivalents of agents/utils/config.py: get_args (:216), load_cfg (:90), retrieve_cfg (:62),
parse_sim_params (:181), set_seed (:35).  Same flags and YAML files; `--cfg_env` / `--cfg_train` may point at
the user's own (e.g. the reference's) cfg directory.  With no YAML given the built-in defaults equal the
reference's cfg/<Task>.yaml (tests/test_host_logic.py checks that when the reference tree is present)."""
import argparse
import os
import random

import numpy as np
import torch
import yaml

from ..model import default_cfg

TASKS = ("TenAnt", "OneAnt", "MultiIngenuity")
MARL_ALGOS = ("mappo", "happo", "hatrpo", "maddpg", "ippo")


def set_np_formatting():
    np.set_printoptions(edgeitems=30, infstr='inf', linewidth=4000, nanstr='nan', precision=2, suppress=False,
                        threshold=10000, formatter=None)


def set_seed(seed, torch_deterministic=False):
    if seed == -1 and torch_deterministic:
        seed = 42
    elif seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    torch.cuda.manual_seed_all(seed)
    if torch_deterministic:
        torch.backends.cudnn.benchmark = False
        torch.backends.cudnn.deterministic = True
        torch.use_deterministic_algorithms(True)          # config.py:54 calls the removed torch.set_deterministic
    return seed


def retrieve_cfg(args, use_rlg_config=False):
    if args.task not in TASKS:
        raise ValueError("Unrecognized task %r; choose from %s" % (args.task, ", ".join(TASKS)))
    return os.path.join(args.logdir, "{}/{}/{}".format(args.task, args.algo, args.algo)), \
        "cfg/{}/config.yaml".format(args.algo), "cfg/{}.yaml".format(args.task)


def load_cfg(args, use_rlg_config=False):
    """config.py:90-178: task YAML + train YAML with the CLI overrides."""
    if args.cfg_env and os.path.exists(args.cfg_env):
        with open(os.path.join(os.getcwd(), args.cfg_env), 'r') as f:
            cfg = yaml.load(f, Loader=yaml.SafeLoader)
    else:
        cfg = default_cfg(args.task)
    if args.cfg_train and os.path.exists(args.cfg_train):
        with open(os.path.join(os.getcwd(), args.cfg_train), 'r') as f:
            cfg_train = yaml.load(f, Loader=yaml.SafeLoader)
    else:
        cfg_train = default_train_cfg(args.algo)
    if args.num_envs > 0:
        cfg["env"]["numEnvs"] = args.num_envs
    if args.episode_length > 0:
        cfg["env"]["episodeLength"] = args.episode_length
    cfg["name"] = args.task
    cfg["headless"] = args.headless
    if "task" in cfg:
        if "randomize" not in cfg["task"]:
            cfg["task"]["randomize"] = args.randomize
        else:
            cfg["task"]["randomize"] = args.randomize or cfg["task"]["randomize"]
    else:
        cfg["task"] = {"randomize": False}
    logdir = args.logdir
    if args.torch_deterministic:
        cfg_train["torch_deterministic"] = True
    if args.seed is not None and "seed" in cfg_train or True:
        cfg_train["seed"] = args.seed if args.seed is not None else cfg_train.get("seed", -1)
    log_id = args.logdir + "_{}".format(args.experiment)
    if args.metadata:
        log_id = args.logdir + "_{}_{}_{}".format(args.task, args.algo, args.experiment)
    logdir = os.path.realpath(log_id)
    return cfg, cfg_train, logdir


def default_train_cfg(algo):
    """cfg/ppo/config.yaml values (the only train config the hot path reads: nsteps, gamma, lam, clips)."""
    return {"seed": -1, "clip_observations": 5.0, "clip_actions": 1.0,
            "policy": {"pi_hid_sizes": [1024, 1024, 512], "vf_hid_sizes": [1024, 1024, 512], "activation": "elu"},
            "learn": {"agent_name": "shadow_hand", "test": False, "resume": 0, "save_interval": 1000, "print_log": True,
                      "max_iterations": 6500, "cliprange": 0.2, "ent_coef": 0, "nsteps": 8, "noptepochs": 5,
                      "nminibatches": 4, "max_grad_norm": 1, "optim_stepsize": 3.e-4, "schedule": "adaptive",
                      "desired_kl": 0.016, "gamma": 0.96, "lam": 0.95, "init_noise_std": 0.8, "log_interval": 1,
                      "asymmetric": False}}


class SimParams:
    """The subset of gymapi.SimParams the tasks read (config.py:181-213)."""

    def __init__(self):
        self.dt = 1. / 60.
        self.substeps = 2
        self.up_axis = 2
        self.gravity = [0.0, 0.0, -9.81]
        self.use_gpu_pipeline = True
        self.physx = {}


def parse_sim_params(args, cfg, cfg_train=None):
    sp = SimParams()
    sp.dt = 1. / 60.
    if "sim" in cfg:
        for k, v in cfg["sim"].items():                    # gymutil.parse_sim_config (config.py:206-207)
            setattr(sp, k, v)
    return sp


def get_args(argv=None, benchmark=False, use_rlg_config=False):
    """The flags of config.py:216-321 plus the gymutil ones it consumes (sim_device, pipeline, ...)."""
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    p.add_argument("--physx", action="store_true")
    p.add_argument("--flex", action="store_true")
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--slices", type=int, default=0)
    p.add_argument("--test", action="store_true", default=False)
    p.add_argument("--play", action="store_true", default=False)
    p.add_argument("--resume", type=int, default=0)
    p.add_argument("--checkpoint", type=str, default="Base")
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--task", type=str, default="TenAnt")
    p.add_argument("--task_type", type=str, default="Python")
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--logdir", type=str, default="logs/")
    p.add_argument("--experiment", type=str, default="Base")
    p.add_argument("--metadata", action="store_true", default=False)
    p.add_argument("--cfg_train", type=str, default="Base")
    p.add_argument("--cfg_env", type=str, default="Base")
    p.add_argument("--num_envs", type=int, default=0)
    p.add_argument("--episode_length", type=int, default=0)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int, default=0)
    p.add_argument("--steps_num", type=int, default=-1)
    p.add_argument("--minibatch_size", type=int, default=-1)
    p.add_argument("--randomize", action="store_true", default=False)
    p.add_argument("--torch_deterministic", action="store_true", default=False)
    p.add_argument("--algo", type=str, default="ppo")
    p.add_argument("--model_dir", type=str, default="")
    args = p.parse_args(argv)
    dev = args.sim_device.split(":")
    args.sim_device_type = dev[0]
    args.compute_device_id = int(dev[1]) if len(dev) > 1 else 0
    args.device_id = args.compute_device_id
    args.device = args.sim_device_type if args.pipeline in ("gpu", "cuda") else "cpu"
    args.use_gpu_pipeline = args.pipeline in ("gpu", "cuda")
    args.physics_engine = "physx"
    args.train = not args.test
    if args.algo in MARL_ALGOS:
        args.task_type = "MultiAgent"
    logdir, cfg_train, cfg_env = retrieve_cfg(args, use_rlg_config)
    if args.logdir == "logs/":
        args.logdir = logdir
    if args.cfg_train == "Base":
        args.cfg_train = cfg_train
    if args.cfg_env == "Base":
        args.cfg_env = cfg_env
    return args
