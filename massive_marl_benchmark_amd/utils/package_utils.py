"""`make(task_name, algo)` -- the entry the reference exposes as `agents.make` (agents/utils/package_utils.py:20-56, used by
train_customize.py:8-13) -- and `train(argv)`, the wiring of train.py:20-96, both without isaacgym: flags (get_args), YAML
(load_cfg), seeds, then parse_task builds the task on this engine and wraps it in VecTaskPython / MultiVecTaskPython.

The learners themselves are the reference's (callers of the hot path: out of scope); `train` imports them
from the caller's checkout of the reference (the `agents` package on PYTHONPATH) and hands them the env exactly as train.py does.
tests/golden/run_reference_learners.py runs `PPO.run` and `Runner.run` that way in the build container."""
from .config import MARL_ALGOS, get_args, load_cfg, parse_sim_params, set_np_formatting, set_seed
from .parse_task import parse_task

SARL_ALGOS = ("ppo", "ddpg", "sac", "td3", "trpo")


def _build(args):
    cfg, cfg_train, logdir = load_cfg(args)
    sim_params = parse_sim_params(args, cfg, cfg_train)
    set_seed(cfg_train.get("seed", -1), cfg_train.get("torch_deterministic", False))
    if args.algo in MARL_ALGOS:
        args.task_type = "MultiAgent"                                           # package_utils.py:31-34
    elif args.algo not in SARL_ALGOS:
        raise ValueError("Unrecognized algorithm %r: expected one of %s" % (args.algo, sorted(MARL_ALGOS) + list(SARL_ALGOS)))
    task, env = parse_task(args, cfg, cfg_train, sim_params, None)
    return task, env, cfg, cfg_train, logdir


def make(task_name, algo, argv=None):
    """package_utils.py:20-56: the wrapped env for `task_name` under `algo`'s task type.  `argv`: further reference flags, e.g.
    ["--sim_device", "cpu", "--pipeline", "cpu", "--rl_device", "cpu", "--num_envs", "64"] for the CPU pipeline."""
    set_np_formatting()
    args = get_args(["--task", task_name, "--algo", algo] + list(argv or []))
    return _build(args)[1]


def train(argv=None):
    """train.py:20-96 for the single-agent and MARL algorithms the reference ships for these tasks."""
    set_np_formatting()
    args = get_args(argv)
    task, env, cfg, cfg_train, logdir = _build(args)
    try:
        if args.algo in MARL_ALGOS:
            from agents.utils.process_marl import process_MultiAgentRL          # the reference's learner glue (process_marl.py)
        else:
            from agents.utils.process_sarl import process_sarl                  # process_sarl.py:7-41
    except ImportError as e:
        raise ImportError("the learners are the reference's own (agents/algorithms): put its checkout on PYTHONPATH -- %s" % e)
    if args.algo in MARL_ALGOS:
        runner = process_MultiAgentRL(args, env=env, config=cfg_train, model_dir=args.model_dir)
        return runner.eval(1000) if args.model_dir != "" else runner.run()
    learner = process_sarl(args, env, cfg_train, logdir)
    iterations = args.max_iterations if args.max_iterations > 0 else cfg_train["learn"]["max_iterations"]
    return learner.run(num_learning_iterations=iterations, log_interval=cfg_train["learn"]["save_interval"])
