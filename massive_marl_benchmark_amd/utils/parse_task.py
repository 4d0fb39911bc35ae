"""parse_task (agents/utils/parse_task.py:25-93): builds (task, env) for the 'Python' and 'MultiAgent' task types."""
from ..tasks.agent_base.multi_vec_task import MultiVecTaskPython
from ..tasks.agent_base.vec_task import VecTaskPython
from ..tasks.multi_ant_circle import MultiAntCircle
from ..tasks.multi_ingenuity import MultiIngenuity
from ..tasks.one_ant import OneAnt
from ..tasks.ten_ant import TenAnt

# (MultiAntCircle is not registered in the reference's parse_task.py:8-10 -- it cannot be imported there; here it can be asked for)
_TASKS = {"TenAnt": TenAnt, "OneAnt": OneAnt, "MultiIngenuity": MultiIngenuity, "MultiAntCircle": MultiAntCircle}


def parse_task(args, cfg, cfg_train, sim_params, agent_index=None):
    device_id = args.device_id
    rl_device = args.rl_device
    cfg["seed"] = cfg_train.get("seed", -1)
    cfg["env"]["seed"] = cfg["seed"]
    if args.task not in _TASKS:
        raise ValueError("Unrecognized task %r" % (args.task,))
    multi = args.task_type == "MultiAgent"
    if args.task_type not in ("Python", "MultiAgent"):
        raise ValueError("task_type %r is not supported (the reference's C++ task types need rlgpu)" % (args.task_type,))
    clip_obs = 7.0 if multi else cfg_train.get("clip_observations", 5.0)     # multi_vec_task.py:22 / vec_task.py:18
    cfg["clip_observations"] = clip_obs
    # the wrapper's observation clamp is fused into the step kernel: BaseTask reads cfg["clip_observations"]
    task = _TASKS[args.task](cfg=cfg, sim_params=sim_params, physics_engine=args.physics_engine, device_type=args.device,
                             device_id=device_id, headless=args.headless, is_multi_agent=multi)
    env = MultiVecTaskPython(task, rl_device, clip_obs, 1.0) if multi else VecTaskPython(task, rl_device, clip_obs, 1.0)
    return task, env
