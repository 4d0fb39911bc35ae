"""OneAnt task (agents/tasks/one_ant.py): 1 ant + 1 box per env, 60-wide observation incl. foot sensors."""
from .agent_base.base_task import BaseTask


class OneAnt(BaseTask):
    TASK_NAME = "OneAnt"

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True,
                 is_multi_agent=False):
        self.cfg = cfg
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.is_multi_agent = is_multi_agent
        self.max_episode_length = cfg["env"]["episodeLength"]
        cfg["env"]["numObservations"] = 60                  # one_ant.py:51-52
        cfg["env"]["numActions"] = 8
        cfg["device_type"], cfg["device_id"], cfg["headless"] = device_type, device_id, headless
        self.num_agents = 1
        super().__init__(cfg, num_agents_default=1)
        n = self.num_envs
        self.num_dof = 8
        self.dof_pos = self.dof_state.view(n, 8, 2)[..., 0]
        self.dof_vel = self.dof_state.view(n, 8, 2)[..., 1]
        self.vec_sensor_tensor = self.engine.tensor("foot_sensors")
