"""TenAnt task (agents/tasks/ten_ant.py): 10 ants + 1 box per env.  Constructor signature and public
attributes follow the reference; the scene, physics, reset_idx, compute_observations and
compute_reward all run inside the fused HIP step kernel."""
from .agent_base.base_task import BaseTask


class TenAnt(BaseTask):
    TASK_NAME = "TenAnt"

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True,
                 is_multi_agent=False, num_ants=10, strict_reference_spaces=False):
        self.cfg = cfg
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.is_multi_agent = is_multi_agent
        self.max_episode_length = cfg["env"]["episodeLength"]
        self.num_ants = int(num_ants)
        # ten_ant.py:53 declares 38 although obs_buf is 388 wide: the real width is
        # exposed unless strict_reference_spaces is requested
        full = 38 * self.num_ants + 8
        if is_multi_agent:
            self.num_agents = self.num_ants
            cfg["env"]["numActions"] = 8                     # ten_ant.py:61-63
            cfg["env"]["numObservations"] = 38
        else:
            self.num_agents = 1
            cfg["env"]["numActions"] = 8 * self.num_ants     # ten_ant.py:66-67
            cfg["env"]["numObservations"] = 38 if strict_reference_spaces else full
        cfg["device_type"], cfg["device_id"], cfg["headless"] = device_type, device_id, headless
        super().__init__(cfg, num_agents_default=self.num_ants)
        self.num_dof = 8
        n = self.num_envs
        self.dof_pos = self.dof_state.view(n, -1, 2)[..., 0]
        self.dof_vel = self.dof_state.view(n, -1, 2)[..., 1]
        for k in range(self.num_ants):                       # ten_ant.py:107-127
            setattr(self, "dof_pos_%d" % (k + 1), self.dof_pos[:, 8 * k:8 * k + 8])
            setattr(self, "dof_vel_%d" % (k + 1), self.dof_vel[:, 8 * k:8 * k + 8])
            setattr(self, "obs_buf_%d" % (k + 1), self.obs_buf[:, 38 * k:38 * k + 38])
        self.box_pos = self.obs_buf[:, 38 * self.num_ants:38 * self.num_ants + 2]
        self.box_quat = self.obs_buf[:, 38 * self.num_ants + 2:38 * self.num_ants + 6]
        self.prev = self.engine.tensor("prev")
