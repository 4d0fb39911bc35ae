"""MultiIngenuity task (agents/tasks/multi_ingenuity.py): 4 coaxial-rotor helicopters per env, Mars gravity."""
from .agent_base.base_task import BaseTask


class MultiIngenuity(BaseTask):
    TASK_NAME = "MultiIngenuity"

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True,
                 is_multi_agent=False, strict_reference_spaces=False):
        self.cfg = cfg
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.is_multi_agent = is_multi_agent
        self.max_episode_length = cfg["env"]["episodeLength"]
        if is_multi_agent:
            self.num_agents = 4
            cfg["env"]["numActions"] = 6                     # multi_ingenuity.py:56-58
            cfg["env"]["numObservations"] = 13
        else:
            self.num_agents = 1
            cfg["env"]["numActions"] = 24                    # multi_ingenuity.py:61-62
            cfg["env"]["numObservations"] = 13 if strict_reference_spaces else 52   # declared 13, produced 52 (:54,:356)
        cfg["device_type"], cfg["device_id"], cfg["headless"] = device_type, device_id, headless
        super().__init__(cfg, num_agents_default=4)
        for k in range(4):
            setattr(self, "obs_buf_%d" % (k + 1), self.obs_buf[:, 13 * k:13 * k + 13])
