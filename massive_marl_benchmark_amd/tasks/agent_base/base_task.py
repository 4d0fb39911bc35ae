"""BaseTask: the step protocol and buffer set of agents/tasks/agent_base/base_task.py:24-149, with the
Isaac Gym calls replaced by one fused engine step (Engine.step = pre_physics_step + simulate +
post_physics_step).  Domain randomisation and the viewer are out of scope."""
import torch

from ...engine import Engine


class BaseTask:
    TASK_NAME = None

    def __init__(self, cfg, num_agents_default, clip_obs=5.0):
        clip_obs = float(cfg.get("clip_observations", clip_obs))   # the VecTask wrapper's clamp, fused into the kernel
        self.device_type = cfg.get("device_type", "cuda")
        self.device_id = cfg.get("device_id", 0)
        if self.device_type not in ("cuda", "GPU"):
            # base_task.py:27-32 selects the CPU pipeline here; this build has no CPU path by design
            raise RuntimeError("device_type=%r: the MI355X engine has no CPU pipeline (no CPU fallback)" % (self.device_type,))
        self.device = "cuda:" + str(self.device_id)
        self.headless = cfg.get("headless", True)
        self.num_envs = cfg["env"]["numEnvs"]
        self.control_freq_inv = cfg["env"].get("controlFrequencyInv", 1)
        if self.control_freq_inv != 1:
            raise NotImplementedError("controlFrequencyInv != 1 (cfg/*.yaml ship 1)")
        self.engine = Engine(self.TASK_NAME, cfg, num_envs=self.num_envs, num_agents=num_agents_default,
                             device=self.device_id, seed=max(int(cfg.get("seed", 0) or 0), 0),
                             env_offset=int(cfg.get("env_offset", 0)), total_envs=cfg.get("total_envs", None),
                             clip_obs=clip_obs, clip_actions=float(cfg["env"].get("clipActions", 1.0)))
        e = self.engine
        self.num_obs = cfg["env"]["numObservations"]
        self.num_states = cfg["env"].get("numStates", 0)
        self.num_actions = cfg["env"]["numActions"]
        # zero-copy views of engine memory (base_task.py:56-67 allocates torch buffers instead)
        self.obs_buf = e.tensor("obs")
        self.obs_buf_clipped = e.tensor("obs_clipped")
        self.states_buf = torch.zeros((self.num_envs, self.num_states), device=self.device, dtype=torch.float)
        self.rew_buf = e.tensor("rew")
        self.reset_buf = e.tensor("reset")
        self.progress_buf = e.tensor("progress")
        self.randomize_buf = torch.zeros(self.num_envs, device=self.device, dtype=torch.long)
        self.root_states = e.tensor("root_states")
        self.initial_root_states = e.tensor("initial_root_states")
        self.dof_state = e.tensor("dof_state")
        self.env_origin = e.tensor("env_origin")
        self._actions = e.tensor("actions")
        self.actions = self._actions
        self.extras = {}
        self.viewer = None
        self.dt = cfg["sim"]["dt"]

    def step(self, actions):
        """base_task.py:129-149.  `actions`: [num_envs, engine action width]; clamping happens in the kernel."""
        if actions.data_ptr() != self._actions.data_ptr():
            self._actions.copy_(actions.reshape(self._actions.shape))
        self.engine.step()

    def get_states(self):
        return self.states_buf

    def render(self, sync_frame_time=False):
        return None
