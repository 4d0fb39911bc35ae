"""VecTask / VecTaskPython (agents/tasks/agent_base/vec_task.py:17-139): the OUTER drop-in boundary that
agents/algorithms/rl/* call.  Same constructor, properties and return values; the clamps of
vec_task.py:127,131 are done inside the step kernel (actions on load, observations into `obs_clipped`)."""
import numpy as np
import torch

from ... import spaces


class VecTask:
    def __init__(self, task, rl_device, clip_observations=5.0, clip_actions=1.0):
        self.task = task
        self.num_environments = task.num_envs
        self.num_agents = 1
        self.num_observations = task.num_obs
        self.num_states = task.num_states
        self.num_actions = task.num_actions
        self.obs_space = spaces.Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self.state_space = spaces.Box(np.ones(self.num_states) * -np.inf, np.ones(self.num_states) * np.inf)
        self.act_space = spaces.Box(np.ones(self.num_actions) * -1., np.ones(self.num_actions) * 1.)
        self.clip_obs = clip_observations
        self.clip_actions = clip_actions
        self.rl_device = rl_device
        cfg = task.engine.config
        if abs(cfg.clip_obs - clip_observations) > 0 or abs(cfg.clip_actions - clip_actions) > 0:
            raise ValueError("the task's engine was created with clip_obs=%g clip_actions=%g; pass the same values here"
                             % (cfg.clip_obs, cfg.clip_actions))

    def step(self, actions):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def get_number_of_agents(self):
        return self.num_agents

    # read-only views the algorithms use (vec_task.py:45-64)
    observation_space = property(lambda self: self.obs_space)
    action_space = property(lambda self: self.act_space)
    num_envs = property(lambda self: self.num_environments)
    num_acts = property(lambda self: self.num_actions)
    num_obs = property(lambda self: self.num_observations)


class VecTaskPython(VecTask):
    def get_state(self):
        return torch.clamp(self.task.states_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    def step(self, actions):
        self.task.step(actions)                                   # clamp to +-clip_actions happens in the kernel
        t = self.task
        return (t.obs_buf_clipped.to(self.rl_device), t.rew_buf.to(self.rl_device), t.reset_buf.to(self.rl_device), t.extras)

    def reset(self):
        actions = 0.01 * (1 - 2 * torch.rand([self.task.num_envs, self.task.num_actions], dtype=torch.float32,
                                              device=self.rl_device))   # vec_task.py:134
        self.task.step(actions)
        return self.task.obs_buf_clipped.to(self.rl_device)
