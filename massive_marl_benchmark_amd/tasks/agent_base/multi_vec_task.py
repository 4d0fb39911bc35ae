"""MultiVecTask / MultiVecTaskPython (agents/tasks/agent_base/multi_vec_task.py:20-175): the MARL wrapper.
The reference hard-codes 10 agents x 38 + 8 shared = 388; here the numbers come from the task (same
defaults).  obs_all is produced by one HIP kernel; state_all / reward_all / done_all are stride-0
expansions of the engine buffers (the reference materialises ten copies)."""
import ctypes

import numpy as np
import torch

from ... import _lib, spaces
from ...engine import current_stream_ptr


class MultiVecTask:
    def __init__(self, task, rl_device, clip_observations=7.0, clip_actions=1.0):
        self.task = task
        self.num_environments = task.num_envs
        self.num_actions = task.num_actions
        self.num_agents = task.num_agents
        e = task.engine
        self.shared_obs = e.obs_dim - 38 * self.num_agents if task.TASK_NAME == "TenAnt" else 0
        self.num_ant_obs = (e.obs_dim - self.shared_obs) // self.num_agents
        self.num_observations = self.num_ant_obs + self.shared_obs            # 46
        self.nums_share_observations = e.obs_dim                               # 388
        self.clip_obs = clip_observations
        self.clip_actions = clip_actions
        self.rl_device = rl_device
        cfg = e.config
        if abs(cfg.clip_obs - clip_observations) > 0 or abs(cfg.clip_actions - clip_actions) > 0:
            raise ValueError("the task's engine was created with clip_obs=%g clip_actions=%g; pass the same values here"
                             % (cfg.clip_obs, cfg.clip_actions))
        self.obs_space = [spaces.Box(low=-np.inf, high=np.inf, shape=(self.num_observations,)) for _ in range(self.num_agents)]
        self.share_observation_space = [spaces.Box(low=-np.inf, high=np.inf, shape=(self.nums_share_observations,))
                                        for _ in range(self.num_agents)]
        self.act_space = tuple([spaces.Box(low=np.ones(self.num_actions) * -clip_actions, high=np.ones(self.num_actions) * clip_actions)
                                for _ in range(self.num_agents)])
        self._obs_all = torch.empty((self.num_environments, self.num_agents, self.num_observations), dtype=torch.float32,
                                    device=e.device)

    def step(self, actions):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def get_number_of_agents(self):
        return self.num_agents

    @property
    def observation_space(self):
        return self.obs_space

    @property
    def action_space(self):
        return self.act_space

    @property
    def num_envs(self):
        return self.num_environments

    @property
    def num_acts(self):
        return self.num_actions

    @property
    def num_obs(self):
        return self.num_observations


class MultiVecTaskPython(MultiVecTask):
    def get_state(self):
        return torch.clamp(self.task.states_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    def _views(self):
        t, e = self.task, self.task.engine
        L, idx, stream = _lib.for_device(e.device)
        _lib.check(L.mms_marl_views(idx, ctypes.c_void_p(t.obs_buf_clipped.contiguous().data_ptr()),
                                    ctypes.c_void_p(self._obs_all.data_ptr()), self.num_environments, self.num_agents,
                                    self.num_ant_obs, self.shared_obs, stream), None, "mms_marl_views", L)
        state_all = t.obs_buf_clipped.unsqueeze(1).expand(-1, self.num_agents, -1)
        return self._obs_all.to(self.rl_device), state_all.to(self.rl_device)

    def step(self, actions):
        if isinstance(actions, (list, tuple)):                      # multi_vec_task.py:96-99
            actions = torch.hstack(list(actions))
        t = self.task
        t.step(actions)
        obs_all, state_all = self._views()
        reward_all = t.rew_buf.view(-1, 1, 1).expand(-1, self.num_agents, 1).to(self.rl_device)
        done_all = t.reset_buf.view(-1, 1).expand(-1, self.num_agents).to(self.rl_device)
        info_all = torch.zeros((self.num_agents, 0))
        return obs_all, state_all, reward_all, done_all, info_all, None

    def reset(self):
        t = self.task
        actions = torch.zeros([self.num_envs, t.engine.num_actions], dtype=torch.float32, device=t.engine.device)  # :147
        t.step(actions)
        obs_all, state_all = self._views()
        return obs_all, state_all, None
