"""ReplayBuffer of the off-policy learners (agents/algorithms/rl/ddpg/storage.py:5-70; td3/storage.py and sac/storage.py hold
the same class): the `[replay_size, num_envs, .]` transition ring that DDPG / TD3 / SAC fill once per env step
(ddpg.py:151-159) -- BASELINE configs[2], MultiIngenuity at 8192 envs.

Same constructor, fields, cursor arithmetic and method names.  What changes is where the rows come from: `slot()` tells the
caller which ring row the next `add_transitions` fills, so the engine can be bound to it (`Engine.bind_obs_out(buf.next_observations[k])`,
`Engine.bind_rollout_out(buf.rewards[k], buf.dones[k])`) and the step kernel writes next_obs / reward / done there itself;
`add_transitions` recognises such rows by their address and copies only what is not in place yet.  At the reference's
replay_size = 10000 (cfg/ddpg/config.yaml) and 8192 envs the ring is 2 x 17 GB of observations: resident in HBM, no host tier.

Reference behaviour kept on purpose: on overflow the cursor becomes (replay_size + 1) % replay_size = 1, not 0 (storage.py:29-33),
so row 0 keeps the first transition for ever and the ring cycles through rows 1 .. replay_size-1; `mini_batch_generator`
draws ROW indices with Python's `random.sample` (the same stream as the reference for the same `random.seed`).
"""
import random

import torch


class ReplayBuffer:
    def __init__(self, num_envs, replay_size, batch_size, num_transitions_per_env, obs_shape, states_shape, actions_shape,
                 device='cpu', sampler='sequential'):
        self.device = device
        self.sampler = sampler
        R, N = replay_size, num_envs
        z = lambda *s: torch.zeros(*s, device=self.device)
        self.observations = z(R, N, *obs_shape)
        self.states = z(R, N, *states_shape)
        self.rewards = z(R, N, 1)
        self.next_observations = z(R, N, *obs_shape)
        self.actions = z(R, N, *actions_shape)
        self.dones = z(R, N, 1).byte()
        self.num_transitions_per_env = num_transitions_per_env
        self.replay_size = R
        self.batch_size = batch_size
        self.num_envs = N
        self.fullfill = False
        self.step = 0

    def slot(self):
        """Ring row the next add_transitions writes (storage.py:29-33 applied ahead of time, without moving the cursor)."""
        return self.step if self.step < self.replay_size else (self.step + 1) % self.replay_size

    def add_transitions(self, observations, states, actions, rewards, next_obs, dones):
        if self.step >= self.replay_size:
            self.step = (self.step + 1) % self.replay_size
            self.fullfill = True
        k = self.step

        def put(dst, src):
            if src.data_ptr() != dst.data_ptr() or src.numel() == 0:
                dst.copy_(src.view(dst.shape))
        put(self.observations[k], observations)
        put(self.states[k], states)
        put(self.actions[k], actions)
        put(self.rewards[k], rewards)
        put(self.next_observations[k], next_obs)
        put(self.dones[k], dones)
        self.step += 1

    def get_statistics(self):
        """storage.py:46-52: mean distance between done flags over the env-major flattening of the WHOLE ring (unwritten rows
        count as not done, the last row as done), and the mean reward of rows [0, step).  Evaluated where the ring lives."""
        done = self.dones.clone()
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1)
        ends = flat.nonzero(as_tuple=False)[:, 0]
        starts = torch.cat((ends.new_tensor([-1]), ends[:-1]))
        return (ends - starts).float().mean(), self.rewards[:self.step].mean()

    def mini_batch_generator(self, num_mini_batches):
        """storage.py:54-70: num_mini_batches lists of batch_size // num_mini_batches distinct ring rows."""
        size = self.batch_size // num_mini_batches
        rows = range(self.replay_size if self.fullfill else self.step)
        return [random.sample(rows, size) for _ in range(num_mini_batches)]
