"""RolloutStorage (agents/algorithms/rl/ppo/storage.py:5-87) with the GAE scan and the advantage
normalisation as HIP kernels (mms_gae_ppo / mms_adv_normalize) instead of T sequential torch launches.

Same fields, shapes and method names, so PPO (agents/algorithms/rl/ppo/ppo.py) uses it unchanged.
Additions, all optional: `observation_slot(t)` hands the engine a slot to write the clamped observation
row into directly (zero-copy rollout), and `process_group` makes the advantage statistics global over
data-parallel ranks (two float64 all-reduced over RCCL between the two kernels)."""
import ctypes

import torch
from torch.utils.data.sampler import BatchSampler, SequentialSampler, SubsetRandomSampler

from .... import _lib


class RolloutStorage:
    def __init__(self, num_envs, num_transitions_per_env, obs_shape, states_shape, actions_shape, device='cpu',
                 sampler='sequential', process_group=None):
        self.device = device
        self.sampler = sampler
        T, N = num_transitions_per_env, num_envs
        z = lambda *s, **k: torch.zeros(*s, device=self.device, **k)
        self.observations = z(T, N, *obs_shape)
        self.states = z(T, N, *states_shape)
        self.rewards = z(T, N, 1)
        self.actions = z(T, N, *actions_shape)
        self.dones = z(T, N, 1).byte()
        self.actions_log_prob = z(T, N, 1)
        self.values = z(T, N, 1)
        self.returns = z(T, N, 1)
        self.advantages = z(T, N, 1)
        self.mu = z(T, N, *actions_shape)
        self.sigma = z(T, N, *actions_shape)
        self.num_transitions_per_env = T
        self.num_envs = N
        self.step = 0
        self.process_group = process_group
        self._stats = torch.zeros(3 + 2 * 2048, dtype=torch.float64, device=self.device)      # MMS_GAE_STATS_DOUBLES (include/mms.h)

    def observation_slot(self, t):
        return self.observations[t]

    def add_transitions(self, observations, states, actions, rewards, dones, values, actions_log_prob, mu, sigma):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        s = self.step
        # rows that the engine (bind_obs_out / bind_rollout_out) or ActorCritic.act (bind_rollout) already wrote in place
        # are recognised by their address and not copied again
        def put(dst, src):
            if src.data_ptr() != dst.data_ptr() or src.numel() == 0:
                dst.copy_(src.view(dst.shape))
        put(self.observations[s], observations)
        put(self.states[s], states)
        put(self.actions[s], actions)
        put(self.rewards[s], rewards)
        put(self.dones[s], dones)
        put(self.values[s], values)
        put(self.actions_log_prob[s], actions_log_prob)
        put(self.mu[s], mu)
        put(self.sigma[s], sigma)
        self.step += 1

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam):
        """storage.py:51-65: reverse GAE scan, then advantages = (ret - V - mean) / (std + 1e-8)."""
        dev = torch.device(self.device)
        L, idx, stream = _lib.for_device(dev)               # "cuda": the HIP build; "cpu": the CPU build (the caller's explicit choice)
        T, N = self.num_transitions_per_env, self.num_envs
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        last_values = last_values.contiguous().view(-1).float()
        if self.process_group is None:
            # one rank: per-block partial sums instead of atomics (fixed summation order: bit-reproducible; nothing to zero)
            _lib.check(L.mms_gae_ppo_normalized(idx, p(self.rewards), p(self.dones), p(self.values), p(last_values), p(self.returns),
                                                p(self.advantages), p(self._stats), T, N, float(gamma), float(lam), stream), None, "mms_gae_ppo_normalized", L)
            return
        _lib.check(L.mms_gae_ppo(idx, p(self.rewards), p(self.dones), p(self.values), p(last_values), p(self.returns),
                                 p(self.advantages), p(self._stats), T, N, float(gamma), float(lam), stream), None, "mms_gae_ppo", L)
        torch.distributed.all_reduce(self._stats[:3], group=self.process_group)   # sum, sum of squares, count
        _lib.check(L.mms_adv_normalize(idx, p(self.advantages), p(self._stats), T * N, stream), None, "mms_adv_normalize", L)

    def get_statistics(self):
        """(mean trajectory length, mean reward) of the stored rollout, storage.py:67-72: trajectories are cut at every done
        and at the end of the buffer, walked env by env."""
        ended = self.dones.detach().to("cpu", copy=True)
        ended[-1] = 1                                                   # the buffer end closes every open trajectory
        per_env = ended.permute(1, 0, 2).reshape(-1)                    # env-major order, as the reference flattens it
        cut = torch.nonzero(per_env, as_tuple=False).flatten()
        lengths = torch.diff(cut, prepend=cut.new_tensor([-1]))
        return lengths.float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches):
        """Index batches over the T x N transitions (storage.py:74-87): in order for 'sequential', shuffled for 'random'."""
        total = self.num_envs * self.num_transitions_per_env
        if self.sampler == "sequential":
            order = SequentialSampler(range(total))
        elif self.sampler == "random":
            order = SubsetRandomSampler(range(total))
        else:
            raise ValueError("sampler must be 'sequential' or 'random', not %r" % (self.sampler,))
        return BatchSampler(order, total // num_mini_batches, drop_last=True)
