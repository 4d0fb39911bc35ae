"""SeparatedReplayBuffer.insert / compute_returns / after_update
(agents/algorithms/marl/utils/separated_buffer.py:12-168) with the GAE scan as a HIP kernel.

Only the path the shipped configs use is accelerated: `use_gae=True`, `use_proper_time_limits=False`, with
PopArt / ValueNorm / no normaliser (cfg/mappo/config.yaml:26-27, cfg/happo/config.yaml:29).  The other
branches raise (they are not reachable from the shipped configs).  The minibatch generators the trainers iterate over
(feed_forward / naive_recurrent / recurrent, separated_buffer.py:170-428) come from generators.MinibatchGenerators."""
import ctypes

import torch

from .... import _lib
from .generators import MinibatchGenerators


def _shape_of(space):
    return tuple(space.shape) if hasattr(space, "shape") else tuple(space)


class SeparatedReplayBuffer(MinibatchGenerators):
    def __init__(self, config, obs_space, share_obs_space, act_space, device):
        self.episode_length = config["episode_length"]
        self.n_rollout_threads = config["n_rollout_threads"]
        self.rnn_hidden_size = config["hidden_size"]
        self.recurrent_N = config["recurrent_N"]
        self.gamma = config["gamma"]
        self.gae_lambda = config["gae_lambda"]
        self._use_gae = config["use_gae"]
        self._use_popart = config["use_popart"]
        self._use_valuenorm = config["use_valuenorm"]
        self._use_proper_time_limits = config["use_proper_time_limits"]
        self.device = device
        T, N = self.episode_length, self.n_rollout_threads
        obs_shape, share_obs_shape = _shape_of(obs_space), _shape_of(share_obs_space)
        z = lambda *s: torch.zeros(*s, device=self.device)
        self.share_obs = z(T + 1, N, *share_obs_shape)
        self.obs = z(T + 1, N, *obs_shape)
        self.rnn_states = z(T + 1, N, self.recurrent_N, self.rnn_hidden_size)
        self.rnn_states_critic = torch.zeros_like(self.rnn_states)
        self.value_preds = z(T + 1, N, 1)
        self.returns = z(T + 1, N, 1)
        self.available_actions = None
        act_shape = act_space.shape[0]
        self.actions = z(T, N, act_shape)
        self.action_log_probs = z(T, N, act_shape)
        self.rewards = z(T, N, 1)
        self.masks = torch.ones(T + 1, N, 1, device=self.device)
        self.bad_masks = torch.ones_like(self.masks)
        self.active_masks = torch.ones_like(self.masks)
        self.factor = torch.ones(T, N, 1, device=self.device)
        self.step = 0

    def update_factor(self, factor):
        self.factor.copy_(factor)

    def insert(self, share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds, rewards, masks,
               bad_masks=None, active_masks=None, available_actions=None):
        s = self.step
        self.share_obs[s + 1].copy_(share_obs)
        self.obs[s + 1].copy_(obs)
        self.rnn_states[s + 1].copy_(rnn_states)
        self.rnn_states_critic[s + 1].copy_(rnn_states_critic)
        self.actions[s].copy_(actions)
        self.action_log_probs[s].copy_(action_log_probs)
        self.value_preds[s].copy_(value_preds)
        self.rewards[s].copy_(rewards)
        self.masks[s + 1].copy_(masks)
        if bad_masks is not None:
            self.bad_masks[s + 1].copy_(bad_masks)
        if active_masks is not None:
            self.active_masks[s + 1].copy_(active_masks)
        self.step = (s + 1) % self.episode_length

    def after_update(self):
        for t in (self.share_obs, self.obs, self.rnn_states, self.rnn_states_critic, self.masks, self.bad_masks, self.active_masks):
            t[0].copy_(t[-1])

    def compute_returns(self, next_value, value_normalizer=None):
        if self._use_proper_time_limits or not self._use_gae:
            raise NotImplementedError("only use_gae=True, use_proper_time_limits=False is accelerated (the shipped configs)")
        dev = torch.device(self.device)
        L, idx, stream = _lib.for_device(dev)               # "cuda": the HIP build; "cpu": the CPU build (the caller's explicit choice)
        self.value_preds[-1] = next_value
        use_norm = 1 if (self._use_popart or self._use_valuenorm) else 0
        if use_norm:
            mean, var = value_normalizer.running_mean_var()
            mean = mean.to(dev).float().contiguous().view(-1)
            var = var.to(dev).float().contiguous().view(-1)
        else:
            mean = var = self.rewards          # unused
        T, N = self.episode_length, self.n_rollout_threads
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(L.mms_gae_marl(idx, p(self.rewards), p(self.value_preds), p(self.masks), p(self.returns), T, N,
                                  float(self.gamma), float(self.gae_lambda), use_norm, p(mean), p(var), stream), None, "mms_gae_marl", L)
