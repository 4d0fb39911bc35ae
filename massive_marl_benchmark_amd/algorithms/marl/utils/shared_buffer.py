"""Rollout-buffer fusion for the MARL runners.

The reference keeps one SeparatedReplayBuffer per agent (agents/algorithms/marl/runner.py:100-112) and, every step,
copies the SAME 388-wide centralised-critic row, the same reward and the same mask into each of the ten buffers
(runner.py:250-255, separated_buffer.py:67-85): ten copies of `share_obs [T+1, N, 388]` = 570 MB at 4096 envs.  It then
walks the envs in a Python loop with a device sync per step (runner.py:141-144).

`SharedRolloutBuffers` owns ONE set of tensors and hands out per-agent objects with the SeparatedReplayBuffer attribute
set (`share_obs`, `obs`, `value_preds`, `returns`, `rewards`, `masks`, `actions`, ... and `insert`, `compute_returns`,
`after_update`) whose tensors are VIEWS of the shared storage, so the reference's collect / train code runs unchanged:

  * `share_obs`  one [T+1, N, 388] tensor; the engine writes the clamped row of step t straight into slot t+1
                 (mms_bind_obs_out): zero copies, stored once instead of A times;
  * `obs`        one [T+1, N, A, 46] tensor filled by mms_marl_views; agent k sees `obs[:, :, k]`;
  * `rewards`, `masks`  stored once ([T, N, 1], [T+1, N, 1]); every agent's view is the same tensor;
  * `value_preds`, `returns`  [T+1, N, A]; one GAE launch for all agents (mms_gae_marl_agents);
  * `insert_step` does the runner's `insert` for all agents at once, and `finished_episode_rewards` replaces the per-env loop.
"""
import ctypes

import torch

from .... import _lib
from .generators import MinibatchGenerators


class _AgentView(MinibatchGenerators):
    """Per-agent facade with the SeparatedReplayBuffer attribute names (separated_buffer.py:36-60)."""

    def __init__(self, parent, k):
        self._p, self._k = parent, k
        p = parent
        self.episode_length, self.n_rollout_threads = p.T, p.N
        self.share_obs = p.share_obs
        self.obs = p.obs[:, :, k]
        self.rnn_states = p.rnn_states[:, :, k]
        self.rnn_states_critic = p.rnn_states_critic[:, :, k]
        self.value_preds = p.value_preds[:, :, k:k + 1]
        self.returns = p.returns[:, :, k:k + 1]
        self.actions = p.actions[:, :, k]
        self.action_log_probs = p.action_log_probs[:, :, k]
        self.rewards = p.rewards
        self.masks = p.masks
        self.bad_masks = p.bad_masks
        self.active_masks = p.active_masks[:, :, k]
        self.factor = p.factor[:, :, k]
        self.available_actions = None

    @property
    def step(self):
        return self._p.step

    def update_factor(self, factor):
        self.factor.copy_(factor)

    def insert(self, share_obs, obs, rnn_states, rnn_states_critic, actions, action_log_probs, value_preds, rewards, masks,
               bad_masks=None, active_masks=None, available_actions=None):
        """API-compatible per-agent insert (separated_buffer.py:67-85).  The shared parts are written by whichever
        agent comes first in a step; prefer SharedRolloutBuffers.insert_step."""
        p, k, s = self._p, self._k, self._p.step
        if k == 0:
            if share_obs.data_ptr() != p.share_obs[s + 1].data_ptr():
                p.share_obs[s + 1].copy_(share_obs)
            p.rewards[s].copy_(rewards)
            p.masks[s + 1].copy_(masks)
        self.obs[s + 1].copy_(obs)
        self.rnn_states[s + 1].copy_(rnn_states)
        self.rnn_states_critic[s + 1].copy_(rnn_states_critic)
        self.actions[s].copy_(actions)
        self.action_log_probs[s].copy_(action_log_probs)
        self.value_preds[s].copy_(value_preds)
        if active_masks is not None:
            self.active_masks[s + 1].copy_(active_masks)
        if k == p.A - 1:
            p.step = (s + 1) % p.T

    def after_update(self):
        if self._k == 0:
            self._p.after_update()

    def compute_returns(self, next_value, value_normalizer=None):
        self._p._pending[self._k] = (next_value, value_normalizer)
        if len(self._p._pending) == self._p.A:
            nv = torch.cat([self._p._pending[k][0].reshape(-1, 1) for k in range(self._p.A)], 1)
            norms = [self._p._pending[k][1] for k in range(self._p.A)]
            self._p._pending = {}
            self._p.compute_returns(nv, norms)


GATHERED = ("share_obs", "obs", "value_preds", "returns", "actions", "action_log_probs", "rewards", "masks", "active_masks", "factor")


def all_gather_envs(tensors, group=None):
    """The one real exchange step of the MARL path: envs are sharded over ranks for the rollout, and when
    TRAINING is agent-parallel (rank g updates agents g, g + world, ...) every rank needs the rollout of ALL envs -- above all
    the centralised critic's `share_obs` rows.  `tensors`: dict name -> [T(+1), N_local, ...] (env dimension 1); returns
    dict name -> [T(+1), world * N_local, ...] with rank r's envs at [r N_local, (r + 1) N_local), i.e. in global env order
    (rank r simulates the global envs [r N, (r + 1) N): the engine's env_offset).  One `all_gather_into_tensor` per buffer --
    over RCCL it drives all seven xGMI links of a GPU at once, which a ring would not; the big one is share_obs,
    (T + 1) N 388 floats = 57 MB per rank at 4096 envs."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = {}
    for name, t in tensors.items():
        t = t.contiguous()
        flat = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)   # ranks concatenated along dim 0
        dist.all_gather_into_tensor(flat, t, group=group)
        # [world, T, N, ...] -> [T, world * N, ...]
        out[name] = flat.view((world,) + tuple(t.shape)).transpose(0, 1).reshape((t.shape[0], world * t.shape[1]) + tuple(t.shape[2:]))
    return out


def happo_factor_chain(order, owner_of, train_agent, factor, group=None):
    """The sequential-update chain of HAPPO / HATRPO (agents/algorithms/marl/runner.py:266-316) with AGENT-PARALLEL training
    (SURVEY.md section 8e): agents are updated one after another in `order` (the reference draws torch.randperm), agent k's
    surrogate is weighted by `factor` = the product, over the agents updated before it, of exp(sum_dims(new_logp - old_logp))
    (:312-313).  `owner_of(k)` names the rank that trains agent k; on that rank `train_agent(k, factor)` is called -- it must
    `update_factor`, evaluate the old log-probs, train, evaluate the new ones (runner.py:277-311) and return
    (old_logp, new_logp), each [T, N, act_dim] -- and the updated [T, N, 1] factor travels to everybody with ONE broadcast per
    agent from its owner (the next owner needs it; the others need it only to stay in step, and a broadcast over RCCL costs what
    a send does at this size: T N floats = 131 KB at 4096 envs).  `order` must be the same on every rank (draw it on rank 0 and
    broadcast it, or seed it).  Without a process group (or world size 1) this is exactly the reference's loop.  Returns factor."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if multi else 0
    for k in [int(x) for x in order]:
        src = owner_of(k) if multi else 0
        if rank == src:
            old_logp, new_logp = train_agent(k, factor)
            factor = factor * torch.exp((new_logp.detach() - old_logp.detach()).sum(dim=-1, keepdim=True))   # runner.py:312-313
        if multi:
            factor = factor.contiguous()
            dist.broadcast(factor, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return factor


class SharedRolloutBuffers:
    def __init__(self, config, env, device):
        self.T = config["episode_length"]
        self.N = config["n_rollout_threads"]
        self.gamma, self.gae_lambda = config["gamma"], config["gae_lambda"]
        self._use_norm = bool(config["use_popart"] or config["use_valuenorm"])
        if config["use_proper_time_limits"] or not config["use_gae"]:
            raise NotImplementedError("only use_gae=True, use_proper_time_limits=False (the shipped configs)")
        self.device = torch.device(device)
        self.env = env
        self.A = env.num_agents
        T, N, A = self.T, self.N, self.A
        od, sd, ad = env.num_observations, env.nums_share_observations, env.action_space[0].shape[0]
        rn, hs = config["recurrent_N"], config["hidden_size"]
        z = lambda *s: torch.zeros(*s, device=self.device)
        self.share_obs = z(T + 1, N, sd)
        self.obs = z(T + 1, N, A, od)
        self.rnn_states = z(T + 1, N, A, rn, hs)
        self.rnn_states_critic = z(T + 1, N, A, rn, hs)
        self.value_preds = z(T + 1, N, A)
        self.returns = z(T + 1, N, A)
        self.actions = z(T, N, A, ad)
        self.action_log_probs = z(T, N, A, ad)
        self.rewards = z(T, N, 1)
        self.masks = torch.ones(T + 1, N, 1, device=self.device)
        self.bad_masks = torch.ones_like(self.masks)
        self.active_masks = torch.ones(T + 1, N, A, 1, device=self.device)
        self.factor = torch.ones(T, N, A, 1, device=self.device)
        self.step = 0
        self._pending = {}
        self.agents = [_AgentView(self, k) for k in range(A)]

    # -- environment side ----------------------------------------------------------------------------------------
    def warmup(self):
        """Runner.warmup (runner.py:187-197): reset the envs, slot 0 of share_obs / obs."""
        self._bind(0)
        self.env.task.step(torch.zeros(self.N, self.env.task.engine.num_actions, device=self.device))   # multi_vec_task.py:147
        self._views(0)

    def env_step(self, actions):
        """envs.step + the observation part of insert: the engine writes share_obs[t+1] and obs[t+1] in place.
        Returns (rewards [N], dones [N]) as views of engine memory (valid until the next step)."""
        s = self.step
        if isinstance(actions, (list, tuple)):
            actions = torch.hstack(list(actions))
        self._bind(s + 1)
        self.env.task.step(actions)
        self._views(s + 1)
        return self.env.task.rew_buf, self.env.task.reset_buf

    def _bind(self, slot):
        self.env.task.engine.bind_obs_out(self.share_obs[slot])

    def _views(self, slot):
        L, idx, stream = _lib.for_device(self.env.task.engine.device)
        _lib.check(L.mms_marl_views(idx, ctypes.c_void_p(self.share_obs[slot].data_ptr()), ctypes.c_void_p(self.obs[slot].data_ptr()),
                                    self.N, self.A, self.env.num_ant_obs, self.env.shared_obs, stream), None, "mms_marl_views", L)

    def insert_step(self, rewards, dones, values, actions, action_log_probs):
        """Runner.insert for all agents (runner.py:222-255) without the per-agent copies of shared data.
        rewards [N], dones [N]; values [N, A]; actions / action_log_probs [N, A, act_dim] (or lists of A tensors)."""
        s = self.step
        if isinstance(actions, (list, tuple)):
            actions = torch.stack(list(actions), 1)
            action_log_probs = torch.stack(list(action_log_probs), 1)
        self.rewards[s].copy_(rewards.view(-1, 1))
        self.masks[s + 1].copy_((dones == 0).to(torch.float32).view(-1, 1))        # dones are per env: dones_env == dones
        # rows that GroupedPolicyInference.collect_into already wrote in place are recognised by their address and not copied again
        def put(dst, src):
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src.reshape(dst.shape))
        put(self.value_preds[s], values)
        put(self.actions[s], actions)
        put(self.action_log_probs[s], action_log_probs)
        self.step = (s + 1) % self.T

    def finished_episode_rewards(self, running, reward_env, dones_env):
        """The per-env Python loop of runner.py:141-144 as tensor ops without a host sync: adds this step's reward to
        `running` [N], returns (sum, count) of the returns of episodes that ended, and zeroes them in `running`."""
        running += reward_env
        done = dones_env != 0
        total, count = (running * done).sum(), done.sum()
        running.mul_((~done).to(running.dtype))
        return total, count

    def all_gather(self, group=None):
        """All ranks' rollouts in global env order (see all_gather_envs): dict of the buffers a trainer reads."""
        return all_gather_envs({k: getattr(self, k) for k in GATHERED}, group)

    # -- learner side --------------------------------------------------------------------------------------------
    def after_update(self):
        for t in (self.share_obs, self.obs, self.rnn_states, self.rnn_states_critic, self.masks, self.bad_masks, self.active_masks):
            t[0].copy_(t[-1])

    def compute_returns(self, next_values, value_normalizers=None):
        """next_values [N, A]; value_normalizers: list of A PopArt / ValueNorm objects (or None)."""
        self.value_preds[-1].copy_(next_values.reshape(self.N, self.A))
        dev = self.device
        if self._use_norm:
            mv = [n.running_mean_var() for n in value_normalizers]
            mean = torch.stack([m.reshape(()) for m, _ in mv]).to(dev).float().contiguous()
            var = torch.stack([v.reshape(()) for _, v in mv]).to(dev).float().contiguous()
        else:
            mean = var = self.rewards
        L, idx, stream = _lib.for_device(dev)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(L.mms_gae_marl_agents(idx, p(self.rewards), p(self.value_preds), p(self.masks), p(self.returns), self.T, self.N, self.A,
                                         float(self.gamma), float(self.gae_lambda), 1 if self._use_norm else 0, p(mean), p(var), stream),
                   None, "mms_gae_marl_agents", L)
