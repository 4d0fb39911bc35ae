"""Minibatch generators of the MARL rollout buffers (agents/algorithms/marl/utils/separated_buffer.py:170-428): what the
reference's trainers iterate over (`buffer.feed_forward_generator(advantages, num_mini_batch)` at
agents/algorithms/marl/mappo_trainer.py:216, happo / hatrpo / ippo likewise; the recurrent variants when
use_recurrent_policy / use_naive_recurrent_policy are set).

One mixin for both buffer classes: `SeparatedReplayBuffer` (its own tensors) and the per-agent views of
`SharedRolloutBuffers` (slices of the shared tensors).  Every generator yields the reference's tuple, in its order:
  share_obs, obs, rnn_states, rnn_states_critic, actions, value_preds, returns, masks, active_masks, old_action_log_probs,
  adv_targ, available_actions [, factor]            (factor only when the buffer has one: HAPPO's update_factor)
with the reference's shapes: [batch, dim] rows; recurrent variants time-major inside a batch with one rnn state per sequence.

Host-side index logic on device tensors (the learner is a caller; out of the hot path) -- no engine call here."""
import torch


def _rows(x):
    """[T, N, ...] -> [T * N, ...] (step-major, the reference's reshape(-1, ...))."""
    return x.reshape(-1, *x.shape[2:])


class MinibatchGenerators:
    # attributes the host class provides: share_obs, obs, rnn_states, rnn_states_critic [T+1, N, ...]; value_preds, returns,
    # masks, active_masks [T+1, N, 1]; actions, action_log_probs [T, N, A]; rewards [T, N, 1]; factor [T, N, 1] or None;
    # available_actions None or [T+1, N, A]; episode_length.

    def _pack(self, pick, adv, rnn_pick):
        """The reference's tuple from `pick(tensor [T(+1), N, ...], drop_last)` -> batch rows and `rnn_pick(tensor)`."""
        avail = None if self.available_actions is None else pick(self.available_actions, True)
        out = (pick(self.share_obs, True), pick(self.obs, True), rnn_pick(self.rnn_states), rnn_pick(self.rnn_states_critic),
               pick(self.actions, False), pick(self.value_preds, True), pick(self.returns, True), pick(self.masks, True),
               pick(self.active_masks, True), pick(self.action_log_probs, False), adv, avail)
        if self.factor is not None:
            out = out + (pick(self.factor, False),)
        return out

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None):
        """separated_buffer.py:170-226: a random permutation of the T * N transitions cut into num_mini_batch index sets."""
        T, N = self.rewards.shape[0:2]
        batch = T * N
        if mini_batch_size is None:
            assert batch >= num_mini_batch, ("PPO requires the number of processes (%d) * number of steps (%d) = %d to be greater than or "
                                             "equal to the number of PPO mini batches (%d)." % (N, T, batch, num_mini_batch))
            mini_batch_size = batch // num_mini_batch
        perm = torch.randperm(batch)
        adv_rows = None if advantages is None else advantages.reshape(-1, 1)
        for b in range(num_mini_batch):
            idx = perm[b * mini_batch_size:(b + 1) * mini_batch_size].to(self.rewards.device)
            pick = lambda x, drop_last: _rows(x[:-1] if drop_last else x)[idx]
            yield self._pack(pick, None if adv_rows is None else adv_rows[idx], lambda x: _rows(x[:-1])[idx])

    def naive_recurrent_generator(self, advantages, num_mini_batch):
        """separated_buffer.py:228-308: whole env trajectories; a batch = N / num_mini_batch envs in random order, rows time-major
        ([T, n, ...] flattened), one initial rnn state per env."""
        T, N = self.rewards.shape[0:2]
        assert N >= num_mini_batch, ("PPO requires the number of processes (%d) to be greater than or equal to the number of PPO mini "
                                     "batches (%d)." % (N, num_mini_batch))
        per = N // num_mini_batch
        perm = torch.randperm(N)
        for start in range(0, N, per):
            envs = perm[start:start + per].to(self.rewards.device)
            n = envs.numel()
            pick = lambda x, drop_last: (x[:-1] if drop_last else x)[:, envs].reshape(T * n, *x.shape[2:])
            yield self._pack(pick, advantages[:, envs].reshape(T * n, *advantages.shape[2:]), lambda x: x[0, envs])

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length):
        """separated_buffer.py:310-428: the env-major sequence of T * N transitions cut into chunks of data_chunk_length consecutive
        steps; a batch = a random set of chunks, stacked [chunk, step] and flattened in that order (the reference's `_flatten`
        of the stacked list), one rnn state per chunk (its first step).  (The reference casts with numpy's 3-argument
        `transpose`, which torch tensors do not have: its evident intent, `[T, N, ...] -> [N, T, ...]`, is what runs here.)"""
        T, N = self.rewards.shape[0:2]
        batch = T * N
        chunks = batch // data_chunk_length
        mb = chunks // num_mini_batch
        assert batch >= data_chunk_length, ("PPO requires the number of processes (%d) * episode length (%d) to be greater than or equal "
                                            "to the number of data chunk length (%d)." % (N, T, data_chunk_length))
        assert chunks >= 2, "need larger batch size"
        perm = torch.randperm(chunks)
        L = data_chunk_length
        dev = self.rewards.device
        steps = torch.arange(L, device=dev)
        cast = lambda x: x.transpose(0, 1).reshape(-1, *x.shape[2:])                  # [T, N, ...] -> [N * T, ...]
        for b in range(num_mini_batch):
            first = perm[b * mb:(b + 1) * mb].to(dev) * L
            idx = (first[:, None] + steps[None, :]).reshape(-1)                       # chunk-major, steps consecutive
            pick = lambda x, drop_last: cast(x[:-1] if drop_last else x)[idx]
            yield self._pack(pick, cast(advantages)[idx], lambda x: cast(x[:-1])[first])
