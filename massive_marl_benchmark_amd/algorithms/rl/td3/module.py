"""MLPActorCritic of TD3 (agents/algorithms/rl/td3/module.py): the DDPG one with twin Q networks `q1`, `q2` (:50-51)."""
from ..ddpg.module import MLPActor, MLPActorCritic as _DDPGActorCritic, MLPQFunction, fused_mlp_forward, mlp  # noqa: F401


class MLPActorCritic(_DDPGActorCritic):
    def _build_q(self, obs_dim, act_dim, hidden_sizes, activation):
        self.q1 = MLPQFunction(obs_dim, act_dim, hidden_sizes, activation)
        self.q2 = MLPQFunction(obs_dim, act_dim, hidden_sizes, activation)
