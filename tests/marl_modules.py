"""Test infrastructure: plain torch modules with the attribute names and state_dict keys of the reference's MAPPO / HAPPO Actor and
Critic (agents/algorithms/marl/actor_critic.py:10-69, 118-155; utils/mlp.py:5-65; utils/act.py:21-23; utils/distributions.py:94-117),
so that the reference's state_dicts (tests/golden/marl_policy_fixture.npz) load into them unchanged, and `torch_forward`, the fp32
torch statement of their forward pass that the grouped HIP operators are compared with.  Not a product path."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class MLPLayer(nn.Module):
    def __init__(self, input_dim, hidden, layer_N):
        super().__init__()
        self._layer_N = layer_N
        self.fc1 = nn.Sequential(nn.Linear(input_dim, hidden), nn.ELU(), nn.LayerNorm(hidden))
        self.fc2 = nn.ModuleList([nn.Sequential(nn.Linear(hidden, hidden), nn.ELU(), nn.LayerNorm(hidden)) for _ in range(layer_N)])


class MLPBase(nn.Module):
    def __init__(self, input_dim, hidden, layer_N):
        super().__init__()
        self._use_feature_normalization = True
        self.feature_norm = nn.LayerNorm(input_dim)
        self.mlp = MLPLayer(input_dim, hidden, layer_N)


class DiagGaussian(nn.Module):
    def __init__(self, hidden, act_dim, std_x_coef=1.0, std_y_coef=0.5):
        super().__init__()
        self.std_x_coef, self.std_y_coef = std_x_coef, std_y_coef
        self.fc_mean = nn.Linear(hidden, act_dim)
        self.log_std = nn.Parameter(torch.ones(act_dim) * std_x_coef)


class ACTLayer(nn.Module):
    def __init__(self, hidden, act_dim):
        super().__init__()
        self.action_out = DiagGaussian(hidden, act_dim)


class Actor(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden=512, layer_N=2):
        super().__init__()
        self._use_recurrent_policy = self._use_naive_recurrent_policy = False
        self.base = MLPBase(obs_dim, hidden, layer_N)
        self.act = ACTLayer(hidden, act_dim)


class Critic(nn.Module):
    def __init__(self, share_obs_dim, hidden=512, layer_N=2):
        super().__init__()
        self._use_recurrent_policy = self._use_naive_recurrent_policy = False
        self.base = MLPBase(share_obs_dim, hidden, layer_N)
        self.v_out = nn.Linear(hidden, 1)


def randomize(module, gen, scale=0.3):
    """Perturb every parameter (LayerNorm affines and biases start at 1 / 0, which would hide mistakes in their use)."""
    with torch.no_grad():
        for p in module.parameters():
            p.add_(scale * torch.randn(p.shape, generator=gen).to(p.device) * (p.abs().mean() + 0.1))


def base_forward(base, x):
    x = F.layer_norm(x, (x.shape[-1],), base.feature_norm.weight, base.feature_norm.bias, base.feature_norm.eps)
    for seq in [base.mlp.fc1] + list(base.mlp.fc2):
        lin, ln = seq[0], seq[2]
        x = F.layer_norm(F.elu(F.linear(x, lin.weight, lin.bias)), (lin.out_features,), ln.weight, ln.bias, ln.eps)
    return x


def torch_forward(actor, critic, obs, share_obs):
    """(mean, std, value) of one agent: Actor.forward's distribution parameters and Critic.forward's value."""
    with torch.no_grad():
        hd = actor.act.action_out
        mean = F.linear(base_forward(actor.base, obs), hd.fc_mean.weight, hd.fc_mean.bias)
        std = torch.sigmoid(hd.log_std / hd.std_x_coef) * hd.std_y_coef
        value = F.linear(base_forward(critic.base, share_obs), critic.v_out.weight, critic.v_out.bias)
    return mean, std, value


def log_prob(mean, std, actions):
    """FixedNormal.log_probs (utils/distributions.py:31-34): the per-dimension Normal log-density, [M, A] (the sum is commented out there)."""
    return torch.distributions.Normal(mean, std).log_prob(actions)
