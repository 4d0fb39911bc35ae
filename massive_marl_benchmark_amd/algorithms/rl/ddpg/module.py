"""MLPActorCritic of DDPG (agents/algorithms/rl/ddpg/module.py:5-61): deterministic tanh actor scaled to the action limit, one Q
network on cat(obs, act), Gaussian exploration noise clipped to the limit.  Same constructor, sub-module names (state_dict keys
`pi.pi.<i>.*`, `q.q.<i>.*`) and `act(obs, deterministic)` contract, so ddpg.py uses it unchanged.

On the HIP device the actor's layers run as `mms_linear2_act` launches (fp32 MFMA, bias + ReLU / tanh in the epilogue) when the
network qualifies (fp32, ReLU / ELU / Tanh / Identity activations, input width a multiple of 4) -- the collection loop of
BASELINE configs[2] is a chain of small kernels and the library path spends three launches per layer.  The exploration noise is
drawn on the device (the reference draws it on the host and copies it over, module.py:59); a different stream of normals,
same distribution.
"""
import ctypes

import torch
import torch.nn as nn

from .... import _lib
from ....engine import current_stream_ptr

_ACT_CODES = {nn.Identity: 0, nn.ELU: 1, nn.ReLU: 2, nn.Tanh: 3}


def mlp(sizes, activation, output_activation=nn.Identity):
    """Linear layers sizes[0] -> ... -> sizes[-1], `activation` between them and `output_activation` at the end (module.py:5-10)."""
    mods = []
    last = len(sizes) - 2
    for j, (fan_in, fan_out) in enumerate(zip(sizes[:-1], sizes[1:])):
        mods.append(nn.Linear(fan_in, fan_out))
        mods.append((output_activation if j == last else activation)())
    return nn.Sequential(*mods)


def fused_mlp_forward(seq, x):
    """`seq(x)` through mms_linear2_act, one launch per Linear + activation pair; None if `seq` does not qualify."""
    mods = list(seq)
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or len(mods) % 2 or torch.is_grad_enabled() and any(p.requires_grad for p in seq.parameters()):
        return None
    pairs = list(zip(mods[0::2], mods[1::2]))
    for lin, act in pairs:
        if not isinstance(lin, nn.Linear) or type(act) not in _ACT_CODES or lin.bias is None or lin.in_features % 4 or lin.weight.dtype != torch.float32:
            return None
        if isinstance(act, nn.ELU) and act.alpha != 1.0:
            return None
    dev = x.device
    L, idx, stream = _lib.for_device(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    h = x.contiguous()
    for lin, act in pairs:
        y = torch.empty(h.shape[0], lin.out_features, device=dev)
        _lib.check(L.mms_linear2_act(idx, h.shape[0], lin.out_features, lin.in_features, p(h), p(lin.weight.detach()), p(lin.bias.detach()), p(y),
                                     None, None, None, None, _ACT_CODES[type(act)], stream), None, "mms_linear2_act", L)
        h = y
    return h


class MLPActor(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden_sizes, activation, act_limit):
        super().__init__()
        self.pi = mlp([obs_dim, *hidden_sizes, act_dim], activation, nn.Tanh)
        self.act_limit = act_limit

    def forward(self, obs):
        out = fused_mlp_forward(self.pi, obs)
        if out is None:
            out = self.pi(obs)
        return out if self.act_limit == 1.0 else self.act_limit * out       # x 1.0 is exact: one launch less for the ant / helicopter tasks


class MLPQFunction(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden_sizes, activation):
        super().__init__()
        self.q = mlp([obs_dim + act_dim, *hidden_sizes, 1], activation)

    def forward(self, obs, act):
        return self.q(torch.cat([obs, act], dim=-1))


class MLPActorCritic(nn.Module):
    def __init__(self, observation_space, action_space, act_noise, device, hidden_sizes=(256, 256), activation=nn.ReLU):
        super().__init__()
        obs_dim, act_dim = observation_space.shape[0], action_space.shape[0]
        self.act_limit = action_space.high[0]
        self.act_noise = act_noise
        self.device = device
        self.pi = MLPActor(obs_dim, act_dim, hidden_sizes, activation, self.act_limit)
        self._build_q(obs_dim, act_dim, hidden_sizes, activation)

    def _build_q(self, obs_dim, act_dim, hidden_sizes, activation):
        self.q = MLPQFunction(obs_dim, act_dim, hidden_sizes, activation)

    def act(self, obs, deterministic=True):
        with torch.no_grad():
            a = self.pi(obs)
            if not deterministic:
                if a.is_cuda:      # mean + std * N(0, 1) in one launch, clamp in place (the collection loop is launch bound)
                    a = torch.normal(a, float(self.act_noise)).clamp_(-self.act_limit, self.act_limit)
                else:
                    a = torch.clamp(a + self.act_noise * torch.randn_like(a), -self.act_limit, self.act_limit)
        return a
