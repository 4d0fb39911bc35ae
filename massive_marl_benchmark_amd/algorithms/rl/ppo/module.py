"""ActorCritic (agents/algorithms/rl/ppo/module.py:9-109) with the sampling tail of `act` as one HIP kernel.

Same constructor, attributes (`actor`, `critic`, `log_std`) and methods (`act`, `act_inference`, `evaluate`) as the
reference, so PPO (agents/algorithms/rl/ppo/ppo.py) uses it unchanged.  The MLPs stay torch modules (rocBLAS / hipBLASLt
GEMMs); what `act` does after them -- Gaussian sample, log-probability, and optionally the stores that
`RolloutStorage.add_transitions` would make -- is `mms_ppo_act`.

Reference semantics kept: `MultivariateNormal(mean, scale_tril=diag(exp(log_std) * exp(log_std)))` (module.py:76-77), i.e.
the scale is sigma squared, and the "sigma" that `act` / `evaluate` return is `log_std.repeat(N, 1)` (:87, :109).  The
noise stream is this build's counter-based generator (seed, global env row, per-row draw counter), not torch's Philox:
sampled actions differ from the reference's draw for the same torch seed, their distribution and log-probabilities do not.
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn

from .... import _lib


def get_activation(act_name):
    table = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "lrelu": nn.LeakyReLU, "tanh": nn.Tanh,
             "sigmoid": nn.Sigmoid}
    if act_name not in table:
        print("invalid activation function!")          # module.py:129-131 prints and returns None
        return None
    return table[act_name]()


def _mlp(in_dim, hidden, out_dim, activation):
    layers, d = [], in_dim
    for h in hidden:
        layers += [nn.Linear(d, h), activation]
        d = h
    layers.append(nn.Linear(d, out_dim))
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    def __init__(self, obs_shape, states_shape, actions_shape, initial_std, model_cfg, asymmetric=False, seed=0, row_offset=0):
        super().__init__()
        self.asymmetric = asymmetric
        if model_cfg is None:
            actor_hidden, critic_hidden, activation = [256, 256, 256], [256, 256, 256], get_activation("selu")
        else:
            actor_hidden, critic_hidden = model_cfg["pi_hid_sizes"], model_cfg["vf_hid_sizes"]
            activation = get_activation(model_cfg["activation"])
        self.actor = _mlp(obs_shape[0], actor_hidden, actions_shape[0], activation)
        self.critic = _mlp(states_shape[0] if asymmetric else obs_shape[0], critic_hidden, 1, activation)
        self.log_std = nn.Parameter(np.log(initial_std) * torch.ones(*actions_shape))
        self.init_weights(self.actor, [np.sqrt(2)] * len(actor_hidden) + [0.01])       # module.py:58-63
        self.init_weights(self.critic, [np.sqrt(2)] * len(critic_hidden) + [1.0])
        self.seed, self.row_offset = int(seed), int(row_offset)     # noise stream key; row_offset = global index of env 0
        self._counters = None
        self._bound = None
        self._side = None       # second HIP stream: the critic MLP runs beside the actor MLP (see act)
        self._trunk = None
        self._act_bufs = None
        self._value_bufs = None
        self._one_bufs = None
        self._wplanes = None    # id(Linear) -> (version tag, P32 planes of its weight)
        self._split_bufs = None

    @staticmethod
    def init_weights(sequential, scales):
        for idx, module in enumerate(m for m in sequential if isinstance(m, nn.Linear)):
            torch.nn.init.orthogonal_(module.weight, gain=scales[idx])

    # Hidden layers of both networks through mms_linear2_act (own fp32-MFMA GEMM, bias + ELU in the epilogue, one launch per layer
    # for both networks) and both last layers inside the sampling kernel (mms_ppo_heads_act): five launches per `act`.  Used when
    # the networks qualify (fp32, ELU, actor and critic of the same hidden shapes, last hidden width a multiple of 64); otherwise
    # the library path below.  Rollout step at 4096 envs: 350 us, against 361 us for the library GEMMs + separate ELU passes with the
    # critic on a second stream (profiles/r01_v12_*).  With fuse_layers off, fuse_head alone puts only the actor's last layer into
    # the sampling kernel (no gain next to the critic's GEMMs: 371.8 us against 365.5 us, profiles/r01_v8_rollout_ab.txt).
    fuse_layers = True
    # The hidden layers on the bf16 matrix pipe with fp32 operands carried as three bf16 planes (csrc/split_kernels.hip,
    # mms_linear_group_act_split): the same fp32 product -- an fp32 number IS the sum of its three planes, six plane products are kept
    # and accumulated in fp32, what is dropped is < 2^-25 of a product -- at ~1.8 x the rate of the exact-fp32 MFMA kernel and with a
    # THIRD of that kernel's error against the float64 product (tests/test_gpu_parity.py::test_split_layers_error).
    # Applies when the batch and every hidden width are multiples of 128; False = the exact-fp32 MFMA kernel (mms_linear2_act).
    split_layers = True
    split_min_tiles = None  # least number of 128 x 128 output tiles of the widest layer for the split path (None: one per CU)
    # Which planes: "f16x2" = two fp16 planes per operand under a power-of-two scale per row (csrc/split16_kernels.hip: three MFMA
    # products, 4 bytes per element; the operand is kept to 2^-22 instead of exactly, the layer's error against float64 is 0.4 x the
    # exact-fp32 kernel's, tests/test_gpu_parity.py::test_split16_layers_error; the scales come from a bound chain over the weights'
    # row norms and cannot overflow) or "bf16x3" = three bf16 planes, every operand exact, six products (0.32 x, ~1.5 x slower).
    split_format = "f16x2"
    fuse_head = True
    two_streams = True      # critic beside the actor on a second stream
    defer_value = False     # opt-in: `act` returns before the critic has finished; the owner calls join() before reading values

    def forward(self):
        raise NotImplementedError

    # -- fused sampling ----------------------------------------------------------------------
    def bind_rollout(self, storage=None, actions_out=None):
        """Optional zero-copy destinations: `storage` (a RolloutStorage: `act` then writes actions / log-prob / value /
        mu / sigma into slot `storage.step` and returns views of those slots, which `add_transitions` recognises and does
        not copy again) and `actions_out` (e.g. the engine's "actions" buffer)."""
        self._bound = (storage, actions_out)

    def _fp32_layers_qualify(self):
        """mms_linear2_act applies: fp32 ELU networks, actor and critic with the same hidden shapes, input widths multiples of 4."""
        a_lin = [m for m in self.actor if isinstance(m, nn.Linear)]
        c_lin = [m for m in self.critic if isinstance(m, nn.Linear)]
        acts = [m for m in list(self.actor) + list(self.critic) if not isinstance(m, nn.Linear)]
        return (len(a_lin) == len(c_lin) and len(a_lin) >= 2 and all(isinstance(m, nn.ELU) and m.alpha == 1.0 for m in acts)
                and all(la.weight.shape == lc.weight.shape and la.weight.dtype == torch.float32 and la.in_features % 4 == 0
                        and la.bias is not None and lc.bias is not None for la, lc in zip(a_lin[:-1], c_lin[:-1])))

    # -- split-operand layers ------------------------------------------------------------------
    @staticmethod
    def _p32_bytes(rows, K):
        return rows * ((K + 31) // 32) * 192               # MMS_P32_BYTES (include/mms.h)

    def _weight_planes(self, lin, L, idx, stream):
        """P32 planes of a Linear layer's weight, re-split when the parameter has changed (its version counter moves with every
        in-place optimizer step) or moved."""
        if self._wplanes is None:
            self._wplanes = {}
        w = lin.weight.detach()
        tag = (w._version, w.data_ptr(), str(w.device))
        hit = self._wplanes.get(id(lin))
        if hit is None or hit[0] != tag:
            planes = hit[1] if hit is not None and hit[1].device == w.device else \
                torch.empty(self._p32_bytes(lin.out_features, lin.in_features), dtype=torch.uint8, device=w.device)
            wc = w.contiguous()
            _lib.check(L.mms_split_planes(idx, lin.out_features, lin.in_features, 0, ctypes.c_void_p(wc.data_ptr()),
                                          ctypes.c_void_p(planes.data_ptr()), stream), None, "mms_split_planes", L)
            self._wplanes[id(lin)] = (tag, planes)
            return planes
        return hit[1]

    @staticmethod
    def _h32_bytes(rows, K):
        return rows * ((K + 31) // 32) * 128               # MMS_H32_BYTES (include/mms.h)

    def _weight_planes16(self, lin, L, idx, stream):
        """H32 planes of a Linear layer's weight, the inverse of its row scales, and the layer's entry of the bound chain
        (largest row 1-norm, largest |bias|: |act(W x + b)| <= that norm max|x| + that bias) as a device tensor [2]; refreshed when the
        weight or the bias has changed (version counters) or moved."""
        if self._wplanes is None:
            self._wplanes = {}
        w, b = lin.weight.detach(), lin.bias.detach()
        tag = (w._version, w.data_ptr(), b._version, b.data_ptr(), str(w.device))
        hit = self._wplanes.get(("h", id(lin)))
        if hit is None or hit[0] != tag:
            N, K = lin.out_features, lin.in_features
            fresh = hit is None or hit[1].device != w.device
            planes = torch.empty(self._h32_bytes(N, K), dtype=torch.uint8, device=w.device) if fresh else hit[1]
            scale = torch.empty(N, device=w.device) if fresh else hit[4]
            inv = torch.empty(N, device=w.device) if fresh else hit[2]
            wc = w.contiguous()
            one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
            _lib.check(L.mms_split_planes16_group(idx, 1, N, K, 0, one(wc), one(planes), one(scale), one(inv), 0, 0, None, None, None, None, 0.0, stream),
                       None, "mms_split_planes16_group", L)
            bound = torch.stack([wc.abs().sum(1).max(), b.abs().max()])
            hit = (tag, planes, inv, bound, scale)
            self._wplanes[("h", id(lin))] = hit
        return hit[1], hit[2], hit[3], hit[0]

    def _split_hidden16(self, nets, inputs, tag, planes=None):
        """As _split_hidden on two scaled fp16 planes per operand (mms_split_planes16_group / mms_linear_group_act_split16).  The
        kernel that splits a network's input also evaluates, per row, the scales of the hidden activations behind it from the
        layers' bound chain."""
        x0 = inputs[0]
        dev, M = x0.device, x0.shape[0]
        L, idx, stream = _lib.for_device(dev)
        G, nl = len(nets), len(nets[0])
        key = (tag, "h", M, tuple(x.shape[1] for x in inputs), str(dev), G, tuple(l.out_features for l in nets[0]))
        if self._split_bufs is None:
            self._split_bufs = {}
        bufs = self._split_bufs.get(key)
        f32 = lambda *sh: torch.empty(*sh, device=dev)
        if bufs is None:
            u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
            bufs = {"x": [u8(self._h32_bytes(M, x.shape[1])) for x in inputs], "xs": [f32(M) for _ in range(G)], "xi": [f32(M) for _ in range(G)],
                    "cs": [f32(G, max(nl - 1, 1), M) for _ in range(G)], "ci": [f32(G, max(nl - 1, 1), M) for _ in range(G)],
                    "h": [[u8(self._h32_bytes(M, l.out_features)) for _ in range(G)] for l in nets[0][:-1]],
                    "out": [f32(M, nets[0][-1].out_features) for _ in range(G)], "chain": {}}
            self._split_bufs[key] = bufs
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        wpl = [[self._weight_planes16(l, L, idx, stream) for l in net] for net in nets]
        # distinct inputs: each is split once, with one bound chain per network it feeds
        src, users = [], []                    # src[g] = (index of the input's first user, chain slot)
        for g, x in enumerate(inputs):
            same = next((h for h in range(g) if inputs[h].data_ptr() == x.data_ptr() and inputs[h].shape == x.shape), None)
            if same is None:
                users.append([g])
                src.append((g, 0))
            else:
                first = src[same][0]
                grp = next(u for u in users if u[0] == first)
                src.append((first, len(grp)))
                grp.append(g)
        given = planes is not None and len(users) == 1          # the caller's planes of the one input every network reads
        for grp in users:
            g0 = grp[0]
            x = inputs[g0]
            assert x.dtype == torch.float32 and x.stride(1) == 1
            nch = len(grp) if nl > 1 else 0
            chain = None
            if nch:
                ckey = tuple(wpl[g][li][3] for g in grp for li in range(nl - 1))
                hit = bufs["chain"].get(g0)
                if hit is None or hit[0] != ckey:
                    hit = (ckey, torch.stack([torch.stack([wpl[g][li][2] for li in range(nl - 1)]) for g in grp]).contiguous())
                    bufs["chain"][g0] = hit
                chain = hit[1]
            if given:
                # rows bounded by 2^14 / scale (the engine clamps them to clip_obs <= that): constant scales, evaluated here with the
                # kernel's own recurrence -- once per refresh of the weights
                pl, scale = planes
                assert pl.dtype == torch.uint8 and pl.numel() == self._h32_bytes(M, x.shape[1]) and pl.device == dev
                ckey = (float(scale), None if chain is None else hit[0])
                if bufs.get("given_key") != ckey:
                    bufs["xi"][g0].fill_(1.0 / float(scale))
                    if nch:
                        bound = torch.full((nch,), 16384.0 / float(scale), device=dev)
                        for li in range(nl - 1):
                            bound = (chain[:, li, 0] * bound + chain[:, li, 1]) * 1.001
                            sc = torch.exp2(14.0 - torch.frexp(bound)[1].float())
                            bufs["cs"][g0][:nch, li] = sc[:, None]
                            bufs["ci"][g0][:nch, li] = (1.0 / sc)[:, None]
                    bufs["given_key"] = ckey
                bufs["x_given"] = pl
                continue
            bufs["given_key"] = None
            _lib.check(L.mms_split_planes16_group(idx, 1, M, x.shape[1], x.stride(0), arr([x]), arr([bufs["x"][g0]]), arr([bufs["xs"][g0]]), arr([bufs["xi"][g0]]),
                                                  nch, nl - 1 if nch else 0, arr([chain]) if nch else None, arr([bufs["cs"][g0]]) if nch else None,
                                                  arr([bufs["ci"][g0]]) if nch else None, None, 0.0, stream), None, "mms_split_planes16_group", L)
        cur = [bufs["x_given"] if given else bufs["x"][src[g][0]] for g in range(G)]
        cur_inv = [bufs["xi"][src[g][0]] for g in range(G)]
        for li in range(nl):
            lins = [net[li] for net in nets]
            last = li == nl - 1
            out = bufs["out"] if last else bufs["h"][li]
            ysc = None if last else arr([bufs["cs"][src[g][0]][src[g][1], li] for g in range(G)])
            _lib.check(L.mms_linear_group_act_split16(idx, G, M, lins[0].out_features, lins[0].in_features, arr(cur), arr([wpl[g][li][0] for g in range(G)]),
                                                      arr([l.bias.detach() for l in lins]), arr(out), arr(cur_inv), arr([wpl[g][li][1] for g in range(G)]), ysc,
                                                      1, 0 if last else 1, None, None, None, None, None, 0, stream), None, "mms_linear_group_act_split16", L)
            if not last:
                cur, cur_inv = out, [bufs["ci"][src[g][0]][src[g][1], li] for g in range(G)]
        return bufs["out"]

    def _split_applies(self, M, lins, networks=2):
        """Shapes the split kernel takes (batch and widths multiples of 128) AND is worth taking: the widest layer must give each CU at
        least one 128 x 128 output tile (`split_min_tiles`: None = the device's CU count) -- below that the kernel's fixed cost per
        launch (first slice, epilogue: ~7-15 us) outweighs its faster k-steps and the exact-fp32 kernel's 64-row tiles are quicker
        (PPO demo at 4096 x [256, 128, 128]: 1.07 against 1.34 M env-steps/s end to end)."""
        if not (self.split_layers and M > 0 and M % 128 == 0 and all(l.out_features % 128 == 0 for l in lins)):
            return False
        need = self.split_min_tiles
        if need is None:
            dev = lins[0].weight.device
            need = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 1
        return max(networks * (M // 128) * (l.out_features // 128) for l in lins) >= need

    def _split_hidden(self, nets, inputs, tag, planes=None):
        """Hidden layers of the networks in `nets` (lists of their hidden Linear layers, the same shapes in every network), one
        mms_linear_group_act_split launch per layer for all of them.  `inputs`: one fp32 [M, K] tensor per network (the same tensor
        twice is split once).  Activations stay in the three-plane format between the layers; the last one leaves fp32 [M, H]."""
        if self.split_format == "f16x2":
            return self._split_hidden16(nets, inputs, tag, planes)
        assert self.split_format == "bf16x3", self.split_format
        x0 = inputs[0]
        dev, M, K = x0.device, x0.shape[0], x0.shape[1]
        L, idx, stream = _lib.for_device(dev)
        G, nl = len(nets), len(nets[0])
        key = (tag, M, K, str(dev), G, tuple(l.out_features for l in nets[0]))
        if self._split_bufs is None:
            self._split_bufs = {}
        bufs = self._split_bufs.get(key)
        if bufs is None:
            u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
            bufs = {"x": [u8(self._p32_bytes(M, K)) for _ in range(G)],
                    "h": [[u8(self._p32_bytes(M, l.out_features)) for _ in range(G)] for l in nets[0][:-1]],
                    "out": [torch.empty(M, nets[0][-1].out_features, device=dev) for _ in range(G)]}
            self._split_bufs[key] = bufs
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        cur = []
        for g, x in enumerate(inputs):
            same = next((h for h in range(g) if inputs[h].data_ptr() == x.data_ptr() and inputs[h].shape == x.shape), None)
            if same is not None:
                cur.append(cur[same])
                continue
            assert x.dtype == torch.float32 and x.stride(1) == 1
            _lib.check(L.mms_split_planes(idx, M, K, x.stride(0), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(bufs["x"][g].data_ptr()), stream),
                       None, "mms_split_planes", L)
            cur.append(bufs["x"][g])
        for li in range(nl):
            lins = [net[li] for net in nets]
            last = li == nl - 1
            out = bufs["out"] if last else bufs["h"][li]
            wp = [self._weight_planes(l, L, idx, stream) for l in lins]
            _lib.check(L.mms_linear_group_act_split(idx, G, M, lins[0].out_features, lins[0].in_features, arr(cur), arr(wp),
                                                    arr([l.bias.detach() for l in lins]), arr(out), 1, 0 if last else 1, None, None, None, None, None, 0, stream),
                       None, "mms_linear_group_act_split", L)
            cur = out
        return bufs["out"]

    def _fused_hidden(self, x, critic_in, planes=None):
        """Hidden layers of BOTH networks, one launch per layer (bias + ELU in the epilogue): mms_linear_group_act_split (three-plane
        operands on the bf16 pipe) when the shapes allow, else mms_linear2_act (fp32 MFMA).
        Returns (actor hidden, critic hidden) or None when the two MLPs are not ELU networks of identical hidden shapes."""
        a_lin = [m for m in self.actor if isinstance(m, nn.Linear)]
        c_lin = [m for m in self.critic if isinstance(m, nn.Linear)]
        if not self._fp32_layers_qualify():
            return None
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        dev = x.device
        L, idx, stream = _lib.for_device(dev)
        ha, hc = x.contiguous(), critic_in.contiguous()
        M = ha.shape[0]
        if self._split_applies(M, a_lin[:-1]) and ha.data_ptr() % 16 == 0 and hc.data_ptr() % 16 == 0:
            out = self._split_hidden([a_lin[:-1], c_lin[:-1]], [ha, hc], "act", planes)
            return out[0], out[1]
        # activations of the hidden layers: allocated once per (batch, device) and reused -- the eager path would otherwise take an
        # allocator round trip per layer and call
        key = (M, str(dev))
        if self._act_bufs is None or self._act_bufs[0] != key:
            self._act_bufs = (key, [(torch.empty(M, la.out_features, device=dev), torch.empty(M, la.out_features, device=dev))
                                    for la in a_lin[:-1]])
        for (la, lc), (ya, yc) in zip(zip(a_lin[:-1], c_lin[:-1]), self._act_bufs[1]):
            _lib.check(L.mms_linear2_act(idx, M, la.out_features, la.in_features, p(ha), p(la.weight.detach()), p(la.bias.detach()), p(ya),
                                         p(hc), p(lc.weight.detach()), p(lc.bias.detach()), p(yc), 1, stream), None, "mms_linear2_act", L)
            ha, hc = ya, yc
        return ha, hc

    def _hidden_one(self, net, x):
        """Hidden layers of ONE network through mms_linear2_act (bias + ELU in the epilogue); activations in persistent buffers."""
        lin = [m for m in net if isinstance(m, nn.Linear)][:-1]
        L, idx, stream = _lib.for_device(x.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        M = x.shape[0]
        key = (M, str(x.device), id(net))
        if self._one_bufs is None or self._one_bufs[0] != key:
            self._one_bufs = (key, [torch.empty(M, l.out_features, device=x.device) for l in lin])
        h = x
        for l, y in zip(lin, self._one_bufs[1]):
            _lib.check(L.mms_linear2_act(idx, M, l.out_features, l.in_features, p(h), p(l.weight.detach()), p(l.bias.detach()), p(y),
                                         None, None, None, None, 1, stream), None, "mms_linear2_act", L)
            h = y
        return h

    def _actor_pass(self, x):
        """The actor MLP up to what the sampling kernel takes: (mean, None) or, when the last Linear layer can run inside
        the kernel (fp32, in_features a multiple of 64, at most 128 actions), (None, hidden)."""
        last = self.actor[-1]
        if (self.fuse_head and x.is_cuda and isinstance(last, nn.Linear) and last.weight.dtype == torch.float32
                and last.in_features % 64 == 0 and last.out_features <= 128 and last.bias is not None):
            if self._trunk is None:
                self._trunk = [self.actor[:-1]]          # in a list: not registered as a second copy of the parameters
            return None, self._trunk[0](x)
        return self.actor(x), None

    def _sample(self, mean, value, hidden=None, vhidden=None):
        """mean [N, A] (or None with hidden [N, H]: the last actor layer runs in the kernel); value [N, 1], or None with vhidden
        [N, VH] (the last critic layer runs in the kernel too), or None alone (the caller stores the value itself).
        Returns act, logp, val, mu, sigma."""
        src = mean if mean is not None else hidden
        N, A = src.shape[0], self.log_std.shape[0]
        dev = src.device
        L, idx, stream = _lib.for_device(dev)              # "cuda": the HIP build; "cpu": the CPU build (the module lives where its owner put it)
        if self._counters is None or self._counters.numel() != N or self._counters.device != dev:
            self._counters = torch.zeros(N, dtype=torch.int64, device=dev)
        storage, actions_out = self._bound if self._bound is not None else (None, None)
        if storage is not None:
            s = storage.step
            act, logp, val = storage.actions[s], storage.actions_log_prob[s], storage.values[s]
            mu, sigma = storage.mu[s], storage.sigma[s]
        else:
            act, mu, sigma = (torch.empty(N, A, device=dev) for _ in range(3))
            logp, val = torch.empty(N, 1, device=dev), torch.empty(N, 1, device=dev)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        value = None if value is None else value.contiguous().float()
        log_std = self.log_std.detach().float().contiguous()
        if mean is not None:
            mean = mean.contiguous().float()
            _lib.check(L.mms_ppo_act(idx, p(mean), p(value), p(log_std), self.seed, p(self._counters), self.row_offset, 1,
                                     p(actions_out), p(act), p(logp), p(val), p(mu), p(sigma), N, A, stream),
                       None, "mms_ppo_act", L)
        else:
            last = self.actor[-1]
            hidden = hidden.contiguous()
            vlast = self.critic[-1]
            vh = None if vhidden is None else vhidden.contiguous()
            _lib.check(L.mms_ppo_heads_act(idx, p(hidden), p(last.weight.detach()), p(last.bias.detach()), last.in_features, p(value), p(vh),
                                           None if vh is None else p(vlast.weight.detach()), None if vh is None else p(vlast.bias.detach()),
                                           0 if vh is None else vlast.in_features, p(log_std), self.seed, p(self._counters), self.row_offset, 1,
                                           p(actions_out), p(act), p(logp), p(val), p(mu), p(sigma), N, A, stream),
                       None, "mms_ppo_heads_act", L)
        return act, logp.view(-1), val, mu, sigma

    def act(self, observations, states, obs_planes=None):
        """module.py:73-87.  obs_planes = (planes, scale), optional and not in the reference: the H32 operand planes of THESE observation
        rows as the engine wrote them beside the rows (Engine.bind_obs_planes / mms_bind_obs_planes16) -- the split layers then skip
        their own pass over the observation.  The caller vouches that the planes belong to `observations`; ignored on every other path."""
        with torch.no_grad():
            dtype = self.log_std.dtype                              # a bf16 copy of the module takes fp32 observations
            critic_in = (states if self.asymmetric else observations).to(dtype)
            if self.fuse_layers and observations.is_cuda and dtype == torch.float32 and not self.defer_value:
                hidden = self._fused_hidden(observations, critic_in, None if self.asymmetric else obs_planes)
                if hidden is not None:
                    ha, hc = hidden
                    la, lc = self.actor[-1], self.critic[-1]
                    if (self.fuse_head and la.in_features % 64 == 0 and la.out_features <= 128 and la.bias is not None
                            and lc.in_features % 4 == 0 and lc.out_features == 1 and lc.bias is not None):
                        return self._sample(None, None, hidden=ha, vhidden=hc)       # both heads + sampling in one launch
                    return self._sample(la(ha), lc(hc))
            if (self.fuse_layers and self.fuse_head and self.defer_value and observations.is_cuda and dtype == torch.float32
                    and self._fp32_layers_qualify() and self.actor[-1].in_features % 64 == 0 and self.actor[-1].out_features <= 128):
                # The action needs the ACTOR only: its layers run alone on this stream (mms_linear2_act, one network), the sampling
                # kernel and the env step follow at once, and the whole critic pass (`value`) runs beside them on the second stream;
                # `join()` is where the owner waits for the values (before the GAE).
                if self._side is None:
                    self._side = torch.cuda.Stream(observations.device)
                cur = torch.cuda.current_stream(observations.device)
                self._side.wait_stream(cur)                                 # the observation row is ready
                ha = self._hidden_one(self.actor, observations.contiguous())
                act, logp, val, mu, sigma = self._sample(None, None, hidden=ha, vhidden=None)
                with torch.cuda.stream(self._side):
                    val.copy_(self.value(critic_in))
                return act, logp, val, mu, sigma
            if not (observations.is_cuda and self.two_streams):
                mean, hidden = self._actor_pass(observations.to(dtype))
                return self._sample(mean, self.critic(critic_in), hidden)
            # The two MLPs are independent: the critic's GEMMs / activations go to a second stream, so its memory-bound
            # activation kernels overlap the actor's compute-bound GEMMs (fork / join edges when captured in a hipGraph).
            if self._side is None:
                self._side = torch.cuda.Stream(observations.device)
            cur = torch.cuda.current_stream(observations.device)
            self._side.wait_stream(cur)
            if not self.defer_value:
                with torch.cuda.stream(self._side):
                    value = self.critic(critic_in)
                mean, hidden = self._actor_pass(observations.to(dtype))
                cur.wait_stream(self._side)
                return self._sample(mean, value, hidden)
            # defer_value: nothing downstream of the action needs the value -- not the env step, not the next actor pass --
            # so the critic keeps running on its stream beside the sampling kernel and the env step; `join()` (called by the
            # owner before the values are read: GAE) is the only point where the current stream waits for it.
            mean, hidden = self._actor_pass(observations.to(dtype))
            act, logp, val, mu, sigma = self._sample(mean, None, hidden)
            with torch.cuda.stream(self._side):
                val.copy_(self.critic(critic_in))
            return act, logp, val, mu, sigma

    def join(self):
        """Make the current stream wait for deferred critic passes (see `defer_value`)."""
        if self._side is not None:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)

    def value(self, critic_in, obs_planes=None):
        """Not in the reference (ppo.py:163 calls `act` once more for the bootstrap value of a rollout and discards the action): the
        critic alone, [N, 1].  fp32 ELU critics on the GPU run their hidden layers through mms_linear2_act (one network; bias + ELU
        in the epilogue) and the 1-wide output layer as a matrix-vector product; anything else goes through the torch module."""
        with torch.no_grad():
            lin = [m for m in self.critic if isinstance(m, nn.Linear)]
            acts = [m for m in self.critic if not isinstance(m, nn.Linear)]
            x = critic_in
            ok = (self.fuse_layers and x.is_cuda and x.dtype == torch.float32 and len(lin) >= 2 and lin[-1].out_features == 1
                  and all(isinstance(m, nn.ELU) and m.alpha == 1.0 for m in acts)
                  and all(l.weight.dtype == torch.float32 and l.in_features % 4 == 0 and l.bias is not None for l in lin)
                  and lin[-1].in_features <= 1024)
            if not ok:
                return self.critic(x)
            L, idx, stream = _lib.for_device(x.device)
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            h, M = x.contiguous(), x.shape[0]
            if self._split_applies(M, lin[:-1], networks=1) and h.data_ptr() % 16 == 0:
                h = self._split_hidden([lin[:-1]], [h], "value", obs_planes)[0]
            else:
                key = (M, str(x.device))
                if self._value_bufs is None or self._value_bufs[0] != key:
                    self._value_bufs = (key, [torch.empty(M, l.out_features, device=x.device) for l in lin[:-1]])
                for l, y in zip(lin[:-1], self._value_bufs[1]):
                    _lib.check(L.mms_linear2_act(idx, M, l.out_features, l.in_features, p(h), p(l.weight.detach()), p(l.bias.detach()), p(y),
                                                 None, None, None, None, 1, stream), None, "mms_linear2_act", L)
                    h = y
            # the 1-wide output layer: one launch of the grouped heads operator without its LayerNorm (eps < 0) instead of a
            # library matrix-vector product + its output fill + a bias add
            out = torch.empty(M, 1, device=x.device)
            one = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
            w, b = lin[-1].weight.detach(), lin[-1].bias.detach()
            _lib.check(L.mms_marl_heads_act(idx, 1, M, lin[-1].in_features, one(h), one(w), one(w), one(w), one(b), (ctypes.c_int32 * 1)(1), None,
                                            one(out), None, None, None, 0, 0, -1.0, stream), None, "mms_marl_heads_act", L)
            return out

    def act_inference(self, observations):
        return self.actor(observations)

    def evaluate(self, observations, states, actions):
        """module.py:93-109 without building the [A, A] covariance: the diagonal Gaussian in closed form (differentiable)."""
        mean = self.actor(observations)
        scale_log = 2.0 * self.log_std                               # log of the scale_tril diagonal, sigma^2
        z = (actions - mean) * torch.exp(-scale_log)
        log_prob = (-0.5 * z * z - scale_log - 0.5 * math.log(2.0 * math.pi)).sum(-1)
        entropy = (0.5 + 0.5 * math.log(2.0 * math.pi) + scale_log).sum(-1).expand(mean.shape[0])
        value = self.critic(states if self.asymmetric else observations)
        return log_prob, entropy, value, mean, self.log_std.repeat(mean.shape[0], 1)
