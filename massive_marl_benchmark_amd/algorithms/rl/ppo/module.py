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
        self._wplanes = None    # id(Linear) -> (version tag, P32 planes of its weight, the Linear)
        self._split_bufs = None
        self._nets = None       # (actor hidden Linears, critic hidden Linears)
        self._h16 = None        # the f16x2 layers' weight-side state (planes, scales, bound chain): _h16_state
        self._addr_tag = None   # addresses / version counters of the parameters the derived buffers were built from
        self._ver_tag = None
        self._calls = None      # prebuilt launch lists of the f16x2 hidden layers, per (buffers, input addresses)
        self._sample_calls = None
        self._qualify = None
        self._step_engine = None
        self._head_wt = None                 # (padded rows, tiles): the actor's last layer stored for coalesced operand loads (_head_tiles)

    _DERIVED = ("_counters", "_side", "_trunk", "_act_bufs", "_value_bufs", "_one_bufs", "_wplanes", "_split_bufs", "_nets", "_h16", "_addr_tag",
                "_ver_tag", "_calls", "_sample_calls", "_qualify", "_bound", "_step_engine", "_head_wt")

    def __deepcopy__(self, memo):
        """A copy starts without derived state: the caches hold device addresses of THIS module's parameters and buffers."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in self._DERIVED else copy.deepcopy(v, memo)
        return new

    @staticmethod
    def init_weights(sequential, scales):
        for idx, module in enumerate(m for m in sequential if isinstance(m, nn.Linear)):
            torch.nn.init.orthogonal_(module.weight, gain=scales[idx])

    # Hidden layers of both networks through mms_linear2_act (own fp32-MFMA GEMM, bias + ELU in the epilogue, one launch per layer
    # for both networks) and both last layers inside the sampling kernel (mms_ppo_heads_act): five launches per `act`.  Used when
    # the networks qualify (fp32, ELU, actor and critic of the same hidden shapes, last hidden width a multiple of 64); otherwise
    # the library path below.  Rollout step at 4096 envs: 350 us, against 361 us for the library GEMMs + separate ELU passes with the
    # critic on a second stream (profiles/r01_v12_bench.json, r01_v12_bench_library_gemms.json).  With fuse_layers off, fuse_head alone puts only the actor's last layer into
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
    def bind_rollout(self, storage=None, actions_out=None, step_engine=None):
        """Optional zero-copy destinations: `storage` (a RolloutStorage: `act` then writes actions / log-prob / value /
        mu / sigma into slot `storage.step` and returns views of those slots, which `add_transitions` recognises and does
        not copy again) and `actions_out` (e.g. the engine's "actions" buffer).
        step_engine (with a storage; optional): an Engine whose NEXT step() follows every `act` -- the output heads and the sampling tail
        then run in that step kernel's prologue (Engine.bind_policy_head / mms_bind_policy_head: the same instruction sequence as
        mms_ppo_heads_act, one launch and one memory round trip less per rollout step) and `act` returns the slot views BEFORE they are
        filled: valid for whoever reads them in stream order behind the step (add_transitions, the learner), not for host code between
        `act` and `step`.  Used where the engine has the layout for it (Engine.takes_policy_head()); elsewhere the heads kernel runs."""
        self._bound = (storage, actions_out)
        self._step_engine = step_engine if (storage is not None and step_engine is not None and step_engine.takes_policy_head()) else None

    def _fp32_layers_qualify(self):
        """mms_linear2_act applies: fp32 ELU networks, actor and critic with the same hidden shapes, input widths multiples of 4."""
        dt = self.log_std.dtype
        if self._qualify is not None and self._qualify[0] == dt:
            return self._qualify[1]
        a_lin = [m for m in self.actor if isinstance(m, nn.Linear)]
        c_lin = [m for m in self.critic if isinstance(m, nn.Linear)]
        acts = [m for m in list(self.actor) + list(self.critic) if not isinstance(m, nn.Linear)]
        ok = (len(a_lin) == len(c_lin) and len(a_lin) >= 2 and all(isinstance(m, nn.ELU) and m.alpha == 1.0 for m in acts)
              and all(la.weight.shape == lc.weight.shape and la.weight.dtype == torch.float32 and la.in_features % 4 == 0
                      and la.bias is not None and lc.bias is not None for la, lc in zip(a_lin[:-1], c_lin[:-1])))
        self._qualify = (dt, ok)
        return ok

    # -- split-operand layers ------------------------------------------------------------------
    @staticmethod
    def _p32_bytes(rows, K):
        return rows * ((K + 31) // 32) * 192               # MMS_P32_BYTES (include/mms.h)

    def _weight_planes(self, lin, L, idx, stream, force=False):
        """P32 planes of a Linear layer's weight (a buffer allocated once per layer), re-split when the parameter has changed (its version
        counter moves with every in-place optimizer step), has moved, or refresh() asks (force)."""
        if self._wplanes is None:
            self._wplanes = {}
        w = lin.weight.detach()
        tag = (w._version, w.data_ptr(), str(w.device))
        hit = self._wplanes.get(id(lin))
        if force or hit is None or hit[0] != tag:
            planes = hit[1] if hit is not None and hit[1].device == w.device else \
                torch.empty(self._p32_bytes(lin.out_features, lin.in_features), dtype=torch.uint8, device=w.device)
            wc = w.contiguous()
            _lib.check(L.mms_split_planes(idx, lin.out_features, lin.in_features, 0, ctypes.c_void_p(wc.data_ptr()),
                                          ctypes.c_void_p(planes.data_ptr()), stream), None, "mms_split_planes", L)
            self._wplanes[id(lin)] = (tag, planes, lin)
            return planes
        return hit[1]

    @staticmethod
    def _h32_bytes(rows, K):
        return rows * ((K + 31) // 32) * 128               # MMS_H32_BYTES (include/mms.h)

    # -- two scaled fp16 planes per operand: everything derived from the parameters lives at a STABLE address and is rebuilt by device
    #    work only (refresh), so that a hipGraph of a rollout that starts with refresh() follows every optimizer step ---------------------
    def _networks(self):
        """(hidden Linear layers of the actor, of the critic): every Linear but the last of each Sequential (cached; the modules'
        structure is fixed after construction)."""
        if self._nets is None:
            self._nets = [[m for m in net if isinstance(m, nn.Linear)][:-1] for net in (self.actor, self.critic)]
        return self._nets

    def _param_tags(self):
        """(version counters, addresses) of every parameter a derived buffer or a cached launch depends on."""
        ps = [q for net in self._networks() for l in net for q in (l.weight, l.bias) if q is not None]
        ps += [q for q in (self.actor[-1].weight, self.actor[-1].bias, self.critic[-1].weight, self.critic[-1].bias, self.log_std) if q is not None]
        return tuple(q._version for q in ps), tuple(q.data_ptr() for q in ps)

    def _h16_state(self, dev):
        """The weights' side of the f16x2 layers on device `dev`: per hidden Linear its H32 planes, row scales and row 1-norms, and per
        network the bound chain's (mult, add) pairs -- one flat tensor [actor L x 2 | critic L x 2], so that the chains of networks that
        share an input are one contiguous [nchains, L, 2] block the split kernel reads where it lies.  Allocated once; refreshed in place."""
        st = self._h16
        if st is not None and st["dev"] == dev:
            return st
        L, idx, _ = _lib.for_device(dev)
        nets = self._networks()
        recs, off, slices = [], 0, []
        for net in nets:
            rs = []
            for l in net:
                assert l.weight.dtype == torch.float32 and l.weight.is_contiguous() and l.bias is not None and l.weight.device == dev
                N, K = l.out_features, l.in_features
                rs.append({"lin": l, "N": N, "K": K, "planes": torch.empty(self._h32_bytes(N, K), dtype=torch.uint8, device=dev),
                           "scale": torch.empty(N, device=dev), "inv": torch.empty(N, device=dev), "l1": torch.empty(N, device=dev)})
            recs.append(rs)
            slices.append((off, max(len(net) - 1, 0)))
            off += 2 * max(len(net) - 1, 0)
        bounds = torch.zeros(max(off, 2), device=dev)
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        # the refresh launches, arguments prebuilt: the planes, row scales and row 1-norms of ALL hidden layers' weights in one launch ...
        flat = [r for rs in recs for r in rs]
        calls = []
        if flat:
            calls.append((L.mms_weight_planes16_group, (idx, len(flat), (ctypes.c_int64 * len(flat))(*[r["N"] for r in flat]),
                                                        (ctypes.c_int32 * len(flat))(*[r["K"] for r in flat]), arr([r["lin"].weight for r in flat]),
                                                        arr([r["planes"] for r in flat]), arr([r["scale"] for r in flat]), arr([r["inv"] for r in flat]),
                                                        arr([r["l1"] for r in flat])), "mms_weight_planes16_group"))
        st = {"dev": dev, "L": L, "idx": idx, "recs": recs, "bounds": bounds, "slices": slices, "calls": calls, "given": {}, "chain_args": {}}
        self._h16 = st
        return st

    def _chain_args(self, st, members):
        """(nchains, L, l1 pointers, bias pointers, counts, chain) of mms_chain_refresh16 for the chains of networks `members`."""
        members = tuple(members)
        hit = st["chain_args"].get(members)
        if hit is None:
            chain, Lh = self._chain(st, list(members))
            ent = [st["recs"][g][li] for g in members for li in range(Lh)]
            arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
            hit = (len(members), Lh, arr([r["l1"] for r in ent]), arr([r["lin"].bias for r in ent]), (ctypes.c_int32 * len(ent))(*[r["N"] for r in ent]),
                   ctypes.c_void_p(chain.data_ptr())) if Lh else None
            st["chain_args"][members] = hit
        return hit

    def _chain(self, st, members):
        """The contiguous [nchains, L, 2] block of the bound chains of networks `members` (indices into (actor, critic)), or None when
        they have a single hidden layer (nothing is stored as planes)."""
        o0, Lh = st["slices"][members[0]]
        if Lh == 0:
            return None, 0
        for j, g in enumerate(members):
            assert st["slices"][g] == (o0 + 2 * Lh * j, Lh), "networks that share an input must be adjacent and equally deep"
        return st["bounds"][o0: o0 + 2 * Lh * len(members)], Lh

    def refresh(self):
        """Rebuild everything the fused paths derive from the parameters -- the hidden layers' operand planes, their row scales, the bound
        chain and the hidden activations' scales of the given-planes path -- from the parameters as they are NOW.  Device work only, on
        the caller's current stream, into buffers whose addresses never change: it may be captured at the head of a hipGraph of a rollout
        (bench.py does), and a replay after an optimizer step then computes with the updated parameters.

        When it runs by itself: at the first `act` of every rollout when a RolloutStorage is bound (bind_rollout: storage.step == 0),
        unconditionally -- a parameter update of ANY kind between two rollouts is followed, also one through `param.data`, which moves no
        version counter (agents/algorithms/marl/hatrpo_trainer.py:122 updates that way; PPO's optimizer.step() does not); on every other
        call of act / value when a parameter's version counter or address has changed.  Call it yourself after writing parameters through
        `.data` without a bound storage, and before replaying a captured graph that does not contain it."""
        dev = self.log_std.device
        vers, addrs = self._param_tags()
        if self._addr_tag is not None and self._addr_tag != (addrs, str(dev)):
            # the parameters moved (.to(device), a new storage): every cached address is void
            self._h16, self._wplanes, self._split_bufs, self._calls, self._sample_calls = None, None, None, None, None
            self._head_wt = None
        self._addr_tag, self._ver_tag = (addrs, str(dev)), vers
        if self._head_wt is not None:
            self._fill_head_tiles()
        st = self._h16
        if st is not None:
            stream = _lib.for_device(st["dev"])[2]
            for fn, args, what in st["calls"]:
                _lib.check(fn(*args, stream), None, what, st["L"])
            # ... then the bound chain: one launch per set of constant-bound scales (it also stores the chain entries of its
            # networks), and one without rows for networks no such set covers
            covered = set()
            for (members, rows, scale), (_, cs, ci, alias) in st["given"].items():
                ca = self._chain_args(st, members)
                if ca is None or alias:
                    continue
                _lib.check(st["L"].mms_chain_refresh16(st["idx"], *ca, 16384.0 / scale, rows, ctypes.c_void_p(cs.data_ptr()), ctypes.c_void_p(ci.data_ptr()), stream),
                           None, "mms_chain_refresh16", st["L"])
                covered.update(members)
            depths = {Lh for _, Lh in st["slices"]}
            rest = [tuple(range(len(st["recs"])))] if len(depths) == 1 else [(g,) for g in range(len(st["recs"]))]
            for grp in rest:
                if not set(grp) <= covered:
                    ca = self._chain_args(st, grp)
                    if ca is not None:
                        _lib.check(st["L"].mms_chain_refresh16(st["idx"], *ca, 0.0, 0, None, None, stream), None, "mms_chain_refresh16", st["L"])
        if self._wplanes:                                # the three-plane format's weights (no scales, no chain)
            for key in list(self._wplanes):
                tag, planes, lin = self._wplanes[key]
                if self.split_format == "bf16x3":
                    self._weight_planes(lin, *_lib.for_device(lin.weight.device), force=True)
                else:
                    self._wplanes[key] = (None, planes, lin)    # not in use now: re-split at their next use

    def _head_tiles(self):
        """The actor's last layer once more, stored [ceil(A / 16)][H / 4][16][4] (struct mms_policy_head.weight_tiles, csrc/head_block.h:
        one operand load of the fused head's matrix phase then reads 1 KB of contiguous memory instead of 16 rows x 64 B): a derived
        buffer at a stable address, rebuilt by refresh() with two device copies."""
        if self._head_wt is None:
            last = self.actor[-1]
            A, H = last.out_features, last.in_features
            nct = (A + 15) // 16
            pad = torch.zeros(nct * 16, H, device=last.weight.device, dtype=torch.float32)
            tiles = torch.empty(nct * (H // 4) * 64, device=last.weight.device, dtype=torch.float32)
            self._head_wt = (pad, tiles)
            self._fill_head_tiles()
        return self._head_wt[1]

    def _fill_head_tiles(self):
        pad, tiles = self._head_wt
        last = self.actor[-1]
        A, H = last.out_features, last.in_features
        with torch.no_grad():
            if A % 16 == 0:                                          # one strided copy (TenAnt: 80 outputs = five whole column tiles)
                tiles.view(-1, H // 4, 16, 4).copy_(last.weight.detach().view(-1, 16, H // 4, 4).permute(0, 2, 1, 3))
            else:
                pad[:A].copy_(last.weight.detach())
                tiles.view(-1, H // 4, 16, 4).copy_(pad.view(-1, 16, H // 4, 4).permute(0, 2, 1, 3))

    def _ensure_fresh(self):
        """refresh() when a parameter's version counter or address has moved since the derived buffers were built (see refresh for what
        that does and does not see), and at the first act of a bound rollout."""
        storage = self._bound[0] if self._bound is not None else None
        if storage is not None and storage.step == 0:
            return self.refresh()
        vers, addrs = self._param_tags()
        if self._ver_tag != vers or self._addr_tag != (addrs, str(self.log_std.device)):
            self.refresh()

    def _split_hidden16(self, nets, inputs, tag, planes=None):
        """As _split_hidden on two scaled fp16 planes per operand (mms_split_planes16_group / mms_linear_group_act_split16).  The
        kernel that splits a network's input also evaluates, per row, the scales of the hidden activations behind it from the
        layers' bound chain.  Buffers per (tag, shapes), launch arguments per (buffers, input addresses): built once, then a call is a
        handful of prebuilt ctypes launches."""
        self._ensure_fresh()
        x0 = inputs[0]
        dev, M = x0.device, x0.shape[0]
        mine = self._networks()
        members = [next(j for j, n in enumerate(mine) if n and n[0] is net[0]) for net in nets]
        fresh = self._h16 is None or self._h16["dev"] != dev
        st = self._h16_state(dev)
        if fresh:
            self.refresh()
        G, nl = len(nets), len(nets[0])
        given = planes is not None and all(x.data_ptr() == x0.data_ptr() and x.shape == x0.shape for x in inputs)
        ckey = (tag, "h", M, tuple((x.data_ptr(), x.shape[1], x.stride(0)) for x in inputs), tuple(members),
                (planes[0].data_ptr(), float(planes[1])) if given else None)
        if self._calls is None:
            self._calls = {}
        hit = self._calls.get(ckey)
        if hit is None:
            hit = self._build_calls16(st, members, inputs, tag, planes if given else None)
            self._calls[ckey] = hit
        calls, out = hit
        stream = _lib.for_device(dev)[2]
        for fn, args, what in calls:
            _lib.check(fn(*args, stream), None, what, st["L"])
        return out

    def _build_calls16(self, st, members, inputs, tag, planes):
        x0 = inputs[0]
        dev, M = x0.device, x0.shape[0]
        L, idx = st["L"], st["idx"]
        recs = [st["recs"][g] for g in members]
        G, nl = len(recs), len(recs[0])
        key = (tag, "h", M, tuple(x.shape[1] for x in inputs), str(dev), tuple(members))
        if self._split_bufs is None:
            self._split_bufs = {}
        bufs = self._split_bufs.get(key)
        f32 = lambda *sh: torch.empty(*sh, device=dev)
        u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
        if bufs is None:
            bufs = {"x": [u8(self._h32_bytes(M, x.shape[1])) for x in inputs], "xs": [f32(M) for _ in range(G)], "xi": [f32(M) for _ in range(G)],
                    "cs": [f32(G, max(nl - 1, 1), M) for _ in range(G)], "ci": [f32(G, max(nl - 1, 1), M) for _ in range(G)],
                    "h": [[u8(self._h32_bytes(M, r["N"])) for _ in range(G)] for r in recs[0][:-1]],
                    "out": [f32(M, recs[0][-1]["N"]) for _ in range(G)], "keep": []}
            self._split_bufs[key] = bufs
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
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
        calls = []
        x_planes = {}
        for grp in users:
            g0 = grp[0]
            x = inputs[g0]
            assert x.dtype == torch.float32 and x.stride(1) == 1
            chain, Lh = self._chain(st, [members[g] for g in grp])
            nch = len(grp) if Lh else 0
            if planes is not None:
                # rows bounded by 2^14 / scale (the engine clamps them to clip_obs <= that): constant scales from the kernels' own
                # recurrence (mms_chain_scales16), re-evaluated by refresh(); the buffers belong to (networks, rows, scale)
                pl, scale = planes
                scale = float(scale)
                assert pl.dtype == torch.uint8 and pl.numel() == self._h32_bytes(M, x.shape[1]) and pl.device == dev
                gm = tuple(members[g] for g in grp)
                gkey = (gm, M, scale)
                if gkey not in st["given"]:
                    # a subset of networks of an existing set with the same rows and scale (the critic alone, for `value`, beside the set of
                    # actor + critic of `act`) shares that set's buffers: one refresh launch serves both
                    host = next(((k, v) for k, v in st["given"].items() if k[1:] == (M, scale) and not v[3] and
                                 any(k[0][j:j + len(gm)] == gm for j in range(len(k[0]) - len(gm) + 1))), None)
                    if host is not None and nch:
                        j = next(j for j in range(len(host[0][0]) - len(gm) + 1) if host[0][0][j:j + len(gm)] == gm)
                        st["given"][gkey] = (host[1][0], host[1][1][j:j + len(gm)], host[1][2][j:j + len(gm)], True)
                    else:
                        xi = torch.full((M,), 1.0 / scale, device=dev)
                        cs, ci = f32(max(nch, 1), max(Lh, 1), M), f32(max(nch, 1), max(Lh, 1), M)
                        st["given"][gkey] = (xi, cs, ci, False)
                        if nch:
                            _lib.check(L.mms_chain_refresh16(idx, *self._chain_args(st, gm), 16384.0 / scale, M, ctypes.c_void_p(cs.data_ptr()),
                                                             ctypes.c_void_p(ci.data_ptr()), _lib.for_device(dev)[2]), None, "mms_chain_refresh16", L)
                xi, cs, ci, _ = st["given"][gkey]
                x_planes[g0] = (pl, xi, cs, ci)
                bufs["keep"].append(pl)
                continue
            x_planes[g0] = (bufs["x"][g0], bufs["xi"][g0], bufs["cs"][g0], bufs["ci"][g0])
            calls.append((L.mms_split_planes16_group, (idx, 1, M, x.shape[1], x.stride(0), arr([x]), arr([bufs["x"][g0]]), arr([bufs["xs"][g0]]), arr([bufs["xi"][g0]]),
                                                       nch, Lh if nch else 0, arr([chain]) if nch else None, arr([bufs["cs"][g0]]) if nch else None,
                                                       arr([bufs["ci"][g0]]) if nch else None, None, 0.0), "mms_split_planes16_group"))
            bufs["keep"].append(x)
        cur = [x_planes[src[g][0]][0] for g in range(G)]
        cur_inv = [x_planes[src[g][0]][1] for g in range(G)]
        for li in range(nl):
            last = li == nl - 1
            out = bufs["out"] if last else bufs["h"][li]
            ysc = None if last else arr([x_planes[src[g][0]][2][src[g][1], li] for g in range(G)])
            calls.append((L.mms_linear_group_act_split16, (idx, G, M, recs[0][li]["N"], recs[0][li]["K"], arr(cur), arr([recs[g][li]["planes"] for g in range(G)]),
                                                           arr([recs[g][li]["lin"].bias for g in range(G)]), arr(out), arr(cur_inv),
                                                           arr([recs[g][li]["inv"] for g in range(G)]), ysc, 1, 0 if last else 1, None, None, None, None, None, 0),
                          "mms_linear_group_act_split16"))
            if not last:
                cur, cur_inv = out, [x_planes[src[g][0]][3][src[g][1], li] for g in range(G)]
        return calls, bufs["out"]

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
        self._ensure_fresh()                                # (the weights' P32 planes are re-split by refresh() too: stable buffers)
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
        # a bound rollout slot with both heads in the kernel (the rollout's hot path): the launch arguments of slot s are built once
        ckey = None
        if storage is not None and mean is None and value is None and hidden.is_contiguous() and (vhidden is None or vhidden.is_contiguous()):
            la, lc = self.actor[-1], self.critic[-1]
            ckey = (storage.actions.data_ptr(), storage.values.data_ptr(), storage.step, hidden.data_ptr(), None if vhidden is None else vhidden.data_ptr(),
                    None if actions_out is None else actions_out.data_ptr(), N, la.weight.data_ptr(), la.bias.data_ptr(), lc.weight.data_ptr(),
                    lc.bias.data_ptr(), self.log_std.data_ptr(), self._counters.data_ptr(), None if self._step_engine is None else id(self._step_engine))
            if self._sample_calls is None:
                self._sample_calls = {}
            hit = self._sample_calls.get(ckey)
            if hit is not None:
                args, ret, head = hit
                if head is not None and self._step_engine is not None:
                    self._step_engine.bind_policy_head(head)         # the engine's next step() evaluates the heads and samples
                else:
                    _lib.check(L.mms_ppo_heads_act(*args, stream), None, "mms_ppo_heads_act", L)
                return ret
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
            args = (idx, p(hidden), p(last.weight.detach()), p(last.bias.detach()), last.in_features, p(value), p(vh),
                    None if vh is None else p(vlast.weight.detach()), None if vh is None else p(vlast.bias.detach()),
                    0 if vh is None else vlast.in_features, p(log_std), self.seed, p(self._counters), self.row_offset, 1,
                    p(actions_out), p(act), p(logp), p(val), p(mu), p(sigma), N, A)
            head = None
            eng = self._step_engine
            if (ckey is not None and eng is not None and vh is not None and value is None and self.log_std.dtype == torch.float32 and eng.device == dev
                    and last.in_features % 512 == 0 and A == eng.num_actions and N == eng.num_envs and self.row_offset == eng.config.env_offset):
                from ....model import MmsPolicyHead
                a_ = lambda t: None if t is None else t.data_ptr()
                head = MmsPolicyHead(a_(hidden), a_(last.weight), a_(last.bias), a_(vh), a_(vlast.weight), a_(vlast.bias), a_(log_std), a_(self._counters),
                                     a_(actions_out), a_(act), a_(logp), a_(val), a_(mu), a_(sigma), self.seed, self.row_offset, last.in_features,
                                     vlast.in_features, A, 1, a_(self._head_tiles()) if last.weight.dtype == torch.float32 else None)
            if head is not None:
                eng.bind_policy_head(head)
            else:
                _lib.check(L.mms_ppo_heads_act(*args, stream), None, "mms_ppo_heads_act", L)
            if ckey is not None and self.log_std.dtype == torch.float32:
                self._sample_calls[ckey] = (args, (act, logp.view(-1), val, mu, sigma), head)
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
