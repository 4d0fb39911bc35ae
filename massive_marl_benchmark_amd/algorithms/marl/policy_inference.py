"""All agents' `get_actions` of a MAPPO / HAPPO / IPPO collect step in a few grouped launches.

The reference's Runner.collect (agents/algorithms/marl/runner.py:186-216) walks the agents and calls, per agent,
`trainer.policy.get_actions(share_obs, obs, rnn_states, rnn_states_critic, masks)` -> Actor.forward
(agents/algorithms/marl/actor_critic.py:43-69) + Critic.forward (:137-155): about thirty small launches per agent, ten agents per
step for TenAnt.  The networks are independent and of the same shape, so here every stage runs ONCE for all of them:

    feature LayerNorm (utils/mlp.py:52-53, 59-60)     mms_layernorm_group   actors (obs, padded to a multiple of 4) | critics (share_obs)
    fc1: Linear + ELU (mlp.py:19-20)                  mms_linear_group_act  actors (K = obs) | critics (K = share_obs)
    its LayerNorm                                     mms_layernorm_group   all networks, in place
    fc2[i]: Linear + ELU + LayerNorm (mlp.py:26-27)   one mms_linear_group_act + one mms_layernorm_group (in place) each, all networks
                                                      -- or, for batch and hidden sizes that are multiples of 128, the LayerNorms BETWEEN
                                                      the layers folded into the layers on either side: the producing layer's epilogue
                                                      leaves row sums (mms_row_stats_group -> mean, rstd), the consuming layer evaluates
                                                      rstd (W~ h - mean s) + c; the normalised activations are never written
    last LayerNorm + fc_mean + sample + log-probs     mms_marl_heads_act    all networks (critics: v_out, no sampling)
      (utils/act.py:75-81, distributions.py:94-117;
       actor_critic.py:153)

`GroupedPolicyInference` takes the agents' Actor / Critic modules AS THEY ARE (the reference's classes, or anything with the same
attributes: `.base.feature_norm`, `.base.mlp.fc1`, `.base.mlp.fc2`, `.act.action_out.{fc_mean, log_std, std_x_coef, std_y_coef}`,
`.v_out`; any number of agents: more than sixteen run as chunks of sixteen).  Biases, LayerNorm affines and head weights are read in
place; DERIVED copies exist of the zero-padded first actor weights (46 -> 48 columns), the folded weights `W diag(gamma)` with their
`s` / `c` vectors, their operand planes, and the action standard deviations -- `refresh()` rebuilds them.  WHEN it runs by itself:
(1) at step 0 of every rollout in both collect forms (`collect(buffers, 0)`, `collect_into(shared)` with `shared.step == 0`),
unconditionally -- the trainers update between rollouts, and the reference's HATRPO does it through `params.data.copy_(...)`
(agents/algorithms/marl/hatrpo_trainer.py:122), which moves NO version counter; (2) in every other inference call when a parameter's
version counter or address has moved (an in-place optimizer step moves them).  What it cannot see: a write through `.data` outside a
collect loop -- call `refresh()` after it (tests/test_marl_policy.py pins all three).  Recurrent policies
(use_recurrent_policy / use_naive_recurrent_policy) and non-Box action spaces are not covered: the constructor raises, nothing falls back silently.

The noise stream is this build's counter-based generator (seed + agent, global env row, per-row draw counter), as in
rl/ppo/module.py: the sampled actions differ from torch's draw for the same torch seed, their distribution and the returned
log-probabilities do not.
"""
import ctypes

import torch
import torch.nn as nn

from ... import _lib


def _blocks(base):
    """[(Linear, LayerNorm)] of an MLPBase: fc1, then fc2[0 .. layer_N)."""
    mlp = base.mlp
    seqs = [mlp.fc1] + list(mlp.fc2)[:getattr(mlp, "_layer_N", len(mlp.fc2))]
    out = []
    for s in seqs:
        lin, act, ln = s[0], s[1], s[2]
        if not (isinstance(lin, nn.Linear) and isinstance(act, nn.ELU) and act.alpha == 1.0 and isinstance(ln, nn.LayerNorm) and len(s) == 3):
            raise NotImplementedError("GroupedPolicyInference: blocks must be Linear + ELU(alpha 1) + LayerNorm (utils/mlp.py:19-27), got %r" % (s,))
        if lin.bias is None or ln.weight is None or ln.bias is None:
            raise NotImplementedError("GroupedPolicyInference: Linear bias and LayerNorm affine parameters are required")
        out.append((lin, ln))
    return out


def _ptrs(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        if t is not None:
            if t.dtype not in (torch.float32, torch.int64) or not t.is_contiguous():
                raise ValueError("GroupedPolicyInference: operands must be contiguous float32")
            arr[i] = t.data_ptr()
    return arr


def _row_ptrs(tensors):
    """Pointer array + common row pitch (floats) of 2-D float32 views whose rows are dense (stride 1 inside a row): a contiguous
    [M, K] matrix, or agent k's rows `block[:, k]` of an [M, agents, K] block."""
    pitch = {t.stride(0) for t in tensors}
    if len(pitch) != 1 or any(t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32 for t in tensors):
        raise ValueError("GroupedPolicyInference: operands must be float32 [M, K] views with dense rows and one common row pitch")
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr, pitch.pop()


def _critic_halves(table, n):
    """For every pointer array over all 2n networks (actors first, critics behind them) the array of its critic half, as `key/c`."""
    for key, arr in list(table.items()):
        if len(arr) == 2 * n and "/c" not in key:
            table[key + "/c"] = (ctypes.c_void_p * n)(*list(arr)[n:])


class GroupedPolicyInference:
    def __init__(self, actors, critics, seed=0, row_offset=0, fold_layernorm=True, split_layers=True, split_format="f16x2"):
        if len(actors) != len(critics) or not actors:
            raise ValueError("one actor and one critic per agent")
        self._chunks = None
        self._plans = None
        self._plist = None
        if len(actors) > 16:
            # MMS_MAX_GROUPS = 32 networks per launch = 16 agents: more agents (the 100-ant swarm) run as chunks of sixteen, each its
            # own set of grouped launches.  The noise key of agent k stays seed + k however the agents are chunked.
            self.n = len(actors)
            self._ranges = [(lo, min(lo + 16, self.n)) for lo in range(0, self.n, 16)]
            self._chunks = [GroupedPolicyInference(actors[lo:hi], critics[lo:hi], seed=int(seed) + lo, row_offset=row_offset,
                                                   fold_layernorm=fold_layernorm, split_layers=split_layers, split_format=split_format)
                            for lo, hi in self._ranges]
            return
        for m in list(actors) + list(critics):
            if getattr(m, "_use_recurrent_policy", False) or getattr(m, "_use_naive_recurrent_policy", False):
                raise NotImplementedError("GroupedPolicyInference: recurrent policies are not covered (actor_critic.py:64-65)")
            if not getattr(m.base, "_use_feature_normalization", hasattr(m.base, "feature_norm")):
                raise NotImplementedError("GroupedPolicyInference: use_feature_normalization = False is not covered")
        self.actors, self.critics = list(actors), list(critics)
        self.n = len(actors)
        self.seed, self.row_offset = int(seed), int(row_offset)
        # The LayerNorms BETWEEN the hidden layers folded into the layers on either side (mms.h: ln_part_out / ln_stat_in): the
        # normalised activations are never written.  Used when the shapes allow (batch and hidden size multiples of 128).
        self.fold_layernorm = bool(fold_layernorm)
        # ... and, on top of the folds, every layer on the bf16 matrix pipe with fp32 operands carried as three bf16 planes
        # (csrc/split_kernels.hip: the same fp32 product, a third of the exact-fp32 MFMA kernel's error, ~1.8 x its rate), the actors'
        # feature LayerNorm folded like the critics', and the output heads finished from per-slot partial dot products that the LAST
        # hidden layer's epilogue leaves (mms_marl_heads_finish): the last activations are never written or re-read.
        self.split_layers = bool(split_layers)
        # The planes: "f16x2" (default) = two fp16 planes per operand under a power-of-two scale per row (csrc/split16_kernels.hip: three
        # products, 4 bytes per element, error against float64 still below the exact-fp32 kernel's); "bf16x3" = three exact bf16 planes.
        # Every layer here sits behind a LayerNorm, so the bound that fixes a hidden activation's scale does not depend on the data:
        # |W~ xhat + c| <= |W~ row|_2 sqrt(K) + |c| with xhat the normalised input (|xhat|_2 <= sqrt(K)).
        if split_format not in ("f16x2", "bf16x3"):
            raise ValueError("split_format: f16x2 or bf16x3")
        self.split_format = split_format
        self.a_blocks = [_blocks(a.base) for a in self.actors]
        self.c_blocks = [_blocks(c.base) for c in self.critics]
        depth = {len(b) for b in self.a_blocks + self.c_blocks}
        hidden = {lin.out_features for b in self.a_blocks + self.c_blocks for (lin, _) in b}
        if len(depth) != 1 or len(hidden) != 1:
            raise NotImplementedError("GroupedPolicyInference: all networks must have the same depth and hidden size")
        self.depth, self.hidden = depth.pop(), hidden.pop()
        heads = [a.act.action_out for a in self.actors]
        for hd in heads:
            if not (hasattr(hd, "fc_mean") and hasattr(hd, "log_std")):
                raise NotImplementedError("GroupedPolicyInference: only the DiagGaussian head (Box action spaces) is covered (utils/act.py:21-23)")
        self.obs_dim = {b[0][0].in_features for b in self.a_blocks}
        self.sobs_dim = {b[0][0].in_features for b in self.c_blocks}
        self.act_dim = {hd.fc_mean.out_features for hd in heads}
        if len(self.obs_dim) != 1 or len(self.sobs_dim) != 1 or len(self.act_dim) != 1:
            raise NotImplementedError("GroupedPolicyInference: agents must share observation / action widths")
        self.obs_dim, self.sobs_dim, self.act_dim = self.obs_dim.pop(), self.sobs_dim.pop(), self.act_dim.pop()
        if self.act_dim > 16 or self.hidden > 1024 or self.hidden % 4:
            raise NotImplementedError("GroupedPolicyInference: at most 16 actions, hidden size a multiple of 4 up to 1024")
        self.eps = float(self.actors[0].base.feature_norm.eps)
        self.device = self.actors[0].act.action_out.fc_mean.weight.device
        self._M, self._D, self.p, self._versions, self._plans = None, None, None, (), None
        self.refresh()

    # -- parameters ---------------------------------------------------------------------------------------------------------------
    def _derived(self):
        """Every copy DERIVED from the parameters, allocated ONCE (stable addresses: the launches of a captured rollout keep pointing at
        them) together with the prebuilt arguments of the launches that rebuild them -- refresh() is device work only:
          folded weights W~ = W diag(gamma) (f32, for the exact-fp32 fold path), their operand planes and inverse row scales, s = W~ 1,
          c = W beta + b, the rows' output bounds (mms_fold_planes16_group: the actors' and critics' first layers behind their feature
          LayerNorms, the H x H layers behind the hidden LayerNorms, the output heads behind the last one); per network and layer the
          power of two of the output rows (mms_fold_scales16_group); the three-plane format's planes of W~ (mms_split_planes_group);
          the zero-padded first weights of the unfolded path and the action standard deviations (torch._foreach_* into fixed buffers)."""
        if self._D is not None:
            return self._D
        dev, n, H, A, depth = self.device, self.n, self.hidden, self.act_dim, self.depth
        L, idx, _ = _lib.for_device(dev)
        z = lambda *sh, **k: torch.zeros(*sh, device=dev, **k)
        d = lambda t: t.detach()
        self.kp_a, self.kp_c = (self.obs_dim + 3) & ~3, (self.sobs_dim + 3) & ~3
        both = self.a_blocks + self.c_blocks
        split = self.split_layers and self.fold_layernorm and self.sobs_dim % 4 == 0 and H % 128 == 0
        h16 = self.split_format == "f16x2"
        pbytes = lambda N, K: N * ((K + 31) // 32) * (128 if h16 else 192)
        u8 = lambda G, N, K: [torch.empty(pbytes(N, K), dtype=torch.uint8, device=dev) for _ in range(G)]
        D = {"split": split, "w1_a": z(n, H, self.kp_a), "w1_c": z(n, H, self.kp_c), "std": z(n, A), "fold": {}, "calls": [], "scale_jobs": []}
        up = lambda ts: (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
        i64 = lambda v: (ctypes.c_int64 * len(v))(*v)
        i32 = lambda v: (ctypes.c_int32 * len(v))(*v)

        def fold_set(ws, gammas, betas, biases, N, K, want_wt, want_planes, want_rb):
            """buffers + one job description of a set of G matrices [N, K] behind LayerNorms (gamma, beta)"""
            G = len(ws)
            Nmax = max(N) if isinstance(N, (list, tuple)) else N
            st = {"G": G, "N": N, "K": K, "wt": z(G, Nmax, K) if want_wt else None, "s": z(G, Nmax), "c": z(G, Nmax), "rb": z(G, Nmax) if want_rb else None,
                  "planes": u8(G, N, K) if want_planes else None, "inv": z(G, N) if (want_planes and h16) else None,
                  "src": (ws, gammas, betas, biases)}
            return st

        def fold_call(sets):
            """ONE mms_fold_planes16_group launch over several sets (at most MMS_MAX_GROUPS matrices)"""
            Ns, Ks, w, gm, bt, bs, pl, iv, sv, cv, rb, wt = [], [], [], [], [], [], [], [], [], [], [], []
            for st in sets:
                ws, gammas, betas, biases = st["src"]
                for g in range(st["G"]):
                    Ns.append(st["N"] if not isinstance(st["N"], (list, tuple)) else st["N"][g])
                    Ks.append(st["K"])
                    w.append(d(ws[g])); gm.append(d(gammas[g])); bt.append(d(betas[g])); bs.append(d(biases[g]))
                    h32 = st["planes"] is not None and h16
                    pl.append(st["planes"][g] if h32 else None); iv.append(st["inv"][g] if h32 else None)
                    sv.append(st["s"][g]); cv.append(st["c"][g]); rb.append(None if st["rb"] is None else st["rb"][g])
                    wt.append(None if st["wt"] is None else st["wt"][g])
            assert len(w) <= 32
            keep = (w, gm, bt, bs)
            return (L.mms_fold_planes16_group, (idx, len(w), i64(Ns), i32(Ks), up(w), up(gm), up(bt), up(bs), up(pl), up(iv), up(sv), up(cv), up(rb), up(wt)),
                    "mms_fold_planes16_group", keep)

        # W~ itself (f32) is read by the exact-fp32 fold path and split by the three-plane format; with the two-plane split path active it
        # is not stored (a quarter of the refresh's traffic: 780 MB for the 100-ant swarm's critics) and batch sizes the split path does not
        # take (not a multiple of 128) run the unfolded layers instead
        need_wt = not (split and h16)
        self._has_wt = need_wt
        fa = [a.base.feature_norm for a in self.actors]
        fc = [c.base.feature_norm for c in self.critics]
        firsts = []
        D["a1"] = None
        if split:
            D["a1"] = fold_set([b[0][0].weight for b in self.a_blocks], [m.weight for m in fa], [m.bias for m in fa], [b[0][0].bias for b in self.a_blocks],
                               H, self.obs_dim, not h16, True, h16)
            firsts.append(D["a1"])
        D["c1"] = None
        if self.sobs_dim % 4 == 0:
            D["c1"] = fold_set([b[0][0].weight for b in self.c_blocks], [m.weight for m in fc], [m.bias for m in fc], [b[0][0].bias for b in self.c_blocks],
                               H, self.sobs_dim, need_wt, split, split and h16)
            firsts.append(D["c1"])
        for st in firsts:                                      # (one launch each: the 46-wide actor rows take the kernel's unaligned path)
            D["calls"].append(fold_call([st]))
        for l in range(1, depth):
            D["fold"][l] = fold_set([b[l][0].weight for b in both], [b[l - 1][1].weight for b in both], [b[l - 1][1].bias for b in both],
                                    [b[l][0].bias for b in both], H, H, need_wt, split, split and h16 and l < depth - 1)
            D["calls"].append(fold_call([D["fold"][l]]))
        D["heads"] = None
        if split:
            last = depth - 1
            hw = [a.act.action_out.fc_mean.weight for a in self.actors] + [c.v_out.weight for c in self.critics]
            hb = [a.act.action_out.fc_mean.bias for a in self.actors] + [c.v_out.bias for c in self.critics]
            hd = fold_set(hw, [b[last][1].weight for b in both], [b[last][1].bias for b in both], hb, [A] * n + [1] * n, H, False, False, False)
            hd["wt"] = z(2 * n, A, H)                                                  # critics: row 0 = v_out, the rest stays zero
            D["heads"] = hd
            D["calls"].append(fold_call([hd]))
            if not h16:                                        # three bf16 planes of the folded weights
                for st in [D["a1"], D["c1"]] + [D["fold"][l] for l in D["fold"]]:
                    wts = list(st["wt"].unbind(0))
                    D["calls"].append((L.mms_split_planes_group, (idx, st["G"], st["N"], st["K"], 0, up(wts), up(st["planes"])), "mms_split_planes_group", wts))
            else:
                # output scales per layer over the 2n networks (actors first): layer 0 = the two first layers, l >= 1 = the H x H layers
                D["scale1"] = {0: z(2 * n)}
                D["scale_jobs"].append((0, list(D["a1"]["rb"].unbind(0)) + list(D["c1"]["rb"].unbind(0))))
                for l in range(1, depth - 1):
                    D["scale1"][l] = z(2 * n)
                    D["scale_jobs"].append((l, list(D["fold"][l]["rb"].unbind(0))))
        self._D = D
        return D

    def refresh(self):
        """Rebuild every copy derived from the parameters from the parameters as they are NOW: device work only (a handful of launches
        on the caller's stream), into buffers whose addresses never change -- it runs at step 0 of every rollout in both collect forms
        and may be captured in a hipGraph with the rollout."""
        if self._chunks is not None:
            for c in self._chunks:
                c.refresh()
            return
        D = self._derived()
        L, idx, stream = _lib.for_device(self.device)
        self._versions = self._param_versions()
        for fn, args, what, _ in D["calls"]:
            _lib.check(fn(*args, stream), None, what, L)
        d = lambda t: t.detach()
        if self.kp_a != self.obs_dim:
            torch._foreach_copy_([D["w1_a"][i, :, :self.obs_dim] for i in range(self.n)], [d(b[0][0].weight) for b in self.a_blocks])
        if self.kp_c != self.sobs_dim:
            torch._foreach_copy_([D["w1_c"][i, :, :self.sobs_dim] for i in range(self.n)], [d(b[0][0].weight) for b in self.c_blocks])
        heads = [a.act.action_out for a in self.actors]
        t = torch._foreach_div([d(hd.log_std).float() for hd in heads], [float(hd.std_x_coef) for hd in heads])
        torch._foreach_sigmoid_(t)
        torch._foreach_mul_(t, [float(hd.std_y_coef) for hd in heads])                     # distributions.py:116
        torch._foreach_copy_(list(D["std"].unbind(0)), t)
        if self.p is None:
            self._bind()
        self._fill_scales()

    def _fill_scales(self):
        """The hidden activations' power-of-two scales of the two-plane path: one value per network and layer (scale1), and -- once a batch
        size has buffers -- the per-row arrays the layer kernel reads."""
        D = self._D
        if not (D and D["split"] and self.split_format == "f16x2"):
            return
        L, idx, stream = _lib.for_device(self.device)
        n = self.n
        have_rows = self._M is not None and getattr(self, "ysc", None) is not None
        up = lambda ts: (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
        for l, rbs in D["scale_jobs"]:
            ysc = list(self.ysc[l].unbind(0)) if have_rows else [None] * (2 * n)
            yinv = list(self.yinv[l].unbind(0)) if have_rows else [None] * (2 * n)
            _lib.check(L.mms_fold_scales16_group(idx, 2 * n, up(rbs), (ctypes.c_int32 * (2 * n))(*([self.hidden] * (2 * n))), self._M if have_rows else 0,
                                                 up(list(D["scale1"][l].unbind(0))), up(ysc), up(yinv), stream), None, "mms_fold_scales16_group", L)

    def _param_versions(self):
        """(data_ptr, version) of every source parameter: what the derived copies of refresh() were built from"""
        if self._plist is None:                                   # (walking twenty modules' parameter trees costs more than reading the counters)
            self._plist = [q for m in self.actors + self.critics for q in m.parameters()]
        return tuple((q.data_ptr(), q._version) for q in self._plist)

    def _ensure_fresh(self):
        if self._chunks is not None:
            for c in self._chunks:
                c._ensure_fresh()
        elif self._versions != self._param_versions():
            if tuple(v[0] for v in self._versions) != tuple(v[0] for v in self._param_versions()):
                self._D, self.p, self._M, self._plans = None, None, None, None      # the parameters moved: every cached address is void
            self.refresh()

    def _bind(self):
        d = lambda t: t.detach()
        D = self._D
        A, C = self.a_blocks, self.c_blocks
        fa = [a.base.feature_norm for a in self.actors]
        fc = [c.base.feature_norm for c in self.critics]
        self._keep = []                                           # (tensors whose addresses sit in the pointer arrays)

        def arr(ts):
            ts = [None if t is None else t for t in ts]
            self._keep.append(ts)
            return _ptrs(ts)
        ub = lambda t: list(t.unbind(0))
        self._w1_a = ub(D["w1_a"]) if self.kp_a != self.obs_dim else [d(b[0][0].weight) for b in A]
        self._w1_c = ub(D["w1_c"]) if self.kp_c != self.sobs_dim else [d(b[0][0].weight) for b in C]
        self._std = ub(D["std"])
        self.p = {
            "fn_a_g": arr([d(m.weight) for m in fa]), "fn_a_b": arr([d(m.bias) for m in fa]),
            "fn_c_g": arr([d(m.weight) for m in fc]), "fn_c_b": arr([d(m.bias) for m in fc]),
            "w1_a": arr(self._w1_a), "b1_a": arr([d(b[0][0].bias) for b in A]),
            "w1_c": arr(self._w1_c), "b1_c": arr([d(b[0][0].bias) for b in C]),
        }
        both = A + C
        for l in range(self.depth):
            self.p["ln%d_g" % l] = arr([d(b[l][1].weight) for b in both])
            self.p["ln%d_b" % l] = arr([d(b[l][1].bias) for b in both])
            if l > 0:
                self.p["w%d" % l] = arr([d(b[l][0].weight) for b in both])
                self.p["b%d" % l] = arr([d(b[l][0].bias) for b in both])
        self._fold = {l: (st["wt"], st["s"], st["c"]) for l, st in D["fold"].items()}
        for l, (Wt, sv, cv) in self._fold.items():
            self.p["fs%d" % l], self.p["fc%d" % l] = arr(ub(sv)), arr(ub(cv))
            if Wt is not None:
                self.p["fw%d" % l] = arr(ub(Wt))
        self._fold_c1 = None
        if D["c1"] is not None:
            st = D["c1"]
            self._fold_c1 = (st["wt"], st["s"], st["c"])
            self.p["fs1_c"], self.p["fc1_c"] = arr(ub(st["s"])), arr(ub(st["c"]))
            if st["wt"] is not None:
                self.p["fw1_c"] = arr(ub(st["wt"]))
        heads = [a.act.action_out.fc_mean for a in self.actors]
        vouts = [c.v_out for c in self.critics]
        self.p["hw"] = arr([d(m.weight) for m in heads] + [d(m.weight) for m in vouts])
        self.p["hb"] = arr([d(m.bias) for m in heads] + [d(m.bias) for m in vouts])
        self.p["std"] = arr(self._std + [None] * self.n)
        self.p["std_none"] = arr([None] * (2 * self.n))
        self._sp = None
        if D["split"]:
            h16 = self.split_format == "f16x2"
            up = lambda ts: (self._keep.append(ts), (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts]))[1]
            self._sp = {"a1": D["a1"], "heads": D["heads"], "scale1": D.get("scale1")}
            self.p["sw_a1"], self.p["sw_c1"] = up(D["a1"]["planes"]), up(D["c1"]["planes"])
            self.p["fs1_a"], self.p["fc1_a"] = arr(ub(D["a1"]["s"])), arr(ub(D["a1"]["c"]))
            for l, st in D["fold"].items():
                self.p["sw%d" % l] = up(st["planes"])
            if h16:
                self.p["swi_a1"], self.p["swi_c1"] = arr(ub(D["a1"]["inv"])), arr(ub(D["c1"]["inv"]))
                for l, st in D["fold"].items():
                    self.p["swi%d" % l] = arr(ub(st["inv"]))
            hd = D["heads"]
            self.p["hwt"], self.p["hs"], self.p["hc"] = arr(ub(hd["wt"])), arr(ub(hd["s"])), arr(ub(hd["c"]))
        self._A = (ctypes.c_int32 * (2 * self.n))(*([self.act_dim] * self.n + [1] * self.n))
        self._A1 = (ctypes.c_int32 * self.n)(*([1] * self.n))
        self._Aa = (ctypes.c_int32 * self.n)(*([self.act_dim] * self.n))
        _critic_halves(self.p, self.n)

    def _buffers(self, M):
        if self._M == M:
            return
        dev, n, H = self.device, self.n, self.hidden
        z = lambda *s, **k: torch.empty(*s, device=dev, **k)
        self.x_a, self.x_c = z(n, M, self.kp_a), z(n, M, self.kp_c)                    # normalised (and padded) inputs
        self.h = [z(2 * n, M, H), z(2 * n, M, H)]                                     # hidden activations, ping-pong
        self.actions, self.logp, self.values = z(n, M, self.act_dim), z(n, M, self.act_dim), z(n, M, 1)
        # draw counters of the noise stream (seed + agent, row, counter): one tensor per batch size, kept across re-allocations of the
        # other buffers, so that alternating batch sizes (training at M, evaluation at M', back to M) never replays noise already used
        if not hasattr(self, "_counters_by_M"):
            self._counters_by_M = {}
        if M not in self._counters_by_M:
            self._counters_by_M[M] = torch.zeros(n, M, dtype=torch.int64, device=dev)
        self.counters = self._counters_by_M[M]
        self.part, self.stat = z(2 * n, max(1, H // 64), M, 2), z(2 * n, M, 2)        # row statistics of the folded LayerNorms
        self.stat_c = z(n, M, 2)                                                       # ... of the critics' feature LayerNorm
        ub = lambda t: list(t.unbind(0))
        self.q = {
            "x_a": _ptrs(ub(self.x_a)), "x_c": _ptrs(ub(self.x_c)),
            "h0": _ptrs(ub(self.h[0])), "h1": _ptrs(ub(self.h[1])),
            "h0_a": _ptrs(ub(self.h[0][:n])), "h0_c": _ptrs(ub(self.h[0][n:])),
            "out": _ptrs(ub(self.actions) + ub(self.values)), "logp": _ptrs(ub(self.logp) + [None] * n),
            "cnt": _ptrs(ub(self.counters) + [None] * n),
            "stat_c": _ptrs(ub(self.stat_c)), "stat_c0": _ptrs([self.stat_c[0]] * n),
            "part": _ptrs(ub(self.part)), "part_a": _ptrs(ub(self.part[:n])), "part_c": _ptrs(ub(self.part[n:])), "stat": _ptrs(ub(self.stat)),
        }
        if self._sp is not None and M % 128 == 0:
            h16 = self.split_format == "f16x2"
            u8 = lambda g, rows, K: [torch.empty(rows * ((K + 31) // 32) * (128 if h16 else 192), dtype=torch.uint8, device=dev) for _ in range(g)]
            self.sx_a, self.sx_c = u8(n, M, self.obs_dim), u8(n, M, self.sobs_dim)
            self.sh = [u8(2 * n, M, H), u8(2 * n, M, H)]
            self.stat_a = z(n, M, 2)
            # per-slot partial dot products of the output heads, rows of A rounded up to 4 floats: an actor's and a critic's differ
            self.head_part_a = z(n, max(1, H // 64), M, (self.act_dim + 3) & ~3)
            self.head_part_c = z(n, max(1, H // 64), M, 4)
            up = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
            self.q.update({"sx_a": up(self.sx_a), "sx_c": up(self.sx_c), "sx_c0": up([self.sx_c[0]] * n), "sh0": up(self.sh[0]), "sh1": up(self.sh[1]),
                           "sh0_a": up(self.sh[0][:n]), "sh0_c": up(self.sh[0][n:]), "stat_a": _ptrs(ub(self.stat_a)),
                           "hpart": _ptrs(ub(self.head_part_a) + ub(self.head_part_c)), "hpart_a": _ptrs(ub(self.head_part_a)), "hpart_c": _ptrs(ub(self.head_part_c))})
            if h16:
                # row scales of the raw inputs (written by the split) and of the hidden activations (constants of the refresh, one per
                # network and layer, laid out per row because that is what the kernel reads)
                self.xs_a, self.xi_a, self.xs_c, self.xi_c = z(n, M), z(n, M), z(n, M), z(n, M)
                self.ysc = {l: z(2 * n, M) for l in self._sp["scale1"]}
                self.yinv = {l: z(2 * n, M) for l in self._sp["scale1"]}
                self.q.update({"xs_a": _ptrs(ub(self.xs_a)), "xi_a": _ptrs(ub(self.xi_a)), "xs_c": _ptrs(ub(self.xs_c)), "xi_c": _ptrs(ub(self.xi_c)),
                               "xi_c0": _ptrs([self.xi_c[0]] * n)})
                for l in self.ysc:
                    self.q["ysc%d" % l], self.q["yinv%d" % l] = _ptrs(ub(self.ysc[l])), _ptrs(ub(self.yinv[l]))
                self.q["ysc0_a"], self.q["yinv0_a"] = _ptrs(ub(self.ysc[0][:n])), _ptrs(ub(self.yinv[0][:n]))
        _critic_halves(self.q, n)
        self._M = M
        self._fill_scales()

    # -- the split path ---------------------------------------------------------------------------------------------------------
    def _split_applies(self, M, sobs_pitch):
        return (self._sp is not None and self.split_layers and self.fold_layernorm and M % 128 == 0 and self.hidden % 128 == 0
                and sobs_pitch == self.sobs_dim and self.sobs_dim >= 8)

    def _forward_split(self, L, idx, stream, M, obs_p, obs_pitch, sobs_p, with_actors, std_p, out_p, logp_p, pitch_p, cnt_p):
        """Every layer through mms_linear_group_act_split16 / mms_linear_group_act_split with all LayerNorms folded and the heads finished from the last layer's
        partials: split + row moments of the raw observations, fc1 of the actors (K = obs) and of the critics (K = share_obs), the
        H x H layers of all networks, mms_marl_heads_finish.  with_actors = False: the critics alone (get_values)."""
        n, H, p, q = self.n, self.hidden, self.p, self.q
        chk = lambda rc, what: _lib.check(rc, None, what, L)
        slots, A, depth = H // 64, self.act_dim, self.depth
        sfx = "" if with_actors else "/c"
        G = 2 * n if with_actors else n
        last_mode = lambda l: 2 if l == depth - 1 else 1
        heads = lambda l, key_w, key_p, dims: (p[key_w], q[key_p], dims) if l == depth - 1 else (None, None, None)
        h16 = self.split_format == "f16x2"
        self._cond = []

        def split(groups, K, pitch, src, dst, xs, xi, stat):
            """planes of the raw rows + the statistics of their feature LayerNorm (one kernel with two fp16 planes: the split reads
            the rows anyway; three bf16 planes: mms_row_moments_group beside the split)"""
            if h16:
                chk(L.mms_split_planes16_group(idx, groups, M, K, pitch, src, q[dst], q[xs], q[xi], 0, 0, None, None, None, q[stat], self.eps, stream),
                    "mms_split_planes16_group")
            else:
                chk(L.mms_split_planes_group(idx, groups, M, K, pitch, src, q[dst], stream), "mms_split_planes_group")
                chk(L.mms_row_moments_group(idx, groups, M, K, pitch, src, q[stat], self.eps, stream), "mms_row_moments_group")

        def layer(groups, K, x, w, b, y, mode, sv, stat, part, hw, hp, hd, xinv, winv, ysc):
            if h16:
                chk(L.mms_linear_group_act_split16(idx, groups, M, H, K, x, w, b, y, xinv, winv, ysc if mode == 1 else None, 1, mode, sv, stat, part, hw, hp, hd, stream),
                    "mms_linear_group_act_split16")
            else:
                chk(L.mms_linear_group_act_split(idx, groups, M, H, K, x, w, b, y, 1, mode, sv, stat, part, hw, hp, hd, stream), "mms_linear_group_act_split")
        g16 = lambda table, key: table[key] if h16 else None
        if with_actors:
            split(n, self.obs_dim, obs_pitch, obs_p, "sx_a", "xs_a", "xi_a", "stat_a")
        shared_rows = len({int(v) for v in sobs_p}) == 1                     # one centralised observation for all critics: one pass
        gs = 1 if shared_rows else n
        split(gs, self.sobs_dim, self.sobs_dim, sobs_p, "sx_c", "xs_c", "xi_c", "stat_c")
        if with_actors:
            hw, hp, hd = heads(0, "hwt", "hpart_a", self._Aa)
            hw = None if hw is None else (ctypes.c_void_p * n)(*list(hw)[:n])
            layer(n, self.obs_dim, q["sx_a"], p["sw_a1"], p["fc1_a"], q["sh0_a"], last_mode(0), p["fs1_a"], q["stat_a"], q["part_a"], hw, hp, hd,
                  g16(q, "xi_a"), g16(p, "swi_a1"), g16(q, "ysc0_a"))
        hw, hp, hd = heads(0, "hwt/c", "hpart_c", self._A1)
        layer(n, self.sobs_dim, q["sx_c0"] if shared_rows else q["sx_c"], p["sw_c1"], p["fc1_c"], q["sh0_c"], last_mode(0), p["fs1_c"],
              q["stat_c0"] if shared_rows else q["stat_c"], q["part_c"], hw, hp, hd, g16(q, "xi_c0" if shared_rows else "xi_c"), g16(p, "swi_c1"), g16(q, "ysc0/c"))
        cur = 0
        for l in range(1, depth):
            chk(L.mms_row_stats_chan_group(idx, G, M, slots, q["part" + sfx], q["stat" + sfx], self.eps, stream), "mms_row_stats_chan_group")
            if self.track_conditioning:
                st = self.stat[(0 if with_actors else n):]
                self._cond.append((st[..., 0].abs() * st[..., 1]).max())
            hw, hp, hd = heads(l, "hwt" + sfx, "hpart" + sfx, self._A if with_actors else self._A1)
            layer(G, H, q["sh%d%s" % (cur, sfx)], p["sw%d%s" % (l, sfx)], p["fc%d%s" % (l, sfx)], q["sh%d%s" % (1 - cur, sfx)], last_mode(l),
                  p["fs%d%s" % (l, sfx)], q["stat" + sfx], q["part" + sfx], hw, hp, hd, g16(q, "yinv%d%s" % (l - 1, sfx)), g16(p, "swi%d%s" % (l, sfx)),
                  g16(q, "ysc%d%s" % (l, sfx)) if l < depth - 1 else None)
            cur = 1 - cur
        A_arr = self._A if with_actors else self._A1
        chk(L.mms_marl_heads_finish(idx, G, M, slots, q["part" + sfx], q["hpart" + sfx], p["hs" + sfx], p["hc" + sfx], A_arr, std_p, out_p, logp_p,
                                    pitch_p, cnt_p, self.seed, self.row_offset, self.eps, stream), "mms_marl_heads_finish")

    def _out_ptrs(self, out):
        """(values, actions, logp lists, output / log-prob pointer arrays over the 2n networks, pitch array or None)"""
        n, q = self.n, self.q
        if out is None:
            return list(self.values.unbind(0)), list(self.actions.unbind(0)), list(self.logp.unbind(0)), q["out"], q["logp"], None
        values = list(out[0])
        actions = list(self.actions.unbind(0)) if out[1] is None else list(out[1])      # (None: scratch -- values_into)
        logp = list(self.logp.unbind(0)) if out[2] is None else list(out[2])
        ap, a_pitch = _row_ptrs(actions)
        lp, l_pitch = _row_ptrs(logp)
        vp, v_pitch = _row_ptrs(values)
        if a_pitch != l_pitch:
            raise ValueError("GroupedPolicyInference: action and log-prob destinations must share one row pitch")
        out_p = (ctypes.c_void_p * (2 * n))(*(list(ap) + list(vp)))
        logp_p = (ctypes.c_void_p * (2 * n))(*(list(lp) + [None] * n))
        pitch = (ctypes.c_int32 * (2 * n))(*([a_pitch] * n + [v_pitch] * n))
        return values, actions, logp, out_p, logp_p, pitch

    # -- inference ----------------------------------------------------------------------------------------------------------------
    def get_actions(self, share_obs, obs, deterministic=False, out=None):
        """share_obs, obs: per-agent lists of [M, share_obs_dim] / [M, obs_dim] float32 tensors (what the Runner hands agent i:
        buffer[i].share_obs[step], buffer[i].obs[step]); rows may be strided (agent i's rows of an [M, agents, K] block are read
        where they lie).  out = (values, actions, action_log_probs): optional per-agent lists of [M, 1] / [M, act_dim] /
        [M, act_dim] destination views with dense rows (e.g. slices of [M, agents, ...] rollout slots), written in place.
        Returns (values, actions, action_log_probs): per-agent lists of
        [M, 1], [M, act_dim], [M, act_dim] views of buffers that the next call overwrites (the reference's
        FixedNormal.log_probs keeps the per-dimension log-densities, utils/distributions.py:31-34)."""
        n = self.n
        if len(share_obs) != n or len(obs) != n:
            raise ValueError("one observation tensor per agent")
        self._ensure_fresh()
        if self._chunks is not None:
            res = ([], [], [])
            for c, (lo, hi) in zip(self._chunks, self._ranges):
                o = None if out is None else tuple(None if x is None else x[lo:hi] for x in out)
                got = c.get_actions(share_obs[lo:hi], obs[lo:hi], deterministic=deterministic, out=o)
                for dst, part in zip(res, got):
                    if part is not None:
                        dst.extend(part)
            return res[0], res[1], (None if deterministic else res[2])
        M = obs[0].shape[0]
        self._buffers(M)
        L, idx, stream = _lib.for_device(self.device)
        p, q = self.p, self.q
        chk = lambda rc, what: _lib.check(rc, None, what, L)
        f32 = lambda t: t.detach() if t.dtype == torch.float32 else t.detach().float()
        obs_f, sobs_f = [f32(t) for t in obs], [f32(t) for t in share_obs]     # (held until the launches below have been issued)
        obs_p, obs_pitch = _row_ptrs(obs_f)
        sobs_p, sobs_pitch = _row_ptrs(sobs_f)
        if self._split_applies(M, sobs_pitch):
            values, actions, logp, out_p, logp_p, pitch = self._out_ptrs(out)
            self._forward_split(L, idx, stream, M, obs_p, obs_pitch, sobs_p, True, p["std_none"] if deterministic else p["std"], out_p, logp_p, pitch, q["cnt"])
            return values, actions, (None if deterministic else logp)
        chk(L.mms_layernorm_group(idx, n, M, self.obs_dim, self.kp_a, obs_pitch, obs_p, p["fn_a_g"], p["fn_a_b"], q["x_a"], self.eps, stream), "mms_layernorm_group")
        H = self.hidden
        fold = self.fold_layernorm and self.depth > 1 and M % 128 == 0 and H % 128 == 0 and self.kp_a >= 8 and self._has_wt
        fold_c1 = fold and self._fold_c1 is not None and sobs_pitch == self.sobs_dim and self.sobs_dim >= 8
        if not fold_c1:
            chk(L.mms_layernorm_group(idx, n, M, self.sobs_dim, self.kp_c, sobs_pitch, sobs_p, p["fn_c_g"], p["fn_c_b"], q["x_c"], self.eps, stream), "mms_layernorm_group")
        slots = H // 64
        chk(L.mms_linear_group_act(idx, n, M, H, self.kp_a, q["x_a"], p["w1_a"], p["b1_a"], q["h0_a"], 1, None, None, q["part_a"] if fold else None, stream), "mms_linear_group_act")
        if fold_c1:
            shared_rows = len({int(v) for v in sobs_p}) == 1                 # one centralised observation for all critics: one pass
            chk(L.mms_row_moments_group(idx, 1 if shared_rows else n, M, self.sobs_dim, sobs_pitch, sobs_p, q["stat_c"], self.eps, stream), "mms_row_moments_group")
            chk(L.mms_linear_group_act(idx, n, M, H, self.sobs_dim, sobs_p, p["fw1_c"], p["fc1_c"], q["h0_c"], 1, p["fs1_c"],
                                       q["stat_c0"] if shared_rows else q["stat_c"], q["part_c"], stream), "mms_linear_group_act")
        else:
            chk(L.mms_linear_group_act(idx, n, M, H, self.kp_c, q["x_c"], p["w1_c"], p["b1_c"], q["h0_c"], 1, None, None, q["part_c"] if fold else None, stream), "mms_linear_group_act")
        cur = 0
        for l in range(self.depth):
            if l > 0 and fold:                                      # LayerNorm l - 1 folded in; leaves the statistics of LayerNorm l
                chk(L.mms_row_stats_group(idx, 2 * n, M, slots, H, q["part"], q["stat"], self.eps, stream), "mms_row_stats_group")
                chk(L.mms_linear_group_act(idx, 2 * n, M, H, H, q["h%d" % cur], p["fw%d" % l], p["fc%d" % l], q["h%d" % (1 - cur)], 1, p["fs%d" % l], q["stat"],
                                           q["part"] if l + 1 < self.depth else None, stream), "mms_linear_group_act")
                cur = 1 - cur
                continue
            if l > 0:
                chk(L.mms_linear_group_act(idx, 2 * n, M, H, H, q["h%d" % cur], p["w%d" % l], p["b%d" % l], q["h%d" % (1 - cur)], 1, None, None, None, stream), "mms_linear_group_act")
                cur = 1 - cur
            if l + 1 < self.depth and not fold:                     # (the last LayerNorm runs inside the heads kernel)
                chk(L.mms_layernorm_group(idx, 2 * n, M, H, H, H, q["h%d" % cur], p["ln%d_g" % l], p["ln%d_b" % l], q["h%d" % cur], self.eps, stream), "mms_layernorm_group")
        last = self.depth - 1
        values, actions, logp, out_p, logp_p, pitch = self._out_ptrs(out)
        chk(L.mms_marl_heads_act(idx, 2 * n, M, H, q["h%d" % cur], p["ln%d_g" % last], p["ln%d_b" % last], p["hw"], p["hb"], self._A,
                                 p["std_none"] if deterministic else p["std"], out_p, logp_p, pitch, q["cnt"], self.seed, self.row_offset, self.eps, stream),
            "mms_marl_heads_act")
        return values, actions, (None if deterministic else logp)

    # -- the Runner's collect step --------------------------------------------------------------------------------------------------
    @classmethod
    def from_trainers(cls, trainers, seed=0, row_offset=0):
        """One object for a Runner's agents: `trainers[i].policy.actor` / `.critic` (runner.py:69-101)."""
        return cls([t.policy.actor for t in trainers], [t.policy.critic for t in trainers], seed=seed, row_offset=row_offset)

    @torch.no_grad()
    def collect(self, buffers, step):
        """The body of Runner.collect (agents/algorithms/marl/runner.py:198-227) for non-recurrent policies: same five return values
        -- values [M, agents, 1], the per-agent action and log-prob lists, and the (untouched) rnn states [M, agents, ...] -- from one
        grouped pass instead of a loop over agents.  `buffers` = the Runner's per-agent SeparatedReplayBuffers.  The padded first
        actor weights and the std vectors are refreshed at step 0 of every episode (the trainers update the parameters in between)."""
        if step == 0:
            self.refresh()
        values, actions, logp = self.get_actions([b.share_obs[step] for b in buffers], [b.obs[step] for b in buffers])
        rnn_states = torch.transpose(torch.stack([b.rnn_states[step] for b in buffers]), 1, 0)
        rnn_states_critic = torch.transpose(torch.stack([b.rnn_states_critic[step] for b in buffers]), 1, 0)
        return torch.stack(values, 1), actions, logp, rnn_states, rnn_states_critic

    @torch.no_grad()
    def collect_into(self, shared, deterministic=False):
        """The collect step over `SharedRolloutBuffers` (utils/shared_buffer.py), zero copy: every agent's observation rows are read
        from `shared.obs[step]` / `shared.share_obs[step]` where the env step left them, and actions, log-probs and values are
        written straight into `shared.actions[step]`, `shared.action_log_probs[step]`, `shared.value_preds[step]`.  Returns the
        [N, agents, act_dim] action slot (what `shared.env_step` takes)."""
        s, n = shared.step, self.n
        if self.refresh_every_rollout:
            if s == 0:
                self.refresh()                              # the trainers updated in between, possibly through .data (no version counter moves)
        else:
            self._ensure_fresh()                            # (inside a rollout the parameters stand still: no check per step otherwise)
        # the marshalled operands of slot s (ten row views per buffer, their pointer arrays) are built once per (buffers, slot): what is
        # left per call is the launches -- the eager collect step then stays ahead of the GPU like the captured one
        key = (shared.obs.data_ptr(), shared.share_obs.data_ptr(), shared.actions.data_ptr(), shared.value_preds.data_ptr(), tuple(shared.obs.shape), s,
               bool(deterministic))

        def make():
            obs, sobs = shared.obs[s], shared.share_obs[s]
            out = ([shared.value_preds[s][:, k:k + 1] for k in range(n)], [shared.actions[s][:, k] for k in range(n)],
                   [shared.action_log_probs[s][:, k] for k in range(n)])
            return [sobs] * n, [obs[:, k] for k in range(n)], out
        self._get_actions_planned(key, make, deterministic)
        return shared.actions[s]

    def _get_actions_planned(self, key, make, deterministic):
        """get_actions(share_obs, obs, deterministic, out) with (share_obs, obs, out) = make(), called only when `key` is new: the split
        path's operand pointer arrays are cached under it (the caller vouches that the key names the tensors)."""
        if self._plans is None:
            self._plans = {}
        plan = self._plans.get(key)
        if plan is None:
            share_obs, obs, out = make()
            if self._chunks is not None:
                plan = ("chunks", [(c, (lambda so=share_obs[lo:hi], ob=obs[lo:hi], ou=tuple(x[lo:hi] for x in out): (so, ob, ou)))
                                   for c, (lo, hi) in zip(self._chunks, self._ranges)])
            else:
                M = obs[0].shape[0]
                self._buffers(M)
                f32 = lambda t: t.detach() if t.dtype == torch.float32 else t.detach().float()
                obs_f, sobs_f = [f32(t) for t in obs], [f32(t) for t in share_obs]
                obs_p, obs_pitch = _row_ptrs(obs_f)
                sobs_p, sobs_pitch = _row_ptrs(sobs_f)
                direct = all(a is b for a, b in zip(obs_f + sobs_f, list(obs) + list(share_obs)))      # (no converted temporaries in the pointer arrays)
                if self._split_applies(M, sobs_pitch) and direct:
                    plan = ("split", M, obs_p, obs_pitch, sobs_p) + tuple(self._out_ptrs(out)[3:]) + ((obs_f, sobs_f, out),)
                else:
                    plan = ("generic", share_obs, obs, out)
            self._plans[key] = plan
        if plan[0] == "chunks":
            for c, mk in plan[1]:
                c._get_actions_planned(key, mk, deterministic)
        elif plan[0] == "generic":
            self.get_actions(plan[1], plan[2], deterministic=deterministic, out=plan[3])
        else:
            _, M, obs_p, obs_pitch, sobs_p, out_p, logp_p, pitch, _keep = plan
            self._buffers(M)
            L, idx, stream = _lib.for_device(self.device)
            self._forward_split(L, idx, stream, M, obs_p, obs_pitch, sobs_p, True, self.p["std_none"] if deterministic else self.p["std"], out_p, logp_p, pitch,
                                self.q["cnt"])

    # collect_into refreshes at step 0 of every rollout (see the module docstring); False leaves only the version-counter check -- for
    # callers that never write parameters through .data and want the ~ms of derived-copy rebuilding only when an optimizer stepped
    refresh_every_rollout = True

    # Diagnostics (off the hot path): with track_conditioning = True every folded pass records, per hidden LayerNorm, the largest
    # |mean| / std of the rows it normalised (three small torch kernels per layer: not for timed runs)
    track_conditioning = False

    @torch.no_grad()
    def fold_conditioning(self):
        """Largest |mean| / std over the rows a folded hidden LayerNorm normalised in the previous pass (a device scalar; None before the
        first folded pass): over ALL hidden LayerNorms when `track_conditioning` was set for that pass, else of the last one only.  The
        folded layer evaluates W~ h - mean s, which amplifies whatever its operands lack by this ratio: with two fp16 planes (operands
        kept to 2^-22) the layer's relative error is ~ratio x 2^-22, with three bf16 planes or the exact-fp32 kernel ~ratio x 2^-24 (the
        rounding of h itself).  Ordinary networks sit below 10; choose split_format="bf16x3" when this reports hundreds
        (tests/test_marl_policy.py: the stress case at 1000)."""
        if self._chunks is not None:
            vals = [c.fold_conditioning() for c in self._chunks]
            vals = [v for v in vals if v is not None]
            return torch.stack(vals).max() if vals else None
        if self._M is None or not hasattr(self, "stat"):
            return None
        if getattr(self, "_cond", None):
            return torch.stack(self._cond).max()
        st = self.stat                                                  # [2n, M, 2] = (mean, 1 / sqrt(var + eps)) of the last hidden LayerNorm's input
        return (st[..., 0].abs() * st[..., 1]).max()

    @torch.no_grad()
    def get_values(self, share_obs, out=None):
        """The critics alone (policy.get_values, mappo_policy.py:77-88; Runner.compute, runner.py:229-241): per-agent lists of
        [M, share_obs_dim] in, [M, 1] values out (views of a buffer the next call overwrites, or the `out` destinations)."""
        n = self.n
        if len(share_obs) != n:
            raise ValueError("one observation tensor per agent")
        self._ensure_fresh()
        if self._chunks is not None:
            res = []
            for c, (lo, hi) in zip(self._chunks, self._ranges):
                res.extend(c.get_values(share_obs[lo:hi], None if out is None else out[lo:hi]))
            return res
        M = share_obs[0].shape[0]
        self._buffers(M)
        L, idx, stream = _lib.for_device(self.device)
        p, q = self.p, self.q
        chk = lambda rc, what: _lib.check(rc, None, what, L)
        f32 = lambda t: t.detach() if t.dtype == torch.float32 else t.detach().float()
        sobs_f = [f32(t) for t in share_obs]                                  # (held until the launches below have been issued)
        sobs_p, sobs_pitch = _row_ptrs(sobs_f)
        H = self.hidden
        if self._split_applies(M, sobs_pitch):
            values = list(self.values.unbind(0)) if out is None else list(out)
            vp, v_pitch = _row_ptrs(values)
            self._forward_split(L, idx, stream, M, None, 0, sobs_p, False, None, vp, None, (ctypes.c_int32 * n)(*([v_pitch] * n)), None)
            return values
        fold = self.fold_layernorm and self.depth > 1 and M % 128 == 0 and H % 128 == 0 and self._has_wt
        fold_c1 = fold and self._fold_c1 is not None and sobs_pitch == self.sobs_dim and self.sobs_dim >= 8
        slots = H // 64
        if fold_c1:
            shared_rows = len({int(v) for v in sobs_p}) == 1
            chk(L.mms_row_moments_group(idx, 1 if shared_rows else n, M, self.sobs_dim, sobs_pitch, sobs_p, q["stat_c"], self.eps, stream), "mms_row_moments_group")
            chk(L.mms_linear_group_act(idx, n, M, H, self.sobs_dim, sobs_p, p["fw1_c"], p["fc1_c"], q["h0_c"], 1, p["fs1_c"],
                                       q["stat_c0"] if shared_rows else q["stat_c"], q["part_c"], stream), "mms_linear_group_act")
        else:
            chk(L.mms_layernorm_group(idx, n, M, self.sobs_dim, self.kp_c, sobs_pitch, sobs_p, p["fn_c_g"], p["fn_c_b"], q["x_c"], self.eps, stream), "mms_layernorm_group")
            chk(L.mms_linear_group_act(idx, n, M, H, self.kp_c, q["x_c"], p["w1_c"], p["b1_c"], q["h0_c"], 1, None, None, q["part_c"] if fold else None, stream), "mms_linear_group_act")
        cur = 0
        for l in range(self.depth):
            if l > 0 and fold:
                chk(L.mms_row_stats_group(idx, n, M, slots, H, q["part/c"], q["stat/c"], self.eps, stream), "mms_row_stats_group")
                chk(L.mms_linear_group_act(idx, n, M, H, H, q["h%d/c" % cur], p["fw%d/c" % l], p["fc%d/c" % l], q["h%d/c" % (1 - cur)], 1, p["fs%d/c" % l],
                                           q["stat/c"], q["part/c"] if l + 1 < self.depth else None, stream), "mms_linear_group_act")
                cur = 1 - cur
                continue
            if l > 0:
                chk(L.mms_linear_group_act(idx, n, M, H, H, q["h%d/c" % cur], p["w%d/c" % l], p["b%d/c" % l], q["h%d/c" % (1 - cur)], 1, None, None, None, stream), "mms_linear_group_act")
                cur = 1 - cur
            if l + 1 < self.depth and not fold:
                chk(L.mms_layernorm_group(idx, n, M, H, H, H, q["h%d/c" % cur], p["ln%d_g/c" % l], p["ln%d_b/c" % l], q["h%d/c" % cur], self.eps, stream), "mms_layernorm_group")
        last = self.depth - 1
        values = list(self.values.unbind(0)) if out is None else list(out)
        vp, v_pitch = _row_ptrs(values)
        pitch = (ctypes.c_int32 * n)(*([v_pitch] * n))
        chk(L.mms_marl_heads_act(idx, n, M, H, q["h%d/c" % cur], p["ln%d_g/c" % last], p["ln%d_b/c" % last], p["hw/c"], p["hb/c"], self._A1,
                                 None, vp, None, pitch, None, self.seed, self.row_offset, self.eps, stream), "mms_marl_heads_act")
        return values

    @torch.no_grad()
    def values_into(self, shared, dst, slot=-1):
        """Critic values of observation slot `slot` (default: the last one -- the bootstrap values of Runner.compute,
        runner.py:229-241, taken from share_obs[-1]) into dst [N, agents]: the critics alone (`get_values`)."""
        n = self.n
        self.get_values([shared.share_obs[slot]] * n, out=[dst[:, k:k + 1] for k in range(n)])
        return dst
