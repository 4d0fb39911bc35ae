"""ActorCritic.refresh() and the device-side refresh entry points (mms_weight_planes16_group, mms_layer_bounds16, mms_chain_scales16) --
one check list for both builds: tests/test_cpu_backend.py runs it on libmms_cpu.so, tests/test_gpu_parity.py on libmms.so.

What the reference does at this point: nothing -- its nn.Linear layers read their parameters in every forward
(/root/reference/agents/algorithms/rl/ppo/module.py:73-87), so a parameter update of any kind is followed by the next `act`.  The fused
layers compute from buffers DERIVED from the parameters (operand planes, row scales, bound chain); these checks pin when those follow."""
import copy
import ctypes

import torch

from massive_marl_benchmark_amd import _lib
from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic
from massive_marl_benchmark_amd.algorithms.rl.ppo.storage import RolloutStorage


def _arr(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def check_refresh_entry_points(device):
    """mms_weight_planes16_group (matrices of different shapes in one launch) and mms_chain_refresh16 against torch and against the chain
    the input split evaluates per row."""
    dev = torch.device(device)
    L, idx, stream = _lib.for_device(dev)
    torch.manual_seed(11)
    shapes = ((256, 388), (128, 1024), (64, 36), (8, 5), (512, 1024), (1024, 388))
    w = [(torch.randn(N, K) * (0.5 + g)).to(dev) for g, (N, K) in enumerate(shapes)]
    b = [torch.randn(N).to(dev) for N, _ in shapes]
    G = len(shapes)
    planes = [torch.zeros(N * ((K + 31) // 32) * 128, dtype=torch.uint8, device=dev) for N, K in shapes]
    scale, inv, l1 = ([torch.zeros(N, device=dev) for N, _ in shapes] for _ in range(3))
    Ns, Ks = (ctypes.c_int64 * G)(*[N for N, _ in shapes]), (ctypes.c_int32 * G)(*[K for _, K in shapes])
    assert L.mms_weight_planes16_group(idx, G, Ns, Ks, _arr(w), _arr(planes), _arr(scale), _arr(inv), _arr(l1), stream) == 0, _lib.last_error(None, L)
    for g, (N, K) in enumerate(shapes):
        # the planes and scales are mms_split_planes16_group's, bit for bit
        planes2, scale2, inv2 = torch.zeros_like(planes[g]), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        assert L.mms_split_planes16_group(idx, 1, N, K, 0, _arr([w[g]]), _arr([planes2]), _arr([scale2]), _arr([inv2]), 0, 0, None, None, None, None, 0.0, stream) == 0
        assert torch.equal(planes[g], planes2) and torch.equal(scale[g], scale2) and torch.equal(inv[g], inv2), (N, K)
        ref = w[g].double().abs().sum(1)
        assert float(((l1[g].double() - ref).abs() / ref).max()) < 1e-6
    # the chain's entries: (largest row 1-norm, largest |bias|), entry e = c L + l; no bias: add = 0
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    nch, Lh, rows = 2, 2, 300
    for bound0 in (8.0, 5.0 / 3.0, 1e-3):
        chain = torch.full((nch, Lh, 2), -1.0, device=dev)
        cs, ci = (torch.zeros(nch, Lh, rows, device=dev) for _ in range(2))
        pick = [0, 1, 4, 5]
        bias = _arr([b[g] for g in pick[:3]] + [b[0]])
        bias[3] = None
        assert L.mms_chain_refresh16(idx, nch, Lh, _arr([l1[g] for g in pick]), bias, (ctypes.c_int32 * 4)(*[shapes[g][0] for g in pick]), p(chain),
                                     bound0, rows, p(cs), p(ci), stream) == 0, _lib.last_error(None, L)
        want = torch.tensor([[float(l1[g].max()), float(b[g].abs().max()) if j < 3 else 0.0] for j, g in enumerate(pick)], device=dev).view(nch, Lh, 2)
        assert torch.equal(chain, want)
        # ... and its constant-bound scales == what the input split leaves for a row whose largest magnitude is that bound
        K = 40
        x = (torch.rand(rows, K) * 2 - 1).to(dev) * bound0 * 0.9
        x[:, 3] = bound0
        xp = torch.zeros(rows * 2 * 128, dtype=torch.uint8, device=dev)
        xs, xi = torch.zeros(rows, device=dev), torch.zeros(rows, device=dev)
        cs2, ci2 = (torch.zeros(nch, Lh, rows, device=dev) for _ in range(2))
        assert L.mms_split_planes16_group(idx, 1, rows, K, 0, _arr([x]), _arr([xp]), _arr([xs]), _arr([xi]), nch, Lh, _arr([chain]), _arr([cs2]), _arr([ci2]),
                                          None, 0.0, stream) == 0
        assert torch.equal(cs, cs2) and torch.equal(ci, ci2)
        assert bool(((cs * ci) == 1.0).all())
        # rows = 0: the entries alone
        chain0 = torch.full((nch, Lh, 2), -1.0, device=dev)
        assert L.mms_chain_refresh16(idx, nch, Lh, _arr([l1[g] for g in pick]), bias, (ctypes.c_int32 * 4)(*[shapes[g][0] for g in pick]), p(chain0),
                                     0.0, 0, None, None, stream) == 0
        assert torch.equal(chain0, want)
    return {sh: t.cpu() for sh, t in zip(shapes, l1)}


def check_fold_entry_points(device):
    """mms_fold_planes16_group / mms_fold_scales16_group against the torch statement of the LayerNorm fold (W~ = W diag(gamma), s = W~ 1,
    c = W beta + b, row bound |W~ row|_2 sqrt(K) + |c|) and against mms_split_planes16_group on W~ (planes, inverse scales: bit for bit)."""
    dev = torch.device(device)
    L, idx, stream = _lib.for_device(dev)
    torch.manual_seed(3)
    shapes = ((128, 46), (128, 388), (256, 256), (8, 128), (1, 128))
    G = len(shapes)
    w = [torch.randn(N, K).to(dev) for N, K in shapes]
    gam = [(1.0 + 0.3 * torch.randn(K)).to(dev) for _, K in shapes]
    bet = [(0.2 * torch.randn(K)).to(dev) for _, K in shapes]
    bias = [torch.randn(N).to(dev) for N, _ in shapes]
    planes = [torch.zeros(N * ((K + 31) // 32) * 128, dtype=torch.uint8, device=dev) for N, K in shapes]
    inv, sv, cv, rb = ([torch.zeros(N, device=dev) for N, _ in shapes] for _ in range(4))
    wt = [torch.zeros(N, K, device=dev) for N, K in shapes]
    Ns, Ks = (ctypes.c_int64 * G)(*[N for N, _ in shapes]), (ctypes.c_int32 * G)(*[K for _, K in shapes])
    assert L.mms_fold_planes16_group(idx, G, Ns, Ks, _arr(w), _arr(gam), _arr(bet), _arr(bias), _arr(planes), _arr(inv), _arr(sv), _arr(cv), _arr(rb), _arr(wt),
                                     stream) == 0, _lib.last_error(None, L)
    for g, (N, K) in enumerate(shapes):
        Wt = w[g] * gam[g][None, :]
        assert torch.equal(wt[g], Wt), (N, K)
        ref_s, ref_c = Wt.double().sum(1), w[g].double() @ bet[g].double() + bias[g].double()
        assert float((sv[g].double() - ref_s).abs().max()) < 1e-5 * (1.0 + float(Wt.abs().sum(1).max()))
        assert float((cv[g].double() - ref_c).abs().max()) < 1e-5 * (1.0 + float(ref_c.abs().max()))
        ref_rb = Wt.double().pow(2).sum(1).sqrt() * K ** 0.5 + ref_c.abs()
        assert float(((rb[g].double() - ref_rb).abs() / ref_rb).max()) < 1e-5
        p2, s2, i2 = torch.zeros_like(planes[g]), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        assert L.mms_split_planes16_group(idx, 1, N, K, 0, _arr([Wt.contiguous()]), _arr([p2]), _arr([s2]), _arr([i2]), 0, 0, None, None, None, None, 0.0, stream) == 0
        assert torch.equal(planes[g], p2) and torch.equal(inv[g], i2), (N, K)
    # without gamma / beta / bias: the plain matrix; optional outputs may be absent
    s0 = [torch.zeros(N, device=dev) for N, _ in shapes]
    assert L.mms_fold_planes16_group(idx, G, Ns, Ks, _arr(w), None, None, None, None, None, _arr(s0), None, None, None, stream) == 0, _lib.last_error(None, L)
    for g in range(G):
        assert float((s0[g].double() - w[g].double().sum(1)).abs().max()) < 1e-4
    # the networks' output scales: 2^(14 - e) with 1.001 max rb <= 2^e, once and per row of the batch
    M = 300
    scale1 = [torch.zeros(1, device=dev) for _ in range(G)]
    ysc, yinv = [torch.zeros(M, device=dev) for _ in range(G)], [torch.zeros(M, device=dev) for _ in range(G)]
    assert L.mms_fold_scales16_group(idx, G, _arr(rb), (ctypes.c_int32 * G)(*[N for N, _ in shapes]), M, _arr(scale1), _arr(ysc), _arr(yinv), stream) == 0
    for g in range(G):
        bound = float(rb[g].max()) * 1.001
        want = 2.0 ** (14 - (torch.frexp(torch.tensor(bound, dtype=torch.float32))[1].item()))
        assert float(scale1[g]) == want and bool((ysc[g] == want).all()) and bool((yinv[g] == 1.0 / want).all()), (g, float(scale1[g]), want)
        assert bound * want <= 2.0 ** 14
    return {sh: t.cpu() for sh, t in zip(shapes, rb)}


def check_module_refresh(device, hid=(128, 128), n=128, obs_dim=36, fmt="f16x2"):
    """Which parameter updates the fused layers follow, and when (both plane formats)."""
    dev = torch.device(device)
    torch.manual_seed(5)
    cfg = {"pi_hid_sizes": list(hid), "vf_hid_sizes": list(hid), "activation": "elu"}
    ac = ActorCritic((obs_dim,), (0,), (8,), 0.8, cfg, seed=3).to(dev)
    ac.split_min_tiles = 0
    ac.split_format = fmt
    obs, states = torch.randn(n, obs_dim, device=dev).clamp(-5, 5), torch.zeros(n, 0, device=dev)
    h16 = fmt == "f16x2"
    addresses = lambda: ([r["planes"].data_ptr() for rs in ac._h16["recs"] for r in rs] + [ac._h16["bounds"].data_ptr()]) if h16 else \
        sorted(v[1].data_ptr() for v in ac._wplanes.values())

    def close(m=None):
        m = m or ac
        ha, hc = m._fused_hidden(obs, obs)                    # (the hidden layers of both networks; on the CPU build `act` keeps to the torch modules)
        rel = lambda got, ref: float((got - ref).abs().max() / (1.0 + ref.abs().max()))         # in units of the output's scale
        with torch.no_grad():
            e = max(rel(ha, m.actor[:-1](obs)), rel(hc, m.critic[:-1](obs)))
            if dev.type == "cuda":
                _, _, v, mu, _ = m.act(obs, states)
                e = max(e, rel(mu, m.actor(obs)), rel(v, m.critic(obs)))
        return e

    assert close() < 1e-5 and ac._split_bufs, "the split path did not run"
    planes_addr = addresses()
    # 1. an optimizer step (in place: version counters move) is followed by the next act, unbound
    opt = torch.optim.SGD(ac.parameters(), lr=0.5)
    for q in ac.parameters():
        q.grad = torch.randn_like(q) * 0.1
    opt.step()
    assert close() < 1e-5
    # 2. a write through .data moves no version counter (hatrpo_trainer.py:122 `params.data.copy_(new)`): unbound, it needs refresh()
    with torch.no_grad():
        for q in ac.parameters():
            q.data.copy_(q.data + 0.1 * torch.randn_like(q))
    stale = close()
    ac.refresh()
    assert close() < 1e-5 and stale > 1e-3, stale            # (the documented limit of the version check, and its remedy)
    # 3. with a bound RolloutStorage the first act of every rollout refreshes unconditionally: .data writes are followed too
    storage = RolloutStorage(n, 2, (obs_dim,), (0,), (8,), device=str(dev))
    ac.bind_rollout(storage, None)
    assert close() < 1e-5
    with torch.no_grad():
        for q in ac.parameters():
            q.data.copy_(q.data + 0.1 * torch.randn_like(q))
    storage.clear()
    assert close() < 1e-5
    ac.bind_rollout(None, None)
    # derived buffers never moved
    assert planes_addr == addresses()
    # 4. a copy starts from its own parameters (no shared derived state), and follows its own updates
    twin = copy.deepcopy(ac)
    assert twin._h16 is None and twin._calls is None and twin._wplanes is None and twin.split_format == fmt
    with torch.no_grad():
        for q in twin.parameters():
            q.add_(0.2 * torch.randn_like(q))
    twin.split_min_tiles = 0
    assert close(twin) < 1e-5
    with torch.no_grad():
        assert float((twin._fused_hidden(obs, obs)[0] - ac.actor[:-1](obs)).abs().max()) > 1e-2
    assert close() < 1e-5
    # 5. value() (the bootstrap pass) shares the weights' planes and follows too
    with torch.no_grad():
        for q in ac.parameters():
            q.add_(0.05 * torch.randn_like(q))
        assert float((ac.value(obs) - ac.critic(obs)).abs().max() / (1.0 + ac.critic(obs).abs().max())) < 1e-5
    assert close() < 1e-5
    return stale
