"""The CPU build of the engine (massive_marl_benchmark_amd/lib/libmms_cpu.so; csrc/cpu/): the kernels' lane math compiled for the
host behind the same C ABI, selected only by asking for it (device_type="cpu" / device=-1).  What it is for: the reference's
`--sim_device cpu` pipeline (agents/tasks/agent_base/base_task.py:27-32; BASELINE configs[0] "OneAnt num_envs=64 PPO, sim_device=cpu
-- plumbing, no GPU").  Checked here: the ABI, parity with the oracle under the gates of tests/parity.py, every reference fixture
through its step path, the drop-in classes end to end on it, and that it is never a fallback."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import parity
from conftest import ROOT, load_golden
from massive_marl_benchmark_amd import _lib
from massive_marl_benchmark_amd.engine import Engine
from massive_marl_benchmark_amd.model import default_cfg, make_config
from oracle.oracle import OracleEngine


def test_cpu_library_exports_the_whole_abi():
    header = open(os.path.join(ROOT, "include", "mms.h")).read()
    declared = set(re.findall(r"\b(mms_[a-z0-9_]+)\s*\(", header)) - {"mms_engine"}
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_CPU_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert declared <= exported, declared - exported
    assert set(_lib.SYMBOLS) <= exported
    assert b"gfx950" not in open(_lib.LIB_CPU_PATH, "rb").read()                  # host code only
    assert _lib.lib_cpu().mms_abi_version() == _lib.lib().mms_abi_version()


def test_never_a_fallback():
    """The two builds refuse each other's device: the HIP library fails for device -1 (and without a GPU), the CPU library fails
    for a HIP ordinal; Engine() without a device argument still means the GPU."""
    h = ctypes.c_void_p()
    cfg = make_config("OneAnt", num_envs=4, device=-1)
    assert _lib.lib().mms_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert "libmms_cpu.so" in _lib.last_error(None)
    cfg = make_config("OneAnt", num_envs=4, device=0)
    assert _lib.lib_cpu().mms_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert "device must be -1" in _lib.last_error(None, _lib.lib_cpu())
    if not torch.cuda.is_available():
        with pytest.raises(_lib.MmsError):
            Engine("OneAnt", num_envs=4)
    z = torch.zeros(4)
    assert _lib.lib_cpu().mms_marl_views(0, ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(z.data_ptr()), 0, 2, 1, 0, None) != 0


def test_abi_error_paths_cpu_build():
    """SURVEY 8b's status-code contract on the CPU build: bad task id, num_agents > 126, unknown tensor name, env id out of range (and
    nothing written), null pointers, the other library's device ... each returns non-zero with a non-empty mms_last_error
    (tests/abi_errors.py; the same list runs on the HIP build in tests/test_gpu_parity.py).  The Python boundary on top of it: a
    VecTask whose clip_observations disagrees with the engine's raises."""
    import abi_errors
    assert abi_errors.check_abi_error_paths(_lib.lib_cpu(), -1) >= 40
    from massive_marl_benchmark_amd.tasks.agent_base.vec_task import VecTaskPython
    from massive_marl_benchmark_amd.tasks.one_ant import OneAnt
    cfg = default_cfg("OneAnt")
    cfg["env"]["numEnvs"] = 4
    task = OneAnt(cfg, None, "physx", "cpu", 0, True)
    with pytest.raises(ValueError, match="clip_obs"):
        VecTaskPython(task, "cpu", 7.0, 1.0)                          # the engine was created with clip_observations 5.0
    with pytest.raises(ValueError, match="clip_obs"):
        VecTaskPython(task, "cpu", 5.0, 0.5)
    with pytest.raises(_lib.MmsError, match="no_such"):
        task.engine.tensor("no_such")
    with pytest.raises(_lib.MmsError, match="out of range"):
        task.engine.set_state("progress", np.zeros((1,), np.int64), env_ids=[4])
    task.engine.close()


def test_package_never_imports_the_oracle():
    """The oracle is the checker: no file of the product package may import, load or name it."""
    pkg = os.path.join(ROOT, "massive_marl_benchmark_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b|libmms_oracle|mms_oracle\.c\"|oracle/_build", text, re.M), os.path.join(dirpath, f)


class CpuImpl:
    """tests/parity.py adapter: the CPU build behind the put / get / post_step / step interface."""

    def __init__(self, task, cfg=None, **kw):
        self.eng = Engine(task, cfg, device="cpu", **kw)
        self.config = self.eng.config

    def put(self, name, arr):
        t = self.eng.tensor(name)
        t.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(arr).reshape(tuple(t.shape)))).to(t.dtype))

    def get(self, name):
        return self.eng.tensor(name).numpy().copy()

    def post_step(self, actions):
        self.eng.tensor("actions").copy_(torch.from_numpy(np.ascontiguousarray(actions, np.float32)))
        self.eng.post_step()

    def step(self, actions):
        self.eng.tensor("actions").copy_(torch.from_numpy(np.ascontiguousarray(actions, np.float32)))
        self.eng.step()

    def close(self):
        self.eng.close()


@pytest.mark.parametrize("check", [parity.fixture_tenant_obs, parity.fixture_tenant_goals, parity.fixture_oneant, parity.fixture_ingenuity, parity.fixture_circle])
def test_reference_fixtures_through_cpu_build(check):
    check(lambda task, cfg=None, **kw: CpuImpl(task, cfg=cfg, **kw), load_golden, "cpu/")


@pytest.mark.parametrize("task,n,steps", [("OneAnt", 64, 100), ("TenAnt", 8, 80), ("MultiIngenuity", 16, 80), ("MultiAntCircle", 16, 80)])
def test_teacher_forced_parity_vs_oracle(task, n, steps):
    """BASELINE configs[0]'s shape (OneAnt, 64 envs) and the other tasks on the CPU build, step for step against the oracle."""
    kw = dict(num_envs=n, seed=5, total_envs=64, env_offset=0)
    if task in ("MultiIngenuity", "MultiAntCircle"):      # (both measure their reward in the GLOBAL frame)
        cfg = default_cfg(task)
        cfg["env"]["envSpacing"] = 0.0
        kw["cfg"] = cfg
    eng, ora = Engine(task, device="cpu", **kw), OracleEngine(task, **kw)
    tf = parity.TeacherForced(ora, lambda k: eng.tensor(k).numpy())
    rng = np.random.default_rng(1)
    for t in range(steps):
        act = rng.uniform(-1.2, 1.2, (n, ora.num_actions)).astype(np.float32)
        if task == "MultiIngenuity":
            act[:, 2::3] = np.abs(act[:, 2::3]) * 0.12
        for name in parity.STATE:
            eng.tensor(name).copy_(torch.from_numpy(ora.tensor(name).copy()))
        tf.before(act)
        eng.tensor("actions").copy_(torch.from_numpy(act))
        eng.step()
        ora.step(act)
        tf.after("%s step %d" % (task, t))
    tf.finish("cpu/teacher_forced/%s" % task, min_live_steps=steps // 2)
    eng.close()


def test_one_ant_cpu_pipeline_ppo_plumbing():
    """BASELINE configs[0]: OneAnt, num_envs = 64, PPO, sim_device = cpu.  The drop-in classes end to end on the CPU build: task
    constructor with device_type="cpu", VecTaskPython, ActorCritic, RolloutStorage (cfg/ppo/config.yaml: nsteps 8, gamma 0.96,
    lam 0.95), one surrogate update -- finite, episodes reset, zero-copy slots hold what the wrapper returned."""
    from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic
    from massive_marl_benchmark_amd.algorithms.rl.ppo.storage import RolloutStorage
    from massive_marl_benchmark_amd.tasks.agent_base.vec_task import VecTaskPython
    from massive_marl_benchmark_amd.tasks.one_ant import OneAnt
    cfg = default_cfg("OneAnt")
    cfg["env"]["numEnvs"] = 64
    cfg["seed"] = 2
    task = OneAnt(cfg, None, "physx", "cpu", 0, True)
    env = VecTaskPython(task, "cpu", 5.0, 1.0)
    assert env.num_envs == 64 and env.observation_space.shape == (60,) and env.action_space.shape == (8,)
    torch.manual_seed(0)
    ac = ActorCritic((60,), (0,), (8,), 0.8, {"pi_hid_sizes": [64, 64], "vf_hid_sizes": [64, 64], "activation": "elu"}, seed=3)
    st = RolloutStorage(64, 8, (60,), (0,), (8,), device="cpu")
    opt = torch.optim.Adam(ac.parameters(), lr=3e-4)
    obs = env.reset()
    states = env.get_state()
    resets = 0
    for it in range(3):
        for _ in range(8):
            a, logp, v, mu, sigma = ac.act(obs, states)
            nxt, rew, done, info = env.step(a)
            st.add_transitions(obs, states, a, rew, done, v, logp, mu, sigma)
            obs = nxt.clone()
            resets += int(done.sum())
        _, _, last, _, _ = ac.act(obs, states)
        st.compute_returns(last, 0.96, 0.95)
        assert bool(torch.isfinite(st.returns).all()) and abs(float(st.advantages.mean())) < 1e-5
        assert abs(float(st.advantages.std()) - 1.0) < 1e-3
        for idx in st.mini_batch_generator(4):
            o = st.observations.view(-1, 60)[idx]
            lp, ent, val, _, _ = ac.evaluate(o, states[:0], st.actions.view(-1, 8)[idx])
            ratio = torch.exp(lp - st.actions_log_prob.view(-1)[idx])
            adv = st.advantages.view(-1)[idx]
            loss = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean() + ((st.returns.view(-1)[idx] - val.view(-1)) ** 2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
        assert bool(torch.isfinite(loss))
        st.clear()
    assert bool(torch.isfinite(obs).all()) and float(obs.abs().max()) <= 5.0
    task.engine.close()


def test_marl_wrappers_and_generators_on_cpu_build():
    """MultiVecTaskPython + SharedRolloutBuffers + the per-agent views' minibatch generators (what mappo_trainer.py:216 iterates
    over) on the CPU build; the shared views and ten SeparatedReplayBuffers driven the reference's way hold the same data, and the
    generators of both yield the reference's 12 / 13-tuples with matching contents under the same torch seed."""
    from massive_marl_benchmark_amd.algorithms.marl.utils.separated_buffer import SeparatedReplayBuffer
    from massive_marl_benchmark_amd.algorithms.marl.utils.shared_buffer import SharedRolloutBuffers
    from massive_marl_benchmark_amd.tasks.agent_base.multi_vec_task import MultiVecTaskPython
    from massive_marl_benchmark_amd.tasks.ten_ant import TenAnt
    n, T, A = 12, 8, 10
    conf = dict(episode_length=T, n_rollout_threads=n, hidden_size=16, recurrent_N=1, gamma=0.99, gae_lambda=0.95, use_gae=True,
                use_popart=False, use_valuenorm=False, use_proper_time_limits=False)

    def make_env():
        cfg = default_cfg("TenAnt")
        cfg["env"]["numEnvs"] = n
        cfg["clip_observations"] = 7.0
        cfg["seed"] = 3
        return MultiVecTaskPython(TenAnt(cfg, None, "physx", "cpu", 0, True, is_multi_agent=True), "cpu")

    g = torch.Generator().manual_seed(0)
    acts = [[torch.rand(n, 8, generator=g) * 2 - 1 for _ in range(A)] for _ in range(T)]
    vals = [torch.randn(n, A, generator=g) for _ in range(T + 1)]
    logp = [[torch.randn(n, 8, generator=g) for _ in range(A)] for _ in range(T)]
    env = make_env()
    bufs = [SeparatedReplayBuffer(conf, env.observation_space[k], env.share_observation_space[k], env.action_space[k], "cpu") for k in range(A)]
    obs, share, _ = env.reset()
    for k in range(A):
        bufs[k].share_obs[0].copy_(share[:, k]); bufs[k].obs[0].copy_(obs[:, k])
    for t in range(T):
        obs, share, rew, dones, _, _ = env.step(acts[t])
        masks = torch.ones(n, A, 1)
        masks[torch.all(dones != 0, dim=1)] = 0
        for k in range(A):
            bufs[k].insert(share[:, k], obs[:, k], torch.zeros(n, 1, 16), torch.zeros(n, 1, 16), acts[t][k], logp[t][k], vals[t][:, k:k + 1],
                           rew[:, k], masks[:, k])
    for k in range(A):
        bufs[k].compute_returns(vals[T][:, k:k + 1], None)
    env.task.engine.close()
    env = make_env()
    sh = SharedRolloutBuffers(conf, env, "cpu")
    sh.warmup()
    for t in range(T):
        rew, dones = sh.env_step(acts[t])
        sh.insert_step(rew, dones, vals[t], acts[t], logp[t])
    sh.compute_returns(vals[T], None)
    for k in (0, 7):
        v, b = sh.agents[k], bufs[k]
        assert torch.equal(v.share_obs, b.share_obs) and torch.equal(v.obs, b.obs) and torch.equal(v.rewards, b.rewards)
        assert torch.allclose(v.returns[:T], b.returns[:T], atol=1e-5)
        adv = b.returns[:-1] - b.value_preds[:-1]
        for gen, args in (("feed_forward_generator", (adv, 4)), ("naive_recurrent_generator", (adv, 3)), ("recurrent_generator", (adv, 2, 4))):
            torch.manual_seed(5)
            one = list(getattr(v, gen)(*args))
            torch.manual_seed(5)
            two = list(getattr(b, gen)(*args))
            assert len(one) == len(two) == args[1]
            for x, y in zip(one, two):
                assert len(x) == len(y) == 13                                         # factor is present in both buffer kinds
                for i, (p, q) in enumerate(zip(x, y)):
                    assert (p is None and q is None) or (p.shape == q.shape and torch.allclose(p, q, atol=1e-5)), (gen, i)
        # shapes of one feed-forward batch: [batch, dim] rows in the reference's order
        x = one_ff = next(iter(v.feed_forward_generator(adv, 4)))
        mb = T * n // 4
        assert x[0].shape == (mb, 388) and x[1].shape == (mb, 46) and x[4].shape == (mb, 8) and x[5].shape == (mb, 1) and x[10].shape == (mb, 1)
        assert x[11] is None and x[12].shape == (mb, 1)
    env.task.engine.close()


def test_multi_ant_circle_wrappers_on_cpu_build():
    """MultiAntCircle (intended semantics, tasks/multi_ant_circle.py) through both wrappers on the CPU build: the single-agent view
    (76 observations, 16 actions), the MultiAgent view (two agents x 38, share_obs 76, no shared tail), resets by fall and by time,
    the ring reward on envs at the global origin, parse_task knows the name."""
    from massive_marl_benchmark_amd.tasks.agent_base.multi_vec_task import MultiVecTaskPython
    from massive_marl_benchmark_amd.tasks.agent_base.vec_task import VecTaskPython
    from massive_marl_benchmark_amd.tasks.multi_ant_circle import MultiAntCircle
    from massive_marl_benchmark_amd.utils.parse_task import _TASKS
    assert _TASKS["MultiAntCircle"] is MultiAntCircle
    cfg = default_cfg("MultiAntCircle")
    cfg["env"]["numEnvs"], cfg["env"]["envSpacing"], cfg["env"]["episodeLength"] = 16, 0.0, 40
    task = MultiAntCircle(cfg, None, "physx", "cpu", 0, True)
    env = VecTaskPython(task, "cpu", 5.0, 1.0)
    assert env.num_obs == 76 and env.num_acts == 16 and task.root_states.shape == (48, 13) and task.ant_root_states.shape == (16, 2, 13)
    obs = env.reset()
    assert obs.shape == (16, 76) and torch.allclose(task.ant_root_states[:, 0, 0], torch.full((16,), 3.0)) and torch.allclose(task.ant_root_states[:, 1, 0], torch.full((16,), -3.0))
    g = torch.Generator().manual_seed(0)
    seen_reset, rewards = 0, []
    for t in range(60):
        obs, rew, done, _ = env.step(torch.rand(16, 16, generator=g) * 2 - 1)
        seen_reset += int(done.sum())
        rewards.append(rew.clone())
        assert torch.equal(task.prev[:, 0:2], task.obs_buf[:, 0:2]) and torch.equal(task.prev[:, 2:4], task.obs_buf[:, 38:40])
    assert seen_reset >= 16 and bool(torch.isfinite(torch.stack(rewards)).all())          # every env timed out at least once (40 steps)
    r = torch.stack(rewards)
    assert float(r.max()) > 1.5 and float(r.min()) <= -2.0 + 1e-6                           # on-ring bonus seen, deaths seen
    assert float(task.root_states.view(16, 3, 13)[:, 2, 1].min()) > 900.0                   # the engine's box stayed out of the way
    task.engine.close()
    cfg = default_cfg("MultiAntCircle")
    cfg["env"]["numEnvs"], cfg["clip_observations"] = 8, 7.0
    task = MultiAntCircle(cfg, None, "physx", "cpu", 0, True, is_multi_agent=True)
    menv = MultiVecTaskPython(task, "cpu")
    assert menv.num_agents == 2 and menv.num_observations == 38 and menv.nums_share_observations == 76
    o, s, _ = menv.reset()
    oa, sa, ra, da, _, _ = menv.step([torch.rand(8, 8, generator=g) * 2 - 1 for _ in range(2)])
    assert oa.shape == (8, 2, 38) and sa.shape == (8, 2, 76) and ra.shape == (8, 2, 1) and da.shape == (8, 2)
    assert torch.equal(oa[:, 1], torch.clamp(task.obs_buf[:, 38:76], -7, 7)) and torch.equal(sa[:, 0], torch.clamp(task.obs_buf, -7, 7))
    task.engine.close()
    with pytest.raises(_lib.MmsError, match="two ants"):
        Engine("MultiAntCircle", num_envs=4, num_agents=3, device="cpu")


def test_make_entry_glue():
    """`make(task, algo)` = the reference's `agents.make` (agents/utils/package_utils.py:20-56) without isaacgym."""
    from massive_marl_benchmark_amd.utils.package_utils import make
    cpu = ["--sim_device", "cpu", "--pipeline", "cpu", "--rl_device", "cpu", "--seed", "1"]
    env = make("OneAnt", "ppo", cpu + ["--num_envs", "8"])
    assert type(env).__name__ == "VecTaskPython" and env.reset().shape == (8, 60)
    env.task.engine.close()
    env = make("TenAnt", "mappo", cpu + ["--num_envs", "4"])
    assert type(env).__name__ == "MultiVecTaskPython" and env.num_agents == 10 and env.reset()[0].shape == (4, 10, 46)
    env.task.engine.close()
    with pytest.raises(ValueError):
        make("OneAnt", "mtppo", cpu)


def test_split_operand_layers_on_cpu_build():
    """mms_split_planes / mms_linear_group_act_split on the CPU build (the same ABI as the HIP kernels of csrc/split_kernels.hip): the
    three bf16 planes of a value sum back to it EXACTLY, columns past K are zero, and a layer on the planes equals torch's
    Linear + ELU; the ActorCritic module takes the split path on the CPU build when the shapes allow and follows an in-place
    parameter update."""
    L = _lib.lib_cpu()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    nbytes = lambda rows, K: rows * ((K + 31) // 32) * 192

    def join(planes, rows, K):
        v = planes.view(torch.bfloat16).view(rows, (K + 31) // 32, 3, 32).float()
        return ((v[:, :, 0] + v[:, :, 1]) + v[:, :, 2]).reshape(rows, -1)[:, :K], v
    torch.manual_seed(3)
    for rows, K, pitch in ((128, 388, 388), (9, 36, 40), (4, 1, 4)):
        x = torch.randn(rows, pitch)
        x[0, 0] = 1.0 + 2.0 ** -23
        planes = torch.full((nbytes(rows, K),), 0xAB, dtype=torch.uint8)
        assert L.mms_split_planes(-1, rows, K, pitch, p(x), p(planes), None) == 0
        back, v = join(planes, rows, K)
        assert torch.equal(back, x[:, :K])
        assert float(v.permute(0, 1, 3, 2).reshape(rows, -1, 3)[:, K:].abs().sum()) == 0.0
    M, N, K = 128, 256, 100
    x = [torch.randn(M, K) for _ in range(2)]
    w = [torch.randn(N, K) / K ** 0.5 for _ in range(2)]
    b = [torch.randn(N) for _ in range(2)]
    xp = [torch.empty(nbytes(M, K), dtype=torch.uint8) for _ in range(2)]
    wp = [torch.empty(nbytes(N, K), dtype=torch.uint8) for _ in range(2)]
    for g in range(2):
        assert L.mms_split_planes(-1, M, K, 0, p(x[g]), p(xp[g]), None) == 0 and L.mms_split_planes(-1, N, K, 0, p(w[g]), p(wp[g]), None) == 0
    for planes_out in (1, 0):
        ys = [torch.empty(nbytes(M, N) if planes_out else M * N * 4, dtype=torch.uint8) for _ in range(2)]
        assert L.mms_linear_group_act_split(-1, 2, M, N, K, arr(xp), arr(wp), arr(b), arr(ys), 1, planes_out, None, None, None, None, None, 0, None) == 0, _lib.last_error(None, L)
        for g in range(2):
            out = join(ys[g], M, N)[0] if planes_out else ys[g].view(torch.float32).view(M, N)
            ref = torch.nn.functional.elu(torch.nn.functional.linear(x[g].double(), w[g].double(), b[g].double()))
            assert float((out.double() - ref).abs().max()) < 2e-6
    assert L.mms_linear_group_act_split(-1, 1, 100, 128, 32, arr(xp[:1]), arr(wp[:1]), arr(b[:1]), arr(ys[:1]), 1, 0, None, None, None, None, None, 0, None) != 0
    assert "multiples of 128" in _lib.last_error(None, L)
    # the module on the CPU build
    from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic
    ac = ActorCritic((36,), (0,), (8,), 0.8, {"pi_hid_sizes": [128, 128], "vf_hid_sizes": [128, 128], "activation": "elu"}, seed=3)
    ac.split_format = "bf16x3"
    obs, states = torch.randn(128, 36), torch.zeros(128, 0)
    hidden = ac._fused_hidden(obs, obs)                               # (on the CPU build one tile is enough: split_min_tiles None -> 1)
    assert hidden is not None and ac._split_bufs
    with torch.no_grad():
        assert float((ac.actor[:-1](obs) - hidden[0]).abs().max()) < 1e-5 and float((ac.critic[:-1](obs) - hidden[1]).abs().max()) < 1e-5
        for q in ac.parameters():
            q.add_(0.05 * torch.randn_like(q))                        # an in-place "optimizer step"
        hidden = ac._fused_hidden(obs, obs)
        assert float((ac.actor[:-1](obs) - hidden[0]).abs().max()) < 1e-5 and float((ac.critic[:-1](obs) - hidden[1]).abs().max()) < 1e-5
        assert float((ac.value(obs) - ac.critic(obs)).abs().max()) < 1e-5 if obs.is_cuda else True


def test_split16_layers_on_cpu_build():
    """mms_split_planes16_group / mms_linear_group_act_split16 on the CPU build (the ABI of csrc/split16_kernels.hip): two fp16 planes
    under a power-of-two row scale keep every element to 2^-22 of itself (also far below the row's largest magnitude: subnormal
    halves), columns past K are zero, the bound chain's scales keep every hidden activation below 2^14 on the fp16 axis, a layer on
    the planes equals torch's Linear + ELU in float64 to fp32 accuracy, and the module's default path follows parameter updates."""
    L = _lib.lib_cpu()
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    nbytes = lambda rows, K: rows * ((K + 31) // 32) * 128

    def join(planes, rows, K, inv):
        v = planes.view(torch.float16).view(rows, (K + 31) // 32, 2, 32).double()
        return ((v[:, :, 0] + v[:, :, 1] / 2048.0).reshape(rows, -1) * inv.double()[:, None])[:, :K], v
    torch.manual_seed(5)
    for rows, K, pitch in ((128, 388, 388), (9, 36, 40), (4, 1, 4)):
        x = torch.randn(rows, pitch) * torch.exp2(torch.randint(-20, 12, (rows, 1)).float())
        x[0, 0] = x[0, :K].abs().max() * 2.0 ** -20                   # far below its row's scale
        if rows > 2:
            x[2] = 0.0                                                # an all-zero row keeps scale 1
        planes = torch.full((nbytes(rows, K),), 0xAB, dtype=torch.uint8)
        sc, iv = torch.empty(rows), torch.empty(rows)
        assert L.mms_split_planes16_group(-1, 1, rows, K, pitch, arr([x]), arr([planes]), arr([sc]), arr([iv]), 0, 0, None, None, None, None, 0.0, None) == 0, _lib.last_error(None, L)
        back, v = join(planes, rows, K, iv)
        ref = x[:, :K].double()
        big = ref.abs().max(1, keepdim=True).values
        assert float(((back - ref).abs() / ref.abs().clamp_min(1e-300)).max()) <= 2.0 ** -21 + 1e-12 or float(((back - ref).abs() / big.clamp_min(1e-300)).max()) < 2.0 ** -34
        assert float(((back - ref).abs() - 2.0 ** -21 * ref.abs()).clamp_min(0).max()) <= float((big * 2.0 ** -35).max())
        assert torch.equal(sc * iv, torch.ones(rows)) and float((ref.abs().max(1).values * sc.double()).max()) <= 2.0 ** 14
        assert float(v.permute(0, 1, 3, 2).reshape(rows, -1, 2)[:, K:].abs().sum()) == 0.0
    # a chain of two layers: scales from the bound chain, planes between the layers
    M, K, H = 128, 100, 256
    x = torch.randn(M, K) * 3
    w = [[torch.randn(H, K) / K ** 0.5, torch.randn(H, H) / H ** 0.5] for _ in range(2)]
    b = [[torch.randn(H), torch.randn(H)] for _ in range(2)]
    wp = [[torch.empty(nbytes(wl.shape[0], wl.shape[1]), dtype=torch.uint8) for wl in net] for net in w]
    winv = [[torch.empty(H) for _ in net] for net in w]
    for g in range(2):
        for li in range(2):
            sc = torch.empty(H)
            assert L.mms_split_planes16_group(-1, 1, H, w[g][li].shape[1], 0, arr([w[g][li]]), arr([wp[g][li]]), arr([sc]), arr([winv[g][li]]), 0, 0, None, None, None, None, 0.0, None) == 0
    chain = torch.stack([torch.stack([torch.stack([w[g][0].abs().sum(1).max(), b[g][0].abs().max()])]) for g in range(2)]).contiguous()    # [2 chains][1 layer][2]
    xp, xs, xi = torch.empty(nbytes(M, K), dtype=torch.uint8), torch.empty(M), torch.empty(M)
    cs, ci = torch.empty(2, 1, M), torch.empty(2, 1, M)
    assert L.mms_split_planes16_group(-1, 1, M, K, 0, arr([x]), arr([xp]), arr([xs]), arr([xi]), 2, 1, arr([chain]), arr([cs]), arr([ci]), None, 0.0, None) == 0, _lib.last_error(None, L)
    hp = [torch.empty(nbytes(M, H), dtype=torch.uint8) for _ in range(2)]
    assert L.mms_linear_group_act_split16(-1, 2, M, H, K, arr([xp, xp]), arr([wp[0][0], wp[1][0]]), arr([b[0][0], b[1][0]]), arr(hp), arr([xi, xi]),
                                          arr([winv[0][0], winv[1][0]]), arr([cs[0, 0], cs[1, 0]]), 1, 1, None, None, None, None, None, 0, None) == 0, _lib.last_error(None, L)
    ys = [torch.empty(M, H) for _ in range(2)]
    assert L.mms_linear_group_act_split16(-1, 2, M, H, H, arr(hp), arr([wp[0][1], wp[1][1]]), arr([b[0][1], b[1][1]]), arr(ys), arr([ci[0, 0], ci[1, 0]]),
                                          arr([winv[0][1], winv[1][1]]), None, 1, 0, None, None, None, None, None, 0, None) == 0, _lib.last_error(None, L)
    for g in range(2):
        h1 = torch.nn.functional.elu(torch.nn.functional.linear(x.double(), w[g][0].double(), b[g][0].double()))
        got1, v = join(hp[g], M, H, ci[g, 0])
        assert float((got1 - h1).abs().max()) < 1e-6 * float(h1.abs().max())      # (the host build's product is an fp32 fma chain)
        assert float(v[:, :, 0].abs().max()) <= 2.0 ** 14             # the hi plane of every hidden activation is inside the bound
        ref = torch.nn.functional.elu(torch.nn.functional.linear(got1, w[g][1].double(), b[g][1].double()))
        assert float((ys[g].double() - ref).abs().max()) < 1e-6 * float(ref.abs().max())
    assert L.mms_linear_group_act_split16(-1, 2, M, H, K, arr([xp, xp]), arr([wp[0][0], wp[1][0]]), arr([b[0][0], b[1][0]]), arr(hp), arr([xi, xi]),
                                          arr([winv[0][0], winv[1][0]]), None, 1, 1, None, None, None, None, None, 0, None) != 0
    assert "y_scale" in _lib.last_error(None, L)
    # the module on the CPU build (default format)
    from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic
    ac = ActorCritic((36,), (0,), (8,), 0.8, {"pi_hid_sizes": [128, 256, 128], "vf_hid_sizes": [128, 256, 128], "activation": "elu"}, seed=3)
    assert ac.split_format == "f16x2"
    obs = torch.randn(128, 36) * 4
    hidden = ac._fused_hidden(obs, obs)
    assert hidden is not None and any(k[1] == "h" for k in ac._split_bufs)
    with torch.no_grad():
        assert float((ac.actor[:-1](obs) - hidden[0]).abs().max()) < 1e-5 and float((ac.critic[:-1](obs) - hidden[1]).abs().max()) < 1e-5
        for q in ac.parameters():
            q.add_(0.05 * torch.randn_like(q))                        # an in-place "optimizer step": weights AND biases move
        hidden = ac._fused_hidden(obs, obs)
        assert float((ac.actor[:-1](obs) - hidden[0]).abs().max()) < 1e-5 and float((ac.critic[:-1](obs) - hidden[1]).abs().max()) < 1e-5
        big = obs * 1e4                                               # rows far outside the training range: the row scale follows
        hidden = ac._fused_hidden(big, big)
        ref = ac.actor[:-1](big)
        assert float((ref - hidden[0]).abs().max()) < 1e-5 * float(ref.abs().max())


def test_obs_planes_from_the_step_on_cpu_build():
    """mms_bind_obs_planes16 on the CPU build: tests/obs_planes_check.py (TenAnt and OneAnt rows)."""
    from obs_planes_check import check_obs_planes
    check_obs_planes("cpu")
    check_obs_planes("cpu", task="OneAnt")


@pytest.mark.skipif(not os.path.isdir("/root/reference/agents"), reason="the reference tree is not present on this machine (it does not travel to the GPU box)")
def test_reference_learners_drop_in_unchanged(tmp_path):
    """The reference's real learners, imported in place and unmodified, over this build's VecTaskPython / MultiVecTaskPython on the
    CPU build: PPO.run (agents/algorithms/rl/ppo/ppo.py:99-175), Runner.run (agents/algorithms/marl/runner.py:114-151) as mappo, happo
    and hatrpo (the sequential update_factor path, :266-316), DDPG.run (rl/ddpg/ddpg.py:116-204) and TD3.run (rl/td3/td3.py) on
    MultiIngenuity -- each with the reference's own storage / module / buffer classes and with this build's drop-in ones, mappo also
    with the grouped policy inference as the Runner's collect step.  The committed log of the same script:
    tests/golden/reference_learners_dropin.log."""
    log = tmp_path / "dropin.log"
    env = dict(os.environ, MMS_DROPIN_LOG=str(log), OMP_NUM_THREADS="4")
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tests", "golden", "run_reference_learners.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    text = log.read_text()
    assert text.count(": ok") == 13 and "GroupedPolicyInference as Runner.collect: ok" in text
    for learner, n in (("PPO.run", 2), ("mappo, unmodified", 3), ("happo, unmodified", 2), ("hatrpo, unmodified", 2), ("DDPG.run", 2), ("TD3.run", 2)):
        assert text.count(learner) == n, (learner, text)


def test_refresh_entry_points_and_module_refresh_on_cpu_build():
    """The device-side refresh of the f16x2 layers' weight planes / bound chain (mms_weight_planes16_group, mms_chain_refresh16),
    of the folded-LayerNorm layers' weight side (mms_fold_planes16_group, mms_fold_scales16_group) and ActorCritic.refresh() on the CPU build: tests/refresh_check.py (the HIP build runs the same list)."""
    import refresh_check
    refresh_check.check_refresh_entry_points("cpu")
    refresh_check.check_fold_entry_points("cpu")
    assert refresh_check.check_module_refresh("cpu") > 1e-3
    assert refresh_check.check_module_refresh("cpu", fmt="bf16x3") > 1e-3


def test_task_step_leaves_no_action_binding():
    """BaseTask.step binds the caller's action tensor for the launch only (mms_bind_actions; the reference clones it, ten_ant.py:887): a
    direct Engine.step() behind it reads the engine's own "actions" buffer again -- the trajectory of task.step(a); write b into
    engine.tensor("actions"); engine.step() equals that of an engine fed a, b through its buffer."""
    from massive_marl_benchmark_amd.tasks.one_ant import OneAnt
    cfg = default_cfg("OneAnt")
    cfg["env"]["numEnvs"] = 8
    cfg["seed"] = 5
    g = torch.Generator().manual_seed(2)
    a, b = (torch.rand(8, 8, generator=g) * 2 - 1 for _ in range(2))
    task = OneAnt(cfg, None, "physx", "cpu", 0, True)
    other = OneAnt(cfg, None, "physx", "cpu", 0, True)
    for t in (task, other):
        t.engine.reset_all()
    keep = a.clone()
    task.step(a)                                                      # binds `a`, steps, unbinds
    a.fill_(123.0)                                                    # (the caller's tensor is the caller's again)
    task.engine.tensor("actions").copy_(b)
    task.engine.step()
    for acts in (keep, b):
        other.engine.tensor("actions").copy_(acts)
        other.engine.step()
    for name in ("root_states", "dof_state", "obs", "rew"):
        assert torch.equal(task.engine.tensor(name), other.engine.tensor(name)), name
    task.engine.close()
    other.engine.close()


def test_policy_head_fused_into_the_step_on_cpu_build():
    """mms_bind_policy_head on the CPU build (tests/head_fusion_check.py; the HIP build runs the same check at 4096 envs, where the step
    kernel has the layout for it): the fused form leaves bit for bit what mms_ppo_heads_act + mms_step leave; and the calls the binding
    refuses."""
    import head_fusion_check
    assert head_fusion_check.check_head_fusion("cpu", 32)
    assert head_fusion_check.check_head_bind_errors("cpu", 32)
    from massive_marl_benchmark_amd.engine import Engine
    one = Engine("OneAnt", num_envs=16, device="cpu")
    assert not one.takes_policy_head()
    one.close()
    odd = Engine("TenAnt", num_envs=24, device="cpu")
    assert not odd.takes_policy_head()
    odd.close()
