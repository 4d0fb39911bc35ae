"""CPU-side checks: the C-ABI library loads and exports every symbol include/mms.h declares, struct layouts
agree, the built-in configuration equals the reference YAML, the MJCF-subset compiler reproduces the built-in
ant description, argument parsing, and the multi-rank logic (gloo, world size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

from massive_marl_benchmark_amd import _lib
from massive_marl_benchmark_amd.model import (ANT_DESCRIPTION, MMS_ABI_VERSION, MmsConfig, default_cfg, load_mjcf_ant, make_config, task_dims)

REF = "/root/reference"


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mms.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mms_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_match_loader_table():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "massive_marl_benchmark_amd", "csrc")])
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (mms_[a-z_0-9]+)", out))
    assert set(declared_symbols()) <= exported, set(declared_symbols()) - exported
    # the shared object carries a gfx950 code object
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_library_loads_without_gpu_and_refuses_cpu():
    L = _lib.lib()                                   # dlopen + symbol binding only; no compute
    assert L.mms_abi_version() == MMS_ABI_VERSION
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device error path is not reachable")
    cfg = make_config("TenAnt", num_envs=4)
    h = ctypes.c_void_p()
    assert L.mms_create(ctypes.byref(cfg), ctypes.byref(h)) != 0          # must fail loudly: no CPU fallback
    assert "HIP" in _lib.last_error(None) or "device" in _lib.last_error(None)
    from massive_marl_benchmark_amd.engine import Engine
    with pytest.raises(_lib.MmsError):
        Engine("TenAnt", num_envs=4)


def test_struct_layout_matches_c():
    src = '#include "include/mms.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(){printf("%zu %zu %zu %zu\\n", sizeof(mms_config), sizeof(mms_model), offsetof(mms_config, model), offsetof(mms_model, gravity));return 0;}'
    exe = "/tmp/mms_layout_test"
    subprocess.run(["gcc", "-x", "c", "-", "-I", ROOT, "-o", exe], input=src.encode(), check=True, cwd=ROOT)
    a, b, c, d = map(int, subprocess.check_output([exe]).split())
    from massive_marl_benchmark_amd.model import MmsModel
    assert (a, b, c, d) == (ctypes.sizeof(MmsConfig), ctypes.sizeof(MmsModel), MmsConfig.model.offset, MmsModel.gravity.offset)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
@pytest.mark.parametrize("task", ["TenAnt", "OneAnt", "MultiIngenuity"])
def test_default_cfg_equals_reference_yaml(task):
    import yaml
    ref = yaml.safe_load(open(os.path.join(REF, "cfg", task + ".yaml")))
    assert default_cfg(task) == ref


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_mjcf_compiler_matches_builtin_description():
    d = load_mjcf_ant(os.path.join(REF, "assets/mjcf/open_ai_assets/ant/nv_ant.xml"))
    for key in ("armature", "damping", "gear", "density", "limb_radius"):
        assert d[key] == ANT_DESCRIPTION[key], key
    assert d["torso"]["sphere_radius"] == ANT_DESCRIPTION["torso"]["sphere_radius"]
    assert d["torso"]["aux_capsules"] == ANT_DESCRIPTION["torso"]["aux_capsules"]
    assert d["legs"] == ANT_DESCRIPTION["legs"]


def test_config_and_dims():
    c = make_config("TenAnt", num_envs=4096, env_offset=4096, total_envs=32768, seed=3)
    assert (c.num_envs, c.num_agents, c.substeps, c.max_episode_length) == (4096, 10, 2, 1000)
    assert abs(c.dt - 0.0166) < 1e-7 and abs(c.model.gravity - 9.81) < 1e-6
    assert task_dims("TenAnt", 10) == (11, 80, 80, 388, 42)
    assert task_dims("OneAnt", 1) == (2, 8, 8, 60, 6)
    assert task_dims("MultiIngenuity", 4) == (4, 16, 24, 52, 12)
    assert abs(make_config("MultiIngenuity").model.gravity - 3.721) < 1e-6


def test_get_args_and_load_cfg():
    from massive_marl_benchmark_amd.utils.config import get_args, load_cfg, parse_sim_params
    args = get_args(["--task", "TenAnt", "--algo", "mappo", "--num_envs", "256", "--seed", "5", "--headless"])
    assert args.task_type == "MultiAgent" and args.device == "cuda" and args.device_id == 0
    cfg, cfg_train, logdir = load_cfg(args)
    assert cfg["env"]["numEnvs"] == 256 and cfg_train["seed"] == 5 and cfg["sim"]["substeps"] == 2
    sp = parse_sim_params(args, cfg, cfg_train)
    assert sp.dt == 0.0166 and sp.substeps == 2
    args = get_args(["--task", "OneAnt", "--algo", "ppo"])
    assert args.task_type == "Python"


def test_actor_critic_matches_reference_fixture():
    """The ActorCritic mirror loaded with the reference network's weights: act_inference / evaluate on the CPU (pure torch)
    against the outputs of the reference's module.py (fixture ppo_act); the sampling tail itself needs the HIP device."""
    import torch
    from conftest import load_golden
    from massive_marl_benchmark_amd.algorithms.rl.ppo.module import ActorCritic
    g = load_golden("ppo_act")
    ac = ActorCritic((48,), (0,), (8,), 0.8, {"pi_hid_sizes": [32, 32, 16], "vf_hid_sizes": [32, 32, 16], "activation": "elu"})
    names = [k for k in g.files if k.startswith(("actor_", "critic_")) or k == "log_std"]
    sd = {}
    for k in names:
        parts = k.split("_")
        key = k if k == "log_std" else "%s.%s.%s" % (parts[0], parts[1], parts[2])
        sd[key] = torch.from_numpy(np.asarray(g[k]))
    ac.load_state_dict(sd)
    obs, act = torch.from_numpy(g["obs"]), torch.from_numpy(g["actions"])
    with torch.no_grad():
        inf = ac.act_inference(obs)
        lp, ent, val, mu, sigma = ac.evaluate(obs, torch.zeros(obs.shape[0], 0), act)
    assert np.max(np.abs(inf.numpy() - g["inference"])) < 1e-5
    assert np.max(np.abs(lp.numpy() - g["eval_log_prob"])) < 1e-4
    assert np.max(np.abs(ent.numpy() - g["eval_entropy"])) < 1e-4
    assert np.max(np.abs(val.numpy() - g["eval_value"])) < 1e-5
    np.testing.assert_array_equal(sigma.numpy(), g["sigma"])
    # a module that lives on torch's "cpu" device samples through the CPU build of the ABI (lib/libmms_cpu.so: the same
    # per-action function as the HIP kernel, rollout_lane.h) -- the owner's explicit choice of device, not a fallback
    act2, logp2, val2, mu2, sig2 = ac.act(obs, torch.zeros(obs.shape[0], 0))
    with torch.no_grad():
        lp3, _, v3, m3, _ = ac.evaluate(obs, torch.zeros(obs.shape[0], 0), act2)
    assert float((lp3 - logp2).abs().max()) < 5e-3 and float((v3 - val2).abs().max()) < 1e-5 and float((m3 - mu2).abs().max()) < 1e-5
    np.testing.assert_array_equal(sig2.numpy(), g["sigma"])


@pytest.mark.parametrize("algo", ["ddpg", "td3", "sac"])
def test_replay_buffer_matches_reference_fixture(algo):
    """ReplayBuffer (algorithms/rl/{ddpg,td3,sac}/storage.py) against what the reference's own class produced for the same 13
    adds into a 5-row ring: cursor / fullfill after every add (the overflow rule that skips row 0), the ring's contents,
    get_statistics and the random.sample mini-batches under the same random.seed."""
    import importlib
    import random

    import torch
    from conftest import load_golden
    g = load_golden("replay_buffer")
    ReplayBuffer = importlib.import_module("massive_marl_benchmark_amd.algorithms.rl.%s.storage" % algo).ReplayBuffer
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    NE, R, OD, AD = 6, 5, 13, 4
    buf = ReplayBuffer(NE, R, 8, 2, (OD,), (0,), (AD,), "cpu", "sequential")
    part = ReplayBuffer(NE, R, 8, 2, (OD,), (0,), (AD,), "cpu", "sequential")
    for k in range(13):
        assert buf.slot() == (k if k < R else g[algo + "_cursor"][k] - 1)
        args = (t(g["in_obs"][k]), torch.zeros(NE, 0), t(g["in_act"][k]), t(g["in_rew"][k]), t(g["in_nobs"][k]), t(g["in_done"][k]))
        buf.add_transitions(*args)
        if k < 3:
            part.add_transitions(*args)
        assert buf.step == g[algo + "_cursor"][k] and int(buf.fullfill) == g[algo + "_full"][k]
    for name in ("observations", "next_observations", "actions", "rewards", "dones"):
        assert np.array_equal(getattr(buf, name).numpy(), g[algo + "_" + name]), name
    ln, rw = buf.get_statistics()
    assert float(ln) == float(g[algo + "_mean_traj_len"]) and abs(float(rw) - float(g[algo + "_mean_reward"])) < 1e-6
    random.seed(32)
    assert np.array_equal(np.array(buf.mini_batch_generator(3)), g[algo + "_batches"])
    ln, rw = part.get_statistics()
    assert float(ln) == float(g[algo + "_part_mean_traj_len"]) and abs(float(rw) - float(g[algo + "_part_mean_reward"])) < 1e-6
    random.seed(33)
    assert np.array_equal(np.array(part.mini_batch_generator(4)), g[algo + "_part_batches"])
    # a row that is already in place (written by the engine into the slot) is not copied: same result, and the ring row
    # keeps its address
    k = part.slot()
    row = part.next_observations[k]
    row.copy_(t(g["in_nobs"][3]))
    part.add_transitions(t(g["in_obs"][3]), torch.zeros(NE, 0), t(g["in_act"][3]), t(g["in_rew"][3]), row, t(g["in_done"][3]))
    assert np.array_equal(part.next_observations[3].numpy(), g["in_nobs"][3]) and part.step == 4


@pytest.mark.parametrize("algo", ["ddpg", "td3"])
def test_offpolicy_actor_critic_matches_reference_fixture(algo):
    """MLPActorCritic (algorithms/rl/{ddpg,td3}/module.py): the reference's state_dict loads key for key, deterministic and noisy
    actions (same CPU generator stream under torch.manual_seed) and the Q values match what the reference produced."""
    import importlib

    import torch
    from conftest import load_golden
    from massive_marl_benchmark_amd import spaces
    g = load_golden("offpolicy_act")
    mod = importlib.import_module("massive_marl_benchmark_amd.algorithms.rl.%s.module" % algo)
    ac = mod.MLPActorCritic(spaces.Box(-np.inf * np.ones(52), np.inf * np.ones(52)), spaces.Box(-np.ones(24), np.ones(24)), 0.1, "cpu",
                            hidden_sizes=[32, 32, 32])
    keys = [str(k) for k in g[algo + "_keys"]]
    assert list(ac.state_dict().keys()) == keys
    ac.load_state_dict({k: torch.from_numpy(g[algo + "_sd_" + k.replace(".", "_")]) for k in keys})
    o = torch.from_numpy(g[algo + "_obs"])
    det = ac.act(o)
    assert not det.requires_grad and np.allclose(det.numpy(), g[algo + "_det"], atol=1e-6)
    torch.manual_seed(42)
    noisy = ac.act(o, deterministic=False)
    assert np.allclose(noisy.numpy(), g[algo + "_noisy"], atol=1e-6) and float(noisy.abs().max()) <= 1.0
    with torch.no_grad():
        qs = [ac.q(o, det)] if algo == "ddpg" else [ac.q1(o, det), ac.q2(o, det)]
    assert np.allclose(torch.stack(qs).numpy(), g[algo + "_q"], atol=1e-5)
    # the training path keeps its graph (ddpg.py: loss_pi = -q(o, pi(o)))
    assert ac.pi(o).requires_grad


def test_sharded_env_grid_matches_single():
    """Env sharding: rank r of R owns envs [r*N, (r+1)*N); its env origins and RNG keys are
    those of the corresponding envs of one big engine.  Checked on the oracle (same host code path as the product)."""
    from oracle.oracle import OracleEngine
    big = OracleEngine("TenAnt", num_envs=16, seed=11)
    parts = [OracleEngine("TenAnt", num_envs=8, seed=11, env_offset=8 * r, total_envs=16) for r in range(2)]
    rng = np.random.default_rng(0)
    for t in range(30):
        a = rng.uniform(-1, 1, (16, 80)).astype(np.float32)
        big.step(a)
        for r, p in enumerate(parts):
            p.step(a[8 * r:8 * r + 8])
    for name in ("obs", "rew", "reset", "progress", "root_states", "dof_state"):
        cat = np.concatenate([p.tensor(name) for p in parts])
        np.testing.assert_array_equal(cat, big.tensor(name), err_msg=name)     # bit-identical: nothing crosses envs


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["MASTER_PORT"], rank=rank, world_size=world)
# each rank holds the rollout of its env shard: the PRODUCT's RolloutStorage(process_group=...) (CPU build of the ABI, explicit
# choice of device "cpu") all-reduces {sum, sum sq, count} between its two kernels, so every rank normalises with the GLOBAL
# statistics -- the result must equal one storage holding both shards
from massive_marl_benchmark_amd.algorithms.rl.ppo.storage import RolloutStorage
T, NL = 8, 32
rng = np.random.default_rng(123)
full = {k: rng.normal(0.5, 2.0, size=(T, world * NL, 1)).astype(np.float32) for k in ("rewards", "values")}
full["dones"] = (rng.random((T, world * NL, 1)) < 0.1).astype(np.uint8)
last = rng.normal(size=(world * NL, 1)).astype(np.float32)
def fill(st, lo, hi):
    st.rewards.copy_(torch.from_numpy(full["rewards"][:, lo:hi])); st.values.copy_(torch.from_numpy(full["values"][:, lo:hi]))
    st.dones.copy_(torch.from_numpy(full["dones"][:, lo:hi]))
    st.compute_returns(torch.from_numpy(last[lo:hi]), 0.96, 0.95)
mine = RolloutStorage(NL, T, (4,), (0,), (2,), device="cpu", process_group=dist.group.WORLD)
fill(mine, rank * NL, (rank + 1) * NL)
whole = RolloutStorage(world * NL, T, (4,), (0,), (2,), device="cpu")
fill(whole, 0, world * NL)
assert torch.equal(mine.returns, whole.returns[:, rank * NL:(rank + 1) * NL])
assert float((mine.advantages - whole.advantages[:, rank * NL:(rank + 1) * NL]).abs().max()) < 1e-6
alone = RolloutStorage(NL, T, (4,), (0,), (2,), device="cpu")             # without the group: shard-local statistics, a different result
fill(alone, rank * NL, (rank + 1) * NL)
assert float((alone.advantages - mine.advantages).abs().max()) > 1e-4
# env partition bookkeeping used by bench.py
from bench import shard_for_rank
off, total = shard_for_rank(rank, world, 4096)
assert (off, total) == (rank * 4096, world * 4096)
# the MARL exchange step: all-gather of the env-sharded rollout into global env order (centralised critic, agent-parallel training)
from massive_marl_benchmark_amd.algorithms.marl.utils.shared_buffer import all_gather_envs
T, NL = 3, 5
glob = {"share_obs": rng.standard_normal((T + 1, world * NL, 7)).astype(np.float32),
        "value_preds": rng.standard_normal((T + 1, world * NL, 10)).astype(np.float32),
        "rewards": rng.standard_normal((T, world * NL, 1)).astype(np.float32)}
local = {k: torch.from_numpy(np.ascontiguousarray(v[:, rank * NL:(rank + 1) * NL])) for k, v in glob.items()}
got = all_gather_envs(local)
for k, v in glob.items():
    assert got[k].shape == v.shape and np.array_equal(got[k].numpy(), v), k
# HAPPO's sequential factor chain with agent-parallel training (runner.py:266-316): rank r trains agents r, r + world, ...
from massive_marl_benchmark_amd.algorithms.marl.utils.shared_buffer import happo_factor_chain
A, T, N, D = 6, 3, 8, 4
g = torch.Generator().manual_seed(17)
old = torch.randn(A, T, N, D, generator=g) * 0.1
new = torch.randn(A, T, N, D, generator=g) * 0.1
order = torch.randperm(A, generator=g)
seen = []
def train_agent(k, factor):
    seen.append((k, factor.clone()))
    return old[k], new[k]
f = happo_factor_chain(order, lambda k: k %% world, train_agent, torch.ones(T, N, 1))
ref = torch.ones(T, N, 1)
want = {}
for k in [int(x) for x in order]:                                   # the reference's single-process loop
    want[k] = ref.clone()
    ref = ref * torch.exp((new[k] - old[k]).sum(-1, keepdim=True))
assert torch.allclose(f, ref, rtol=1e-6, atol=0)
assert [k for k, _ in seen] == [int(x) for x in order if int(x) %% world == rank]
for k, fk in seen:
    assert torch.allclose(fk, want[k], rtol=1e-6, atol=0), k       # every owner saw the factor of the agents updated before its own
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_statistics_and_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = str(29500 + (os.getpid() % 2000))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()


def test_kernel_resources_of_the_built_library():
    """Static guard on the built HIP library (no GPU needed): the gfx950 code object is cut out of libmms.so's offload bundle and its
    kernel metadata read with llvm-readelf.  The TenAnt step layouts must stay at <= 168 VGPRs (three waves per SIMD: the whole
    4096-env grid resident in one round; and the 512-thread swarm layout loses 30 % above it, profiles/r02_swarm_occupancy.txt)
    without scratch (a single spilled dword costs 0.8 MB of HBM writes per launch), and no policy kernel may spill."""
    import re
    import struct
    import subprocess
    from massive_marl_benchmark_amd import _lib
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not found")
    data = open(_lib.LIB_PATH, "rb").read()
    import tempfile
    kernels, pos = {}, 0
    while True:                                                     # one offload bundle per object file of the library
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if i < 0:
            break
        pos = i + 24
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off, code = i + 32, None
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple:
                code = data[i + o:i + o + s]
        if not code:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(code)
            f.flush()
            notes = subprocess.check_output([readelf, "--notes", f.name]).decode()
        for block in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(_Z\S+)", block).group(1)
            kernels[name] = (int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1)), int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)))
    assert kernels, "no gfx950 code object in libmms.so"
    step = {k: v for k, v in kernels.items() if "ant_step_kernelILi0E" in k}
    assert len(step) >= 8, sorted(kernels)
    for k, (vgpr, scratch) in step.items():
        # (the instantiation with the policy head in its prologue, <..., false, true> -- a translation unit of its own, step_head_kernels.hip,
        #  so that it cannot move the others' allocation -- keeps up to THREE address dwords across the physics loop in scratch: each stored
        #  once and reloaded once or twice per lane, against the 13-us launch it replaces; the others: none)
        head = k.endswith("Lb0ELb1EEEvNS_8StepArgsE")
        assert vgpr <= 168 and scratch <= (16 if head else 0), (k, vgpr, scratch)
    for k, (vgpr, scratch) in kernels.items():
        if "linear_act" in k or "linear_split" in k or "split16_planes" in k or "split_planes" in k or "marl_heads" in k or "layernorm_rows" in k or "ppo_head_act" in k:
            assert scratch == 0, (k, vgpr, scratch)
        if "linear_split" in k:                                     # 512-thread blocks, two waves per SIMD: 256 registers is all there is
            assert vgpr <= 256, (k, vgpr)
    assert sum(1 for k in kernels if "linear_split16_kernel" in k) >= 12 and sum(1 for k in kernels if "linear_split_kernel" in k) >= 8
