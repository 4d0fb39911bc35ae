"""CPU check of the kernels' lane decomposition: the per-lane functions the HIP kernels are built from
(massive_marl_benchmark_amd/csrc/mms_lane.h) run on the host over the lanes of each env
(tests/emu/emu_step.cpp) and are compared with the oracle, step for step on identical state
(teacher forced) and free running.  Tolerance 1e-4 abs fp32."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import parity
from conftest import random_dr_params, shove_ants_into_box
from massive_marl_benchmark_amd.model import MmsConfig
from oracle.oracle import F, I64, OracleEngine, f32, fp, ip

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_SRC = os.path.join(HERE, "emu", "emu_step.cpp")
EMU_LIB = os.path.join(HERE, "emu", "_libemu.so")
NAMES = ["actions", "obs", "obs_clipped", "rew", "reset", "progress", "root_states", "initial_root_states", "dof_state",
         "env_origin", "prev", "reset_noise", "foot_sensors", "reset_count"]


@pytest.fixture(scope="module")
def emu():
    hdr = os.path.join(HERE, "..", "massive_marl_benchmark_amd", "csrc", "cpu", "lane_step.h")
    if not os.path.exists(EMU_LIB) or os.path.getmtime(EMU_LIB) < max(os.path.getmtime(EMU_SRC), os.path.getmtime(hdr), os.path.getmtime(os.path.join(os.path.dirname(hdr), "..", "mms_lane.h"))):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                               "-o", EMU_LIB, EMU_SRC])
    lib = ctypes.CDLL(EMU_LIB)
    lib.emu_step.argtypes = [ctypes.POINTER(MmsConfig), F, F, F, F, I64, I64, F, F, F, F, F, F, F, I64, ctypes.c_int,
                             ctypes.c_int, ctypes.c_int, F]
    return lib


class EmuEngine:
    def __init__(self, lib, task, **kw):
        self.lib = lib
        self.ref = OracleEngine(task, **kw)          # only used to get identically initialised buffers
        self.config = self.ref.config
        self.buf = {n: self.ref.tensor(n).copy() for n in NAMES}
        self.obs_dim, self.prev_dim = self.ref.obs_dim, self.ref.prev_dim
        self.dr = None                               # [N * A, 33] physical domain randomisation, or None (nominal)

    def step(self, actions, physics=True):
        b = self.buf
        b["actions"][...] = actions
        self.lib.emu_step(ctypes.byref(self.config), fp(b["actions"]), fp(b["obs"]), fp(b["obs_clipped"]), fp(b["rew"]),
                          ip(b["reset"]), ip(b["progress"]), fp(b["root_states"]), fp(b["initial_root_states"]), fp(b["dof_state"]),
                          fp(b["env_origin"]), fp(b["prev"]), fp(b["reset_noise"]), fp(b["foot_sensors"]), ip(b["reset_count"]),
                          1 if physics else 0, self.obs_dim, self.prev_dim, None if self.dr is None else fp(self.dr))


STATE = parity.STATE


def task_kw(task, **kw):
    """MultiIngenuity envs away from the global origin die on every step (the reward measures distances in the GLOBAL frame,
    multi_ingenuity.py:381-453: SURVEY section 0 fact 6): the physics tests keep its envs at the origin."""
    if task in ("MultiIngenuity", "MultiAntCircle"):     # (MultiAntCircle's ring is drawn round the GLOBAL origin too: multi_ant_circle.py:424)
        from massive_marl_benchmark_amd.model import default_cfg
        cfg = default_cfg(task)
        cfg["env"]["envSpacing"] = 0.0
        kw["cfg"] = cfg
    return kw


def drive(o, e, tf, act, what):
    for name in STATE:                                    # identical state in
        e.buf[name][...] = o.tensor(name)
    tf.before(act)
    o.step(act)
    e.step(act)
    tf.after(what)


@pytest.mark.parametrize("task,n,steps", [("TenAnt", 6, 120), ("OneAnt", 8, 120), ("MultiIngenuity", 8, 120), ("MultiAntCircle", 8, 120)])
def test_teacher_forced_parity(emu, task, n, steps):
    kw = task_kw(task, num_envs=n, seed=5, total_envs=64, env_offset=3)
    o = OracleEngine(task, **kw)
    e = EmuEngine(emu, task, **kw)
    tf = parity.TeacherForced(o, lambda k: e.buf[k])
    rng = np.random.default_rng(1)
    for t in range(steps):
        act = f32(rng.uniform(-1.2, 1.2, (n, o.num_actions)))
        if task == "MultiIngenuity":
            act[:, 2::3] = np.abs(act[:, 2::3]) * 0.12      # near hover thrust so that episodes last
        drive(o, e, tf, act, "%s step %d" % (task, t))
    tf.finish("emu/teacher_forced/%s" % task, min_live_steps=steps // 2)
    assert tf.resets > n or task != "TenAnt"


@pytest.mark.parametrize("task,n,rule", [("TenAnt", 5, "average"), ("OneAnt", 6, "average"), ("TenAnt", 4, "min"), ("OneAnt", 5, "min")])
def test_ant_box_contact_parity(emu, task, n, rule):
    """Teacher-forced parity while ants are pressed against the box (with ant-box friction under `average`, frictionless under
    `min`); the box must feel them (its x velocity goes negative)."""
    from massive_marl_benchmark_amd.model import default_cfg
    cfg = default_cfg(task)
    cfg["env"]["frictionCombine"] = rule
    kw = dict(cfg=cfg, num_envs=n, seed=11, total_envs=64, env_offset=7)
    o = OracleEngine(task, **kw)
    e = EmuEngine(emu, task, **kw)
    tf = parity.TeacherForced(o, lambda k: e.buf[k])
    rng = np.random.default_rng(4)
    zero = f32(np.zeros((n, o.num_actions)))
    for _ in range(12):                                    # reset, then let the box settle on the ground
        o.step(zero)
    shove_ants_into_box(o, rng)
    pushed = 0.0
    A = o.num_agents
    for t in range(40):
        act = f32(rng.uniform(-1, 1, (n, o.num_actions)))
        drive(o, e, tf, act, "%s contact step %d" % (task, t))
        box_vx = o.tensor("root_states").reshape(n, A + 1, 13)[:, A, 7]
        pushed = min(pushed, float(box_vx.min()))
        if t == 20:
            shove_ants_into_box(o, rng)                    # again, from a different configuration
    tf.finish("emu/ant_box_contact/%s/%s" % (task, rule))
    assert pushed < -1e-3, pushed


@pytest.mark.parametrize("task,n", [("TenAnt", 5), ("OneAnt", 8)])
def test_domain_randomised_physics_parity(emu, task, n):
    """Per-ant mass / damping scales and joint-limit offsets (cfg/TenAnt.yaml:97-122 ranges): lane code == oracle."""
    kw = dict(num_envs=n, seed=13, total_envs=64, env_offset=2)
    o = OracleEngine(task, **kw)
    e = EmuEngine(emu, task, **kw)
    rng = np.random.default_rng(9)
    dr = random_dr_params(rng, n * o.num_agents)
    o.tensor("dr_params")[...] = dr
    o.set_dr(True)
    e.dr = dr.copy()
    tf = parity.TeacherForced(o, lambda k: e.buf[k], dr=dr)
    for t in range(80):
        act = f32(rng.uniform(-1.2, 1.2, (n, o.num_actions)))
        drive(o, e, tf, act, "%s DR step %d" % (task, t))
    tf.finish("emu/domain_randomised/%s" % task)
    # and the randomisation does something: the same run with the nominal model ends elsewhere
    nominal = OracleEngine(task, **kw)
    rng = np.random.default_rng(9)
    random_dr_params(rng, n * o.num_agents)
    for t in range(80):
        nominal.step(f32(rng.uniform(-1.2, 1.2, (n, o.num_actions))))
    assert np.max(np.abs(nominal.tensor("dof_state") - o.tensor("dof_state"))) > 1e-2


@pytest.mark.parametrize("task,n", [("TenAnt", 4), ("OneAnt", 6)])
def test_box_ground_friction_parity(emu, task, n):
    """cfg env.boxGroundFriction = 0.5: the general 6x6 box solve of the lane code against the oracle, ants pushing the box."""
    from massive_marl_benchmark_amd.model import default_cfg
    cfg = default_cfg(task)
    cfg["env"]["boxGroundFriction"] = 0.5
    kw = dict(cfg=cfg, num_envs=n, seed=17, total_envs=64, env_offset=5)
    o = OracleEngine(task, **kw)
    e = EmuEngine(emu, task, **kw)
    assert abs(o.config.model.boxgnd_mu - 0.5) < 1e-7
    tf = parity.TeacherForced(o, lambda k: e.buf[k])
    rng = np.random.default_rng(3)
    zero = f32(np.zeros((n, o.num_actions)))
    for _ in range(12):
        o.step(zero)
    shove_ants_into_box(o, rng)
    for t in range(50):
        act = f32(rng.uniform(-1, 1, (n, o.num_actions)))
        drive(o, e, tf, act, "%s box friction step %d" % (task, t))
    tf.finish("emu/box_ground_friction/%s" % task)


@pytest.mark.parametrize("task,n,steps", [("TenAnt", 4, 25), ("OneAnt", 4, 25), ("MultiIngenuity", 4, 60)])
def test_free_running_parity(emu, task, n, steps):
    kw = dict(num_envs=n, seed=9)
    o = OracleEngine(task, **kw)
    e = EmuEngine(emu, task, **kw)
    rng = np.random.default_rng(2)
    for t in range(steps):
        act = f32(rng.uniform(-1, 1, (n, o.num_actions)))
        if task == "MultiIngenuity":
            act[:, 2::3] = np.abs(act[:, 2::3]) * 0.12
        o.step(act)
        e.step(act)
        # contact dynamics amplify the per-step rounding differences exponentially: free running is only a
        # short-horizon sanity bound on the poses (the per-step gate is test_teacher_forced_parity)
        ro, re_ = o.tensor("root_states"), e.buf["root_states"]
        assert np.max(np.abs(ro[:, :7] - re_[:, :7])) < 2e-2, t
        np.testing.assert_array_equal(o.tensor("progress"), e.buf["progress"])


def test_glue_fixture_through_lanes(emu):
    """The reference step-glue fixture through the lane code path (physics off)."""
    from conftest import random_dr_params, shove_ants_into_box, angle_close, load_golden
    g = load_golden("tenant_step_glue")
    S, n = g["actions"].shape[0], g["actions"].shape[1]
    e = EmuEngine(emu, "TenAnt", num_envs=n, clip_obs=5.0, external_noise=True)
    for t in range(S):
        loc = g["sim_root"][t].reshape(n, 11, 13).copy()
        loc[:, :, 0:3] -= g["env_origin"][:, None, :]
        e.buf["root_states"][...] = loc.reshape(n * 11, 13)
        e.buf["dof_state"][...] = g["sim_dof"][t]
        e.buf["reset_noise"][:, :8] = g["noise_pos"][t]
        e.buf["reset_noise"][:, 8:] = g["noise_vel"][t]
        if t == 2:
            e.buf["progress"][7] = 998
        e.step(g["actions"][t], physics=False)
        np.testing.assert_array_equal(e.buf["reset"], g["reset"][t])
        np.testing.assert_array_equal(e.buf["progress"], g["progress"][t])
        ang = np.zeros(388, bool)
        for k in range(10):
            ang[38 * k + 9:38 * k + 12] = True
        assert np.max(np.abs(e.buf["obs"][:, ~ang] - g["obs"][t][:, ~ang])) < 2e-4
        assert angle_close(e.buf["obs"][:, ang], g["obs"][t][:, ang], 0) < 2e-4
        assert np.max(np.abs(e.buf["rew"] - g["rew"][t])) < 0.4


class EmuImpl:
    """parity.py adapter: the lane emulation behind the put / get / post_step / step interface of the fixture checks."""

    def __init__(self, lib, task, cfg=None, **kw):
        self.e = EmuEngine(lib, task, cfg=cfg, **kw)
        self.config = self.e.config

    def put(self, name, arr):
        self.e.buf[name][...] = np.asarray(arr).reshape(self.e.buf[name].shape)

    def get(self, name):
        return self.e.buf[name].copy()

    def post_step(self, actions):
        self.e.step(f32(actions), physics=False)

    def step(self, actions):
        self.e.step(f32(actions))

    def close(self):
        self.e.ref.close()


@pytest.mark.parametrize("check", [parity.fixture_tenant_obs, parity.fixture_tenant_goals, parity.fixture_oneant, parity.fixture_ingenuity, parity.fixture_circle])
def test_reference_fixtures_through_lanes(emu, check):
    """Every reference fixture of the task functions through the lane code's step path (the CPU rehearsal of the GPU test)."""
    from conftest import load_golden
    check(lambda task, cfg=None, **kw: EmuImpl(emu, task, cfg=cfg, **kw), load_golden, "emu/")
