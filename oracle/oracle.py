"""ctypes binding of the CPU oracle (oracle/mms_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes
import os
import subprocess

import numpy as np

from massive_marl_benchmark_amd.model import MmsConfig, MmsModel, make_config  # struct layout = include/mms.h

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmms_oracle.so")
_LIB64_PATH = os.path.join(_HERE, "_build", "libmms_oracle_f64.so")
_lib = None
_lib64 = None

F = ctypes.POINTER(ctypes.c_float)
I64 = ctypes.POINTER(ctypes.c_int64)
U8 = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    src = os.path.join(_HERE, "mms_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.mo_create.restype = ctypes.c_void_p
        _lib.mo_create.argtypes = [ctypes.POINTER(MmsConfig)]
        _lib.mo_destroy.argtypes = [ctypes.c_void_p]
        _lib.mo_tensor.restype = ctypes.c_void_p
        _lib.mo_tensor.argtypes = [ctypes.c_void_p, ctypes.c_char_p, I64]
        _lib.mo_step.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _lib.mo_dims.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
        _lib.mo_rand_uniform.restype = ctypes.c_float
        _lib.mo_rand_uniform.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
        cf, ci, c64 = ctypes.c_float, ctypes.c_int, ctypes.c_int64
        _lib.mo_rand_normal.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
        _lib.mo_rand_normal.restype = ctypes.c_float
        _lib.mo_ppo_act.argtypes = [c64, ci, F, F, ctypes.c_uint64, I64, c64, ci, F, F, F]
        _lib.mo_ppo_log_prob.argtypes = [c64, ci, F, F, F, ci, F, F]
        _lib.mo_helpers_batch.argtypes = [c64] + [F] * 13
        _lib.mo_tenant_obs_batch.argtypes = [c64, F, F, F, F, F, cf, F, F]
        _lib.mo_tenant_goals_batch.argtypes = [c64, F, F, F, F, F, F]
        _lib.mo_tenant_reward_batch.argtypes = [c64, F, I64, I64, F, F, F, F, F, F, F, I64]
        _lib.mo_circle_reward_batch.argtypes = [c64, F, I64, I64, F, F, F, F, I64, F]
        _lib.mo_oneant_obs_batch.argtypes = [c64] + [F] * 12
        _lib.mo_oneant_reward_batch.argtypes = [c64, F, I64, I64, F, F, F, F, F, F, F, F, I64]
        _lib.mo_ingenuity_thrust_batch.argtypes = [c64, F, cf, F]
        _lib.mo_ingenuity_reward_batch.argtypes = [c64, F, I64, F, I64]
        _lib.mo_marl_views.argtypes = [c64, ci, ci, ci, cf, F, F]
        _lib.mo_gae_ppo.argtypes = [ci, c64, F, U8, F, F, cf, cf, F, F, ci]
        _lib.mo_gae_marl.argtypes = [ci, c64, F, F, F, cf, cf, ci, cf, cf, F]
        _lib.mo_ant_substep.argtypes = [ctypes.POINTER(MmsModel), cf, F, F, F, F, F, F]
        _lib.mo_ant_substep_dr.argtypes = [ctypes.POINTER(MmsModel), cf, F, F, F, F, F, F, F]
        _lib.mo_set_dr.argtypes = [ctypes.c_void_p, ci]
        _lib.mo_box_substep.argtypes = [ctypes.POINTER(MmsModel), cf, F, F]
        _lib.mo_heli_substep.argtypes = [ctypes.POINTER(MmsModel), cf, F, F]
        _lib.mo_ant_momentum.argtypes = [ctypes.POINTER(MmsModel), F, F, F]
        assert _lib.mo_sizeof_config() == ctypes.sizeof(MmsConfig), "mms_config layout mismatch"
    return _lib


def lib64():
    """The physics alone compiled in double (oracle/Makefile, -DMO_F64): the yardstick of the parity gates."""
    global _lib64
    if _lib64 is None:
        build()
        if not os.path.exists(_LIB64_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib64 = ctypes.CDLL(_LIB64_PATH)
        D = ctypes.POINTER(ctypes.c_double)
        _lib64.mo_physics_f64.argtypes = [ctypes.POINTER(MmsConfig), ctypes.c_int64, F, F, F, F, F, I64, D, D, D]
        assert _lib64.mo_sizeof_config() == ctypes.sizeof(MmsConfig), "mms_config layout mismatch"
    return _lib64


def physics_f64(config, actions, root_states, dof_state, reset, foot_sensors=None, dr_params=None):
    """One control step of physics in double from fp32 inputs: (root [rows,13], dof [rows,2], sensors or None) as float64.
    Envs with reset != 0 pass through unchanged (mo_step skips their physics too)."""
    n = int(config.num_envs)
    act, root, dof = f32(actions), f32(root_states), f32(dof_state)
    rs = i64(reset)
    ro, do = np.zeros(root.shape, np.float64), np.zeros(dof.shape, np.float64)
    D = ctypes.POINTER(ctypes.c_double)
    sens = f32(foot_sensors) if foot_sensors is not None else None
    so = np.zeros(sens.shape, np.float64) if sens is not None else None
    dr = f32(dr_params) if dr_params is not None else None
    lib64().mo_physics_f64(ctypes.byref(config), n, fp(act), fp(root), fp(dof), fp(sens) if sens is not None else None,
                           fp(dr) if dr is not None else None, ip(rs), ro.ctypes.data_as(D), do.ctypes.data_as(D),
                           so.ctypes.data_as(D) if so is not None else None)
    return ro, do, so


def fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(F)


def ip(a):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(I64)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


_I64_NAMES = ("reset", "progress", "reset_count")


class OracleEngine:
    """Same buffers and step protocol as the product engine, on the CPU."""

    def __init__(self, task, cfg=None, **kw):
        self._init_from_config(task, make_config(task, cfg, **kw))

    def _init_from_config(self, task, config):
        self.task = task
        self.config = config
        self._h = lib().mo_create(ctypes.byref(self.config))
        if not self._h:
            raise RuntimeError("mo_create failed")
        d = (ctypes.c_int32 * 6)()
        lib().mo_dims(self._h, d)
        self.actors, self.dofs, self.num_actions, self.obs_dim, self.prev_dim, self.num_agents = list(d)
        self.num_envs = self.config.num_envs

    def tensor(self, name):
        n = ctypes.c_int64()
        p = lib().mo_tensor(self._h, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        ct = ctypes.c_int64 if name in _I64_NAMES else ctypes.c_float
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ct)), shape=(n.value,))
        N = self.num_envs
        shapes = {"actions": (N, self.num_actions), "obs": (N, self.obs_dim), "obs_clipped": (N, self.obs_dim),
                  "root_states": (N * self.actors, 13), "initial_root_states": (N * self.actors, 13),
                  "dof_state": (N * self.dofs, 2), "env_origin": (N, 3), "prev": (N, self.prev_dim),
                  "reset_noise": (N, 16), "foot_sensors": (N * self.num_agents, 24), "dr_params": (N * self.num_agents, 33)}
        return arr.reshape(shapes.get(name, (n.value,)))

    def set_dr(self, enable=True):
        lib().mo_set_dr(self._h, 1 if enable else 0)

    def step(self, actions=None, physics=True):
        if actions is not None:
            self.tensor("actions")[...] = actions
        lib().mo_step(self._h, 1 if physics else 0)

    def close(self):
        if self._h:
            lib().mo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
