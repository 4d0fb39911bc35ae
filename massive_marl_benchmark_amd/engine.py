"""Engine: owns one `mms_handle` and exposes its buffers as NON-owning torch tensors (zero copy).

This is what replaces `gymapi.acquire_gym() ... prepare_sim()` plus the `acquire_*_tensor` /
`gymtorch.wrap_tensor` calls of the reference tasks (agents/tasks/ten_ant.py:84-104)."""
import ctypes

import torch

from . import _lib
from .model import MmsPolicyHead, MmsTensor, make_config, task_dims

_TORCH_DTYPES = {0: (torch.float32, "<f4"), 1: (torch.int64, "<i8"), 2: (torch.int32, "<i4"), 3: (torch.uint8, "|u1")}


class _DevicePtr:
    """Minimal __cuda_array_interface__ carrier: lets torch.as_tensor view engine memory without copying."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        self._owner = owner


def current_stream_ptr(device):
    device = torch.device(device)
    if device.type == "cpu":
        return None                                   # the CPU build ignores streams: every call completes before it returns
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_NP_DTYPES = {0: "float32", 1: "int64", 2: "int32", 3: "uint8"}


class Engine:
    def __init__(self, task, cfg=None, num_envs=None, num_agents=None, device=0, seed=0, env_offset=0, total_envs=None,
                 clip_obs=5.0, clip_actions=1.0, external_noise=False):
        """device: a HIP ordinal (the HIP build, lib/libmms.so) -- or "cpu" / -1: the CPU build of the same ABI
        (lib/libmms_cpu.so, the reference's `--sim_device cpu` pipeline), an explicit choice, never a fallback."""
        self.task = task
        self.is_cpu = device in ("cpu", -1)
        if self.is_cpu:
            self.device_index, self.device, self._L = -1, torch.device("cpu"), _lib.lib_cpu()
        else:
            if not torch.cuda.is_available():
                raise _lib.MmsError("no HIP device visible to torch: the HIP engine has no CPU fallback (the CPU build is an explicit "
                                    "choice: device='cpu')")
            self.device_index = int(device)
            self.device = torch.device("cuda", self.device_index)
            self._L = _lib.lib()
        self.config = make_config(task, cfg, num_envs=num_envs, num_agents=num_agents, device=self.device_index, seed=seed,
                                  env_offset=env_offset, total_envs=total_envs, clip_obs=clip_obs, clip_actions=clip_actions,
                                  external_noise=external_noise)
        self.num_envs = self.config.num_envs
        self.num_agents = self.config.num_agents
        self.actors, self.dofs, self.num_actions, self.obs_dim, self.prev_dim = task_dims(task, self.num_agents)
        self._h = ctypes.c_void_p()
        _lib.check(self._L.mms_create(ctypes.byref(self.config), ctypes.byref(self._h)), None, "mms_create", self._L)
        self._tensors = {}
        self._bound = None

    # -- buffers -----------------------------------------------------------------------------
    def tensor(self, name):
        t = self._tensors.get(name)
        if t is None:
            mt = MmsTensor()
            self._check(self._L.mms_get_tensor(self._h, name.encode(), ctypes.byref(mt)), "mms_get_tensor")
            dtype, typestr = _TORCH_DTYPES[mt.dtype]
            shape = [mt.shape[i] for i in range(mt.ndim)]
            if self.is_cpu:
                import numpy as np
                n = int(np.prod(shape)) if shape else 1
                raw = (ctypes.c_uint8 * (n * np.dtype(_NP_DTYPES[mt.dtype]).itemsize)).from_address(mt.ptr)
                t = torch.from_numpy(np.frombuffer(raw, dtype=_NP_DTYPES[mt.dtype]).reshape(shape))      # engine memory, not a copy
            else:
                t = torch.as_tensor(_DevicePtr(mt.ptr, shape, typestr, self), device=self.device)
            assert t.data_ptr() == mt.ptr and t.dtype == dtype
            self._tensors[name] = t
        return t

    # -- stepping ----------------------------------------------------------------------------
    def _check(self, rc, what):
        _lib.check(rc, self._h, what, self._L)

    def step(self):
        self._check(self._L.mms_step(self._h, current_stream_ptr(self.device)), "mms_step")

    def post_step(self):
        self._check(self._L.mms_post_step(self._h, current_stream_ptr(self.device)), "mms_post_step")

    def reset_all(self):
        self._check(self._L.mms_reset_all(self._h, current_stream_ptr(self.device)), "mms_reset_all")

    def bind_obs_out(self, tensor_or_none):
        """Extra destination for the clamped observation row (e.g. a rollout-buffer slot [N, obs_dim])."""
        if tensor_or_none is None:
            ptr = None
        else:
            t = tensor_or_none
            assert t.device.type == self.device.type and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.num_envs * self.obs_dim
            ptr = ctypes.c_void_p(t.data_ptr())
        self._bound = tensor_or_none          # keep it alive
        self._check(self._L.mms_bind_obs_out(self._h, ptr), "mms_bind_obs_out")

    def bind_obs_planes(self, planes_or_none, scale=2048.0):
        """Extra destination for the clamped observation row as the policy layers' operand planes (H32: uint8 tensor of
        num_envs * ceil(obs_dim / 32) * 128 bytes, two fp16 planes of row * scale; mms_bind_obs_planes16): ActorCritic.act(obs, states,
        obs_planes=(planes, scale)) then skips its own split of the observation.  `scale`: a power of two with clip_obs * scale <= 2^14."""
        if planes_or_none is None:
            ptr = None
        else:
            t = planes_or_none
            assert t.device.type == self.device.type and t.dtype == torch.uint8 and t.is_contiguous() and \
                t.numel() == self.num_envs * ((self.obs_dim + 31) // 32) * 128
            ptr = ctypes.c_void_p(t.data_ptr())
        self._bound_planes = planes_or_none   # keep it alive
        self._check(self._L.mms_bind_obs_planes16(self._h, ptr, float(scale)), "mms_bind_obs_planes16")

    def bind_actions(self, tensor_or_none):
        """The step reads `tensor` ([N, num_actions] f32 on the engine's device, contiguous) in place instead of the engine's own
        "actions" buffer (mms_bind_actions); None returns to that buffer.  The clamp to +-clip_actions is in the kernel."""
        if tensor_or_none is None:
            ptr = None
        else:
            t = tensor_or_none
            assert t.device.type == self.device.type and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.num_envs * self.num_actions
            ptr = ctypes.c_void_p(t.data_ptr())
        self._bound_actions = tensor_or_none  # keep it alive
        self._check(self._L.mms_bind_actions(self._h, ptr), "mms_bind_actions")

    def takes_policy_head(self):
        """Does the next step accept a bound policy head (mms_bind_policy_head: the 16-envs-per-workgroup TenAnt layout)?  Asked once by the
        policy module; the answer of a probe binding with placeholder-free validation is what the library itself says."""
        if getattr(self, "_takes_head", None) is None:
            probe = MmsPolicyHead()                              # all pointers NULL: passes the layout check, fails the pointer check
            rc = self._L.mms_bind_policy_head(self._h, ctypes.byref(probe))
            self._takes_head = rc != 0 and "null pointer" in _lib.last_error(self._h, self._L)
        return self._takes_head

    def bind_policy_head(self, head):
        """`head`: an MmsPolicyHead (prebuilt by the policy module, one per rollout slot) or None.  The binding is consumed by the next step()."""
        self._check(self._L.mms_bind_policy_head(self._h, None if head is None else ctypes.byref(head)), "mms_bind_policy_head")

    def set_obs_outputs(self, raw=True, clipped=True):
        """Which engine-owned observation rows the step writes ("obs", "obs_clipped"); a rollout that binds a slot with
        bind_obs_out needs neither while the slot is bound."""
        self._check(self._L.mms_set_obs_outputs(self._h, int(bool(raw)), int(bool(clipped))), "mms_set_obs_outputs")

    def set_dr(self, enable=True):
        """Use the per-ant physical parameters in tensor("dr_params") (mass / damping scales, joint-limit offsets)."""
        self._check(self._L.mms_set_dr(self._h, int(bool(enable))), "mms_set_dr")

    def bind_rollout_out(self, rewards=None, dones=None):
        """Extra destinations for the step's reward (f32 [N]) and done flag (u8 [N]), e.g. RolloutStorage.rewards[t] /
        dones[t]; None disables either."""
        def ptr(t, dtype):
            if t is None:
                return None
            assert t.device.type == self.device.type and t.dtype == dtype and t.is_contiguous() and t.numel() == self.num_envs
            return ctypes.c_void_p(t.data_ptr())
        self._bound_rollout = (rewards, dones)
        self._check(self._L.mms_bind_rollout_out(self._h, ptr(rewards, torch.float32), ptr(dones, torch.uint8)), "mms_bind_rollout_out")

    def set_state(self, name, src, env_ids=None):
        """Tests / fixtures: copy `src` (torch tensor on this device, or a numpy array) into a named buffer."""
        L = self._L
        if torch.is_tensor(src):
            src = src.contiguous()
            ptr, is_host = ctypes.c_void_p(src.data_ptr()), 0 if src.is_cuda else 1
        else:
            import numpy as np
            src = np.ascontiguousarray(src)
            ptr, is_host = ctypes.c_void_p(src.ctypes.data), 1
        if env_ids is None:
            ids, n = None, 0
        else:
            arr = (ctypes.c_int64 * len(env_ids))(*[int(i) for i in env_ids])
            ids, n = arr, len(env_ids)
        self._check(L.mms_set_state(self._h, name.encode(), ptr, is_host, ids, n, current_stream_ptr(self.device)), "mms_set_state")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._tensors.clear()
            self._L.mms_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
