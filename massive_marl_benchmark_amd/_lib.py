"""ctypes loader of the engine libraries (csrc/ -> lib/libmms.so, lib/libmms_cpu.so).

`lib()` is the HIP build.  There is no CPU fallback: if it is missing or no HIP device is usable the product path raises.
`lib_cpu()` is the CPU build of the same C ABI (csrc/cpu/: the kernels' lane math compiled for the host) -- the reference's
`--sim_device cpu` pipeline (base_task.py:27-32).  It is reached only by asking for it: `device_type="cpu"` in a task
constructor, `Engine(device="cpu")`, storage / modules created on torch's "cpu" device.  Nothing selects it automatically."""
import ctypes
import os

from .model import MmsConfig, MmsTensor

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMS_LIB", os.path.join(_HERE, "lib", "libmms.so"))   # MMS_LIB: A/B builds of the same library
LIB_CPU_PATH = os.path.join(_HERE, "lib", "libmms_cpu.so")
_lib = None
_lib_cpu = None

# every symbol include/mms.h declares (tests check that the library exports all of them)
SYMBOLS = ["mms_create", "mms_destroy", "mms_get_tensor", "mms_step", "mms_post_step", "mms_reset_all", "mms_set_state",
           "mms_bind_obs_out", "mms_bind_obs_planes16", "mms_bind_actions", "mms_bind_policy_head", "mms_set_dr", "mms_set_obs_outputs", "mms_bind_rollout_out", "mms_ppo_act", "mms_ppo_heads_act", "mms_linear2_act", "mms_linear_group_act", "mms_split_planes", "mms_split_planes_group", "mms_linear_group_act_split", "mms_split_planes16_group", "mms_weight_planes16_group", "mms_chain_refresh16", "mms_fold_planes16_group", "mms_fold_scales16_group", "mms_linear_group_act_split16", "mms_row_stats_chan_group", "mms_marl_heads_finish", "mms_row_stats_group", "mms_row_moments_group", "mms_layernorm_group", "mms_marl_heads_act", "mms_marl_views", "mms_gae_ppo", "mms_adv_normalize", "mms_gae_ppo_normalized", "mms_layer_clock_probe", "mms_gae_marl", "mms_gae_marl_agents",
           "mms_last_error", "mms_abi_version"]


class MmsError(RuntimeError):
    pass


def _bind(path):
    L = ctypes.CDLL(path)
    vp, ci, c64, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    L.mms_create.argtypes = [ctypes.POINTER(MmsConfig), ctypes.POINTER(vp)]
    L.mms_destroy.argtypes = [vp]
    L.mms_get_tensor.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(MmsTensor)]
    L.mms_step.argtypes = [vp, vp]
    L.mms_post_step.argtypes = [vp, vp]
    L.mms_reset_all.argtypes = [vp, vp]
    L.mms_set_state.argtypes = [vp, ctypes.c_char_p, vp, ci, ctypes.POINTER(c64), c64, vp]
    L.mms_bind_obs_out.argtypes = [vp, vp]
    L.mms_bind_actions.argtypes = [vp, vp]
    L.mms_bind_policy_head.argtypes = [vp, vp]
    L.mms_bind_obs_planes16.argtypes = [vp, vp, cf]
    L.mms_set_obs_outputs.argtypes = [vp, ctypes.c_int32, ctypes.c_int32]
    L.mms_set_dr.argtypes = [vp, ctypes.c_int32]
    L.mms_bind_rollout_out.argtypes = [vp, vp, vp]
    L.mms_ppo_act.argtypes = [ci, vp, vp, vp, ctypes.c_uint64, vp, c64, ctypes.c_int32, vp, vp, vp, vp, vp, vp, c64, ctypes.c_int32, vp]
    L.mms_ppo_heads_act.argtypes = [ci, vp, vp, vp, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_int32, vp, ctypes.c_uint64, vp, c64, ctypes.c_int32, vp, vp, vp, vp,
                                    vp, vp, c64, ctypes.c_int32, vp]
    L.mms_linear2_act.argtypes = [ci, c64, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int32, vp]
    L.mms_linear_group_act.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_int32, vp, vp, vp, vp]
    L.mms_split_planes.argtypes = [ci, c64, ctypes.c_int32, ctypes.c_int32, vp, vp, vp]
    L.mms_split_planes_group.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, ctypes.c_int32, vp, vp, vp]
    L.mms_linear_group_act_split.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32,
                                             vp, vp, vp, vp, vp, vp, vp]
    L.mms_split_planes16_group.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, cf, vp]
    L.mms_weight_planes16_group.argtypes = [ci, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mms_fold_planes16_group.argtypes = [ci, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mms_fold_scales16_group.argtypes = [ci, ctypes.c_int32, vp, vp, c64, vp, vp, vp, vp]
    L.mms_chain_refresh16.argtypes = [ci, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, cf, c64, vp, vp, vp]
    L.mms_linear_group_act_split16.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32,
                                               vp, vp, vp, vp, vp, vp, vp]
    L.mms_row_stats_chan_group.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, vp, vp, cf, vp]
    L.mms_marl_heads_finish.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint64, c64, cf, vp]
    L.mms_row_moments_group.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, ctypes.c_int32, vp, vp, cf, vp]
    L.mms_row_stats_group.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, ctypes.c_int32, vp, vp, cf, vp]
    L.mms_layernorm_group.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, cf, vp]
    L.mms_marl_heads_act.argtypes = [ci, ctypes.c_int32, c64, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint64, c64, cf, vp]
    L.mms_marl_views.argtypes = [ci, vp, vp, c64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    L.mms_gae_ppo.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int32, c64, cf, cf, vp]
    L.mms_adv_normalize.argtypes = [ci, vp, vp, c64, vp]
    L.mms_gae_ppo_normalized.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, ctypes.c_int32, c64, cf, cf, vp]
    L.mms_layer_clock_probe.argtypes = [ci, vp, ctypes.c_int32]
    L.mms_gae_marl.argtypes = [ci, vp, vp, vp, vp, ctypes.c_int32, c64, cf, cf, ctypes.c_int32, vp, vp, vp]
    L.mms_gae_marl_agents.argtypes = [ci, vp, vp, vp, vp, ctypes.c_int32, c64, ctypes.c_int32, cf, cf, ctypes.c_int32, vp, vp, vp]
    L.mms_last_error.argtypes = [vp]
    L.mms_last_error.restype = ctypes.c_char_p
    L.mms_abi_version.restype = ci
    return L


def lib():
    """The HIP build (lib/libmms.so)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmsError("HIP engine library not built: %s is missing (run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C massive_marl_benchmark_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    _lib = _bind(LIB_PATH)
    return _lib


def lib_cpu():
    """The CPU build of the same ABI (lib/libmms_cpu.so): explicit opt-in only, see the module docstring."""
    global _lib_cpu
    if _lib_cpu is not None:
        return _lib_cpu
    if not os.path.exists(LIB_CPU_PATH):
        raise MmsError("CPU engine library not built: %s is missing (make -C massive_marl_benchmark_amd/csrc)" % LIB_CPU_PATH)
    _lib_cpu = _bind(LIB_CPU_PATH)
    return _lib_cpu


def for_device(device):
    """(library, device argument, stream argument) for work on torch device `device`: the HIP build with the device ordinal and the
    current stream for "cuda", the CPU build with -1 and no stream for "cpu" -- the caller chose the device, nothing falls back."""
    import torch
    dev = torch.device(device)
    if dev.type == "cpu":
        return lib_cpu(), -1, None
    if dev.type != "cuda":
        raise MmsError("unsupported device %r" % (device,))
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return lib(), idx, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def last_error(handle=None, L=None):
    msg = (L or lib()).mms_last_error(handle)
    return msg.decode() if msg else ""


def check(rc, handle=None, what="", L=None):
    if rc != 0:
        raise MmsError("%s failed: %s" % (what, last_error(handle, L)))
