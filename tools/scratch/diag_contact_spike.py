"""Diagnostic: the ant-box contact parity case on the GPU with per-step error dump and the inputs of the worst step saved."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity
from conftest import shove_ants_into_box
from oracle.oracle import OracleEngine, physics_f64
from massive_marl_benchmark_amd.engine import Engine
from massive_marl_benchmark_amd.model import default_cfg

task, n, rule = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cfg = default_cfg(task); cfg["env"]["frictionCombine"] = rule
kw = dict(cfg=cfg, num_envs=n, seed=11, total_envs=64, env_offset=7)
eng, ora = Engine(task, device=0, **kw), OracleEngine(task, **kw)
rng = np.random.default_rng(4)
zero = np.zeros((n, ora.num_actions), np.float32)
for _ in range(12):
    ora.step(zero)
shove_ants_into_box(ora, rng)
worst = (0, None)
for t in range(40):
    act = rng.uniform(-1, 1, (n, ora.num_actions)).astype(np.float32)
    inp = {k: ora.tensor(k).copy() for k in parity.STATE}
    for k in parity.STATE:
        eng.tensor(k).copy_(torch.from_numpy(inp[k]).cuda())
    eng.tensor("actions").copy_(torch.from_numpy(act).cuda())
    eng.step(); ora.step(act); torch.cuda.synchronize()
    r64, d64, _ = physics_f64(ora.config, act, inp["root_states"], inp["dof_state"], inp["reset"], inp["foot_sensors"], None)
    g_r, g_d = eng.tensor("root_states").cpu().numpy(), eng.tensor("dof_state").cpu().numpy()
    o_r, o_d = ora.tensor("root_states"), ora.tensor("dof_state")
    ev_g = max(np.abs(g_r[:, 7:] - r64[:, 7:]).max(), np.abs(g_d[:, 1] - d64[:, 1]).max())
    ev_o = max(np.abs(o_r[:, 7:] - r64[:, 7:]).max(), np.abs(o_d[:, 1] - d64[:, 1]).max())
    print(t, "gpu %.3e oracle32 %.3e" % (ev_g, ev_o))
    if ev_g > worst[0]:
        worst = (ev_g, dict(t=t, act=act, g_r=g_r, g_d=g_d, o_r=o_r.copy(), o_d=o_d.copy(), r64=r64, d64=d64, **{"in_" + k: v for k, v in inp.items()}))
    if t == 20:
        shove_ants_into_box(ora, rng)
np.savez(os.path.join(ROOT, "gpurun_out", "diag_spike_%s_%d_%s.npz" % (task, n, rule)), **worst[1])
w = worst[1]
dr = np.abs(w["g_r"][:, 7:] - w["r64"][:, 7:]); dd = np.abs(w["g_d"][:, 1] - w["d64"][:, 1])
print("worst step", w["t"], "root idx", np.unravel_index(dr.argmax(), dr.shape), dr.max(), "dof idx", dd.argmax(), dd.max())
