#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz from the REFERENCE's own functions.

Runs only where /root/reference exists (this build container).  Nothing from the reference is
copied into the repo: the reference modules are imported in place (SURVEY.md section 8c recipe)

  * `agents` is registered as an empty namespace package pointing at /root/reference/agents so its
    `__init__` (which needs gym / tensorboard / isaacgym) never runs;
  * `isaacgym` and `gym` resolve to the name-only stand-ins in tests/golden/_isaacgym_stub
    (our code; `isaacgym.torch_utils` restates the helper semantics of SURVEY.md appendix A.4);
  * the two reward functions that cannot run on CPU tensors (`abs(bool - 1)`, ten_ant.py:1074...,
    one_ant.py:505) are loaded from an in-memory patched copy written to a temp dir OUTSIDE the
    repo, with exactly one mechanical substitution `abs(X - 1)` -> `(~X).float()`; the count of
    substitutions is stored in each fixture's `meta`.
  * numpy>=2 removed `np.Inf`, which multi_vec_task.py:43 uses: `np.Inf = np.inf` is set before import.

The fixtures pin "reference task code o our helper restatement" (helpers have separate KATs).
What is saved: inputs and expected outputs only (arrays), never reference text.
"""
import contextlib
import importlib.util
import io
import os
import re
import sys
import tempfile
import types

import numpy as np
import torch

REF = os.environ.get("MMS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
N = 64
PI = float(np.pi)


# --------------------------------------------------------------------------------------
# import plumbing
# --------------------------------------------------------------------------------------
def _setup_imports():
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    sys.path.insert(0, os.path.join(HERE, "_isaacgym_stub"))
    for name, sub in (("agents", "agents"), ("agents.tasks", "agents/tasks"),
                      ("agents.tasks.agent_base", "agents/tasks/agent_base"),
                      ("agents.utils", "agents/utils"),
                      ("agents.algorithms", "agents/algorithms")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m
    # matplotlib / PIL are imported by the task files for unused names
    for name in ("matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image"):
        try:
            __import__(name)
        except Exception:  # pragma: no cover
            m = types.ModuleType(name)
            m.axis = None
            m.Image = None
            sys.modules[name] = m


def _load_by_path(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_patched(modname, relpath, tmpdir):
    """Load a reference task file with the bool-minus-int substitution (see module docstring)."""
    src = open(os.path.join(REF, relpath)).read()
    pat = re.compile(r"abs\((ant_push(?:_\d+)?) - 1\)")
    new, count = pat.subn(r"(~\1).float()", src)
    path = os.path.join(tmpdir, modname + ".py")
    with open(path, "w") as f:
        f.write(new)
    return _load_by_path(modname, path), count


@contextlib.contextmanager
def _quiet():
    """TorchScript print() inside ten_ant.compute_ant_observations writes to fd 1."""
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)


def _np(t):
    return t.detach().cpu().numpy()


ONLY = set(sys.argv[1:])        # fixture names given on the command line: write only those


def _save(name, meta, **arrays):
    if ONLY and name not in ONLY:
        return
    out = {k: (_np(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    out["meta"] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote %s.npz (%d arrays)" % (name, len(out)))


# --------------------------------------------------------------------------------------
# plausible random state
# --------------------------------------------------------------------------------------
def rand_quats(n, g):
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    k = n // 4
    # near identity block
    q[:k] = torch.tensor([0.0, 0.0, 0.0, 1.0]) + 0.05 * torch.randn(k, 4, generator=g)
    # yaw-only block
    yaw = (torch.rand(k, generator=g) * 2 - 1) * PI
    q[k:2 * k] = torch.stack([torch.zeros(k), torch.zeros(k), torch.sin(yaw / 2), torch.cos(yaw / 2)], -1)
    # near gimbal lock (pitch ~ +-90 deg)
    m = max(2, n // 16)
    s = 0.70710678
    q[2 * k:2 * k + m] = torch.tensor([0.0, s, 0.0, s]) + 1e-3 * torch.randn(m, 4, generator=g)
    q[2 * k + m:2 * k + 2 * m] = torch.tensor([0.0, -s, 0.0, s]) + 1e-3 * torch.randn(m, 4, generator=g)
    return q / q.norm(dim=-1, keepdim=True)


def rand_root(n, g, xy_scale=8.0, zlo=0.25, zhi=0.9):
    r = torch.zeros(n, 13)
    r[:, 0:2] = (torch.rand(n, 2, generator=g) * 2 - 1) * xy_scale
    r[:, 2] = zlo + (zhi - zlo) * torch.rand(n, generator=g)
    r[:, 3:7] = rand_quats(n, g)
    r[:, 7:10] = torch.randn(n, 3, generator=g)
    r[:, 10:13] = 2.0 * torch.randn(n, 3, generator=g)
    return r


ANT_LOWER = torch.tensor([-0.698132, 0.523599, -0.698132, -1.745329, -0.698132, -1.745329, -0.698132, 0.523599])
ANT_UPPER = torch.tensor([0.698132, 1.745329, 0.698132, -0.523599, 0.698132, -0.523599, 0.698132, 1.745329])


def rand_dofs(n, g):
    u = torch.rand(n, 8, generator=g)
    pos = (ANT_LOWER - 0.05) + u * (ANT_UPPER - ANT_LOWER + 0.1)
    # a block exactly at / beyond the upper limit so `> 0.99` fires
    pos[: n // 8] = ANT_UPPER + 0.01 * torch.rand(n // 8, 8, generator=g)
    vel = 3.0 * torch.randn(n, 8, generator=g)
    return pos, vel


def yaw_quat(yaw):
    z = torch.zeros_like(yaw)
    return torch.stack([z, z, torch.sin(yaw / 2), torch.cos(yaw / 2)], -1)


# --------------------------------------------------------------------------------------
def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not found at %s: fixtures can only be generated in the build container" % REF)
    _setup_imports()
    tmpdir = tempfile.mkdtemp(prefix="mms_fixture_")
    import isaacgym.torch_utils as tu
    tj = __import__("agents.utils.torch_jit_utils", fromlist=["x"])
    ten_ant = __import__("agents.tasks.ten_ant", fromlist=["x"])
    one_ant = __import__("agents.tasks.one_ant", fromlist=["x"])
    ingen = __import__("agents.tasks.multi_ingenuity", fromlist=["x"])
    ten_ant_p, n_sub_ten = _load_patched("ten_ant_patched", "agents/tasks/ten_ant.py", tmpdir)
    one_ant_p, n_sub_one = _load_patched("one_ant_patched", "agents/tasks/one_ant.py", tmpdir)
    assert n_sub_ten == 10 and n_sub_one == 1, (n_sub_ten, n_sub_one)
    meta_common = "torch %s; reference %s; substitutions ten_ant=%d one_ant=%d" % (
        torch.__version__, REF, n_sub_ten, n_sub_one)

    # ---------------- helpers_kat ----------------
    g = torch.Generator().manual_seed(1)
    q = rand_quats(N, g)
    q2 = rand_quats(N, g)
    v = torch.randn(N, 3, generator=g)
    roll, pitch, yaw = tu.get_euler_xyz(q)
    x = torch.randn(N, 8, generator=g)
    _save("helpers_kat", meta_common + "; fns: isaacgym stub torch_utils + reference torch_jit_utils.quat_axis",
          q=q, q2=q2, v=v, x=x,
          quat_mul=tu.quat_mul(q, q2), quat_conjugate=tu.quat_conjugate(q),
          quat_rotate=tu.quat_rotate(q, v), quat_rotate_inverse=tu.quat_rotate_inverse(q, v),
          roll=roll, pitch=pitch, yaw=yaw, normalize=tu.normalize(v),
          normalize_tiny=tu.normalize(v * 1e-12),
          unscale=tu.unscale(x, ANT_LOWER, ANT_UPPER),
          tensor_clamp=tu.tensor_clamp(x, ANT_LOWER, ANT_UPPER),
          quat_axis0=tj.quat_axis(q, 0), quat_axis2=tj.quat_axis(q, 2),
          lower=ANT_LOWER, upper=ANT_UPPER)

    # ---------------- tenant_obs ----------------
    g = torch.Generator().manual_seed(2)
    root = rand_root(N, g)
    root[: N // 8, 0] += 4000.0  # global-frame coordinates far from the origin
    dof_pos, dof_vel = rand_dofs(N, g)
    actions = torch.rand(N, 8, generator=g) * 2 - 1
    targets = torch.zeros(N, 3)
    inv_start_rot = tu.quat_conjugate(torch.tensor([0.0, 0.0, 0.0, 1.0])).repeat(N, 1)
    basis0 = torch.tensor([1.0, 0.0, 0.0]).repeat(N, 1)
    basis1 = torch.tensor([0.0, 0.0, 1.0]).repeat(N, 1)
    with _quiet():
        obs = ten_ant.compute_ant_observations(torch.zeros(N, 38), root.clone(), targets, inv_start_rot,
                                               dof_pos, dof_vel, ANT_LOWER, ANT_UPPER, 0.2, actions,
                                               0.0166, 0.1, basis0, basis1, 2)
    _save("tenant_obs", meta_common + "; fn: ten_ant.compute_ant_observations (ten_ant.py:1304-1350)",
          root=root, dof_pos=dof_pos, dof_vel=dof_vel, actions=actions, lower=ANT_LOWER, upper=ANT_UPPER,
          dof_vel_scale=0.2, obs=obs)

    # ---------------- tenant_goals ----------------
    g = torch.Generator().manual_seed(3)
    box = rand_root(N, g, xy_scale=6.0, zlo=0.45, zhi=0.6)
    yaw_only = (torch.rand(N // 2, generator=g) * 2 - 1) * 1.4
    box[: N // 2, 3:7] = yaw_quat(yaw_only)
    box_pos, box_quat, g1, g2, g3, g4 = ten_ant.compute_box_pos(box)
    g5, g6, g7, g8, g9, g10 = ten_ant.compute_other_goal(box)
    angle = ten_ant.compute_box_angle(box[:, 3:7])
    bx, by, bz = ten_ant.compute_box_quat(box[:, 3:7])
    qd = ten_ant.compute_box_quat_dist(0.0, 1.0, 0.0, bx, by, bz)
    goals = torch.stack([g1, g2, g3, g4, g5, g6, g7, g8, g9, g10], 1)
    a = torch.randn(N, 2, generator=g)
    b = torch.randn(N, 2, generator=g)
    _save("tenant_goals", meta_common + "; fns: compute_box_pos/compute_other_goal/compute_box_angle/"
          "compute_box_quat/compute_box_quat_dist/l2_dist (ten_ant.py:935-986,1353-1393)",
          box_root=box, box_pos=box_pos, box_quat=box_quat, goals=goals, angle=angle,
          quat_xyz=torch.stack([bx, by, bz], -1), quat_dist=qd, l2_a=a, l2_b=b, l2=ten_ant.l2_dist(a, b))

    # ---------------- tenant_reward ----------------
    def tenant_reward_case(seed, n):
        g = torch.Generator().manual_seed(seed)
        box_now = rand_root(n, g, xy_scale=3.0, zlo=0.45, zhi=0.55)
        box_now[:, 3:7] = yaw_quat((torch.rand(n, generator=g) * 2 - 1) * 0.6)
        # rows 0..5: box at the target with small yaw -> every goal arrives, quat_dist > 0.9 (success branch)
        box_now[:6, 0:2] = 0.05 * torch.randn(6, 2, generator=g)
        box_now[:6, 3:7] = yaw_quat(0.01 * torch.randn(6, generator=g))
        box_prev = box_now.clone()
        box_prev[:, 0:2] += 0.02 * torch.randn(n, 2, generator=g)
        box_prev[:, 3:7] = yaw_quat(torch.atan2(box_now[:, 5], box_now[:, 6]) * 2 + 0.01 * torch.randn(n, generator=g))
        bp, bq, c1, c2, c3, c4 = ten_ant.compute_box_pos(box_now)
        c5, c6, c7, c8, c9, c10 = ten_ant.compute_other_goal(box_now)
        goals_now = [c1, c2, c3, c4, c5, c6, c7, c8, c9, c10]
        bpp, _, d1, d2, d3, d4 = ten_ant.compute_box_pos(box_prev)
        d5, d6, d7, d8, d9, d10 = ten_ant.compute_other_goal(box_prev)
        goals_prev = [d1, d2, d3, d4, d5, d6, d7, d8, d9, d10]
        actions = torch.rand(n, 80, generator=g) * 2 - 1
        obs_list, pos_before = [], []
        for k in range(10):
            root = rand_root(n, g, xy_scale=2.0, zlo=0.33, zhi=0.9)
            # place the ant relative to its goal at distances straddling 1.5
            dist = 0.2 + 2.6 * torch.rand(n, generator=g)
            ang = torch.rand(n, generator=g) * 2 * PI
            root[:, 0] = goals_now[k][:, 0] + dist * torch.cos(ang)
            root[:, 1] = goals_now[k][:, 1] + dist * torch.sin(ang)
            # a few fallen ants (z < 0.31), only in some rows
            fallen = torch.rand(n, generator=g) < 0.03
            root[fallen, 2] = 0.26 + 0.04 * torch.rand(int(fallen.sum()), generator=g)
            # upright block so up_proj > 0.93 fires, plus random orientations
            dp, dv = rand_dofs(n, g)
            with _quiet():
                o = ten_ant.compute_ant_observations(torch.zeros(n, 38), root.clone(), torch.zeros(n, 3),
                                                     torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(n, 1), dp, dv,
                                                     ANT_LOWER, ANT_UPPER, 0.2, actions[:, 8 * k:8 * k + 8],
                                                     0.0166, 0.1, torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1),
                                                     torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1), 2)
            obs_list.append(o)
            pos_before.append(root[:, 0:2] + 0.03 * torch.randn(n, 2, generator=g))
        reset_in = (torch.rand(n, generator=g) < 0.1).long()
        progress = torch.randint(0, 1002, (n,), generator=g)
        progress[:8] = torch.tensor([0, 500, 997, 998, 999, 1000, 1001, 3])
        bt = [torch.tensor([0.0, s * (1.5 + 3.0 * j)]).repeat(n, 1) for j in range(5) for s in (-1.0, 1.0)]
        rew, reset = ten_ant_p.compute_ant_reward(
            *obs_list, reset_in, progress, actions,
            0.1, 0.5, 0.005, 0.05, 0.1, 0.31, -2.0, 1000,
            *pos_before, *goals_prev, bpp, bp, 0.0166, 1.0, bq, 0.0, 1.0, 0.0, 0.0, 500.0,
            torch.zeros(n, 2), *bt, 500.0, *goals_now)
        return dict(obs=torch.stack(obs_list, 1), reset_in=reset_in, progress=progress, actions=actions,
                    pos_before=torch.stack(pos_before, 1), goal_before=torch.stack(goals_prev, 1),
                    box_before=bpp, box_pos=bp, box_quat=bq, goals=torch.stack(goals_now, 1),
                    box_targets=torch.stack(bt, 1), rew=rew, reset=reset)

    _save("tenant_reward", meta_common + "; fn: patched ten_ant.compute_ant_reward (ten_ant.py:988-1301); "
          "scalars up_weight=0.1 heading_weight=0.5 actions_cost=0.005 energy_cost=0.05 joints_at_limit=0.1 "
          "termination_height=0.31 death_cost=-2 max_episode_length=1000 quat_reward_scale=0 "
          "ant_dist_reward_scale=500 goal_dist_reward_scale=500 goal=(0,1,0)",
          **tenant_reward_case(4, N))

    # ---------------- tenant_step_glue: post_physics_step x3 on synthetic state, incl. resets ----------------
    glue = tenant_glue(ten_ant_p, tu)
    _save("tenant_step_glue", meta_common + "; scripted TenAnt.pre_physics_step/post_physics_step "
          "(ten_ant.py:886-926, reset_idx :810-884) on a dummy object: state tensors supplied per step, "
          "no physics.  Model of the engine boundary: set_*_indexed write the simulator's internal state; "
          "the wrapped tensors change only at refresh_* (SURVEY.md A.3).", **glue)

    # ---------------- oneant_obs / oneant_reward ----------------
    g = torch.Generator().manual_seed(6)
    root = rand_root(N, g)
    boxr = rand_root(N, g, xy_scale=5.0, zlo=0.45, zhi=0.6)
    dof_pos, dof_vel = rand_dofs(N, g)
    actions = torch.rand(N, 8, generator=g) * 2 - 1
    sensors = 5.0 * torch.randn(N, 24, generator=g)
    potentials = -torch.rand(N, generator=g) * 300
    o, pot, prev_pot, up_vec, heading_vec, ant_pos = one_ant.compute_ant_observations(
        torch.zeros(N, 60), root.clone(), boxr.clone(), torch.zeros(N, 3), potentials,
        torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(N, 1), dof_pos, dof_vel, ANT_LOWER, ANT_UPPER, 0.2,
        sensors, actions, 0.0166, 0.1, torch.tensor([1.0, 0.0, 0.0]).repeat(N, 1),
        torch.tensor([0.0, 0.0, 1.0]).repeat(N, 1), 2)
    bpos, bquat = one_ant.compute_box_pos(boxr)
    _save("oneant_obs", meta_common + "; fns: one_ant.compute_ant_observations (one_ant.py:563-618), compute_box_pos (:621-627)",
          root=root, box_root=boxr, dof_pos=dof_pos, dof_vel=dof_vel, actions=actions, sensors=sensors,
          potentials_in=potentials, obs=o, potentials=pot, prev_potentials=prev_pot, up_vec=up_vec,
          heading_vec=heading_vec, ant_pos=ant_pos, box_pos=bpos, box_quat=bquat, lower=ANT_LOWER, upper=ANT_UPPER)

    g = torch.Generator().manual_seed(7)
    boxr = rand_root(N, g, xy_scale=2.0, zlo=0.45, zhi=0.55)
    boxr[:, 3:7] = yaw_quat((torch.rand(N, generator=g) * 2 - 1) * 0.8)
    boxr[:6, 0:2] = 0.1 * torch.randn(6, 2, generator=g)
    boxr[:6, 3:7] = yaw_quat(0.02 * torch.randn(6, generator=g))
    root = rand_root(N, g, xy_scale=2.0, zlo=0.27, zhi=0.9)
    dist = 0.2 + 2.6 * torch.rand(N, generator=g)
    ang = torch.rand(N, generator=g) * 2 * PI
    root[:, 0] = boxr[:, 0] + dist * torch.cos(ang)
    root[:, 1] = boxr[:, 1] + dist * torch.sin(ang)
    dof_pos, dof_vel = rand_dofs(N, g)
    actions = torch.rand(N, 8, generator=g) * 2 - 1
    sensors = 5.0 * torch.randn(N, 24, generator=g)
    potentials = -torch.rand(N, generator=g) * 300
    o, pot, prev_pot, _, _, ant_pos = one_ant.compute_ant_observations(
        torch.zeros(N, 60), root.clone(), boxr.clone(), torch.zeros(N, 3), potentials,
        torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(N, 1), dof_pos, dof_vel, ANT_LOWER, ANT_UPPER, 0.2,
        sensors, actions, 0.0166, 0.1, torch.tensor([1.0, 0.0, 0.0]).repeat(N, 1),
        torch.tensor([0.0, 0.0, 1.0]).repeat(N, 1), 2)
    bpos, bquat = one_ant.compute_box_pos(boxr)
    pos_before = ant_pos + 0.03 * torch.randn(N, 2, generator=g)
    box_before = bpos + 0.02 * torch.randn(N, 2, generator=g)
    reset_in = (torch.rand(N, generator=g) < 0.1).long()
    progress = torch.randint(0, 1002, (N,), generator=g)
    progress[:8] = torch.tensor([0, 500, 997, 998, 999, 1000, 1001, 3])
    rew, reset = one_ant_p.compute_ant_reward(
        o, reset_in, progress, actions, 0.1, 0.5, pot, prev_pot, 0.005, 0.05, 0.1, 0.31, -2.0, 1000,
        pos_before, box_before, ant_pos.clone(), bpos, 0.0166, 1.0, bquat, 0.0, 1.0, 0.0, 1.0, 500.0,
        torch.zeros(N, 2), 500.0)
    _save("oneant_reward", meta_common + "; fn: patched one_ant.compute_ant_reward (one_ant.py:465-560); "
          "quat_reward_scale=1, other scalars as tenant_reward",
          obs=o, reset_in=reset_in, progress=progress, actions=actions, potentials=pot, prev_potentials=prev_pot,
          pos_before=pos_before, box_before=box_before, ant_pos=ant_pos, box_pos=bpos, box_quat=bquat,
          rew=rew, reset=reset)

    # ---------------- ingenuity_reward / ingenuity_thrust ----------------
    g = torch.Generator().manual_seed(8)
    goals = [torch.tensor(p).repeat(N, 1) for p in ([4.0, 2.0, 1.0], [4.0, -2.0, 1.0], [4.0, 6.0, 1.0], [4.0, -6.0, 1.0])]
    roots = []
    for k in range(4):
        r = rand_root(N, g, xy_scale=1.0, zlo=0.3, zhi=2.5)
        r[:, 0:3] = goals[k] + torch.randn(N, 3, generator=g) * torch.tensor([3.0, 3.0, 0.8])
        r[: N // 16, 0] = goals[k][: N // 16, 0] + 8.5  # too far -> die
        roots.append(r)
    reset_in = (torch.rand(N, generator=g) < 0.1).long()
    progress = torch.randint(0, 1002, (N,), generator=g)
    progress[:8] = torch.tensor([0, 500, 997, 998, 999, 1000, 1001, 3])
    rew, reset = ingen.compute_ingenuity_reward(*roots, *[r[:, :3].clone() for r in roots], *goals,
                                                reset_in, progress, 1000.0)
    _save("ingenuity_reward", meta_common + "; fn: multi_ingenuity.compute_ingenuity_reward (multi_ingenuity.py:381-453)",
          roots=torch.stack(roots, 1), goals=torch.stack(goals, 1), reset_in=reset_in, progress=progress,
          rew=rew, reset=reset)

    g = torch.Generator().manual_seed(9)
    dummy = ingen.MultiIngenuity.__new__(ingen.MultiIngenuity)
    dummy.device = "cpu"
    dummy.dt = 0.0166
    dummy.num_envs = N
    dummy.thrust_lower_limit, dummy.thrust_upper_limit, dummy.thrust_lateral_component = 0, 2000, 0.2
    dummy.thrusts = torch.zeros(N, 8, 3)
    dummy.forces = torch.zeros(N, 24, 3)
    dummy.sim = None
    dummy.gym = types.SimpleNamespace(apply_rigid_body_force_tensors=lambda *a, **k: None)
    acts = (torch.rand(N, 24, generator=g) * 2 - 1) * 1.3  # beyond +-1 so the clamps fire
    dummy.pre_physics_step(acts)
    _save("ingenuity_thrust", meta_common + "; MultiIngenuity.pre_physics_step (multi_ingenuity.py:268-339) on a dummy object",
          actions=acts, dt=0.0166, thrusts=dummy.thrusts, forces=dummy.forces)

    # ---------------- vec_wrappers ----------------
    mvt = __import__("agents.tasks.agent_base.multi_vec_task", fromlist=["x"])
    vt = __import__("agents.tasks.agent_base.vec_task", fromlist=["x"])
    g = torch.Generator().manual_seed(10)
    obs_buf = 6.0 * torch.randn(N, 388, generator=g)
    rew_buf = torch.randn(N, generator=g)
    reset_buf = (torch.rand(N, generator=g) < 0.2).long()

    class _Task:
        num_envs, num_actions, num_obs, num_states = N, 8, 38, 0
        extras = {}

        def __init__(self):
            self.obs_buf, self.rew_buf, self.reset_buf = obs_buf, rew_buf, reset_buf
            self.states_buf = torch.zeros(N, 0)
            self.seen_actions = None

        def step(self, a):
            self.seen_actions = a.clone()

    with contextlib.redirect_stdout(io.StringIO()):
        env = mvt.MultiVecTaskPython(_Task(), "cpu")
    acts = [1.5 * (torch.rand(N, 8, generator=g) * 2 - 1) for _ in range(10)]
    obs_all, state_all, reward_all, done_all, info_all, _ = env.step(acts)
    seen_step = env.task.seen_actions.clone()
    r_obs, r_state, _ = env.reset()
    with contextlib.redirect_stdout(io.StringIO()):
        t1 = _Task()
        t1.num_obs, t1.num_actions = 388, 80
        env1 = vt.VecTaskPython(t1, "cpu", 5.0, 1.0)
    a80 = 1.5 * (torch.rand(N, 80, generator=g) * 2 - 1)
    o1, r1, d1, _ = env1.step(a80)
    _save("vec_wrappers", meta_common + "; MultiVecTaskPython.step/reset (multi_vec_task.py:94-175), "
          "VecTaskPython.step (vec_task.py:126-131) on a dummy task; np.Inf shim applied",
          obs_buf=obs_buf, rew_buf=rew_buf, reset_buf=reset_buf, actions=torch.stack(acts, 1),
          seen_actions=seen_step, reset_seen_actions=env.task.seen_actions, obs_all=obs_all, state_all=state_all, reward_all=reward_all,
          done_all=done_all, reset_obs=r_obs, reset_state=r_state,
          single_actions=a80, single_seen_actions=t1.seen_actions, single_obs=o1, single_rew=r1, single_done=d1)

    # ---------------- ppo_gae ----------------
    storage_mod = _load_by_path("ref_ppo_storage", os.path.join(REF, "agents/algorithms/rl/ppo/storage.py"))
    g = torch.Generator().manual_seed(11)
    T, NE = 8, N
    st = storage_mod.RolloutStorage(NE, T, (388,), (0,), (80,), "cpu", "sequential")
    rec = dict(obs=[], act=[], rew=[], done=[], val=[], logp=[], mu=[], sigma=[])
    for t in range(T):
        o = torch.randn(NE, 388, generator=g)
        a = torch.randn(NE, 80, generator=g)
        r = 5.0 * torch.randn(NE, generator=g)
        d = (torch.rand(NE, generator=g) < 0.1).long()
        v = torch.randn(NE, 1, generator=g)
        lp = torch.randn(NE, generator=g)
        mu = torch.randn(NE, 80, generator=g)
        sg = torch.rand(NE, 80, generator=g)
        st.add_transitions(o, torch.zeros(NE, 0), a, r, d, v, lp, mu, sg)
        for k, x in zip(rec, (o, a, r, d, v, lp, mu, sg)):
            rec[k].append(x)
    last_values = torch.randn(NE, 1, generator=g)
    # get_statistics() does `done = self.dones.cpu(); done[-1] = 1` (storage.py:68-69): with the storage on the
    # CPU `.cpu()` is not a copy, so it would overwrite the last row of dones.  The reference runs with the
    # storage on the GPU, where it is a copy -- reproduce that by taking the statistics from a deep copy.
    import copy
    stats_len, stats_rew = copy.deepcopy(st).get_statistics()
    st.compute_returns(last_values, 0.96, 0.95)
    _save("ppo_gae", meta_common + "; RolloutStorage.add_transitions/compute_returns/get_statistics "
          "(algorithms/rl/ppo/storage.py:32-72), gamma=0.96 lam=0.95",
          rewards=torch.stack(rec["rew"]), dones=torch.stack(rec["done"]), values=torch.stack(rec["val"]),
          last_values=last_values, gamma=0.96, lam=0.95, returns=st.returns, advantages=st.advantages,
          stored_rewards=st.rewards, stored_dones=st.dones, stored_logp=st.actions_log_prob,
          logp=torch.stack(rec["logp"]), mean_traj_len=stats_len, mean_reward=stats_rew)

    # ---------------- ppo_act ----------------
    # ActorCritic.act / evaluate (algorithms/rl/ppo/module.py:73-109) on a small network: pins the scale_tril = diag(sigma^2)
    # reading of the covariance (:76-77), the log-probability, the entropy and what is returned as "sigma" (:87).
    module_mod = _load_by_path("ref_ppo_module", os.path.join(REF, "agents/algorithms/rl/ppo/module.py"))
    torch.manual_seed(21)
    with _quiet():
        ac = module_mod.ActorCritic((48,), (0,), (8,), 0.8, {"pi_hid_sizes": [32, 32, 16], "vf_hid_sizes": [32, 32, 16],
                                                           "activation": "elu"}, asymmetric=False)
    with torch.no_grad():
        ac.log_std.copy_(torch.linspace(-0.6, 0.1, 8))
    g = torch.Generator().manual_seed(22)
    o = torch.randn(N, 48, generator=g)
    torch.manual_seed(23)
    with torch.no_grad():
        a_act, lp_act, v_act, mu_act, sg_act = ac.act(o, torch.zeros(N, 0))
        lp_ev, ent_ev, v_ev, mu_ev, sg_ev = ac.evaluate(o, torch.zeros(N, 0), a_act)
        a_inf = ac.act_inference(o)
    sd = {k.replace(".", "_"): v for k, v in ac.state_dict().items()}
    _save("ppo_act", meta_common + "; ActorCritic.act/evaluate/act_inference (algorithms/rl/ppo/module.py:73-109), "
          "obs 48, actions 8, hidden [32,32,16] elu, initial_std 0.8 then log_std = linspace(-0.6, 0.1, 8)",
          obs=o, actions=a_act, log_prob=lp_act, value=v_act, mu=mu_act, sigma=sg_act, eval_log_prob=lp_ev, eval_entropy=ent_ev,
          eval_value=v_ev, inference=a_inf, **sd)

    # ---------------- replay_buffer ----------------
    # ReplayBuffer of DDPG / TD3 / SAC (algorithms/rl/{ddpg,td3,sac}/storage.py): the ring's wrap-around rule (:29-33: on the first
    # overflow the cursor becomes (replay_size + 1) % replay_size = 1, slot 0 keeps its old transition and the cursor then
    # runs on to replay_size again), get_statistics (:46-52) and the `random.sample` mini-batches (:54-70).
    import random as _random
    rb = {}
    for algo in ("ddpg", "td3", "sac"):
        mod = _load_by_path("ref_%s_storage" % algo, os.path.join(REF, "agents/algorithms/rl/%s/storage.py" % algo))
        NE, R, OD, AD = 6, 5, 13, 4
        buf = mod.ReplayBuffer(NE, R, 8, 2, (OD,), (0,), (AD,), "cpu", "sequential")
        g = torch.Generator().manual_seed(31)
        ins = dict(obs=[], act=[], rew=[], nobs=[], done=[])
        cursor, full = [], []
        for t in range(13):
            o, a = torch.randn(NE, OD, generator=g), torch.randn(NE, AD, generator=g)
            r, no = torch.randn(NE, generator=g), torch.randn(NE, OD, generator=g)
            d = (torch.rand(NE, generator=g) < 0.3).long()
            buf.add_transitions(o, torch.zeros(NE, 0), a, r, no, d)
            for k, x in zip(ins, (o, a, r, no, d)):
                ins[k].append(x)
            cursor.append(buf.step); full.append(int(buf.fullfill))
        import copy
        stats_len, stats_rew = copy.deepcopy(buf).get_statistics()      # see ppo_gae above: .cpu() aliases a CPU storage
        _random.seed(32)
        batches = buf.mini_batch_generator(3)
        part = copy.deepcopy(mod.ReplayBuffer(NE, R, 8, 2, (OD,), (0,), (AD,), "cpu", "sequential"))
        for t in range(3):
            part.add_transitions(ins["obs"][t], torch.zeros(NE, 0), ins["act"][t], ins["rew"][t], ins["nobs"][t], ins["done"][t])
        p_len, p_rew = copy.deepcopy(part).get_statistics()
        _random.seed(33)
        p_batches = part.mini_batch_generator(4)
        rb[algo] = dict(cursor=np.array(cursor), full=np.array(full), observations=buf.observations, next_observations=buf.next_observations,
                        actions=buf.actions, rewards=buf.rewards, dones=buf.dones, mean_traj_len=stats_len, mean_reward=stats_rew,
                        batches=np.array(batches), part_mean_traj_len=p_len, part_mean_reward=p_rew, part_batches=np.array(p_batches))
        if algo == "ddpg":
            rb["in"] = {k: torch.stack(v) for k, v in ins.items()}
    flat = {"in_" + k: v for k, v in rb["in"].items()}
    for algo in ("ddpg", "td3", "sac"):
        flat.update({algo + "_" + k: v for k, v in rb[algo].items()})
    _save("replay_buffer", meta_common + "; ReplayBuffer.add_transitions/get_statistics/mini_batch_generator "
          "(algorithms/rl/{ddpg,td3,sac}/storage.py), 6 envs, replay_size 5, batch_size 8, 13 adds (random.seed(32) before the "
          "3 mini-batches of the full buffer; a second buffer with 3 adds, random.seed(33), 4 mini-batches)", **flat)

    # ---------------- offpolicy_act ----------------
    # MLPActorCritic.act of DDPG and TD3 (algorithms/rl/{ddpg,td3}/module.py:35-61): deterministic and with exploration noise
    # (torch.manual_seed(42) right before the noisy call; the noise is torch.randn(shape) of the CPU generator), and the Q values.
    from gym import spaces as _spaces
    flat = {}
    for algo in ("ddpg", "td3"):
        mod = _load_by_path("ref_%s_module" % algo, os.path.join(REF, "agents/algorithms/rl/%s/module.py" % algo))
        torch.manual_seed(41)
        ac = mod.MLPActorCritic(_spaces.Box(-np.inf * np.ones(52), np.inf * np.ones(52)), _spaces.Box(-np.ones(24), np.ones(24)), 0.1, "cpu",
                                hidden_sizes=[32, 32, 32])
        g = torch.Generator().manual_seed(43)
        o = 2.0 * torch.randn(N, 52, generator=g)
        det = ac.act(o)
        torch.manual_seed(42)
        noisy = ac.act(o, deterministic=False)
        with torch.no_grad():
            qs = [ac.q(o, det)] if algo == "ddpg" else [ac.q1(o, det), ac.q2(o, det)]
        flat.update({algo + "_sd_" + k.replace(".", "_"): v for k, v in ac.state_dict().items()})
        flat.update({algo + "_obs": o, algo + "_det": det, algo + "_noisy": noisy, algo + "_q": torch.stack(qs)})
        flat[algo + "_keys"] = np.array(list(ac.state_dict().keys()))
    _save("offpolicy_act", meta_common + "; MLPActorCritic.act / q (algorithms/rl/{ddpg,td3}/module.py), obs 52, actions 24, hidden "
          "[32,32,32] ReLU, act_noise 0.1, act_limit 1", **flat)

    # ---------------- marl_gae ----------------
    sb = __import__("agents.algorithms.marl.utils.separated_buffer", fromlist=["x"])
    popart = _load_by_path("ref_popart", os.path.join(REF, "agents/algorithms/marl/utils/popart.py"))
    valuenorm = _load_by_path("ref_valuenorm", os.path.join(REF, "agents/algorithms/marl/utils/valuenorm.py"))
    from gym import spaces
    out = {}
    for tag, use_popart, use_vn in (("popart", True, False), ("valuenorm", False, True), ("plain", False, False)):
        g = torch.Generator().manual_seed(12)
        cfg = dict(episode_length=8, n_rollout_threads=N, hidden_size=64, recurrent_N=1, gamma=0.99,
                   gae_lambda=0.95, use_gae=True, use_popart=use_popart, use_valuenorm=use_vn,
                   use_proper_time_limits=False)
        buf = sb.SeparatedReplayBuffer(cfg, spaces.Box(-np.inf, np.inf, (46,)), spaces.Box(-np.inf, np.inf, (388,)),
                                       spaces.Box(-np.ones(8), np.ones(8)), "cpu")
        ins = dict(share_obs=[], obs=[], actions=[], logp=[], values=[], rewards=[], masks=[])
        for t in range(8):
            so = torch.randn(N, 388, generator=g)
            ob = torch.randn(N, 46, generator=g)
            ac = torch.randn(N, 8, generator=g)
            lp = torch.randn(N, 8, generator=g)
            vp = torch.randn(N, 1, generator=g)
            rw = 5.0 * torch.randn(N, 1, generator=g)
            mk = (torch.rand(N, 1, generator=g) > 0.1).float()
            buf.insert(so, ob, torch.zeros(N, 1, 64), torch.zeros(N, 1, 64), ac, lp, vp, rw, mk)
            for k, x in zip(ins, (so, ob, ac, lp, vp, rw, mk)):
                ins[k].append(x)
        next_value = torch.randn(N, 1, generator=g)
        norm = None
        if use_popart:
            norm = popart.PopArt(1)
        elif use_vn:
            norm = valuenorm.ValueNorm(1)
        if norm is not None:
            # give the normaliser non-trivial running statistics
            samples = 3.0 * torch.randn(256, 1, generator=g) + 1.5
            if use_popart:
                norm(samples)
            else:
                norm.update(samples)
            mean, var = norm.running_mean_var()
        else:
            mean, var = torch.zeros(1), torch.ones(1)
        buf.compute_returns(next_value, norm)
        if tag == "popart":
            out.update(rewards=torch.stack(ins["rewards"]), values=torch.stack(ins["values"]),
                       masks_in=torch.stack(ins["masks"]), next_value=next_value,
                       share_obs_in=torch.stack(ins["share_obs"])[:, :4], obs_in=torch.stack(ins["obs"])[:, :4],
                       stored_share_obs=buf.share_obs[:, :4], stored_obs=buf.obs[:, :4])
        out["returns_" + tag] = buf.returns
        out["value_preds_" + tag] = buf.value_preds
        out["masks_" + tag] = buf.masks
        out["norm_mean_" + tag] = mean
        out["norm_var_" + tag] = var
    _save("marl_gae", meta_common + "; SeparatedReplayBuffer.insert/compute_returns "
          "(algorithms/marl/utils/separated_buffer.py:67-85,124-168), gamma=0.99 gae_lambda=0.95, "
          "use_proper_time_limits=False; normaliser = PopArt / ValueNorm / none", gamma=0.99, gae_lambda=0.95, **out)


# --------------------------------------------------------------------------------------
def tenant_glue(ten_ant_p, tu):
    """Drive the reference TenAnt.pre/post_physics_step on a dummy object for 3 steps."""
    n = 16
    T = ten_ant_p.TenAnt
    env = T.__new__(T)
    env.device = "cpu"
    env.num_envs, env.num_agents, env.num_dof, env.num_dof_1 = n, 10, 8, 8
    env.randomize = False
    env.dof_vel_scale, env.contact_force_scale, env.power_scale = 0.2, 0.1, 1.0
    env.heading_weight, env.up_weight = 0.5, 0.1
    env.actions_cost_scale, env.energy_cost_scale, env.joints_at_limit_cost_scale = 0.005, 0.05, 0.1
    env.death_cost, env.termination_height, env.max_episode_length = -2.0, 0.31, 1000
    env.move_reward_scale, env.quat_reward_scale = 1.0, 0.0
    env.ant_dist_reward_scale = env.goal_dist_reward_scale = 500.0
    env.dt, env.up_axis_idx = 0.0166, 2
    env.x_goal, env.y_goal, env.z_goal = 0.0, 1.0, 0.0
    env.dof_limits_lower, env.dof_limits_upper = ANT_LOWER.clone(), ANT_UPPER.clone()
    env.joint_gears = torch.full((80,), 15.0)
    env.sim = None

    # initial (construction-time) poses: ten ants then the box, env-grid origin added (global frame)
    g = torch.Generator().manual_seed(5)
    init_root = torch.zeros(n * 11, 13)
    init_root[:, 6] = 1.0
    starts = [(6.0, s * (1.5 + 3.0 * j), 1.0) for j in range(5) for s in (-1.0, 1.0)]
    npr = int(np.sqrt(n))
    origin = torch.zeros(n, 3)
    for i in range(n):
        origin[i, 0] = (i % npr) * 80.0
        origin[i, 1] = (i // npr) * 80.0
    for i in range(n):
        for k in range(10):
            init_root[i * 11 + k, 0:3] = torch.tensor(starts[k]) + origin[i]
        init_root[i * 11 + 10, 0:3] = torch.tensor([4.0, 0.0, 1.0]) + origin[i]

    internal = dict(root=init_root.clone(), dof=torch.zeros(n * 80, 2))
    env.root_states = init_root.clone()          # wrapped tensor = snapshot as of the last refresh
    env.dof_state = torch.zeros(n * 80, 2)
    env.initial_root_states = init_root.clone()

    def refresh_root(_):
        env.root_states.copy_(internal["root"])

    def refresh_dof(_):
        env.dof_state.copy_(internal["dof"])

    def set_root_indexed(_, src, idx, cnt):
        idx = idx.long()
        internal["root"][idx] = src[idx]

    def set_dof_indexed(_, src, idx, cnt):
        s3 = src.view(n * 10, 8, 2)
        internal["dof"].view(n * 10, 8, 2)[_actor_to_antrow(idx.long())] = s3[_actor_to_antrow(idx.long())]

    def _actor_to_antrow(actor_idx):
        return (actor_idx // 11) * 10 + (actor_idx % 11)

    env.gym = types.SimpleNamespace(
        refresh_dof_state_tensor=refresh_dof, refresh_actor_root_state_tensor=refresh_root,
        refresh_force_sensor_tensor=lambda _: None, set_actor_root_state_tensor_indexed=set_root_indexed,
        set_dof_state_tensor_indexed=set_dof_indexed, set_dof_actuation_force_tensor=lambda *a: None)
    for k in range(10):
        setattr(env, "dof_pos_%d" % (k + 1), env.dof_state.view(n, -1, 2)[:, 8 * k:8 * k + 8, 0])
        setattr(env, "dof_vel_%d" % (k + 1), env.dof_state.view(n, -1, 2)[:, 8 * k:8 * k + 8, 1])
        setattr(env, "obs_buf_%d" % (k + 1), torch.zeros(n, 38))
        setattr(env, "ant_indices_%d" % (k + 1), torch.arange(n) * 11 + k)
        setattr(env, "pos_before_%d" % (k + 1), torch.zeros(2))
        setattr(env, "goal_before_%d" % (k + 1), torch.zeros(2))
        setattr(env, "goal_%d" % (k + 1), torch.zeros(n, 2))
        j, s = k // 2, (-1.0 if k % 2 == 0 else 1.0)
        setattr(env, "box_targets_%d" % (k + 1), torch.tensor([0.0, s * (1.5 + 3.0 * j)]).repeat(n, 1))
    env.box_indices = torch.arange(n) * 11 + 10
    env.box_before = torch.zeros(2)
    zero = torch.tensor([0.0])
    env.initial_dof_pos = torch.where(env.dof_limits_lower > zero, env.dof_limits_lower,
                                      torch.where(env.dof_limits_upper < zero, env.dof_limits_upper,
                                                  torch.zeros(n, 8)))
    env.box_pos = torch.zeros(n, 2)
    env.box_quat = torch.zeros(n, 4)
    env.box_quat_before = torch.zeros(n, 4)
    env.targets = torch.zeros(n, 3)
    env.box_targets = torch.zeros(n, 2)
    env.inv_start_rot = tu.quat_conjugate(torch.tensor([0.0, 0.0, 0.0, 1.0])).repeat(n, 1)
    env.basis_vec0 = torch.tensor([1.0, 0.0, 0.0]).repeat(n, 1)
    env.basis_vec1 = torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1)
    env.obs_buf = torch.zeros(n, 38)
    env.rew_buf = torch.zeros(n)
    env.reset_buf = torch.ones(n, dtype=torch.long)
    env.progress_buf = torch.zeros(n, dtype=torch.long)
    env.randomize_buf = torch.zeros(n, dtype=torch.long)

    rec = dict(sim_root=[], sim_dof=[], actions=[], noise_pos=[], noise_vel=[], obs=[], rew=[], reset=[],
               progress=[], pos_before=[], goal_before=[], box_before=[], reset_in=[], progress_in=[],
               root_after=[], dof_after=[])
    n_steps = 4
    for t in range(n_steps):
        actions = torch.rand(n, 80, generator=g) * 2 - 1
        env.pre_physics_step(actions)
        # "simulate": supply a synthetic post-physics internal state (plausible perturbation of the last one)
        sim_root = internal["root"].clone()
        sim_root[:, 0:3] += 0.05 * torch.randn(n * 11, 3, generator=g)
        sim_root[:, 2] = sim_root[:, 2].clamp(min=0.33)
        q = sim_root[:, 3:7] + 0.05 * torch.randn(n * 11, 4, generator=g)
        sim_root[:, 3:7] = q / q.norm(dim=-1, keepdim=True)
        sim_root[:, 7:13] = torch.randn(n * 11, 6, generator=g)
        sim_dof = internal["dof"].clone()
        sim_dof[:, 0] += 0.05 * torch.randn(n * 80, generator=g)
        sim_dof[:, 1] = torch.randn(n * 80, generator=g)
        if t == 1:   # make envs 2 and 5 fall this step -> they reset at step 2
            sim_root[2 * 11 + 3, 2] = 0.2
            sim_root[5 * 11 + 0, 2] = 0.25
        if t == 2:   # episode timeout for env 7 -> reset at step 3
            env.progress_buf[7] = 998
        internal["root"], internal["dof"] = sim_root.clone(), sim_dof.clone()
        rec["reset_in"].append(env.reset_buf.clone())
        rec["progress_in"].append(env.progress_buf.clone())
        # reproduce the two torch.rand draws of reset_idx to record them as injectable noise
        ids = env.reset_buf.nonzero(as_tuple=False).flatten()
        torch.manual_seed(100 + t)
        npos = torch.zeros(n, 8)
        nvel = torch.zeros(n, 8)
        if len(ids) > 0:
            npos[ids] = 0.4 * torch.rand(len(ids), 8) - 0.2
            nvel[ids] = 0.2 * torch.rand(len(ids), 8) - 0.1
        torch.manual_seed(100 + t)
        with _quiet():
            env.post_physics_step()
        rec["sim_root"].append(sim_root)
        rec["sim_dof"].append(sim_dof)
        rec["actions"].append(actions)
        rec["noise_pos"].append(npos)
        rec["noise_vel"].append(nvel)
        rec["obs"].append(env.obs_buf.clone())
        rec["rew"].append(env.rew_buf.clone())
        rec["reset"].append(env.reset_buf.clone())
        rec["progress"].append(env.progress_buf.clone())
        rec["pos_before"].append(torch.stack([getattr(env, "pos_before_%d" % (k + 1)) for k in range(10)], 1))
        rec["goal_before"].append(torch.stack([getattr(env, "goal_before_%d" % (k + 1)) for k in range(10)], 1))
        rec["box_before"].append(env.box_before.clone())
        rec["root_after"].append(internal["root"].clone())
        rec["dof_after"].append(internal["dof"].clone())
    out = {k: torch.stack(v) for k, v in rec.items()}
    out["init_root"] = init_root
    out["env_origin"] = origin
    return out


if __name__ == "__main__":
    main()
