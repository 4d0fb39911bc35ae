#!/usr/bin/env python3
"""Fixtures for MultiAntCircle (agents/tasks/multi_ant_circle.py) -- INTENDED SEMANTICS, parity unpinned.

The reference cannot import this task: its @torch.jit.script functions call numpy on tensors and do arithmetic on bool tensors,
`compute_reward` passes 19 arguments to a 16-parameter function (:298-318), the task is not registered (utils/parse_task.py:8-10)
and no cfg/MultiAntCircle.yaml ships.  This script loads a TEMP COPY of the file with the mechanical substitutions listed in
SUBSTITUTIONS -- every one of them recorded in the fixture's meta string with its count -- and stores inputs / outputs of the
two functions of the path:

    circle_obs.npz      compute_ant_observations (multi_ant_circle.py:505-543): the same 38 entries per ant as TenAnt's
    circle_reward.npz   compute_ant_reward (:400-502) called with its OWN 16 parameters

One substitution is a reading, not a repair: `np.linalg.norm(pos)` on an [N, 2] tensor would be ONE Frobenius norm over all envs;
the per-env norm (`torch.norm(pos, dim=-1)`) is taken as what is meant (a reward that couples every env of a vectorised task cannot
be).  Runs only where /root/reference exists; nothing of the reference is copied into the repo (the temp copy is deleted).

    python tests/golden/make_circle_fixture.py
"""
import os
import re
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf  # noqa: E402  (import plumbing: namespace packages + the name-only isaacgym stand-in)

SUBSTITUTIONS = [
    (r"np\.linalg\.norm\((pos_\d)\)", r"torch.norm(\1, dim=-1)", "np.linalg.norm(pos_k) -> torch.norm(pos_k, dim=-1)  [per-env norm: a reading]"),
    (r"np\.abs\(np\.arctan2\(b,a\)\*180/np\.pi\)", r"torch.abs(torch.atan2(b,a)*180/3.141592653589793)", "np.abs(np.arctan2(b,a)*180/np.pi) -> torch.abs(torch.atan2(b,a)*180/pi)"),
    (r"f = -c\b", r"f = -(c.float())", "f = -c (negation of a bool tensor) -> -(c.float())"),
    (r"\(\(clockwise_(\d)\.mul\(is_oncircle_\1\)\) - 1\)", r"((clockwise_\1.mul(is_oncircle_\1)).float() - 1)", "(bool) - 1 -> (bool).float() - 1"),
    (r"# type: \(Tensor\) -> \[Tensor\]", r"# type: (Tensor) -> Tensor", "type comment of compute_angle: [Tensor] -> Tensor"),
]


def load_patched(tmpdir):
    src = open(os.path.join(mf.REF, "agents/tasks/multi_ant_circle.py")).read()
    counts = []
    for pat, rep, what in SUBSTITUTIONS:
        src, n = re.subn(pat, rep, src)
        counts.append((what, n))
        assert n > 0, ("substitution did not apply", what)
    path = os.path.join(tmpdir, "multi_ant_circle_patched.py")
    with open(path, "w") as f:
        f.write(src)
    return mf._load_by_path("multi_ant_circle_patched", path), counts


def main():
    if not os.path.isdir(mf.REF):
        sys.exit("reference tree not present: this script runs in the build container only")
    mf._setup_imports()
    tmpdir = tempfile.mkdtemp(prefix="mms_circle_fixture_")
    try:
        with mf._quiet():
            mod, counts = load_patched(tmpdir)
        meta = ("torch %s; agents/tasks/multi_ant_circle.py through a temp copy with substitutions: %s; INTENDED SEMANTICS, the reference cannot run this task"
                % (torch.__version__, "; ".join("%s (x%d)" % c for c in counts)))
        g = torch.Generator().manual_seed(4242)
        # ---- observations: two ants, random poses, the same helper inputs as the task builds (:136-143, :322-341)
        N = 64
        root = mf.rand_root(N, g, xy_scale=4.0)
        root[:8, 0:2] *= 0.01                                               # near the target (the origin): heading / angle edge
        dof_pos, dof_vel = mf.rand_dofs(N, g)
        lo = torch.tensor([-0.698132, 0.523599, -0.698132, -1.745329, -0.698132, -1.745329, -0.698132, 0.523599])
        hi = torch.tensor([0.698132, 1.745329, 0.698132, -0.523599, 0.698132, -0.523599, 0.698132, 1.745329])
        actions = torch.rand(N, 8, generator=g) * 2 - 1
        targets = torch.zeros(N, 3)
        inv_start_rot = torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(N, 1)
        basis0, basis1 = torch.tensor([1.0, 0, 0]).repeat(N, 1), torch.tensor([0, 0, 1.0]).repeat(N, 1)
        with mf._quiet():
            obs = mod.compute_ant_observations(torch.zeros(N, 38), root.clone(), targets, inv_start_rot, dof_pos, dof_vel, lo, hi, 0.2,
                                               actions, 0.0166, 0.1, basis0, basis1, 2)
        np.savez(os.path.join(HERE, "circle_obs.npz"), meta=np.array(meta + "; fn: compute_ant_observations (:505-543)"), root=mf._np(root),
                 dof_pos=mf._np(dof_pos), dof_vel=mf._np(dof_vel), dof_lower=mf._np(lo), dof_upper=mf._np(hi), actions=mf._np(actions), obs=mf._np(obs))
        # ---- reward from CONSISTENT states (so that the same case can go through an engine's post-step): two ants per env on / off
        #      the ring, both turning senses, fallen ants, joints at their limits, timeouts; observations by the function above
        N = 512
        ang = torch.rand(N, 2, generator=g) * 2 * np.pi
        rad = 3.0 + (torch.rand(N, 2, generator=g) - 0.5) * 1.2            # 2.4 .. 3.6: inside, on and outside the 2.7-3.3 ring
        rad[:16] = torch.tensor([2.7, 3.3]).repeat(16, 1) + (torch.rand(16, 2, generator=g) - 0.5) * 1e-3   # the ring's edges
        dang = (torch.rand(N, 2, generator=g) - 0.5) * 0.1
        dang[16:32] = 0.0                                                   # no angular motion: "clockwise" is a strict inequality
        pos_now = torch.stack([rad * torch.cos(ang + dang), rad * torch.sin(ang + dang)], -1)     # [N, 2 ants, 2]
        pos_prev = torch.stack([rad * torch.cos(ang), rad * torch.sin(ang)], -1)
        pos_now[32:48, :, 1] *= 1e-4                                        # on the x axis: the 0 / 360 degree seam
        pos_now[:, 1] = -pos_now[:, 1]                                      # ant 2 is looked at through pos_2 = -obs_2[:, :2] (:428)
        roots, dps, dvs, obs = [], [], [], []
        actions = torch.rand(N, 16, generator=g) * 2 - 1
        for k in range(2):
            root = mf.rand_root(N, g, xy_scale=1.0, zlo=0.2, zhi=0.8)       # some below the 0.31 termination height
            root[:, 0:2] = pos_now[:, k]
            upright = torch.rand(N, generator=g) < 0.6                      # up_proj on both sides of 0.93
            tilt = mf.rand_quats(N, g) * 0.12
            tilt[:, 3] = 1.0
            root[upright, 3:7] = (tilt / tilt.norm(dim=-1, keepdim=True))[upright]
            dp, dv = mf.rand_dofs(N, g)
            dp[::7] = hi                                                    # joints at the upper limit: dof_at_limit_cost
            with mf._quiet():
                o = mod.compute_ant_observations(torch.zeros(N, 38), root.clone(), torch.zeros(N, 3), torch.tensor([0.0, 0.0, 0.0, 1.0]).repeat(N, 1),
                                                 dp, dv, lo, hi, 0.2, actions[:, 8 * k:8 * k + 8].contiguous(), 0.0166, 0.1,
                                                 torch.tensor([1.0, 0, 0]).repeat(N, 1), torch.tensor([0, 0, 1.0]).repeat(N, 1), 2)
            roots.append(root); dps.append(dp); dvs.append(dv); obs.append(o)
        progress = torch.randint(0, 1002, (N,), generator=g)
        progress[:8] = torch.tensor([997, 998, 999, 1000, 1001, 0, 1, 500])
        reset_buf = torch.zeros(N, dtype=torch.int64)
        with mf._quiet():
            rew, reset = mod.compute_ant_reward(obs[0], obs[1], reset_buf, progress, actions, 0.1, 0.5, 0.005, 0.05, 0.1, 0.31, -2.0, 1000.0,
                                                pos_prev[:, 0].contiguous(), pos_prev[:, 1].contiguous(), 0.0166)
            angle = mod.compute_angle(pos_now[:, 0].contiguous())
        np.savez(os.path.join(HERE, "circle_reward.npz"), meta=np.array(meta + "; fn: compute_ant_observations (:505-543) on two ants' states, then compute_ant_reward "
                                                                               "(:400-502) with its own 16 parameters; compute_angle (:385-398)"),
                 root_1=mf._np(roots[0]), root_2=mf._np(roots[1]), dof_pos_1=mf._np(dps[0]), dof_pos_2=mf._np(dps[1]), dof_vel_1=mf._np(dvs[0]),
                 dof_vel_2=mf._np(dvs[1]), obs1=mf._np(obs[0]), obs2=mf._np(obs[1]), actions=mf._np(actions), progress=mf._np(progress),
                 pos_before_1=mf._np(pos_prev[:, 0]), pos_before_2=mf._np(pos_prev[:, 1]), rew=mf._np(rew), reset=mf._np(reset), angle_1=mf._np(angle),
                 params=np.array([0.1, 0.5, 0.005, 0.05, 0.1, 0.31, -2.0, 1000.0, 0.0166], np.float64))
        print("wrote circle_obs.npz, circle_reward.npz;", meta)
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


if __name__ == "__main__":
    main()
