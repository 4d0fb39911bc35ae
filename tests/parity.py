"""Shared parity gates: the same checks run on the CPU lane emulation (tests/test_lane_emulation.py), on the CPU build of the
lane code (tests/test_cpu_backend.py) and on the HIP kernels (tests/test_gpu_parity.py, through the C ABI).

Three independent questions, three gates:

1. PHYSICS.  The step map is stiff (k = 1e4..2e4 N/m on 0.07 kg feet): two correct fp32 implementations cannot agree entry for
   entry at 1e-4 -- the fp32 ORACLE itself sits ~1e-3 rad/s (median) from the same step evaluated in double.  So the yardstick
   is that double evaluation (oracle/mms_oracle.c compiled with -DMO_F64, same model, same fp32 inputs):
       err(implementation, f64)  <=  RATIO x err(fp32 oracle, f64) + floor        on the median,
       err(implementation, f64)  <=  RATIO_TAIL x err(fp32 oracle, f64) + floor   on the 99th percentile
   over all (env, step) pairs of a test -- one sample per pair, the env's largest error -- for velocities (relative to
   max(1, |v|)), poses and -- OneAnt -- the foot sensors.  (The tail is an order statistic of a heavy-tailed sample: the lane
   emulation on the CPU -- a clean fp32 evaluation with no approximate functions -- measures 0.3 .. 2.8 x the oracle's own p99.
   The largest single sample is capped, AND the far tail is bounded (round 3): the share of samples above 10 x the fp32 oracle's own
   p99 may be at most 2 x the oracle's share + 1e-3, p99.9 is recorded for both, and for the WORST sample of every test the double
   evaluation is repeated on inputs perturbed by +-1 ulp (fp32) -- `worst_f64_sensitivity_1ulp` in the margins file: the model has
   rare states where one ulp on an input moves the DOUBLE result by O(1) rad/s -- a foot caught between the ground and the tilted
   box inside the 0.5 mm activation ramp -- and whichever side of such a kink an implementation's rounding lands on is not an
   error; the sensitivity next to the worst error makes that a measurement instead of an explanation.)  A wrong term,
   index or sign is O(0.1 .. 10) on most steps and fails by orders of magnitude; an implementation that is merely sloppier
   than a clean fp32 evaluation (approximate reciprocals, polynomial sin / cos) fails too once it is 2x worse.
2. EPILOGUE (reset, observations, reward, caches), decoupled from the physics' conditioning: the oracle's post-physics glue is
   evaluated on the implementation's OWN post-step state and must reproduce its observation row to 1e-4 (angles modulo 2 pi),
   its reset flags exactly and its reward to rounding.
3. Integer outputs (reset, progress, reset_count) bit-exact against the teacher.

Every test records what it measured (not only pass / fail): MARGINS is written to gpurun_out/parity_margins.json at session end
(tests/conftest.py) and committed under profiles/ per round."""
import json
import os

import numpy as np

from oracle.oracle import OracleEngine, physics_f64

RATIO = 2.0              # the implementation may be this much farther from the double result than the fp32 oracle is (median)
RATIO_TAIL = 3.0         # ... and this much on the 99th percentile (measured: <= 1.5 lane emulation, <= 1.8 MI355X)
FAR_FACTOR = 10.0        # "far tail": samples above FAR_FACTOR x the fp32 oracle's own p99.  The implementation's share of such samples may be
FAR_RATIO = 2.0          # at most FAR_RATIO x the fp32 oracle's share + FAR_FLOOR: beyond p99 the samples are no longer only capped
FAR_FLOOR = 1e-3         # (or two samples, whichever is more)
VEL_FLOOR = 2e-6         # additive floors: a few ulp (the helicopters in free flight sit at 5e-8 on both sides)
POSE_FLOOR = 2e-6
SENS_FLOOR = 2e-5
VEL_CAP = 0.5            # any single step, any entry (relative to max(1, |v|)): blow-up guard
POSE_CAP = 5e-3
OBS_TOL = 1e-4           # epilogue on identical state
REW_FLIP_BUDGET = 1e-3   # fraction of (env, step) pairs whose reward may differ by more than rounding (hard thresholds in the
                         # reward: |ant - goal| < 1.5, up_proj > 0.93, |box - target| < 0.5; ten_ant.py:1073-1079,1193)

MARGINS = {}
STATE = ["root_states", "dof_state", "prev", "reset", "progress", "foot_sensors", "reset_count"]


def record(name, **vals):
    MARGINS[name] = {k: (float(v) if (np.isscalar(v) and not isinstance(v, str)) else v) for k, v in vals.items()}


def dump_margins(path):
    if not MARGINS:
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(MARGINS)
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)


def angle_err(a, b):
    return np.abs(((np.asarray(a, np.float64) - np.asarray(b, np.float64)) + np.pi) % (2 * np.pi) - np.pi)


def angle_columns(task, num_agents):
    """(columns holding yaw / roll / angle_to_target, per-ant stride) of the observation row."""
    if task in ("TenAnt", "MultiAntCircle"):
        return [38 * k + j for k in range(num_agents) for j in (9, 10, 11)]
    if task == "OneAnt":
        return [7, 8, 9]
    return []


def euler_conditioning(q):
    """min over (roll, yaw) of the radius of their atan2 arguments (get_euler_xyz): the angle error is rounding / radius."""
    x, y, z, w = (q[..., i].astype(np.float64) for i in range(4))
    rr = np.hypot(2.0 * (w * x + y * z), w * w - x * x - y * y + z * z)
    ry = np.hypot(2.0 * (w * z + x * y), w * w + x * x - y * y - z * z)
    return np.minimum(rr, ry)


def clone_oracle(ora):
    """A second oracle engine with the same configuration (the checker of the decoupled epilogue gate)."""
    o = OracleEngine.__new__(OracleEngine)
    OracleEngine._init_from_config(o, ora.task, ora.config)
    return o


class TeacherForced:
    """Step-for-step comparison on identical inputs.  `ora` is the teacher (its trajectory is the one followed), `get(name)`
    returns the implementation's buffer as a numpy array AFTER its step.  Protocol per step:
        tf.before(act)          # inputs captured from the teacher (the caller has already pushed them to the implementation)
        ora.step(act); impl.step(act)
        tf.after("label")
    and tf.finish("record name") at the end applies the distribution gates and records the margins."""

    def __init__(self, ora, get, dr=None):
        self.ora, self.get, self.task = ora, get, ora.task
        self.chk = clone_oracle(ora)
        self.dr = dr
        self.n, self.A = ora.num_envs, ora.num_agents
        self.ant = self.task != "MultiIngenuity"
        self.ang_cols = angle_columns(self.task, self.A)
        self.log = {k: [] for k in ("gv", "ov", "gp", "op", "gs", "os", "dv", "dp", "obs", "ang", "rew")}
        self.flips = self.pairs = self.resets = self.live_steps = 0
        self.worst = None           # (relative velocity error, its inputs): the sample whose f64 sensitivity finish() measures
        gmax = float(np.max(np.abs(ora.tensor("env_origin")))) + 30.0
        self.ulp = float(np.spacing(np.float32(gmax)))

    def before(self, act):
        o = self.ora
        self.inp = {k: o.tensor(k).copy() for k in STATE}
        self.act = np.ascontiguousarray(act, np.float32)

    def _split(self, root, dof):
        root = root.reshape(self.n, -1, 13)
        dof = dof.reshape(self.n, -1, 2)
        pose = [root[..., 0:7].reshape(self.n, -1)]
        vel = [root[..., 7:13].reshape(self.n, -1), dof[..., 1]]
        if self.ant:
            pose.append(dof[..., 0])            # (the helicopters' visual rotor angles grow without bound: velocity only)
        return np.concatenate(pose, 1), np.concatenate(vel, 1)

    def after(self, what):
        o, inp, n = self.ora, self.inp, self.n
        got = {k: np.asarray(self.get(k)) for k in STATE + ["obs", "obs_clipped", "rew"]}
        live = inp["reset"] == 0
        self.resets += int((~live).sum())
        # ---- 3. integer outputs ----
        for k in ("reset", "progress", "reset_count"):
            np.testing.assert_array_equal(got[k], o.tensor(k), err_msg="%s: %s" % (what, k))
        # ---- 1. physics against the double evaluation ----
        r64, d64, s64 = physics_f64(o.config, self.act, inp["root_states"], inp["dof_state"], inp["reset"],
                                    inp["foot_sensors"] if self.ant else None, self.dr)
        p64, v64 = self._split(r64, d64)
        pg, vg = self._split(got["root_states"], got["dof_state"])
        po, vo = self._split(o.tensor("root_states"), o.tensor("dof_state"))
        if live.any():
            self.live_steps += 1
            vs = np.maximum(1.0, np.abs(v64[live]))
            # one sample per (env, step): the env's largest error.  (Statistics over per-STEP maxima would let one ill-conditioned
            # env decide a whole step: the model has rare states -- a foot caught between the ground and the tilted box inside the
            # 0.5 mm activation ramp -- where ONE ulp on an input moves the double result itself by 2 rad/s.)
            gv, ov = np.max(np.abs(vg[live] - v64[live]) / vs, 1), np.max(np.abs(vo[live] - v64[live]) / vs, 1)
            gp, op = np.max(np.abs(pg[live] - p64[live]), 1), np.max(np.abs(po[live] - p64[live]), 1)
            assert gv.max() < VEL_CAP, (what, "velocity vs f64", float(gv.max()))
            assert gp.max() < POSE_CAP, (what, "pose vs f64", float(gp.max()))
            if self.worst is None or float(gv.max()) > self.worst["err_impl"]:
                e = int(np.flatnonzero(live)[int(np.argmax(gv))])
                rows = lambda a: np.array(a.reshape(n, -1)[e], copy=True)
                self.worst = {"err_impl": float(gv.max()), "err_oracle32": float(ov[int(np.argmax(gv))]), "what": what, "env": e,
                              "act": rows(self.act), "root": rows(inp["root_states"]), "dof": rows(inp["dof_state"]),
                              "sens": rows(inp["foot_sensors"]) if self.ant else None,
                              "dr": None if self.dr is None else rows(np.asarray(self.dr, np.float32))}
            self.log["gv"].extend(gv.tolist()); self.log["ov"].extend(ov.tolist())
            self.log["gp"].extend(gp.tolist()); self.log["op"].extend(op.tolist())
            self.log["dv"].append(float(np.max(np.abs(vg[live] - vo[live]) / vs)))
            self.log["dp"].append(float(np.max(np.abs(pg[live] - po[live]))))
            if self.task == "OneAnt":
                sg, so = got["foot_sensors"].reshape(n, -1)[live], o.tensor("foot_sensors").reshape(n, -1)[live]
                st = s64.reshape(n, -1)[live]
                ss = np.maximum(1.0, np.abs(st))
                self.log["gs"].extend(np.max(np.abs(sg - st) / ss, 1).tolist()); self.log["os"].extend(np.max(np.abs(so - st) / ss, 1).tolist())
        if (~live).any():                       # reset rows: integer hashing is bit-exact, 0.4 u - 0.2 may contract into one fma
            assert np.max(np.abs(pg[~live] - po[~live])) < 3e-7 and np.max(np.abs(vg[~live] - vo[~live])) < 3e-7, (what, "reset state")
        # ---- 2. epilogue on the implementation's own post-step state ----
        c = self.chk
        c.tensor("root_states")[...] = got["root_states"]
        c.tensor("dof_state")[...] = got["dof_state"]
        c.tensor("foot_sensors")[...] = got["foot_sensors"]
        c.tensor("prev")[...] = inp["prev"]
        c.tensor("reset")[...] = 0
        c.tensor("progress")[...] = got["progress"] - 1
        c.step(self.act, physics=False)
        np.testing.assert_array_equal(got["reset"], c.tensor("reset"), err_msg="%s: reset flags from own state" % what)
        obs_g, obs_c = got["obs"], c.tensor("obs")
        cols = np.ones(obs_g.shape[1], bool)
        cols[self.ang_cols] = False
        e_obs = float(np.max(np.abs(obs_g[:, cols] - obs_c[:, cols]) / np.maximum(1.0, np.abs(obs_c[:, cols]))))
        assert e_obs < OBS_TOL, (what, "observation from own state", e_obs)
        self.log["obs"].append(e_obs)
        if self.ang_cols:
            q = got["root_states"].reshape(n, -1, 13)[:, :self.A, 3:7]
            tol = OBS_TOL + 2e-6 / np.maximum(euler_conditioning(q), 1e-6)            # [n, A]
            e_ang = angle_err(obs_g[:, self.ang_cols], obs_c[:, self.ang_cols]).reshape(n, self.A, 3)
            assert np.all(e_ang <= tol[..., None]), (what, "angles from own state", float(e_ang.max()))
            self.log["ang"].append(float(e_ang.max()))
        np.testing.assert_array_equal(got["obs_clipped"], np.clip(obs_g, -o.config.clip_obs, o.config.clip_obs), err_msg=what)
        pv_g, pv_c = got["prev"], c.tensor("prev")
        assert np.max(np.abs(pv_g - pv_c) / np.maximum(1.0, np.abs(pv_c))) < OBS_TOL, (what, "caches from own state")
        # reward = 500 x differences of global-frame fp32 positions (reference behaviour): on identical state only the rounding
        # of those differences is left -- one ulp of a coordinate per term, 2 terms per ant
        rew_tol = 500.0 * self.ulp * 2 * max(self.A, 1) + 2e-4 * np.abs(c.tensor("rew")) + 5e-4
        d_rew = np.abs(got["rew"] - c.tensor("rew"))
        self.flips += int(np.sum(d_rew > rew_tol))
        self.pairs += n
        self.log["rew"].append(float(np.max(np.where(d_rew > rew_tol, 0.0, d_rew))))

    def f64_sensitivity(self, trials=3):
        """How far does the DOUBLE evaluation of the worst sample's step move when its fp32 inputs move by one ulp?  The env's state is
        replicated over all n envs, env 0 unperturbed, every other copy with each input entry moved by -1 / 0 / +1 ulp at random;
        returns the largest relative velocity change against the unperturbed result over `trials` x (n - 1) perturbations."""
        w, n = self.worst, self.n
        if w is None or n < 2:
            return None
        rng = np.random.default_rng(12345)
        worst = 0.0
        for _ in range(trials):
            def spread(row):
                a = np.repeat(row[None, :].astype(np.float32), n, 0)
                step = rng.integers(-1, 2, a.shape)
                step[0] = 0
                up, dn = np.nextafter(a, np.float32(np.inf)), np.nextafter(a, np.float32(-np.inf))
                return np.where(step > 0, up, np.where(step < 0, dn, a)).astype(np.float32)
            root, dof, act = spread(w["root"]), spread(w["dof"]), spread(w["act"])
            sens = spread(w["sens"]) if w["sens"] is not None else None
            dr = np.repeat(w["dr"][None, :], n, 0).reshape(-1, 33) if w["dr"] is not None else None
            r64, d64, _ = physics_f64(self.ora.config, act, root.reshape(-1, 13), dof.reshape(-1, 2), np.zeros(n, np.int64),
                                      sens.reshape(self.inp["foot_sensors"].shape) if sens is not None else None, dr)
            _, v = self._split(r64, d64)
            worst = max(worst, float(np.max(np.abs(v[1:] - v[0]) / np.maximum(1.0, np.abs(v[0])))))
        return worst

    def finish(self, name, min_live_steps=10):
        L = self.log
        assert self.live_steps >= min_live_steps, ("physics was exercised on too few steps", self.live_steps)
        st = {}
        for tag, g, o_, floor in (("vel", "gv", "ov", VEL_FLOOR), ("pose", "gp", "op", POSE_FLOOR), ("sens", "gs", "os", SENS_FLOOR)):
            if not L[g]:
                continue
            for q, fn in (("median", np.median), ("p99", lambda x: np.percentile(x, 99))):
                a, b = float(fn(L[g])), float(fn(L[o_]))
                st["%s_%s_impl" % (tag, q)] = a
                st["%s_%s_oracle32" % (tag, q)] = b
            st["%s_max_impl" % tag] = float(np.max(L[g]))
            st["%s_max_oracle32" % tag] = float(np.max(L[o_]))
            st["%s_p999_impl" % tag] = float(np.percentile(L[g], 99.9))
            st["%s_p999_oracle32" % tag] = float(np.percentile(L[o_], 99.9))
            far = FAR_FACTOR * st["%s_p99_oracle32" % tag] + floor
            st["%s_far_share_impl" % tag] = float(np.mean(np.asarray(L[g]) > far))
            st["%s_far_share_oracle32" % tag] = float(np.mean(np.asarray(L[o_]) > far))
            st["%s_samples" % tag] = len(L[g])
        st.update(vel_vs_oracle32_median=float(np.median(L["dv"])), vel_vs_oracle32_max=float(np.max(L["dv"])),
                  pose_vs_oracle32_median=float(np.median(L["dp"])), pose_vs_oracle32_max=float(np.max(L["dp"])),
                  obs_own_state_max=float(np.max(L["obs"])), angle_own_state_max=float(np.max(L["ang"])) if L["ang"] else 0.0,
                  reward_own_state_max=float(np.max(L["rew"])), reward_flips=self.flips, pairs=self.pairs, resets=self.resets,
                  live_steps=self.live_steps, ratio_allowed=RATIO, ratio_tail_allowed=RATIO_TAIL)
        if self.worst is not None:
            st.update(worst_vel_err_impl=self.worst["err_impl"], worst_vel_err_oracle32=self.worst["err_oracle32"], worst_sample=self.worst["what"],
                      worst_f64_sensitivity_1ulp=self.f64_sensitivity(), far_factor=FAR_FACTOR, far_ratio_allowed=FAR_RATIO, far_floor=FAR_FLOOR)
        record(name, **st)
        for tag in ("vel", "pose", "sens"):
            if tag + "_far_share_impl" in st:
                # (the floor is at least two samples: in a test with < 2000 samples one sample already is a share of 1e-3)
                assert st[tag + "_far_share_impl"] <= FAR_RATIO * st[tag + "_far_share_oracle32"] + max(FAR_FLOOR, 2.0 / st[tag + "_samples"]), \
                    (name, tag, "far-tail share", st[tag + "_far_share_impl"], st[tag + "_far_share_oracle32"])
        for tag, floor in (("vel", VEL_FLOOR), ("pose", POSE_FLOOR), ("sens", SENS_FLOOR)):
            for q in ("median", "p99"):
                k = "%s_%s" % (tag, q)
                if k + "_impl" in st:
                    allowed = (RATIO if q == "median" else RATIO_TAIL) * st[k + "_oracle32"] + floor
                    assert st[k + "_impl"] <= allowed, (name, k, st[k + "_impl"], st[k + "_oracle32"])
        assert self.flips <= max(2, REW_FLIP_BUDGET * self.pairs), (name, "reward threshold flips", self.flips, self.pairs)
        self.chk.close()
        return st


# ------------------------------------------------------------------------------------------------------------------
# reference fixtures through an implementation's step path (physics off): the adapter `impl` offers
#   impl.put(name, array)  impl.get(name) -> array  impl.post_step(actions)  impl.step(actions)  impl.config  impl.close()
# and is built by make(task, cfg=None, **kw) -- the GPU engine, the CPU build of the lane code, or the lane emulation.
# ------------------------------------------------------------------------------------------------------------------
def _cfg(task, spacing=0.0, **env):
    from massive_marl_benchmark_amd.model import default_cfg
    cfg = default_cfg(task)
    cfg["env"]["envSpacing"] = spacing          # every env at the global origin: the fixtures' coordinates are global
    cfg["env"].update(env)
    return cfg


def fixture_tenant_obs(make, load_golden, tag):
    """tests/golden/tenant_obs.npz -- the reference's compute_ant_observations (ten_ant.py:1304-1350), gimbal-lock rows
    included -- through the fused step's epilogue: fixture row i is ant (i % 10) of env (i // 10)."""
    g = load_golden("tenant_obs")
    rows = g["root"].shape[0]
    n = (rows + 9) // 10
    impl = make("TenAnt", cfg=_cfg("TenAnt"), num_envs=n)
    root = impl.get("root_states").reshape(n, 11, 13).copy()
    dof = impl.get("dof_state").reshape(n, 10, 8, 2).copy()
    act = np.zeros((n, 10, 8), np.float32)
    idx = np.arange(rows)
    root[idx // 10, idx % 10] = g["root"]
    dof[idx // 10, idx % 10, :, 0] = g["dof_pos"]
    dof[idx // 10, idx % 10, :, 1] = g["dof_vel"]
    act[idx // 10, idx % 10] = g["actions"]
    impl.put("root_states", root.reshape(n * 11, 13))
    impl.put("dof_state", dof.reshape(n * 80, 2))
    impl.put("reset", np.zeros(n, np.int64))
    impl.post_step(act.reshape(n, 80))
    obs = impl.get("obs")[:, :380].reshape(n * 10, 38)[:rows]
    ref = g["obs"]
    ang = [9, 10, 11]
    rest = [i for i in range(38) if i not in ang]
    seen = np.clip(g["actions"], -1, 1)                            # the wrapper clamp is fused into the step (vec_task.py:127)
    ref = ref.copy()
    ref[:, 30:38] = seen
    e_rest = float(np.max(np.abs(obs[:, rest] - ref[:, rest])))
    q = g["root"][:, 3:7]
    ok = np.abs(2.0 * (q[:, 3] * q[:, 1] - q[:, 2] * q[:, 0])) < 0.999
    e_ok = float(angle_err(obs[ok][:, ang], ref[ok][:, ang]).max())
    e_lock = float(angle_err(obs[~ok][:, ang], ref[~ok][:, ang]).max()) if (~ok).any() else 0.0
    record(tag + "fixture_tenant_obs", max_abs=e_rest, angle=e_ok, angle_gimbal_lock_rows=e_lock, rows=rows, gimbal_rows=int((~ok).sum()))
    assert e_rest < 1e-4 and e_ok < 1e-4 and e_lock < 5e-3, (e_rest, e_ok, e_lock)       # (same bars as the oracle's own test)
    assert (~ok).any()
    impl.close()


def fixture_tenant_goals(make, load_golden, tag):
    """tests/golden/tenant_goals.npz (compute_box_pos / compute_other_goal, ten_ant.py:1353-1393): the box row of each env is a
    fixture row; the step's observation tail holds box_pos / box_quat and its goal cache the ten goals."""
    g = load_golden("tenant_goals")
    n = g["box_root"].shape[0]
    impl = make("TenAnt", cfg=_cfg("TenAnt"), num_envs=n)
    root = impl.get("root_states").reshape(n, 11, 13).copy()
    root[:, 10] = g["box_root"]
    impl.put("root_states", root.reshape(n * 11, 13))
    impl.put("reset", np.zeros(n, np.int64))
    impl.post_step(np.zeros((n, 80), np.float32))
    obs, prev = impl.get("obs"), impl.get("prev")
    np.testing.assert_array_equal(obs[:, 380:382], g["box_pos"])
    np.testing.assert_array_equal(obs[:, 382:386], g["box_quat"])
    np.testing.assert_array_equal(obs[:, 386:388], 0.0)
    e = float(np.max(np.abs(prev[:, 20:40].reshape(n, 10, 2) - g["goals"])))
    np.testing.assert_array_equal(prev[:, 40:42], g["box_pos"])
    record(tag + "fixture_tenant_goals", goals_max_abs=e, rows=n)
    assert e < 1e-4, e
    impl.close()


def fixture_oneant(make, load_golden, tag):
    """tests/golden/oneant_obs.npz (one_ant.py:563-627) and oneant_reward.npz (one_ant.py:465-560) through the OneAnt step."""
    g = load_golden("oneant_obs")
    n = g["root"].shape[0]
    impl = make("OneAnt", cfg=_cfg("OneAnt"), num_envs=n)
    root = np.stack([g["root"], g["box_root"]], 1).reshape(n * 2, 13)
    dof = np.stack([g["dof_pos"], g["dof_vel"]], -1).reshape(n * 8, 2)
    prev = impl.get("prev").copy()
    prev[:, 4] = g["potentials_in"]
    impl.put("root_states", root)
    impl.put("dof_state", dof)
    impl.put("foot_sensors", g["sensors"])
    impl.put("prev", prev)
    impl.put("reset", np.zeros(n, np.int64))
    impl.post_step(g["actions"])
    obs, ref = impl.get("obs"), g["obs"].copy()
    ref[:, 52:60] = np.clip(g["actions"], -1, 1)
    q = g["root"][:, 3:7]
    ok = np.abs(2.0 * (q[:, 3] * q[:, 1] - q[:, 2] * q[:, 0])) < 0.999
    rest = [i for i in range(60) if i not in (7, 8, 9)]
    e_rest = float(np.max(np.abs(obs[:, rest] - ref[:, rest])))
    e_ang = float(angle_err(obs[ok][:, 7:10], ref[ok][:, 7:10]).max())
    pv = impl.get("prev")
    e_pot = float(np.max(np.abs(pv[:, 4] - g["potentials"]) / np.maximum(1, np.abs(g["potentials"]))))
    np.testing.assert_array_equal(pv[:, 5], g["prev_potentials"])
    np.testing.assert_array_equal(pv[:, 0:2], g["ant_pos"])
    np.testing.assert_array_equal(pv[:, 2:4], g["box_pos"])
    record(tag + "fixture_oneant_obs", max_abs=e_rest, angle=e_ang, potentials_rel=e_pot, rows=n)
    assert e_rest < 1e-4 and e_ang < 1e-4 and e_pot < 1e-6, (e_rest, e_ang, e_pot)
    impl.close()

    # reward: a consistent state is rebuilt from the fixture's observation row (identity orientation, so the `up` term is
    # present in every row and is swapped for the fixture's own), caches and box pose come from the fixture
    g = load_golden("oneant_reward")
    n = g["obs"].shape[0]
    impl = make("OneAnt", cfg=_cfg("OneAnt"), num_envs=n)
    M = impl.config.model
    lower, upper = np.array(M.dof_lower[:], np.float32), np.array(M.dof_upper[:], np.float32)
    o = g["obs"]
    root = np.zeros((n, 2, 13), np.float32)
    root[:, :, 6] = 1.0
    root[:, 0, 0:2] = g["ant_pos"]
    root[:, 0, 2] = o[:, 0]
    root[:, 1, 0:2] = g["box_pos"]
    root[:, 1, 2] = 0.5
    root[:, 1, 3:7] = g["box_quat"]
    dof = np.zeros((n, 8, 2), np.float32)
    dof[..., 0] = 0.5 * (o[:, 12:20] * (upper - lower) + upper + lower)
    dof[..., 1] = o[:, 20:28] / 0.2
    prev = np.zeros((n, 6), np.float32)
    prev[:, 0:2], prev[:, 2:4] = g["pos_before"], g["box_before"]
    impl.put("root_states", root.reshape(n * 2, 13))
    impl.put("dof_state", dof.reshape(n * 8, 2))
    impl.put("prev", prev)
    impl.put("reset", np.zeros(n, np.int64))
    impl.put("progress", (g["progress"] - 1).astype(np.int64))
    impl.post_step(g["actions"])
    got = impl.get("obs")
    assert np.max(np.abs(got[:, 12:28] - o[:, 12:28])) < 1e-5                       # the reward's inputs are reproduced
    fallen = o[:, 0] < 0.31
    up_fix = 0.1 * (o[:, 10] > 0.93)
    expect = np.where(fallen, -2.0, g["rew"] - up_fix + 0.1)
    e = float(np.max(np.abs(impl.get("rew") - expect)))
    np.testing.assert_array_equal(impl.get("reset"), np.where(fallen | (g["progress"] >= 999), 1, 0))
    record(tag + "fixture_oneant_reward", max_abs=e, rows=n, fallen_rows=int(fallen.sum()))
    assert e < 2e-3, e                                                               # 500 x sqrt rounding x 2 terms (oracle test: 5e-4 per term)
    impl.close()


def fixture_circle(make, load_golden, tag):
    """tests/golden/circle_reward.npz (multi_ant_circle.py:385-543 through a patched temp copy: INTENDED semantics, the reference cannot
    run this task -- tests/golden/make_circle_fixture.py lists the substitutions) through the MultiAntCircle step: two ants' states in,
    the 76-wide observation row, the ring reward, the reset flags and the position caches out."""
    g = load_golden("circle_reward")
    n = g["root_1"].shape[0]
    impl = make("MultiAntCircle", cfg=_cfg("MultiAntCircle"), num_envs=n)
    root = impl.get("root_states").reshape(n, 3, 13).copy()                  # (row 2: the engine's inert box, far away)
    root[:, 0], root[:, 1] = g["root_1"], g["root_2"]
    dof = np.stack([np.stack([g["dof_pos_1"], g["dof_vel_1"]], -1), np.stack([g["dof_pos_2"], g["dof_vel_2"]], -1)], 1).reshape(n * 16, 2)
    prev = np.concatenate([g["pos_before_1"], g["pos_before_2"]], 1).astype(np.float32)
    impl.put("root_states", root.reshape(n * 3, 13))
    impl.put("dof_state", dof.astype(np.float32))
    impl.put("prev", prev)
    impl.put("reset", np.zeros(n, np.int64))
    impl.put("progress", (g["progress"] - 1).astype(np.int64))
    impl.post_step(g["actions"])
    obs = impl.get("obs")
    ref = np.concatenate([g["obs1"], g["obs2"]], 1)
    ang = [9, 10, 11, 38 + 9, 38 + 10, 38 + 11]
    rest = [i for i in range(76) if i not in ang]
    e_rest = float(np.max(np.abs(obs[:, rest] - ref[:, rest])))
    ok = np.ones(n, bool)
    for q in (g["root_1"][:, 3:7], g["root_2"][:, 3:7]):
        ok &= np.abs(2.0 * (q[:, 3] * q[:, 1] - q[:, 2] * q[:, 0])) < 0.999
    e_ang = float(angle_err(obs[ok][:, ang], ref[ok][:, ang]).max())
    d_rew = np.abs(impl.get("rew") - g["rew"])
    # the ring reward is a step function of the position (+-3 per ant): rows where torch's atan2 / norm and the implementation's
    # land on different sides of a threshold are counted, not averaged in
    flips = int(np.sum(d_rew > 1e-3))
    e_rew = float(np.max(np.where(d_rew > 1e-3, 0.0, d_rew)))
    np.testing.assert_array_equal(impl.get("reset"), g["reset"])
    pv = impl.get("prev")
    np.testing.assert_array_equal(pv[:, 0:2], obs[:, 0:2])
    np.testing.assert_array_equal(pv[:, 2:4], obs[:, 38:40])
    record(tag + "fixture_circle", obs_max_abs=e_rest, angle=e_ang, reward_max_abs=e_rew, reward_threshold_flips=flips, rows=n)
    assert e_rest < 1e-4 and e_ang < 1e-4 and e_rew < 1e-4 and flips <= 2, (e_rest, e_ang, e_rew, flips)
    impl.close()


def fixture_ingenuity(make, load_golden, tag):
    """tests/golden/ingenuity_reward.npz (multi_ingenuity.py:381-453) through the epilogue, and ingenuity_thrust.npz
    (multi_ingenuity.py:268-339) through ONE PHYSICS SUBSTEP: a helicopter at rest in mid-air obeys m a_com = R f0 + R f1 + m g and
    the torque balance about its origin, so the applied rotor forces are read back from its velocities after the step."""
    g = load_golden("ingenuity_reward")
    n = g["roots"].shape[0]
    impl = make("MultiIngenuity", cfg=_cfg("MultiIngenuity"), num_envs=n)
    impl.put("root_states", g["roots"].reshape(n * 4, 13))
    impl.put("reset", np.zeros(n, np.int64))
    impl.put("progress", (g["progress"] - 1).astype(np.int64))
    impl.post_step(np.zeros((n, 24), np.float32))
    e = float(np.max(np.abs(impl.get("rew") - g["rew"])))
    np.testing.assert_array_equal(impl.get("reset"), g["reset"])
    np.testing.assert_array_equal(impl.get("obs"), g["roots"].reshape(n, 52))       # obs = the raw root rows (multi_ingenuity.py:351-357)
    record(tag + "fixture_ingenuity_reward", max_abs=e, rows=n)
    assert e < 1e-4, e
    impl.close()

    g = load_golden("ingenuity_thrust")
    n = g["actions"].shape[0]
    cfg = _cfg("MultiIngenuity")
    cfg["sim"]["substeps"] = 1
    cfg["sim"]["dt"] = float(g["dt"])
    impl = make("MultiIngenuity", cfg=cfg, num_envs=n)
    M = impl.config.model
    m, grav, cz = float(M.heli_mass), float(M.gravity), float(M.heli_com_z)
    z0, z1 = float(M.heli_rotor_z[0]), float(M.heli_rotor_z[1])
    ixx = float(M.heli_inertia[0])
    h = float(g["dt"])
    root = np.zeros((n, 4, 13), np.float32)
    root[:, :, 6] = 1.0
    root[:, :, 2] = 5.0                                            # mid-air: no ground contact
    worst = 0.0
    for mask in ((1, 0), (0, 1), (1, 1)):                         # rotor 0 alone, rotor 1 alone, both
        act = g["actions"].reshape(n, 4, 2, 3).copy()
        for r in range(2):
            if not mask[r]:
                act[:, :, r, :] = 0.0
        impl.put("root_states", root.reshape(n * 4, 13))
        impl.put("reset", np.zeros(n, np.int64))
        impl.put("progress", np.zeros(n, np.int64))
        impl.step(act.reshape(n, 24))
        out = impl.get("root_states").reshape(n, 4, 13).astype(np.float64)
        a_lin, alpha = out[..., 7:10] / h, out[..., 10:13] / h     # from rest: v = h a (origin), w = h alpha
        c = np.array([0.0, 0.0, cz])
        F = m * (a_lin + np.cross(alpha, c)) + np.array([0.0, 0.0, m * grav])     # total thrust (identity orientation)
        thr = np.clip(g["thrusts"].reshape(n, 4, 2, 3).astype(np.float64), None, None)
        want = sum(thr[:, :, r] * mask[r] for r in range(2))
        worst = max(worst, float(np.max(np.abs(F - want))))
        # torque about the COM: (x_r - c) x f_r summed = I alpha  ->  lateral force moments pin the split between the rotors
        tau = sum(np.cross(np.array([0.0, 0.0, (z0, z1)[r] - cz]), thr[:, :, r] * mask[r]) for r in range(2))
        worst_t = float(np.max(np.abs(ixx * alpha[..., :2] - tau[..., :2])))
        assert worst_t < 2e-4, ("thrust moment", mask, worst_t)
    record(tag + "fixture_ingenuity_thrust", force_max_abs=worst, rows=n, max_thrust=float(np.abs(g["thrusts"]).max()))
    assert worst < 2e-3, worst                                      # thrusts reach 33 N: 6e-5 relative
    impl.close()
