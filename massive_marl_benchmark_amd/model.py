"""Physical model constants and the `mms_config` / `mms_model` structs of include/mms.h.

The ant description restates the numbers of the reference asset
`assets/mjcf/open_ai_assets/ant/nv_ant.xml`: 9 bodies, 8 hinge DOF in
tree-DFS order hip_1, ankle_1, ..., hip_4, ankle_4; `inertiafromgeom`, density 5, angles in degrees.
`load_mjcf_ant()` is a small MJCF-subset compiler that derives the same description from any such
file; tests compare the two when the reference tree is present.  Masses and inertias are computed
here from the geometry (overlapping geoms counted, as the physics engines do).

Scene constants follow agents/tasks/ten_ant.py:339-358,491-495 (TenAnt), one_ant.py:234,264-266
(OneAnt), multi_ingenuity.py:124-164 and assets/.../ingenuity/ingenuity.xml (MultiIngenuity).
Contact / limit compliance values are this build's own model parameters;
the reference delegates them to PhysX.
"""
import ctypes
import math
import xml.etree.ElementTree as ET

MMS_ABI_VERSION = 4
TASK_IDS = {"TenAnt": 0, "OneAnt": 1, "MultiIngenuity": 2, "MultiAntCircle": 3}


class MmsModel(ctypes.Structure):
    _fields_ = [
        ("torso_mass", ctypes.c_float), ("torso_ixx", ctypes.c_float), ("torso_izz", ctypes.c_float),
        ("torso_radius", ctypes.c_float),
        ("leg_mass", ctypes.c_float), ("leg_ia", ctypes.c_float), ("leg_it", ctypes.c_float),
        ("foot_mass", ctypes.c_float), ("foot_ia", ctypes.c_float), ("foot_it", ctypes.c_float),
        ("limb_radius", ctypes.c_float), ("leg_len", ctypes.c_float), ("foot_len", ctypes.c_float),
        ("hip_pos", (ctypes.c_float * 3) * 4),
        ("limb_dir", (ctypes.c_float * 3) * 4),
        ("ankle_axis", (ctypes.c_float * 3) * 4),
        ("dof_lower", ctypes.c_float * 8), ("dof_upper", ctypes.c_float * 8),
        ("dof_init", ctypes.c_float * 8), ("gear", ctypes.c_float * 8),
        ("armature", ctypes.c_float), ("joint_damping", ctypes.c_float),
        ("limit_k", ctypes.c_float), ("limit_c", ctypes.c_float), ("limit_ramp", ctypes.c_float),
        ("gnd_k", ctypes.c_float), ("gnd_c", ctypes.c_float), ("gnd_mu", ctypes.c_float),
        ("slip_eps", ctypes.c_float), ("pen_ramp", ctypes.c_float),
        ("antbox_k", ctypes.c_float), ("antbox_c", ctypes.c_float), ("antbox_mu", ctypes.c_float),
        ("boxgnd_k", ctypes.c_float), ("boxgnd_c", ctypes.c_float), ("boxgnd_mu", ctypes.c_float),
        ("box_half", ctypes.c_float * 3), ("box_mass", ctypes.c_float), ("box_inertia", ctypes.c_float * 3),
        ("heli_mass", ctypes.c_float), ("heli_inertia", ctypes.c_float * 3), ("heli_com_z", ctypes.c_float),
        ("heli_rotor_z", ctypes.c_float * 2), ("heli_half", ctypes.c_float), ("heli_max_angvel", ctypes.c_float),
        ("heli_gnd_k", ctypes.c_float), ("heli_gnd_c", ctypes.c_float),
        ("gravity", ctypes.c_float),
    ]


class MmsConfig(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("task", ctypes.c_int32), ("num_envs", ctypes.c_int32),
        ("num_agents", ctypes.c_int32), ("device", ctypes.c_int32), ("substeps", ctypes.c_int32),
        ("max_episode_length", ctypes.c_int32), ("external_noise", ctypes.c_int32),
        ("env_offset", ctypes.c_int64), ("total_envs", ctypes.c_int64), ("seed", ctypes.c_uint64),
        ("env_spacing", ctypes.c_float), ("dt", ctypes.c_float),
        ("clip_actions", ctypes.c_float), ("clip_obs", ctypes.c_float),
        ("dof_vel_scale", ctypes.c_float), ("contact_force_scale", ctypes.c_float), ("power_scale", ctypes.c_float),
        ("heading_weight", ctypes.c_float), ("up_weight", ctypes.c_float), ("actions_cost", ctypes.c_float),
        ("energy_cost", ctypes.c_float), ("joints_at_limit_cost", ctypes.c_float),
        ("death_cost", ctypes.c_float), ("termination_height", ctypes.c_float),
        ("quat_reward_scale", ctypes.c_float), ("ant_dist_reward_scale", ctypes.c_float),
        ("goal_dist_reward_scale", ctypes.c_float),
        ("ant_start_x", ctypes.c_float), ("ant_start_z", ctypes.c_float),
        ("box_start", ctypes.c_float * 3),
        ("model", MmsModel),
    ]


class MmsTensor(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("shape", ctypes.c_int64 * 4), ("ndim", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("device", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class MmsPolicyHead(ctypes.Structure):
    """struct mms_policy_head (include/mms.h): the operands of mms_ppo_heads_act, bound to the next mms_step (mms_bind_policy_head)."""
    _fields_ = [("hidden", ctypes.c_void_p), ("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("vhidden", ctypes.c_void_p), ("vweight", ctypes.c_void_p), ("vbias", ctypes.c_void_p),
                ("log_std", ctypes.c_void_p), ("counters", ctypes.c_void_p),
                ("actions_out", ctypes.c_void_p), ("act_slot", ctypes.c_void_p), ("logp_slot", ctypes.c_void_p), ("value_slot", ctypes.c_void_p),
                ("mu_slot", ctypes.c_void_p), ("sigma_slot", ctypes.c_void_p),
                ("seed", ctypes.c_uint64), ("row_offset", ctypes.c_int64),
                ("H", ctypes.c_int32), ("VH", ctypes.c_int32), ("A", ctypes.c_int32), ("reference_scale", ctypes.c_int32),
                ("weight_tiles", ctypes.c_void_p)]


# ----------------------------------------------------------------------------------------------
# ant description (restated from nv_ant.xml; see module docstring)
# ----------------------------------------------------------------------------------------------
ANT_DENSITY = 5.0
ANT_FRICTION = 1.5        # <default><geom friction="1.5 0.1 0.1"> (nv_ant.xml:8)
BOX_FRICTION = 0.0        # shape_props[0].friction = 0. (ten_ant.py:548, one_ant.py:282)
ANT_DESCRIPTION = {
    "torso": {"sphere_radius": 0.25,
              "aux_capsules": [((0.0, 0.0, 0.0), (0.2, 0.2, 0.0)), ((0.0, 0.0, 0.0), (-0.2, 0.2, 0.0)),
                               ((0.0, 0.0, 0.0), (-0.2, -0.2, 0.0)), ((0.0, 0.0, 0.0), (0.2, -0.2, 0.0))],
              "capsule_radius": 0.08},
    # leg l: hip position in the torso frame, leg capsule end (= foot body origin) in the leg frame,
    # foot capsule end in the foot frame, ankle axis, joint ranges in degrees
    "legs": [
        {"hip_pos": (0.2, 0.2, 0.0), "leg_to": (0.2, 0.2, 0.0), "foot_to": (0.4, 0.4, 0.0),
         "hip_axis": (0, 0, 1), "hip_range": (-40, 40), "ankle_axis": (-1, 1, 0), "ankle_range": (30, 100)},
        {"hip_pos": (-0.2, 0.2, 0.0), "leg_to": (-0.2, 0.2, 0.0), "foot_to": (-0.4, 0.4, 0.0),
         "hip_axis": (0, 0, 1), "hip_range": (-40, 40), "ankle_axis": (1, 1, 0), "ankle_range": (-100, -30)},
        {"hip_pos": (-0.2, -0.2, 0.0), "leg_to": (-0.2, -0.2, 0.0), "foot_to": (-0.4, -0.4, 0.0),
         "hip_axis": (0, 0, 1), "hip_range": (-40, 40), "ankle_axis": (-1, 1, 0), "ankle_range": (-100, -30)},
        {"hip_pos": (0.2, -0.2, 0.0), "leg_to": (0.2, -0.2, 0.0), "foot_to": (0.4, -0.4, 0.0),
         "hip_axis": (0, 0, 1), "hip_range": (-40, 40), "ankle_axis": (1, 1, 0), "ankle_range": (30, 100)},
    ],
    "armature": 0.01, "damping": 0.1, "gear": 15.0, "density": ANT_DENSITY, "limb_radius": 0.08,
}


def _norm(v):
    n = math.sqrt(sum(x * x for x in v))
    return tuple(x / n for x in v)


def capsule_mass_inertia(length, radius, density):
    """Mass, axial and transverse inertia (about the COM) of a capsule whose segment has `length`."""
    mc = density * math.pi * radius * radius * length
    mh = density * 4.0 / 3.0 * math.pi * radius ** 3
    ia = mc * radius * radius / 2.0 + mh * 2.0 * radius * radius / 5.0
    it = mc * (length * length / 12.0 + radius * radius / 4.0) + \
        mh * (2.0 * radius * radius / 5.0 + length * length / 4.0 + 3.0 * length * radius / 8.0)
    return mc + mh, ia, it


def load_mjcf_ant(path):
    """MJCF-subset compiler: returns a description with the same keys as ANT_DESCRIPTION."""
    root = ET.parse(path).getroot()
    dj = root.find("default/joint")
    dg = root.find("default/geom")
    torso = root.find("worldbody/body")
    fl = lambda s: tuple(float(x) for x in s.split())
    desc = {"torso": {"aux_capsules": []}, "legs": [],
            "armature": float(dj.get("armature")), "damping": float(dj.get("damping")),
            "density": float(dg.get("density"))}
    for g in torso.findall("geom"):
        if g.get("type") == "sphere":
            desc["torso"]["sphere_radius"] = float(g.get("size"))
        else:
            ft = fl(g.get("fromto"))
            desc["torso"]["aux_capsules"].append((ft[:3], ft[3:]))
            desc["torso"]["capsule_radius"] = float(g.get("size"))
    for leg in torso.findall("body"):
        j1 = leg.find("joint")
        g1 = leg.find("geom")
        foot = leg.find("body")
        j2 = foot.find("joint")
        g2 = foot.find("geom")
        assert fl(foot.get("pos")) == fl(g1.get("fromto"))[3:]
        desc["legs"].append({
            "hip_pos": fl(leg.get("pos")), "leg_to": fl(g1.get("fromto"))[3:], "foot_to": fl(g2.get("fromto"))[3:],
            "hip_axis": tuple(int(x) for x in fl(j1.get("axis"))), "hip_range": tuple(int(x) for x in fl(j1.get("range"))),
            "ankle_axis": tuple(int(x) for x in fl(j2.get("axis"))),
            "ankle_range": tuple(int(x) for x in fl(j2.get("range")))})
        desc["limb_radius"] = float(g1.get("size"))
    gears = {float(m.get("gear")) for m in root.findall("actuator/motor")}
    assert len(gears) == 1
    desc["gear"] = gears.pop()
    return desc


def fill_ant(m, desc=ANT_DESCRIPTION):
    rho = desc["density"]
    r_t = desc["torso"]["sphere_radius"]
    r_c = desc["limb_radius"]
    m_sph = rho * 4.0 / 3.0 * math.pi * r_t ** 3
    ixx = izz = 0.4 * m_sph * r_t * r_t
    mass = m_sph
    for (p0, p1) in desc["torso"]["aux_capsules"]:
        d = tuple(b - a for a, b in zip(p0, p1))
        length = math.sqrt(sum(x * x for x in d))
        u = _norm(d)
        c = tuple((a + b) / 2.0 for a, b in zip(p0, p1))
        mc, ia, it = capsule_mass_inertia(length, desc["torso"]["capsule_radius"], rho)
        cc = sum(x * x for x in c)
        # Ic = it*1 + (ia-it) u u^T, shifted to the torso origin; products of inertia cancel by symmetry
        ixx += it + (ia - it) * u[0] * u[0] + mc * (cc - c[0] * c[0])
        izz += it + (ia - it) * u[2] * u[2] + mc * (cc - c[2] * c[2])
        mass += mc
    m.torso_mass, m.torso_ixx, m.torso_izz, m.torso_radius = mass, ixx, izz, r_t
    leg_len = math.sqrt(sum(x * x for x in desc["legs"][0]["leg_to"]))
    foot_len = math.sqrt(sum(x * x for x in desc["legs"][0]["foot_to"]))
    m.leg_mass, m.leg_ia, m.leg_it = capsule_mass_inertia(leg_len, r_c, rho)
    m.foot_mass, m.foot_ia, m.foot_it = capsule_mass_inertia(foot_len, r_c, rho)
    m.limb_radius, m.leg_len, m.foot_len = r_c, leg_len, foot_len
    for l, leg in enumerate(desc["legs"]):
        assert tuple(leg["hip_axis"]) == (0, 0, 1)
        u = _norm(leg["leg_to"])
        assert max(abs(a - b) for a, b in zip(u, _norm(leg["foot_to"]))) < 1e-12
        ax = _norm(leg["ankle_axis"])
        for i in range(3):
            m.hip_pos[l][i] = leg["hip_pos"][i]
            m.limb_dir[l][i] = u[i]
            m.ankle_axis[l][i] = ax[i]
        lo, hi = (math.radians(x) for x in leg["hip_range"])
        m.dof_lower[2 * l], m.dof_upper[2 * l] = min(lo, hi), max(lo, hi)
        lo, hi = (math.radians(x) for x in leg["ankle_range"])
        m.dof_lower[2 * l + 1], m.dof_upper[2 * l + 1] = min(lo, hi), max(lo, hi)
    for d in range(8):
        lo, hi = m.dof_lower[d], m.dof_upper[d]
        m.dof_init[d] = lo if lo > 0.0 else (hi if hi < 0.0 else 0.0)      # ten_ant.py:133-137
        m.gear[d] = desc["gear"]
    m.armature, m.joint_damping = desc["armature"], desc["damping"]


def build_model(task, num_agents, dt, substeps, gravity):
    m = MmsModel()
    fill_ant(m)
    h = dt / substeps
    # --- compliance parameters (this build's model) ---
    m.limit_k, m.limit_c, m.limit_ramp = 5000.0, 20.0, 5.0e-3
    m.gnd_k, m.gnd_c, m.gnd_mu, m.slip_eps, m.pen_ramp = 2.0e4, 300.0, 1.0, 1.0e-2, 5.0e-4
    if task == "OneAnt":
        bx, by, bz = 1.0, 1.0, 1.0                                         # one_ant.py:264
    else:
        bx, by, bz = 1.0, 2.8 * num_agents, 1.0                            # ten_ant.py:493 (28 m for 10 ants)
    mass = 1.0 * bx * by * bz                                              # density 1 (ten_ant.py:492)
    m.box_mass = mass
    m.box_half[0], m.box_half[1], m.box_half[2] = bx / 2, by / 2, bz / 2
    m.box_inertia[0] = mass / 12.0 * (by * by + bz * bz)
    m.box_inertia[1] = mass / 12.0 * (bx * bx + bz * bz)
    m.box_inertia[2] = mass / 12.0 * (bx * bx + by * by)
    # ant-box contacts are implicit on the ant side and explicit on the box side: keep k h^2 / m_box small
    m.antbox_k = min(1.0e4, 0.25 * mass / (h * h))
    m.antbox_c = min(100.0, 0.25 * mass / h)
    m.boxgnd_k = 3.0e3 * mass
    m.boxgnd_c = 60.0 * mass
    m.boxgnd_mu = m.antbox_mu = 0.0         # set by make_config from the materials and the combine rule
    # --- helicopter: chassis box 0.12^3 density 50 + two rotor discs r 0.15, half height 0.005, density 1000
    mc = 50.0 * 0.12 ** 3
    mr = 1000.0 * math.pi * 0.15 ** 2 * 0.01
    z = (0.0, 0.025)
    tot = mc + 2 * mr
    com_z = mr * (z[0] + z[1]) / tot
    i_ch = mc / 6.0 * 0.12 ** 2
    i_rt = mr * (3 * 0.15 ** 2 + 0.01 ** 2) / 12.0
    i_ra = mr * 0.15 ** 2 / 2.0
    ixx = i_ch + mc * com_z ** 2 + sum(i_rt + mr * (zz - com_z) ** 2 for zz in z)
    m.heli_mass = tot
    m.heli_inertia[0], m.heli_inertia[1], m.heli_inertia[2] = ixx, ixx, i_ch + 2 * i_ra
    m.heli_com_z = com_z
    m.heli_rotor_z[0], m.heli_rotor_z[1] = z
    m.heli_half = 0.06
    m.heli_max_angvel = 4.0 * math.pi                                      # multi_ingenuity.py:149
    m.heli_gnd_k, m.heli_gnd_c = 2.0e4, 300.0
    m.gravity = gravity
    return m


# ----------------------------------------------------------------------------------------------
# default task configuration: the values of cfg/TenAnt.yaml, cfg/OneAnt.yaml, cfg/MultiIngenuity.yaml
# (identical except env_name / envSpacing).  Users pass their own YAML through utils.config.load_cfg;
# tests check this dict against the reference YAML when the reference tree is present.
# ----------------------------------------------------------------------------------------------
def default_cfg(task):
    # (MultiAntCircle ships no YAML in the reference: the keys its constructor reads, multi_ant_circle.py:33-52, with TenAnt's values)
    name, spacing = {"TenAnt": ("ten_ant", 40), "OneAnt": ("one_ant", 5), "MultiIngenuity": ("multi_ingenuity", 2.5),
                     "MultiAntCircle": ("multi_ant_circle", 10)}[task]
    dr_prop = lambda rng, op, dist: {"range": rng, "operation": op, "distribution": dist}
    return {
        "env": {
            "env_name": name, "numEnvs": 128, "envSpacing": spacing, "episodeLength": 1000,
            "enableDebugVis": False, "cameraDebug": True, "pointCloudDebug": True, "aggregateMode": 1,
            "stiffnessScale": 1.0, "forceLimitScale": 1.0, "useRelativeControl": False, "dofSpeedScale": 20.0,
            "actionsMovingAverage": 1.0, "controlFrequencyInv": 1,
            "startPositionNoise": 0.01, "startRotationNoise": 0.0, "resetPositionNoise": 0.01,
            "resetRotationNoise": 0.0, "resetDofPosRandomInterval": 0.2, "resetDofVelRandomInterval": 0.0,
            "AgentIndex": "[[0]]",
            "distRewardScale": 50, "rotRewardScale": 1.0, "rotEps": 0.1, "actionPenaltyScale": -0.0002,
            "reachGoalBonus": 250, "fallDistance": 0.4, "fallPenalty": 0.0, "clipActions": 1.0, "powerScale": 1.0,
            "headingWeight": 0.5, "upWeight": 0.1, "actionsCost": 0.005, "energyCost": 0.05,
            "dofVelocityScale": 0.2, "contactForceScale": 0.1, "jointsAtLimitCost": 0.1, "deathCost": -2.0,
            "terminationHeight": 0.31,
            "plane": {"staticFriction": 1.0, "dynamicFriction": 1.0, "restitution": 0.0},
            "asset": {"assetRoot": "../assets", "assetFileName": "mjcf/open_ai_assets/ant/nv_ant.xml"},
        },
        "sim": {
            "dt": 0.0166, "substeps": 2, "up_axis": 2, "gravity": [0.0, 0.0, -9.81],
            "physx": {"num_threads": 4, "solver_type": 1, "num_position_iterations": 8, "num_velocity_iterations": 0,
                      "contact_offset": 0.002, "rest_offset": 0.0, "bounce_threshold_velocity": 0.2,
                      "max_depenetration_velocity": 1000.0, "default_buffer_size_multiplier": 5.0},
            "flex": {"num_outer_iterations": 5, "num_inner_iterations": 20, "warm_start": 0.8, "relaxation": 0.75},
        },
        "task": {
            "randomize": False,
            "randomization_params": {
                "frequency": 600,
                "observations": dr_prop([0, .002], "additive", "gaussian"),
                "actions": dr_prop([0., .02], "additive", "gaussian"),
                "actor_params": {"ant": {
                    "color": True,
                    "rigid_body_properties": {"mass": dict(dr_prop([0.5, 1.5], "scaling", "uniform"), setup_only=True)},
                    "dof_properties": {
                        "damping": dr_prop([0.5, 1.5], "scaling", "uniform"),
                        "stiffness": dr_prop([0.5, 1.5], "scaling", "uniform"),
                        "lower": dr_prop([0, 0.01], "additive", "gaussian"),
                        "upper": dr_prop([0, 0.01], "additive", "gaussian")}}},
            },
        },
    }


def make_config(task, cfg=None, num_envs=None, num_agents=None, device=0, seed=0, env_offset=0,
                total_envs=None, clip_obs=5.0, clip_actions=1.0, external_noise=False):
    """Build the mms_config for `task` from a cfg dict with the reference YAML's keys."""
    if cfg is None:
        cfg = default_cfg(task)
    env, sim = cfg["env"], cfg["sim"]
    n = int(num_envs if num_envs is not None else env["numEnvs"])
    if num_agents is None:
        num_agents = {"TenAnt": 10, "OneAnt": 1, "MultiIngenuity": 4, "MultiAntCircle": 2}[task]
    c = MmsConfig()
    c.abi_version = MMS_ABI_VERSION
    c.task = TASK_IDS[task]
    c.num_envs, c.num_agents, c.device = n, int(num_agents), int(device)
    c.substeps = int(sim.get("substeps", 2))
    c.max_episode_length = int(env["episodeLength"])
    c.external_noise = 1 if external_noise else 0
    c.env_offset = int(env_offset)
    c.total_envs = int(total_envs if total_envs is not None else n)
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    c.env_spacing = float(env["envSpacing"])
    c.dt = float(sim["dt"])
    c.clip_actions, c.clip_obs = float(clip_actions), float(clip_obs)
    c.dof_vel_scale = float(env["dofVelocityScale"])
    c.contact_force_scale = float(env["contactForceScale"])
    c.power_scale = float(env["powerScale"])
    c.heading_weight, c.up_weight = float(env["headingWeight"]), float(env["upWeight"])
    c.actions_cost, c.energy_cost = float(env["actionsCost"]), float(env["energyCost"])
    c.joints_at_limit_cost = float(env["jointsAtLimitCost"])
    c.death_cost, c.termination_height = float(env["deathCost"]), float(env["terminationHeight"])
    # ten_ant.py:55-59 / one_ant.py:56-60
    c.quat_reward_scale = 1.0 if task == "OneAnt" else 0.0
    c.ant_dist_reward_scale = c.goal_dist_reward_scale = 500.0
    if task == "MultiAntCircle":
        # two ants at (+-3, 0, 1) (multi_ant_circle.py:216-219); the scene has no box: the engine's box actor rests far outside any
        # ant's reach (env-local y = 1000 m; collisions are per env) and never enters an observation or the reward
        c.ant_start_x, c.ant_start_z = 3.0, 1.0
        c.box_start[0], c.box_start[1], c.box_start[2] = 0.0, 1000.0, 0.5
    elif task == "OneAnt":
        c.ant_start_x, c.ant_start_z = -6.0, 1.0
        c.box_start[0], c.box_start[1], c.box_start[2] = -4.0, 0.0, 1.0
    else:
        c.ant_start_x, c.ant_start_z = 6.0, 1.0
        c.box_start[0], c.box_start[1], c.box_start[2] = 4.0, 0.0, 1.0
    gravity = 3.721 if task == "MultiIngenuity" else -float(sim.get("gravity", [0, 0, -9.81])[2])
    c.model = build_model(task, int(num_agents), c.dt, c.substeps, gravity)
    # Contact friction = combine(material a, material b).  Materials: the ant's geoms 1.5 (nv_ant.xml:8, the MJCF default geom
    # friction), the ground plane cfg env.plane.dynamicFriction (ten_ant.py:233-238), the box 0 (ten_ant.py:547-551,
    # one_ant.py:281-285), the helicopter 1.0.  Combine rule: PhysX's documented default is the AVERAGE of the two materials --
    # ant-ground 1.25, box-ground 0.5, ant-box 0.75 -- and the reference's own OneAnt training log is only consistent with a box
    # that stops when it is no longer pushed (tools/ref_logs.py).  `env.frictionCombine: "min"` (not a key
    # of the reference YAML) gives the other reading: a box that is frictionless against everything, ant-ground 1.0.
    # `env.boxGroundFriction` (not a reference key either) overrides the box-ground value alone.
    rule = str(env.get("frictionCombine", "average"))
    if rule not in ("average", "min"):
        raise ValueError("env.frictionCombine must be 'average' or 'min', got %r" % rule)
    comb = (lambda a, b: 0.5 * (a + b)) if rule == "average" else min
    plane_mu = float(env.get("plane", {}).get("dynamicFriction", 1.0))
    body_mu = 1.0 if task == "MultiIngenuity" else ANT_FRICTION
    c.model.gnd_mu = comb(body_mu, plane_mu)
    c.model.boxgnd_mu = float(env.get("boxGroundFriction", comb(BOX_FRICTION, plane_mu)))
    c.model.antbox_mu = comb(ANT_FRICTION, BOX_FRICTION)
    return c


def task_dims(task, num_agents):
    """(actors_per_env, dofs_per_env, num_actions, obs_dim, prev_dim)"""
    a = num_agents
    if task == "TenAnt":
        return a + 1, 8 * a, 8 * a, 38 * a + 8, 4 * a + 2
    if task == "OneAnt":
        return 2, 8, 8, 60, 6
    if task == "MultiAntCircle":
        return a + 1, 8 * a, 8 * a, 38 * a, 2 * a
    return a, 4 * a, 6 * a, 13 * a, 3 * a
