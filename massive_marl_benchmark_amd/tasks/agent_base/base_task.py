"""BaseTask: the step protocol and buffer set of agents/tasks/agent_base/base_task.py:24-149, with the
Isaac Gym calls replaced by one fused engine step (Engine.step = pre_physics_step + simulate +
post_physics_step).

Domain randomisation (base_task.py:216-410): the non-physical part -- observation and action noise lambdas,
gaussian / uniform, additive / scaling, with the linear / constant schedules and the correlated term
(base_task.py:246-316) -- is implemented with the same parameters and semantics.  Per-actor physical randomisation
(mass, damping, limits) and the viewer are out of scope this round."""
import operator

import numpy as np
import torch

from ...engine import Engine


class BaseTask:
    TASK_NAME = None

    def __init__(self, cfg, num_agents_default, clip_obs=5.0):
        clip_obs = float(cfg.get("clip_observations", clip_obs))   # the VecTask wrapper's clamp, fused into the kernel
        self.device_type = cfg.get("device_type", "cuda")
        self.device_id = cfg.get("device_id", 0)
        if self.device_type in ("cuda", "GPU"):
            self.device, engine_device = "cuda:" + str(self.device_id), self.device_id
        elif self.device_type == "cpu":
            # base_task.py:27-32: `--sim_device cpu` selects the CPU pipeline.  Here: the CPU build of the engine (lib/libmms_cpu.so,
            # the kernels' lane math compiled for the host) -- because the caller asked for it, never as a fallback.
            self.device, engine_device = "cpu", "cpu"
        else:
            raise RuntimeError("device_type=%r: expected 'cuda' / 'GPU' (the HIP engine) or 'cpu' (the CPU build)" % (self.device_type,))
        self.headless = cfg.get("headless", True)
        self.num_envs = cfg["env"]["numEnvs"]
        self.control_freq_inv = cfg["env"].get("controlFrequencyInv", 1)
        if self.control_freq_inv != 1:
            raise NotImplementedError("controlFrequencyInv != 1 (cfg/*.yaml ship 1)")
        self.engine = Engine(self.TASK_NAME, cfg, num_envs=self.num_envs, num_agents=num_agents_default,
                             device=engine_device, seed=max(int(cfg.get("seed", 0) or 0), 0),
                             env_offset=int(cfg.get("env_offset", 0)), total_envs=cfg.get("total_envs", None),
                             clip_obs=clip_obs, clip_actions=float(cfg["env"].get("clipActions", 1.0)))
        e = self.engine
        self.num_obs = cfg["env"]["numObservations"]
        self.num_states = cfg["env"].get("numStates", 0)
        self.num_actions = cfg["env"]["numActions"]
        # zero-copy views of engine memory (base_task.py:56-67 allocates torch buffers instead)
        self.obs_buf = e.tensor("obs")
        self.obs_buf_clipped = e.tensor("obs_clipped")
        self.states_buf = torch.zeros((self.num_envs, self.num_states), device=self.device, dtype=torch.float)
        self.rew_buf = e.tensor("rew")
        self.reset_buf = e.tensor("reset")
        self.progress_buf = e.tensor("progress")
        self.randomize_buf = torch.zeros(self.num_envs, device=self.device, dtype=torch.long)
        self.root_states = e.tensor("root_states")
        self.initial_root_states = e.tensor("initial_root_states")
        self.dof_state = e.tensor("dof_state")
        self.env_origin = e.tensor("env_origin")
        self._actions = e.tensor("actions")
        self.actions = self._actions
        self.extras = {}
        self.viewer = None
        self.dt = cfg["sim"]["dt"]
        # domain randomisation state (base_task.py:70-80)
        self.dr_randomizations = {}
        self.first_randomization = True
        self.last_step = -1
        self.last_rand_step = -1
        self.frame_count = 0
        self._engine_obs = self.obs_buf
        self._engine_obs_clipped = self.obs_buf_clipped
        self._clip_obs = clip_obs
        task_cfg = cfg.get("task", {}) or {}
        self.randomize = bool(task_cfg.get("randomize", False))
        self.randomization_params = task_cfg.get("randomization_params", {})
        if self.randomize:
            self.apply_randomizations(self.randomization_params)          # ten_ant.py:226-227

    def step(self, actions):
        """base_task.py:129-149.  `actions`: [num_envs, engine action width]; clamping happens in the kernel."""
        if self.dr_randomizations.get('actions', None):
            actions = self.dr_randomizations['actions']['noise_lambda'](actions)
        # the engine reads the caller's tensor where it lies (mms_bind_actions) when it can: fp32, contiguous, on the engine's device;
        # anything else is copied into the engine's own "actions" buffer (the reference clones it, ten_ant.py:887)
        a = actions
        if (a.data_ptr() != self._actions.data_ptr() and a.dtype == torch.float32 and a.is_contiguous() and a.device == self._actions.device
                and a.numel() == self._actions.numel() and a.data_ptr() % 8 == 0):
            self.engine.bind_actions(a)
            self.actions = a.view(self._actions.shape)
            self.actions_read_in_place = True                         # (this step's launch reads the caller's tensor; tests look at it)
        else:
            self.actions_read_in_place = False
            self.engine.bind_actions(None)
            if a.data_ptr() != self._actions.data_ptr():
                self._actions.copy_(a.reshape(self._actions.shape))
            self.actions = self._actions
        if self.randomize and bool(self.reset_buf.any()):
            self.apply_randomizations(self.randomization_params)          # reset_idx does this (ten_ant.py:812-813)
        self.engine.step()
        # the binding ends with the step: a later direct Engine.step() / post_step / reset_all, or a writer of engine.tensor("actions")
        # (ActorCritic.bind_rollout(storage, engine.tensor("actions"))), finds the engine reading its own buffer again.  `self.actions`
        # stays an ALIAS of the caller's tensor where the reference holds a clone (ten_ant.py:887): an in-place edit of that tensor
        # after step() shows in task.actions
        self.engine.bind_actions(None)
        self.frame_count += 1
        self.randomize_buf += 1
        if self.dr_randomizations.get('observations', None):
            self.obs_buf = self.dr_randomizations['observations']['noise_lambda'](self._engine_obs)
            self.obs_buf_clipped = torch.clamp(self.obs_buf, -self._clip_obs, self._clip_obs)
        else:
            self.obs_buf, self.obs_buf_clipped = self._engine_obs, self._engine_obs_clipped

    def apply_randomizations(self, dr_params):
        """base_task.py:216-410: the 'observations' / 'actions' noise lambdas (same parameters and schedules) and the
        'actor_params' physical parameters (see _randomize_actor_params); 'sim_params' (gravity) is not supported."""
        rand_freq = dr_params.get("frequency", 1)
        self.last_step = self.frame_count                                  # gym.get_frame_count
        if self.first_randomization:
            do_nonenv_randomize = True
            env_ids = torch.arange(self.num_envs, device=self.device)
        else:
            do_nonenv_randomize = (self.last_step - self.last_rand_step) >= rand_freq
            rand_envs = torch.logical_and(self.randomize_buf >= rand_freq, self.reset_buf.bool())
            env_ids = torch.nonzero(rand_envs, as_tuple=False).squeeze(-1)
            self.randomize_buf[rand_envs] = 0
        if do_nonenv_randomize:
            self.last_rand_step = self.last_step
        for name in ("observations", "actions"):
            if name in dr_params and do_nonenv_randomize:
                prm = dr_params[name]
                dist, op_type = prm["distribution"], prm["operation"]
                sched_type = prm.get("schedule", None)
                sched_step = prm.get("schedule_steps", None) if "schedule" in prm else None
                op = operator.add if op_type == 'additive' else operator.mul
                if sched_type == 'linear':
                    sched_scaling = 1.0 / sched_step * min(self.last_step, sched_step)
                elif sched_type == 'constant':
                    sched_scaling = 0 if self.last_step < sched_step else 1
                else:
                    sched_scaling = 1
                if dist == 'gaussian':
                    mu, var = prm["range"]
                    mu_corr, var_corr = prm.get("range_correlated", [0., 0.])
                    if op_type == 'additive':
                        mu, var, mu_corr, var_corr = (x * sched_scaling for x in (mu, var, mu_corr, var_corr))
                    elif op_type == 'scaling':
                        var = var * sched_scaling
                        mu = mu * sched_scaling + 1.0 * (1.0 - sched_scaling)
                        var_corr = var_corr * sched_scaling
                        mu_corr = mu_corr * sched_scaling + 1.0 * (1.0 - sched_scaling)

                    def noise_lambda(tensor, param_name=name, op=op):
                        params = self.dr_randomizations[param_name]
                        corr = params.get('corr', None)
                        if corr is None:
                            corr = torch.randn_like(tensor)
                            params['corr'] = corr
                        corr = corr * params['var_corr'] + params['mu_corr']
                        return op(tensor, corr + torch.randn_like(tensor) * params['var'] + params['mu'])

                    self.dr_randomizations[name] = {'mu': mu, 'var': var, 'mu_corr': mu_corr, 'var_corr': var_corr,
                                                    'noise_lambda': noise_lambda}
                elif dist == 'uniform':
                    lo, hi = prm["range"]
                    lo_corr, hi_corr = prm.get("range_correlated", [0., 0.])
                    if op_type == 'additive':
                        lo, hi, lo_corr, hi_corr = (x * sched_scaling for x in (lo, hi, lo_corr, hi_corr))
                    elif op_type == 'scaling':
                        lo, hi, lo_corr, hi_corr = (x * sched_scaling + 1.0 * (1.0 - sched_scaling) for x in (lo, hi, lo_corr, hi_corr))

                    def noise_lambda(tensor, param_name=name, op=op):
                        params = self.dr_randomizations[param_name]
                        corr = params.get('corr', None)
                        if corr is None:
                            corr = torch.randn_like(tensor)
                            params['corr'] = corr
                        corr = corr * (params['hi_corr'] - params['lo_corr']) + params['lo_corr']
                        return op(tensor, corr + torch.rand_like(tensor) * (params['hi'] - params['lo']) + params['lo'])

                    self.dr_randomizations[name] = {'lo': lo, 'hi': hi, 'lo_corr': lo_corr, 'hi_corr': hi_corr,
                                                    'noise_lambda': noise_lambda}
        if dr_params.get("actor_params") and env_ids.numel() > 0:
            self._randomize_actor_params(dr_params["actor_params"], env_ids)
        self.first_randomization = False

    # -- physical parameters (base_task.py:343-395) ------------------------------------------------------------------
    def _dr_sample(self, prm, shape):
        """One draw of isaacgym.gymutil.generate_random_samples / apply_random_samples (not in the reference tree; restated
        from its documented behaviour: 'uniform' / 'loguniform' over range = [lo, hi], 'gaussian' with range = [mean, std];
        a linear / constant schedule blends additive samples towards 0 and scaling samples towards 1)."""
        lo, hi = prm["range"]
        dist = prm.get("distribution", "uniform")
        if dist == "gaussian":
            x = np.random.normal(lo, hi, shape)
        elif dist == "loguniform":
            x = np.exp(np.random.uniform(np.log(lo), np.log(hi), shape))
        elif dist == "uniform":
            x = np.random.uniform(lo, hi, shape)
        else:
            raise ValueError("unknown randomisation distribution %r" % dist)
        sched, steps = prm.get("schedule", None), prm.get("schedule_steps", None)
        k = 1.0
        if sched == "linear":
            k = min(self.last_step, steps) / float(steps)
        elif sched == "constant":
            k = 0.0 if self.last_step < steps else 1.0
        if prm["operation"] == "scaling":
            return x * k + 1.0 * (1.0 - k)
        if prm["operation"] == "additive":
            return x * k
        raise ValueError("unknown randomisation operation %r" % prm["operation"])

    def _randomize_actor_params(self, actor_params, env_ids):
        """Fills the engine's per-ant parameter blocks for `env_ids` (include/mms.h: mms_set_dr).  Supported, as in
        cfg/TenAnt.yaml:97-122: rigid_body_properties.mass (scaling), dof_properties.damping (scaling) / lower / upper
        (additive); dof stiffness has no effect in effort mode (ten_ant.py:274) and colour is visual.  Actor key `ant` addresses
        every ant of the env (the reference looks the actor up by that name -- base_task.py:346 -- which exists in OneAnt only;
        TenAnt names its actors ant_1 .. ant_10, also accepted here as keys)."""
        if self.TASK_NAME == "MultiIngenuity":
            raise NotImplementedError("actor_params randomisation: the helicopter task has no randomised physical parameters")
        A = self.engine.num_agents
        dr = self.engine.tensor("dr_params").view(self.num_envs, A, -1)
        ids = env_ids.to(self.device)
        block = dr[ids].cpu().numpy()                                     # [n, A, 33]
        n = block.shape[0]
        for actor, props in actor_params.items():
            if actor == "ant":
                ants = list(range(A))
            elif actor.startswith("ant_") and actor[4:].isdigit() and 1 <= int(actor[4:]) <= A:
                ants = [int(actor[4:]) - 1]
            else:
                raise KeyError("actor_params: no actor named %r in this task" % actor)
            for prop_name, attrs in props.items():
                if prop_name == "color":
                    continue
                if prop_name not in ("rigid_body_properties", "dof_properties"):
                    raise NotImplementedError("actor_params.%s.%s is not a parameter of this engine" % (actor, prop_name))
                for attr, prm in attrs.items():
                    if prm.get("setup_only", False) and not self.first_randomization:
                        continue                                          # randomised once, before the simulation starts
                    if prop_name == "rigid_body_properties" and attr == "mass" and prm["operation"] == "scaling":
                        cols, count = slice(0, 9), 9
                    elif prop_name == "dof_properties" and attr == "damping" and prm["operation"] == "scaling":
                        cols, count = slice(9, 17), 8
                    elif prop_name == "dof_properties" and attr == "lower" and prm["operation"] == "additive":
                        cols, count = slice(17, 25), 8
                    elif prop_name == "dof_properties" and attr == "upper" and prm["operation"] == "additive":
                        cols, count = slice(25, 33), 8
                    elif prop_name == "dof_properties" and attr == "stiffness":
                        continue
                    else:
                        raise NotImplementedError("actor_params.%s.%s.%s (%s) is not supported" % (actor, prop_name, attr, prm["operation"]))
                    for a in ants:
                        block[:, a, cols] = self._dr_sample(prm, (n, count))
        dr[ids] = torch.from_numpy(block).to(self.device)
        self.engine.set_dr(True)

    def get_states(self):
        return self.states_buf

    def render(self, sync_frame_time=False):
        return None
