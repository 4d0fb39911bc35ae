"""MultiAntCircle task (agents/tasks/multi_ant_circle.py): two ants per env at (+-3, 0, 1), no box in the scene, 38 observation
entries per ant (TenAnt's), reward for walking round the r = 3 ring about the GLOBAL origin (:400-502).

INTENDED SEMANTICS, parity unpinned: the reference cannot import or construct this task -- numpy calls and bool arithmetic inside
its @torch.jit.script functions (:385-398, :423-433), a 19-argument call of a 16-parameter function (:298-318), not registered in
utils/parse_task.py:8-10, no cfg/MultiAntCircle.yaml, an asset path that does not exist (:178).  What runs here is what those
functions say once they are made to run, with the substitutions recorded in tests/golden/make_circle_fixture.py; cfg defaults
(model.default_cfg("MultiAntCircle")) are the keys the constructor reads (:33-52) with TenAnt's values.

Constructor signature and public attributes follow the reference; the scene, physics, reset_idx, compute_observations and
compute_reward run inside the fused HIP step kernel (ant_step_kernel<MMS_TASK_MULTI_ANT_CIRCLE, ...>).  The engine keeps an inert box
actor far from the ants (its ant kernels' lane layouts carry box lanes): `root_states` is [N * 3, 13] with the reference's two ant rows
first -- `ant_root_states` is the [N, 2, 13] view of them."""
from .agent_base.base_task import BaseTask


class MultiAntCircle(BaseTask):
    TASK_NAME = "MultiAntCircle"

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True,
                 is_multi_agent=False, strict_reference_spaces=False):
        self.cfg = cfg
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.is_multi_agent = is_multi_agent
        self.max_episode_length = cfg["env"]["episodeLength"]
        # multi_ant_circle.py:54 declares 38 although obs_buf is the concatenation of both ants' rows (:341): the real width is exposed
        # unless strict_reference_spaces is requested (as for TenAnt)
        if is_multi_agent:
            self.num_agents = 2
            cfg["env"]["numActions"] = 8                     # multi_ant_circle.py:62-64
            cfg["env"]["numObservations"] = 38
        else:
            self.num_agents = 1
            cfg["env"]["numActions"] = 16                    # multi_ant_circle.py:66-68
            cfg["env"]["numObservations"] = 38 if strict_reference_spaces else 76
        cfg["device_type"], cfg["device_id"], cfg["headless"] = device_type, device_id, headless
        super().__init__(cfg, num_agents_default=2)
        self.num_dof = 8
        n = self.num_envs
        self.ant_root_states = self.root_states.view(n, 3, 13)[:, :2]
        self.dof_pos = self.dof_state.view(n, -1, 2)[..., 0]
        self.dof_vel = self.dof_state.view(n, -1, 2)[..., 1]
        for k in range(2):                                   # multi_ant_circle.py:108-113, 132-133
            setattr(self, "dof_pos_%d" % (k + 1), self.dof_pos[:, 8 * k:8 * k + 8])
            setattr(self, "dof_vel_%d" % (k + 1), self.dof_vel[:, 8 * k:8 * k + 8])
            setattr(self, "obs_buf_%d" % (k + 1), self.obs_buf[:, 38 * k:38 * k + 38])
        self.prev = self.engine.tensor("prev")               # pos_before_1, pos_before_2 (:367-368, :382-383)
