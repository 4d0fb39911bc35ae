/*
 * mms.h -- C ABI of the MI355X-native multi-agent physics + rollout engine ("mms").
 *
 * This is the INNER drop-in boundary: what the reference's task files obtain
 * today from the closed Isaac Gym binary through `self.gym.*` calls.  The reference has no FFI of
 * its own for this path (it is Python calling a vendor binary), so each entry point cites the
 * reference call sites it replaces.  Plain pointers and sizes only; no torch types.
 *
 *   reference call site (under /root/reference)                         replaced by
 *   ------------------------------------------------------------------  -----------------------
 *   gymapi.acquire_gym / create_sim / add_ground / create_env /          mms_create
 *     load_asset / create_box / create_actor / prepare_sim
 *     (agents/tasks/agent_base/base_task.py:25,83,122;
 *      agents/tasks/ten_ant.py:205-633; one_ant.py:150-312;
 *      multi_ingenuity.py:120-226)
 *   acquire_actor_root_state_tensor / acquire_dof_state_tensor /         mms_get_tensor
 *     acquire_force_sensor_tensor + gymtorch.wrap_tensor
 *     (ten_ant.py:84-104; one_ant.py:78-103; multi_ingenuity.py:77-95)
 *   pre_physics_step + gym.simulate + fetch_results +                    mms_step
 *     post_physics_step (refresh_*, reset_idx, compute_observations,
 *     compute_reward)  (base_task.py:129-149; ten_ant.py:886-926)
 *   set_actor_root_state_tensor_indexed / set_dof_state_tensor_indexed   mms_set_state (tests),
 *     (ten_ant.py:856-868)                                               in-kernel reset
 *   RolloutStorage.compute_returns (algorithms/rl/ppo/storage.py:51-65)  mms_gae_ppo
 *   SeparatedReplayBuffer.compute_returns                                mms_gae_marl
 *     (algorithms/marl/utils/separated_buffer.py:153-164)
 *   MultiVecTaskPython.step slicing (agent_base/multi_vec_task.py:       mms_marl_views
 *     105-142)
 *   apply_randomizations actor_params: set_actor_rigid_body_properties   mms_set_dr + "dr_params"
 *     / set_actor_dof_properties (base_task.py:343-395)
 *   ActorCritic.act sampling tail + RolloutStorage.add_transitions       mms_ppo_act, mms_ppo_heads_act,
 *     (algorithms/rl/ppo/module.py:73-87; storage.py:33-47)              mms_bind_rollout_out
 *   ActorCritic hidden layers (module.py:27-52)                          mms_linear2_act, mms_linear_group_act_split,
 *                                                                        mms_split_planes(_group), mms_split_planes16_group,
 *                                                                        mms_linear_group_act_split16, mms_row_stats_chan_group,
 *                                                                        mms_marl_heads_finish
 *   Actor / Critic forward of every MAPPO / HAPPO agent                  mms_linear_group_act, mms_layernorm_group,
 *     (algorithms/marl/actor_critic.py:43-69, 137-155; runner.py:186-216)  mms_row_stats_group, mms_marl_heads_act
 *
 * Ownership: the engine owns every buffer it reports through mms_get_tensor for the lifetime of the
 * handle; callers wrap them as NON-owning views and must keep the handle alive while any view exists.
 * Threading: a handle is not thread-safe; one handle per process / GPU.  All work is enqueued on the
 * caller's HIP stream; no entry point synchronises the device except mms_create / mms_destroy.
 * Status: 0 = ok; non-zero = error, text from mms_last_error().
 * There is NO CPU fallback: mms_create of libmms.so fails when no HIP device is usable.  A SEPARATE library, libmms_cpu.so, exports
 * the same symbols for the reference's `--sim_device cpu` pipeline (base_task.py:27-32): the same lane math compiled for the host
 * (csrc/cpu/), selected only by device = -1 / device_type = "cpu", never automatically.  Device arguments of the free functions
 * (mms_gae_*, mms_ppo_*, mms_linear2_act, mms_marl_views) follow the same rule; streams are ignored by the CPU build.
 * Every entry point leaves the caller's current HIP device unchanged.
 */
#ifndef MMS_H
#define MMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMS_ABI_VERSION 4
#define MMS_DR_FLOATS 33       /* per-ant physical domain-randomisation block, see mms_set_dr */

enum mms_task { MMS_TASK_TEN_ANT = 0, MMS_TASK_ONE_ANT = 1, MMS_TASK_MULTI_INGENUITY = 2,
                /* agents/tasks/multi_ant_circle.py: two ants per env at (+-3, 0, 1), no box in the scene, 38 observation entries per ant
                 * (the same as TenAnt's), reward for walking round the r = 3 ring.  The reference cannot import or construct this task
                 * (numpy calls and bool arithmetic inside @torch.jit.script, a 19-argument call of a 16-parameter function, not
                 * registered, no YAML): what is built is its INTENDED semantics, pinned by fixtures from a patched temp copy
                 * (tests/golden/make_circle_fixture.py).  The engine keeps an inert box actor far from the ants (the ant kernels'
                 * layouts carry box lanes): root_states is [N * 3, 13], the reference's two ant rows first. */
                MMS_TASK_MULTI_ANT_CIRCLE = 3 };
enum mms_dtype { MMS_F32 = 0, MMS_I64 = 1, MMS_I32 = 2, MMS_U8 = 3 };

/* Physical model (numbers are produced by massive_marl_benchmark_amd/model.py
 * from the MJCF geometry).  All SI units, float32. */
typedef struct mms_model {
    /* ant: torso + 4 x (leg, foot); body frames are aligned at q = 0 */
    float torso_mass, torso_ixx, torso_izz, torso_radius;
    float leg_mass, leg_ia, leg_it;      /* capsule axial / transverse inertia about its COM */
    float foot_mass, foot_ia, foot_it;
    float limb_radius, leg_len, foot_len; /* capsule radius and segment lengths */
    float hip_pos[4][3];                 /* hip joint position in the torso frame */
    float limb_dir[4][3];                /* unit capsule direction of leg and foot at q = 0 */
    float ankle_axis[4][3];              /* unit ankle axis in the leg frame */
    float dof_lower[8], dof_upper[8], dof_init[8], gear[8];
    float armature, joint_damping, limit_k, limit_c, limit_ramp;
    /* contact (compliant, linearly-implicit) */
    float gnd_k, gnd_c, gnd_mu, slip_eps, pen_ramp;
    float antbox_k, antbox_c;
    float antbox_mu;                     /* ant-box Coulomb friction (regularised like the ground's); 0 = frictionless contact */
    float boxgnd_k, boxgnd_c;
    float boxgnd_mu;                     /* box-ground Coulomb friction; 0 = frictionless */
    /* box */
    float box_half[3], box_mass, box_inertia[3];
    /* helicopter (MultiIngenuity): one rigid body */
    float heli_mass, heli_inertia[3], heli_com_z, heli_rotor_z[2], heli_half, heli_max_angvel;
    float heli_gnd_k, heli_gnd_c;
    float gravity;                       /* positive magnitude along -z */
} mms_model;

typedef struct mms_config {
    int32_t abi_version;                 /* MMS_ABI_VERSION */
    int32_t task;                        /* enum mms_task */
    int32_t num_envs;                    /* envs owned by this handle (this GPU's shard) */
    int32_t num_agents;                  /* ants (10 / 1) or helicopters (4) per env */
    int32_t device;                      /* HIP device ordinal (>= 0) for libmms.so; -1 for libmms_cpu.so, the explicit opt-in CPU build */
    int32_t substeps;                    /* physics substeps per control step (cfg sim.substeps = 2) */
    int32_t max_episode_length;          /* cfg env.episodeLength */
    int32_t external_noise;              /* 1: reset noise is read from the "reset_noise" tensor */
    int64_t env_offset;                  /* global index of local env 0 (multi-GPU sharding) */
    int64_t total_envs;                  /* global env count: env-grid row length = (int)sqrt(total) */
    uint64_t seed;
    float env_spacing;                   /* cfg env.envSpacing */
    float dt;                            /* cfg sim.dt */
    float clip_actions, clip_obs;        /* wrapper clamps (vec_task.py:18,127,131) */
    float dof_vel_scale, contact_force_scale, power_scale;
    float heading_weight, up_weight, actions_cost, energy_cost, joints_at_limit_cost;
    float death_cost, termination_height;
    float quat_reward_scale, ant_dist_reward_scale, goal_dist_reward_scale;
    float ant_start_x, ant_start_z;      /* ant k starts at (x, s_k*(1.5+3*(k/2)), z), s = -,+,-,+ ... */
    float box_start[3];
    mms_model model;
} mms_config;

typedef struct mms_tensor {
    void*   ptr;                         /* device pointer */
    int64_t shape[4];
    int32_t ndim;
    int32_t dtype;                       /* enum mms_dtype */
    int32_t device;                      /* HIP device ordinal */
    int32_t reserved;
} mms_tensor;

typedef struct mms_engine* mms_handle;

/* Builds the scene for cfg->num_envs environments on cfg->device and allocates all state.
 * Replaces create_sim ... prepare_sim (see table above). */
int mms_create(const mms_config* cfg, mms_handle* out);
int mms_destroy(mms_handle h);

/* Named buffers.  Common: "actions" [N,num_actions] f32 (input of mms_step), "obs" [N,obs_dim] f32,
 * "obs_clipped" [N,obs_dim] f32 (clamped to +-clip_obs), "rew" [N] f32, "reset" [N] i64,
 * "progress" [N] i64, "root_states" [N*actors,13] f32 (env-local frame), "dof_state" [N*dofs,2] f32,
 * "env_origin" [N,3] f32, "prev" [N,prev_dim] f32 (pos_before / goal_before / box_before caches),
 * "reset_noise" [N,16] f32, "foot_sensors" [N*A,24] f32, "initial_root_states" [N*actors,13] f32,
 * "dr_params" [N*A,MMS_DR_FLOATS] f32 (see mms_set_dr),
 * "reset_count" [N] i64 (number of resets of each env so far: the counter of the reset-noise RNG, keyed with
 * the seed and the GLOBAL env index, so results do not depend on how envs are sharded over GPUs). */
int mms_get_tensor(mms_handle h, const char* name, mms_tensor* out);

/* One VecTask step: clamp actions, physics substeps, progress += 1, reset flagged envs,
 * observations, reward / reset flags, cache update -- one fused launch, stream ordered.
 * Replaces BaseTask.step (base_task.py:129-149). */
int mms_step(mms_handle h, void* hip_stream);

/* Same protocol with the physics skipped: the post-physics glue applied to whatever state the
 * buffers hold.  Used by the parity tests that supply state per step (fixture tenant_step_glue). */
int mms_post_step(mms_handle h, void* hip_stream);

/* Flags every env for reset (reset_buf := 1), as at construction (base_task.py:62-63). */
int mms_reset_all(mms_handle h, void* hip_stream);

/* Copies caller data (device or host pointer, `src_is_host`) into a named buffer.  env_ids == NULL
 * copies the whole buffer; otherwise `src` holds n rows (one per listed env) that are scattered.
 * Tests / fixtures only. */
int mms_set_state(mms_handle h, const char* name, const void* src, int src_is_host,
                  const int64_t* env_ids, int64_t n, void* hip_stream);

/* Optional extra destination for the clamped observation row, e.g. slot t of a rollout buffer
 * [T,N,obs_dim]; NULL disables.  The pointer must stay valid until the next bind. */
int mms_bind_obs_out(mms_handle h, void* dst);

/* Optional extra destination for the clamped observation row AS THE POLICY LAYERS' OPERAND PLANES (format H32 of
 * mms_linear_group_act_split16 below: f16 [N, ceil(obs_dim / 32), 2, 32], MMS_H32_BYTES(N, obs_dim) bytes, 16-byte aligned): hi and lo
 * planes of row * scale.  The row is bounded by clip_observations, so the scale is a constant (a power of two, clip_observations * scale
 * <= 2^14) and the policy needs no split pass over the observation (x_inv = 1 / scale for every row).  NULL disables.  Ant tasks only. */
int mms_bind_obs_planes16(mms_handle h, void* planes, float scale);

/* Optional source of the actions: mms_step reads `src` (f32 [N, num_actions], on the engine's device) IN PLACE instead of the
 * engine's own "actions" buffer -- the tensor VecTaskPython.step(actions) was handed (vec_task.py:126-131 clamps and copies it into
 * the task; here the clamp is in the kernel and the copy is gone).  NULL returns to the "actions" buffer.  The pointer must stay
 * valid until the next bind. */
int mms_bind_actions(mms_handle h, const float* src);

/* Physical domain randomisation of the ants (cfg/TenAnt.yaml:97-122 actor_params, applied by base_task.py:343-395 through
 * set_actor_rigid_body_properties / set_actor_dof_properties).  The caller fills "dr_params" [N*A, MMS_DR_FLOATS] per ant:
 *   [0] torso, [1..4] leg, [5..8] foot mass scale (the inertia scales with the mass: recomputeInertia is the setter's default),
 *   [9..16] joint damping scale, [17..24] lower-limit offset, [25..32] upper-limit offset (rad), joints in DOF order;
 * mms_set_dr(h, 1) makes mms_step use it (a separate kernel instantiation: the nominal path carries none of it).  The
 * observations keep the nominal limits, as the reference's do (ten_ant.py:588-599 reads them once at construction).
 * DOF stiffness is not a parameter of this path: the tasks drive the joints in effort mode (ten_ant.py:274). */
int mms_set_dr(mms_handle h, int32_t enable);

/* Which of the engine-owned observation rows mms_step writes: "obs" (raw, = task.obs_buf of the reference) and "obs_clipped"
 * (what VecTaskPython.step returns, vec_task.py:131).  Both on by default.  A rollout that binds a slot with mms_bind_obs_out
 * needs neither while the slot is bound: one 1552-B row per env-step instead of three.  A row that is switched off keeps its
 * last contents. */
int mms_set_obs_outputs(mms_handle h, int32_t raw, int32_t clipped);

/* Optional extra destinations for the step's reward (f32 [N]) and done flag (u8 [N], = reset_buf after the step), e.g.
 * RolloutStorage.rewards[t] / dones[t] (storage.py:40-41): add_transitions then has nothing left to copy for them.
 * NULL disables either.  Pointers must stay valid until the next bind. */
int mms_bind_rollout_out(mms_handle h, float* rew_out, uint8_t* done_out);

/* MARL wrapper views (multi_vec_task.py:105-142): obs_all [N,A,per_agent+shared] from a clamped
 * observation buffer [N, A*per_agent+shared]. */
int mms_marl_views(int device, const float* obs_clipped, float* obs_all, int64_t n, int32_t agents,
                   int32_t per_agent, int32_t shared, void* hip_stream);

/* PPO GAE (storage.py:51-65).  rewards/values/returns/advantages are [T,N] f32, dones [T,N] u8,
 * last_values [N].  Writes returns and UN-normalised advantages, and stats[0..2] =
 * {sum(adv), sum(adv^2), count} as float64 so that ranks can all-reduce them.  */
int mms_gae_ppo(int device, const float* rewards, const uint8_t* dones, const float* values,
                const float* last_values, float* returns, float* advantages, double* stats,
                int32_t T, int64_t N, float gamma, float lam, void* hip_stream);
/* advantages := (advantages - mean) / (std + 1e-8) with the unbiased std from stats. */
int mms_adv_normalize(int device, float* advantages, const double* stats, int64_t count, void* hip_stream);
/* mms_gae_ppo + mms_adv_normalize for ONE rank (nothing to all-reduce in between), the whole of storage.py:51-65: the scan leaves
 * per-block partial sums instead of float64 atomics and the normalisation sums them in a fixed order -- bit-reproducible, and stats
 * needs no zeroing (the atomics' 24-byte memset costs two fill kernels inside a captured rollout).  stats: f64 [MMS_GAE_STATS_DOUBLES]
 * = {sum(adv), sum(adv^2), count} of the un-normalised advantages, then scratch for the partials. */
#define MMS_GAE_STATS_DOUBLES (3 + 2 * 2048)
int mms_gae_ppo_normalized(int device, const float* rewards, const uint8_t* dones, const float* values, const float* last_values,
                           float* returns, float* advantages, double* stats, int32_t T, int64_t N, float gamma, float lam,
                           void* hip_stream);

/* Diagnostics: the clock the chip holds inside the two-plane layer kernel (mms_linear_group_act_split16).  With `out` (device memory,
 * 2 x slots uint64, 8-byte aligned) every such launch issued after this call stores, by its workgroup 0, {shader cycles, 100-MHz ticks}
 * of that workgroup's life into out[2 (n % slots)], n = 0, 1, ... counting launches from this call: cycles / ticks x 0.1 = GHz.  The
 * reference has no counterpart (torch modules, agents/algorithms/rl/ppo/module.py:27-52); bench.py reports the figure beside
 * `roofline_policy_layers` because the kernel runs power-limited: its MFMA rate has to be read against the clock it is granted, not
 * against the 2.4 GHz the dense peak is quoted at.  out = NULL switches the probe off (the default).  Process-wide, not thread-safe;
 * launches captured in a hipGraph keep the slot they were captured with.  The CPU build accepts the call and stores nothing. */
int mms_layer_clock_probe(int device, uint64_t* out, int32_t slots);

/* MARL GAE (separated_buffer.py:153-164, use_proper_time_limits=False): value_preds [T+1,N]
 * (row T already holds next_value), masks [T+1,N], rewards [T,N], returns [T+1,N];
 * denormalisation x*sqrt(var)+mean when use_norm (PopArt / ValueNorm). */
int mms_gae_marl(int device, const float* rewards, const float* value_preds, const float* masks,
                 float* returns, int32_t T, int64_t N, float gamma, float lam,
                 int32_t use_norm, const float* norm_mean, const float* norm_var, void* hip_stream);

/* The same for all A agents of all envs in one launch (rollout-buffer fusion):
 * value_preds / returns [T+1,N,A] (agent fastest), rewards [T,N] and masks [T+1,N] stored once per env instead of
 * once per agent buffer (runner.py:250-255 inserts the same reward / mask into ten buffers); norm_mean / norm_var [A]. */
int mms_gae_marl_agents(int device, const float* rewards, const float* value_preds, const float* masks,
                        float* returns, int32_t T, int64_t N, int32_t A, float gamma, float lam,
                        int32_t use_norm, const float* norm_mean, const float* norm_var, void* hip_stream);

/* The tail of ActorCritic.act (algorithms/rl/ppo/module.py:73-87) fused with RolloutStorage.add_transitions
 * (storage.py:33-47): Gaussian sample, log-probability and the stores of one rollout step in ONE launch
 * (SURVEY.md section 8f item 4).  mean [N,A] and value [N] are the outputs of the actor / critic MLPs.
 *   noise_ij ~ N(0,1): counter-based (seed, global row = row_offset + i, counters[i], j), Box-Muller; counters[i] += 1
 *   scale_j  = exp(log_std_j)^2 when reference_scale != 0 -- module.py:76-77 hands diag(sigma^2) to scale_tril -- else exp(log_std_j)
 *   action   = mean + scale * noise;   log_prob_i = sum_j (-0.5 noise_ij^2 - log(scale_j) - 0.5 log(2 pi))
 * Destinations (any may be NULL): actions_out [N,A] (e.g. the engine's "actions" buffer), act_slot / mu_slot / sigma_slot
 * [N,A], logp_slot / value_slot [N].  sigma_slot receives log_std broadcast, which is what module.py:87 returns as sigma.
 * counters is a device array [N] of int64 so that a captured hipGraph draws fresh noise on every replay. */
int mms_ppo_act(int device, const float* mean, const float* value, const float* log_std, uint64_t seed, int64_t* counters,
                int64_t row_offset, int32_t reference_scale, float* actions_out, float* act_slot, float* logp_slot,
                float* value_slot, float* mu_slot, float* sigma_slot, int64_t N, int32_t A, void* hip_stream);

/* mms_ppo_act with the last Linear layers of both networks folded in.  Actor (module.py:29-30: nn.Linear(pi_hid_sizes[-1],
 * actions)): mean = hidden @ weight^T + bias on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products and sums), then the
 * same sampling and stores; hidden [N,H] f32 is the output of the last activation, weight [A,H] and bias [A] are torch's Linear
 * parameters; H must be a multiple of 64, A <= 128.  Critic (module.py:49: nn.Linear(vf_hid_sizes[-1], 1)) when vhidden is given:
 * value_i = vhidden[i, :] . vweight + vbias[0] (vhidden [N,VH] f32 = output of the critic's last activation, VH a multiple of 4)
 * goes to value_slot and `value` is ignored; vhidden = NULL: `value` [N] (or NULL) is stored as in mms_ppo_act. */
int mms_ppo_heads_act(int device, const float* hidden, const float* weight, const float* bias, int32_t H, const float* value,
                      const float* vhidden, const float* vweight, const float* vbias, int32_t VH, const float* log_std, uint64_t seed,
                      int64_t* counters, int64_t row_offset, int32_t reference_scale, float* actions_out, float* act_slot,
                      float* logp_slot, float* value_slot, float* mu_slot, float* sigma_slot, int64_t N, int32_t A, void* hip_stream);

/* mms_ppo_heads_act FUSED INTO THE STEP: with a head bound, the next mms_step evaluates the policy's output heads and the sampling tail
 * for its own 16 envs per workgroup in the step kernel's prologue -- the arithmetic of mms_ppo_heads_act, instruction for instruction
 * (csrc/head_block.h is the body of both) -- writes the action / log-prob / value / mu / sigma slots and takes the sampled actions from
 * LDS instead of reading an action tensor: one launch and one memory round trip less per rollout step (pre_physics_step,
 * ten_ant.py:886-891, then consumes what ActorCritic.act, module.py:73-87, would have returned).  The binding holds for ONE step: the
 * step that consumes it clears it (the slot pointers move with every rollout step).  Available where the step kernel runs its 16-envs-
 * per-workgroup TenAnt layout (num_agents 10, num_envs a multiple of 16 and >= 16 per CU, no physical DR) with H a multiple of 512 and
 * A = 80; anywhere else the call fails and the caller launches mms_ppo_heads_act.  Fields as the arguments of mms_ppo_heads_act;
 * actions_out may name the engine's "actions" buffer (kept for task.actions) or be NULL. */
typedef struct mms_policy_head {
    const float* hidden; const float* weight; const float* bias;        /* actor: last hidden activations [N, H], last layer [A, H], [A] */
    const float* vhidden; const float* vweight; const float* vbias;     /* critic: [N, VH], [VH], [1] */
    const float* log_std;                                                /* [A] */
    int64_t* counters;                                                   /* [N] draw counters */
    float* actions_out; float* act_slot; float* logp_slot; float* value_slot; float* mu_slot; float* sigma_slot;
    uint64_t seed;
    int64_t row_offset;
    int32_t H, VH, A, reference_scale;
    const float* weight_tiles;   /* optional (NULL: `weight` is read): the actor's last layer once more, stored [ceil(A / 16)][H / 4][16][4] --
                                  * element ((ct (H / 4) + k / 4) 16 + i) 4 + k % 4 = weight[16 ct + i][k], zero for outputs >= A; 16-byte aligned.
                                  * One operand load of the head's matrix phase then reads 1 KB of contiguous memory (csrc/head_block.h). */
} mms_policy_head;
int mms_bind_policy_head(mms_handle h, const mms_policy_head* head);   /* NULL: unbind */

/* One hidden layer of the PPO policy for BOTH networks in one launch (module.py:27-52: nn.Linear + activation, actor and
 * critic of the same shape): y_g = act(x_g @ w_g^T + b_g), g = 0, 1, on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact
 * fp32 products and sums) with bias and activation in the epilogue.  x [M,K], w [N,K] and b [N] in torch's Linear layout,
 * y [M,N], all f32 and contiguous; K a multiple of 4; act 0 = identity, 1 = ELU(alpha 1) (cfg/ppo/config.yaml:9),
 * 2 = ReLU (the DDPG / TD3 actor, rl/ddpg/module.py:37), 3 = tanh (its output layer, :18).
 * x1 = w1 = b1 = y1 = NULL runs a single problem.  x0 and x1 may be the same buffer (first layer). */
int mms_linear2_act(int device, int64_t M, int32_t N, int32_t K, const float* x0, const float* w0, const float* b0, float* y0,
                    const float* x1, const float* w1, const float* b1, float* y1, int32_t act, void* hip_stream);

/* ---- MAPPO / HAPPO policy inference: every layer of ALL agents' networks per launch ------------------------------------------
 * The reference's collect step (algorithms/marl/runner.py:186-216) calls, agent by agent, policy.get_actions -> Actor.forward
 * (algorithms/marl/actor_critic.py:43-69) and Critic.forward (:137-155): MLPBase (algorithms/utils/mlp.py:38-65) = LayerNorm of
 * the input, then layer_N + 1 blocks of Linear + ELU + LayerNorm (:5-36); ACTLayer / DiagGaussian (utils/act.py:75-81,
 * utils/distributions.py:94-117) = Linear mean head, std = sigmoid(log_std / std_x_coef) * std_y_coef, sample, summed
 * log-probability; v_out = Linear(hidden, 1).  The three operators below run one such stage for up to MMS_MAX_GROUPS networks
 * of the same shape at once (ten actors + ten critics of TenAnt's MAPPO: 20).  The pointer arguments are HOST arrays of
 * `groups` device pointers.  Recurrent policies (use_recurrent_policy) are not covered. */
#define MMS_MAX_GROUPS 32

/* y_g = act(x_g @ w_g^T + b_g), g < groups: mms_linear2_act for any number of networks (same kernel, same shapes rules).
 * The three ln_* arguments (all NULL: none) fold the LayerNorms on either side of the layer into it, so that the normalised
 * activations are never written (they need act = ELU and M and N multiples of 128):
 *   ln_part_out[g] [N/64, M, 2]: the epilogue also leaves per output row and per 64-column slot the sum and the sum of squares of
 *     the activations; mms_row_stats_group turns them into (mean, rstd) per row.  No atomics: the result does not depend on scheduling.
 *   ln_stat_in[g] [M, 2] + ln_s[g] [N]: x_g is the PRE-LayerNorm activation h and the layer W (LN(h) gamma + beta) + b is evaluated
 *     as rstd (W~ h - mean s) + c: the caller passes W~ = W diag(gamma) as w_g, s = W~ 1 as ln_s[g] and c = W beta + b as b_g. */
int mms_linear_group_act(int device, int32_t groups, int64_t M, int32_t N, int32_t K, const float* const* x, const float* const* w,
                         const float* const* b, float* const* y, int32_t act, const float* const* ln_s, const float* const* ln_stat_in,
                         float* const* ln_part_out, void* hip_stream);

/* stat_g[r] = (mean, 1 / sqrt(var + eps)) of row r from the `slots` partial (sum, sum of squares) pairs of mms_linear_group_act's
 * ln_part_out (width = that layer's N; biased variance, as nn.LayerNorm). */
int mms_row_stats_group(int device, int32_t groups, int64_t M, int32_t slots, int32_t width, const float* const* part, float* const* stat,
                        float eps, void* hip_stream);

/* stat_g[r] = (mean, 1 / sqrt(var + eps)) of row r of x_g [M, K] (row pitch x_pitch floats, 0 = K; K <= 4096): the statistics of a
 * LayerNorm whose normalised output is never needed because the layer behind it takes them through ln_stat_in -- the feature
 * LayerNorm of the centralised observation, which every critic of an env shares. */
int mms_row_moments_group(int device, int32_t groups, int64_t M, int32_t K, int32_t x_pitch, const float* const* x, float* const* stat,
                          float eps, void* hip_stream);

/* nn.LayerNorm over the last dimension (biased variance, eps inside the root): y_g[r, 0:K] = LN(x_g[r, 0:K]) * gamma_g + beta_g,
 * y_g[r, K:Kp] = 0.  x rows have pitch x_pitch floats (0 = K; A * K reads one agent's rows of an [N, A, K] block where they lie),
 * y rows pitch Kp >= K (Kp > K pads a 46-wide observation to the multiple of 4 the layer kernel wants); y_g == x_g with
 * Kp == x_pitch == K is the in-place form.  K <= 4096 (the 100-ant swarm's centralised observation is 3808 wide). */
int mms_layernorm_group(int device, int32_t groups, int64_t M, int32_t K, int32_t Kp, int32_t x_pitch, const float* const* x,
                        const float* const* gamma, const float* const* beta, float* const* y, float eps, void* hip_stream);

/* (eps < 0: no LayerNorm -- the plain output layer on h_g; gamma / beta are not read.)
 * The last LayerNorm + the output layer (+ the Gaussian sample) of each network: out_g[r, j] = b_g[j] + sum_k w_g[j, k] *
 * LN(h_g[r])[k], j < A[g] <= 16, H <= 1024.  std[g] != NULL ([A[g]] standard deviations): out_g = that mean + std z with z ~ N(0,1)
 * from the counter-based stream keyed (seed + g, row_offset + r, counters[g][r], j) (counters[g][r] += 1; counters or counters[g]
 * NULL: counter 0), and logp[g][r, j] = log N(out_j | mean_j, std_j), PER DIMENSION, [M, A[g]] (FixedNormal.log_probs,
 * distributions.py:31-34, does not sum over the action dimensions).  std == NULL or
 * std[g] == NULL: out_g is the plain output (the critic's value; a deterministic action).  out_pitch[g] (NULL: A[g]) = floats
 * between consecutive rows of out_g and logp_g, so that agent k's actions land in an [N, agents, A] rollout slot directly.  The noise stream is this build's, not
 * torch's Philox: sampled actions differ from the reference's draw for the same torch seed, their distribution does not. */
int mms_marl_heads_act(int device, int32_t groups, int64_t M, int32_t H, const float* const* h, const float* const* gamma,
                       const float* const* beta, const float* const* w, const float* const* b, const int32_t* A, const float* const* std,
                       float* const* out, float* const* logp, const int32_t* out_pitch, int64_t* const* counters, uint64_t seed,
                       int64_t row_offset, float eps, void* hip_stream);

/* ---- the same layers on the bf16 matrix pipe with fp32 operands carried as three bf16 planes (csrc/split_kernels.hip) -----------
 * An fp32 number is exactly a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1); the product is formed as the six
 * bf16 MFMA products a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0 with fp32 accumulation (the dropped terms are < 2^-25 |a b|): the fp32
 * product of module.py:27-52 at 16 / 6 of the fp32 MFMA rate, with an error against the float64 product that is not larger than the
 * exact-fp32 MFMA kernel's (tests/test_gpu_parity.py measures both on the same inputs).
 * Plane format "P32": bf16 [rows, KC, 3, 32], KC = ceil(K / 32): per row and per 32 consecutive k the three planes, 192 contiguous bytes;
 * columns past K are zero.  MMS_P32_BYTES(rows, K) bytes. */
#define MMS_P32_BYTES(rows, K) ((size_t)(rows) * (size_t)(((K) + 31) / 32) * 192)

/* planes <- split(x): x [rows, K] f32 with row pitch x_pitch floats (0 = K; rows 16-byte aligned).  Weights once per optimizer step,
 * the observation once per env step; hidden activations are left in P32 by the layer that produces them (out_planes below). */
int mms_split_planes(int device, int64_t rows, int32_t K, int32_t x_pitch, const float* x, void* planes, void* hip_stream);

/* The same for `groups` matrices of one shape in one launch (x and planes: HOST arrays of device pointers).  Rows need not be 16-byte
 * aligned here (agent k's rows of an [M, agents, 46] observation block): unaligned sources take scalar loads. */
int mms_split_planes_group(int device, int32_t groups, int64_t rows, int32_t K, int32_t x_pitch, const float* const* x, void* const* planes,
                           void* hip_stream);

/* y_g = act(x_g @ w_g^T + b_g), g < groups, as mms_linear_group_act, with x_g [M, K] and w_g [N, K] given as P32 planes and b_g [N] f32.
 * M and N multiples of 128.  out_mode 0: y_g [M, N] f32; 1: y_g left as P32 planes of [M, N] (the next split layer's x); 2: no y_g --
 * only the output head's partial dot products (below).
 * LayerNorm folds (ln_s, ln_stat_in, ln_part_out: all NULL or all given; act must be ELU): as mms_linear_group_act's, except that
 *   ln_part_out[g] [N/64, M, 2] receives per row and 64-column slot (sum, sum of squared deviations FROM THE SLOT'S OWN MEAN) -- the
 *   two-pass form, free of E[x^2] - mean^2 cancellation; mms_row_stats_chan_group combines the slots (Chan's formula).
 * out_mode 2 (needs the folds): head_w[g] [head_dims[g], N] f32 = the output head's weight with the last LayerNorm's gamma folded in,
 *   head_dims[g] <= 16 (host array, one entry per network: an actor's action count, 1 for a critic); head_part[g] [N/64, M, stride_g],
 *   stride_g = head_dims[g] rounded up to 4, receives sum over the slot's columns of y[r, n] head_w[j, n]: the head of every network is
 *   finished by mms_marl_heads_finish (whose A[g] = head_dims[g]) without the activations ever being written.  head_dims may be NULL in
 *   the other output modes. */
int mms_linear_group_act_split(int device, int32_t groups, int64_t M, int32_t N, int32_t K, const void* const* x, const void* const* w,
                               const float* const* b, void* const* y, int32_t act, int32_t out_mode, const float* const* ln_s,
                               const float* const* ln_stat_in, float* const* ln_part_out, const float* const* head_w, float* const* head_part,
                               const int32_t* head_dims, void* hip_stream);

/* ---- ... and with two scaled fp16 planes per operand (csrc/split16_kernels.hip): the default of the policy modules -----------------
 * x s = hi + lo 2^-11 with s a power of two per ROW, hi = f16(x s), lo = f16((x s - hi) 2^11): the operand is kept to 2^-22 |x| (worst
 * case; 4e-8 rms) instead of exactly, in 4 bytes instead of 6, and the product is THREE f16 MFMA products (hi hi, hi lo, lo hi) with fp32
 * accumulation -- the layers are bound by the operand stream into LDS, so the k-loop takes two thirds of the three-plane kernel's time,
 * and three LDS stages fit.  Error against the float64 product: 0.39-0.46 x the exact-fp32 MFMA kernel's on the PPO policy's layers,
 * below it down to K = 32 (tests/test_gpu_parity.py::test_split16_layers_error).  What the two missing bits cost: behind a folded
 * LayerNorm the operand error is amplified by |mean| / std of the row like the rounding of the activations themselves (x 4 against the
 * fp32 passes at |mean| / std = 1000, tests/test_marl_policy.py); the three-plane entry points above have exact operands.
 * Scales: a plane row's scale puts a bound of the row's magnitudes at 2^14 (fp16 ends at 65504).  Inputs (observations, weight rows): the
 * row's own largest magnitude, found by the split.  Hidden activations: a-priori, |act(W x + b)| <= (largest row 1-norm of W) max|x| +
 * max|b| (ELU, ReLU, tanh, identity: |act(y)| <= |y|), evaluated per row by the split of the network's input (`chain`); behind a
 * LayerNorm the bound does not depend on the data (|W~ xhat + c| <= |W~ row|_2 sqrt(K) + |c|) and the caller passes constant scales.
 * Plane format "H32": f16 [rows, KC, 2, 32], KC = ceil(K / 32): hi and lo of 32 consecutive k of a row, 128 contiguous bytes; columns
 * past K are zero.  MMS_H32_BYTES(rows, K) bytes. */
#define MMS_H32_BYTES(rows, K) ((size_t)(rows) * (size_t)(((K) + 31) / 32) * 128)

/* planes_g <- split(x_g) for `groups` matrices [rows, K] f32 (row pitch x_pitch floats, 0 = K; unaligned rows take scalar loads), with
 * scale[g][r] = the row's power of two and inv[g][r] = 1 / scale (f32 [rows]; entries of the two arrays may be NULL).
 * nchains > 0: chain[g] = f32 [nchains, L, 2] of (mult, add) per layer; for each chain c and layer l < L the bound
 * b_{l+1} = (mult_l b_l + add_l) 1.001, b_0 = the row's largest magnitude, gives chain_scale[g][c, l, r] = 2^(14 - e), b_{l+1} <= 2^e, and
 * chain_inv = its inverse (f32 [nchains, L, rows] each): the y_scale / next x_inv of the layers below.
 * stat != NULL: stat[g] f32 [rows, 2] receives (mean, 1 / sqrt(var + eps)) of every row -- the statistics of an nn.LayerNorm over the
 * rows (two-pass form), which the grouped MARL inference folds into its first layers: the split reads the rows anyway. */
int mms_split_planes16_group(int device, int32_t groups, int64_t rows, int32_t K, int32_t x_pitch, const float* const* x, void* const* planes,
                             float* const* scale, float* const* inv, int32_t nchains, int32_t L, const float* const* chain,
                             float* const* chain_scale, float* const* chain_inv, float* const* stat, float eps, void* hip_stream);

/* The weights' side of the same layers, refreshed ON THE DEVICE after every parameter update (no host synchronisation, every output at
 * the caller's address: the two calls may sit at the head of a captured hipGraph of a rollout, so that a replay after an optimizer step
 * computes with the updated parameters -- ActorCritic.refresh(), algorithms/rl/ppo/module.py; the reference re-reads its nn.Linear
 * parameters in every forward, agents/algorithms/rl/ppo/module.py:73-87):
 * mms_weight_planes16_group: planes_g / scale_g / inv_g <- split of w_g [N[g], K[g]] f32 (contiguous; N, K: HOST arrays, the shapes may
 *   differ from group to group: all hidden layers of both networks in ONE launch) exactly as mms_split_planes16_group, and
 *   l1[g][n] = sum_k |w_g[n, k]| (f32 [N[g]]; l1 or l1[g] NULL: not written);
 * mms_chain_refresh16: entry e = c L + l of the bound chain, chain[e] = (max_i l1[e][i], max_i |bias[e][i]|), i < n[e] (bias or bias[e]
 *   NULL: 0) -- `chain` [nchains, L, 2] is what mms_split_planes16_group takes; l1, bias, n: HOST arrays of nchains * L <= MMS_MAX_GROUPS
 *   entries.  rows > 0: also the chain's scales for rows whose input bound is the CONSTANT bound0 (observation rows clamped to clip_obs
 *   whose planes the step kernel wrote, mms_bind_obs_planes16: bound0 = 2^14 / scale): chain_scale / chain_inv [nchains, L, rows] as
 *   mms_split_planes16_group would leave them for a row whose largest magnitude is bound0. */
int mms_weight_planes16_group(int device, int32_t groups, const int64_t* N, const int32_t* K, const float* const* w, void* const* planes,
                              float* const* scale, float* const* inv, float* const* l1, void* hip_stream);
int mms_chain_refresh16(int device, int32_t nchains, int32_t L, const float* const* l1, const float* const* bias, const int32_t* n,
                        float* chain, float bound0, int64_t rows, float* chain_scale, float* chain_inv, void* hip_stream);

/* mms_linear_group_act_split with x_g, w_g (and, out_mode 1, y_g) in the H32 format.  x_inv[g] f32 [M] and w_inv[g] f32 [N] (16-byte
 * aligned) are the inverse row scales of the operands; y_scale[g] f32 [M] (out_mode 1) is the scale the output rows are stored with --
 * it must put a bound of the row's activations at or below 2^14; the next layer's x_inv is its inverse.  Everything else (out_mode, the
 * LayerNorm folds, the output-head partials) as mms_linear_group_act_split. */
int mms_linear_group_act_split16(int device, int32_t groups, int64_t M, int32_t N, int32_t K, const void* const* x, const void* const* w,
                                 const float* const* b, void* const* y, const float* const* x_inv, const float* const* w_inv,
                                 const float* const* y_scale, int32_t act, int32_t out_mode, const float* const* ln_s,
                                 const float* const* ln_stat_in, float* const* ln_part_out, const float* const* head_w, float* const* head_part,
                                 const int32_t* head_dims, void* hip_stream);

/* The weights' side of a layer BEHIND A FOLDED LayerNorm, refreshed on the device after every update (GroupedPolicyInference.refresh():
 * no host synchronisation, every output at the caller's address; the reference re-reads its parameters in every forward,
 * agents/algorithms/utils/mlp.py:19-27,44-60).  For g < groups and row n of w_g [N[g], K[g]] (f32, contiguous; N, K: HOST arrays, the
 * shapes may differ from group to group):
 *   W~[n, k] = w[n, k] gamma_g[k] (gamma or gamma[g] NULL: w itself) -> planes_g (H32) with inv_g[n] = 1 / its row scale (planes or
 *   planes[g] NULL: no planes), wt[g] = W~ as f32 [N, K] (NULL: not stored), s[g][n] = sum_k W~[n, k], c[g][n] = sum_k w[n, k] beta_g[k] +
 *   bias_g[n] (NULL: without that term), rb[g][n] = |W~ row|_2 sqrt(K) + |c[n]|: the row's bound of W~ xhat + c over normalised inputs.
 * mms_fold_scales16_group: scale_g = 2^(14 - e) with 1.001 max_{i < n[g]} rb[g][i] <= 2^e -> scale1[g][0] (f32 [1], optional) and
 *   ysc[g][0..M) = scale_g, yinv[g][0..M) = 1 / scale_g (f32 [M] each, optional): the y_scale / next x_inv rows of
 *   mms_linear_group_act_split16 for a layer whose output bound does not depend on the data. */
int mms_fold_planes16_group(int device, int32_t groups, const int64_t* N, const int32_t* K, const float* const* w, const float* const* gamma,
                            const float* const* beta, const float* const* bias, void* const* planes, float* const* inv, float* const* s,
                            float* const* c, float* const* rb, float* const* wt, void* hip_stream);
int mms_fold_scales16_group(int device, int32_t groups, const float* const* rb, const int32_t* n, int64_t M, float* const* scale1,
                            float* const* ysc, float* const* yinv, void* hip_stream);

/* stat_g[r] = (mean, 1 / sqrt(var + eps)) of row r from mms_linear_group_act_split's ln_part_out (`slots` = N / 64 slots of 64). */
int mms_row_stats_chan_group(int device, int32_t groups, int64_t M, int32_t slots, const float* const* part, float* const* stat, float eps,
                             void* hip_stream);

/* The output heads behind an out_mode-2 layer: mean_gj[r] = rstd_r (sum_slots head_part[g][slot, r, j] - mean_r hs[g][j]) + hc[g][j] with
 * (mean_r, rstd_r) from part[g] (the last hidden layer's ln_part_out), hs = head_w 1 and hc = w beta + b (both [A[g]]); then exactly
 * mms_marl_heads_act's sampling: std / out / logp / out_pitch / counters / seed / row_offset have the same meaning. */
int mms_marl_heads_finish(int device, int32_t groups, int64_t M, int32_t slots, const float* const* part, const float* const* head_part,
                          const float* const* hs, const float* const* hc, const int32_t* A, const float* const* std, float* const* out,
                          float* const* logp, const int32_t* out_pitch, int64_t* const* counters, uint64_t seed, int64_t row_offset, float eps,
                          void* hip_stream);

const char* mms_last_error(mms_handle h);   /* h may be NULL: error of the last failed mms_create */
int mms_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MMS_H */
