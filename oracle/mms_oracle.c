/*
 * mms_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library.  The product (massive_marl_benchmark_amd/) never imports, links or calls it and has
 * no CPU fallback.
 *
 * What it is: a plain-C, scalar, fp32 restatement of the hot path of
 * SafeRL-Lab/Massive-MARL-Benchmark (paths below are relative to /root/reference):
 *
 *   helpers              isaacgym.torch_utils semantics (not in tree) and
 *                        agents/utils/torch_jit_utils.py:13-50
 *   TenAnt obs           agents/tasks/ten_ant.py:1304-1350, assembly :806-808
 *   TenAnt goals         agents/tasks/ten_ant.py:935-986, 1353-1393
 *   TenAnt reward/reset  agents/tasks/ten_ant.py:988-1301 (abs(bool-1) read as 1-[d<1.5])
 *   TenAnt step glue     agents/tasks/ten_ant.py:810-926, agents/tasks/agent_base/base_task.py:129-149
 *   OneAnt obs/reward    agents/tasks/one_ant.py:465-627, glue :371-436
 *   MultiIngenuity       agents/tasks/multi_ingenuity.py:228-453
 *   wrappers             agents/tasks/agent_base/vec_task.py:121-139, multi_vec_task.py:94-175
 *   PPO GAE              agents/algorithms/rl/ppo/storage.py:51-65
 *   MARL GAE             agents/algorithms/marl/utils/separated_buffer.py:153-164
 *
 * PARITY PINNING.  Everything above is pinned by the golden vectors in tests/golden (npz files), which
 * were produced by importing the reference's own functions (tests/golden/make_fixtures.py).
 * The articulated rigid-body step itself (`gym.simulate`, base_task.py:139) is NVIDIA Isaac Gym /
 * PhysX, a closed third-party binary that is absent from the reference tree (README.md:4, no
 * version pin) and from this pipeline: for the physics, PARITY IS UNPINNED.  The physics below is
 * this build's own model definition: reduced-coordinate articulated-body
 * algorithm, linearly-implicit joint damping / joint limits / compliant contacts, semi-implicit
 * Euler, `substeps` substeps per control step.  The HIP kernels are checked against THIS code.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mms.h"

#define MO_EXPORT __attribute__((visibility("default")))
#define PI_F 3.14159265358979323846f
#define TWO_PI_F 6.28318530717958647692f

/* The physics (and the small vector helpers it uses) is written on `real`.  Default build: real = float -- the oracle the HIP
 * kernels are compared with.  -DMO_F64 (oracle/Makefile: _build/libmms_oracle_f64.so) compiles ONLY the physics, in double, behind
 * mo_physics_f64: the same model and the same fp32 inputs, evaluated with 29 more bits.  The GPU parity tests use it as the
 * yardstick: the HIP kernel may be at most a small multiple of as far from the double result as the fp32 oracle itself is. */
#ifdef MO_F64
typedef double real;
#define r_sqrt sqrt
#define r_fmax fmax
#define r_fmin fmin
#define r_fabs fabs
#define r_cos cos
#define r_sin sin
#else
typedef float real;
#define r_sqrt sqrtf
#define r_fmax fmaxf
#define r_fmin fminf
#define r_fabs fabsf
#define r_cos cosf
#define r_sin sinf
#endif
static inline real r_clamp(real x, real lo, real hi) { return r_fmax(r_fmin(x, hi), lo); }
#define R3(name, src) const real name[3] = {(src)[0], (src)[1], (src)[2]}   /* fp32 model vector as `real` (a copy; exact) */

/* ------------------------------------------------------------------------------------------ */
/* small vector / matrix helpers                                                               */
/* ------------------------------------------------------------------------------------------ */
static inline void cross3(const real a[3], const real b[3], real o[3]) {
    real x = a[1] * b[2] - a[2] * b[1];
    real y = a[2] * b[0] - a[0] * b[2];
    real z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline real dot3(const real a[3], const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void matvec3(const real R[3][3], const real v[3], real o[3]) {
    real x = R[0][0] * v[0] + R[0][1] * v[1] + R[0][2] * v[2];
    real y = R[1][0] * v[0] + R[1][1] * v[1] + R[1][2] * v[2];
    real z = R[2][0] * v[0] + R[2][1] * v[1] + R[2][2] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void matTvec3(const real R[3][3], const real v[3], real o[3]) {
    real x = R[0][0] * v[0] + R[1][0] * v[1] + R[2][0] * v[2];
    real y = R[0][1] * v[0] + R[1][1] * v[1] + R[2][1] * v[2];
    real z = R[0][2] * v[0] + R[1][2] * v[1] + R[2][2] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void matmul3(const real A[3][3], const real B[3][3], real O[3][3]) {
    real T[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
    memcpy(O, T, sizeof(T));
}
/* rotation matrix of a unit quaternion (xyzw) */
static void quat_to_mat(const real q[4], real R[3][3]) {
    real x = q[0], y = q[1], z = q[2], w = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - w * z);       R[0][2] = 2.f * (x * z + w * y);
    R[1][0] = 2.f * (x * y + w * z);       R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - w * x);
    R[2][0] = 2.f * (x * z - w * y);       R[2][1] = 2.f * (y * z + w * x);       R[2][2] = 1.f - 2.f * (x * x + y * y);
}
/* Rodrigues: rotation by `ang` about unit axis `a` */
static void axis_angle_to_mat(const real a[3], real ang, real R[3][3]) {
    real c = r_cos(ang), s = r_sin(ang), t = 1.f - c;
    R[0][0] = c + t * a[0] * a[0];        R[0][1] = t * a[0] * a[1] - s * a[2]; R[0][2] = t * a[0] * a[2] + s * a[1];
    R[1][0] = t * a[0] * a[1] + s * a[2]; R[1][1] = c + t * a[1] * a[1];        R[1][2] = t * a[1] * a[2] - s * a[0];
    R[2][0] = t * a[0] * a[2] - s * a[1]; R[2][1] = t * a[1] * a[2] + s * a[0]; R[2][2] = c + t * a[2] * a[2];
}
/* q <- normalize(q + h/2 * (w,0) (x) q), world-frame angular velocity */
static void quat_integrate(real q[4], const real w[3], real h) {
    real x = q[0], y = q[1], z = q[2], s = q[3];
    real hx = 0.5f * h * w[0], hy = 0.5f * h * w[1], hz = 0.5f * h * w[2];
    real nx = x + (hx * s + hy * z - hz * y);
    real ny = y + (hy * s + hz * x - hx * z);
    real nz = z + (hz * s + hx * y - hy * x);
    real ns = s - (hx * x + hy * y + hz * z);
    real inv = 1.f / r_sqrt(nx * nx + ny * ny + nz * nz + ns * ns);
    q[0] = nx * inv; q[1] = ny * inv; q[2] = nz * inv; q[3] = ns * inv;
}

/* MultiIngenuity thrust (multi_ingenuity.py:268-339): thrusts[8][3] for one env from actions[24]; rotor (2h + r) uses
 * actions[6h + 3r .. 6h + 3r + 2].  Pinned through mo_ingenuity_thrust (fixture ingenuity_thrust). */
static void ingenuity_thrust_real(const real actions[24], real dt, real thrusts[24]) {
    for (int h = 0; h < 4; h++)
        for (int r = 0; r < 2; r++) {
            const real* a = actions + 6 * h + 3 * r;
            real vert = r_clamp(a[2] * 2000.f, -2000.f, 2000.f);
            real lx = r_clamp(a[0], -0.2f, 0.2f), ly = r_clamp(a[1], -0.2f, 0.2f);
            real* t = thrusts + 3 * (2 * h + r);
            t[2] = dt * vert;
            t[0] = t[2] * lx;
            t[1] = t[2] * ly;
        }
}

#ifndef MO_F64   /* the fixture-pinned restatements below are fp32 only */
/* ------------------------------------------------------------------------------------------ */
/* isaacgym.torch_utils helper semantics, scalar                               */
/* ------------------------------------------------------------------------------------------ */
MO_EXPORT void mo_quat_mul(const float a[4], const float b[4], float o[4]) {
    float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3], x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    o[3] = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2;
    o[0] = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2;
    o[1] = w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2;
    o[2] = w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2;
}
MO_EXPORT void mo_quat_conjugate(const float a[4], float o[4]) { o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3]; }
/* quat_rotate: v(2w^2-1) + 2w(q x v) + 2q(q.v); the inverse negates the cross term */
static void quat_rot_impl(const float q[4], const float v[3], float o[3], float sign) {
    float w = q[3], c[3];
    float a = 2.0f * w * w - 1.0f;
    cross3(q, v, c);
    float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
    for (int i = 0; i < 3; i++) o[i] = v[i] * a + sign * (c[i] * w * 2.0f) + q[i] * d * 2.0f;
}
MO_EXPORT void mo_quat_rotate(const float q[4], const float v[3], float o[3]) { quat_rot_impl(q, v, o, 1.f); }
MO_EXPORT void mo_quat_rotate_inverse(const float q[4], const float v[3], float o[3]) { quat_rot_impl(q, v, o, -1.f); }
MO_EXPORT void mo_normalize3(const float v[3], float o[3]) {
    float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (n < 1e-9f) n = 1e-9f;
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}
/* python-style remainder by 2*pi: result in [0, 2pi) */
static inline float wrap_2pi(float a) {
    float r = fmodf(a, TWO_PI_F);
    if (r < 0.f) r += TWO_PI_F;
    return r;
}
MO_EXPORT void mo_get_euler_xyz(const float q[4], float* roll, float* pitch, float* yaw) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float sinr_cosp = 2.0f * (w * x + y * z);
    float cosr_cosp = w * w - x * x - y * y + z * z;
    float r = atan2f(sinr_cosp, cosr_cosp);
    float sinp = 2.0f * (w * y - z * x);
    float p = (fabsf(sinp) >= 1.f) ? copysignf(PI_F / 2.0f, sinp) : asinf(sinp);
    float siny_cosp = 2.0f * (w * z + x * y);
    float cosy_cosp = w * w + x * x - y * y - z * z;
    float yw = atan2f(siny_cosp, cosy_cosp);
    *roll = wrap_2pi(r); *pitch = wrap_2pi(p); *yaw = wrap_2pi(yw);
}
static inline float unscale1(float x, float lo, float hi) { return (2.0f * x - hi - lo) / (hi - lo); }
static inline float clampf(float x, float lo, float hi) { return fmaxf(fminf(x, hi), lo); }

/* batch wrappers used by the helper KATs */
MO_EXPORT void mo_helpers_batch(int64_t n, const float* q, const float* q2, const float* v, float* quat_mul,
                                float* quat_conj, float* rot, float* rot_inv, float* roll, float* pitch,
                                float* yaw, float* norm, float* axis0, float* axis2) {
    const float e0[3] = {1, 0, 0}, e2[3] = {0, 0, 1};
    for (int64_t i = 0; i < n; i++) {
        mo_quat_mul(q + 4 * i, q2 + 4 * i, quat_mul + 4 * i);
        mo_quat_conjugate(q + 4 * i, quat_conj + 4 * i);
        mo_quat_rotate(q + 4 * i, v + 3 * i, rot + 3 * i);
        mo_quat_rotate_inverse(q + 4 * i, v + 3 * i, rot_inv + 3 * i);
        mo_get_euler_xyz(q + 4 * i, roll + i, pitch + i, yaw + i);
        mo_normalize3(v + 3 * i, norm + 3 * i);
        mo_quat_rotate(q + 4 * i, e0, axis0 + 3 * i); /* torch_jit_utils.py:45-50 quat_axis */
        mo_quat_rotate(q + 4 * i, e2, axis2 + 3 * i);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* ant observation core (ten_ant.py:1304-1350 / one_ant.py:563-618 share it)                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float vel_loc[3], angvel_loc[3], yaw, roll, angle_to_target, up_proj, heading_proj;
} ant_obs_core;

/* root: global-frame pos(3) quat(4) linvel(3) angvel(3).  targets = (0,0,0), inv_start_rot = identity,
 * basis_vec0 = (1,0,0), basis_vec1 = (0,0,1) (ten_ant.py:164-170). */
static void ant_obs_core_compute(const float root[13], ant_obs_core* o) {
    const float* p = root;
    const float* q = root + 3;
    const float targets[3] = {0.f, 0.f, 0.f};
    float to_target[3] = {targets[0] - p[0], targets[1] - p[1], 0.0f};
    float target_dirs[3];
    mo_normalize3(to_target, target_dirs);                     /* torch_jit_utils.py:19 */
    const float inv_start_rot[4] = {-0.f, -0.f, -0.f, 1.f};
    float tq[4];
    mo_quat_mul(q, inv_start_rot, tq);                         /* :21 */
    const float b1[3] = {0, 0, 1}, b0[3] = {1, 0, 0};
    float up_vec[3], heading_vec[3];
    mo_quat_rotate(tq, b1, up_vec);
    mo_quat_rotate(tq, b0, heading_vec);
    o->up_proj = up_vec[2];
    o->heading_proj = heading_vec[0] * target_dirs[0] + heading_vec[1] * target_dirs[1] + heading_vec[2] * target_dirs[2];
    mo_quat_rotate_inverse(tq, root + 7, o->vel_loc);          /* :32-33 */
    mo_quat_rotate_inverse(tq, root + 10, o->angvel_loc);
    float pitch;
    mo_get_euler_xyz(tq, &o->roll, &pitch, &o->yaw);
    float walk_target_angle = atan2f(targets[2] - p[2], targets[0] - p[0]); /* :38-39 (z and x, sic) */
    o->angle_to_target = walk_target_angle - o->yaw;
}

MO_EXPORT void mo_tenant_ant_obs(const float root[13], const float dof_pos[8], const float dof_vel[8],
                                 const float lower[8], const float upper[8], float dof_vel_scale,
                                 const float actions[8], float obs[38]) {
    ant_obs_core c;
    ant_obs_core_compute(root, &c);
    obs[0] = root[0]; obs[1] = root[1]; obs[2] = root[2];
    for (int i = 0; i < 3; i++) { obs[3 + i] = c.vel_loc[i]; obs[6 + i] = c.angvel_loc[i]; }
    obs[9] = c.yaw; obs[10] = c.roll; obs[11] = c.angle_to_target; obs[12] = c.up_proj; obs[13] = c.heading_proj;
    for (int j = 0; j < 8; j++) {
        obs[14 + j] = unscale1(dof_pos[j], lower[j], upper[j]);
        obs[22 + j] = dof_vel[j] * dof_vel_scale;
        obs[30 + j] = actions[j];
    }
}
MO_EXPORT void mo_tenant_obs_batch(int64_t n, const float* root, const float* dof_pos, const float* dof_vel,
                                   const float* lower, const float* upper, float dof_vel_scale,
                                   const float* actions, float* obs) {
    for (int64_t i = 0; i < n; i++)
        mo_tenant_ant_obs(root + 13 * i, dof_pos + 8 * i, dof_vel + 8 * i, lower, upper, dof_vel_scale,
                          actions + 8 * i, obs + 38 * i);
}

/* ten_ant.py:935-947 compute_box_angle, :1353-1393 goals.  goals[k] for k = 0..A-1:
 * goal_{2j+1} = box + (1.5+3j) d, goal_{2j+2} = box - (1.5+3j) d, d = (sin, -cos). */
static inline float box_angle(const float q[4]) {
    float qw = q[3], qz = q[2];
    float y = 2.f * qw * qz;
    float x = 1.f - 2.f * qz * qz;
    return atanf(y / x);
}
MO_EXPORT void mo_tenant_goals(const float box_root[13], int agents, float box_pos[2], float box_quat[4], float* goals) {
    box_pos[0] = box_root[0]; box_pos[1] = box_root[1];
    for (int i = 0; i < 4; i++) box_quat[i] = box_root[3 + i];
    float ang = box_angle(box_quat);
    float sv = sinf(ang), cv = -cosf(ang);
    for (int k = 0; k < agents; k++) {
        float off = 1.5f + 3.0f * (float)(k / 2);
        float s = (k % 2 == 0) ? 1.f : -1.f;
        goals[2 * k + 0] = (s > 0.f) ? box_pos[0] + off * sv : box_pos[0] - off * sv;
        goals[2 * k + 1] = (s > 0.f) ? box_pos[1] + off * cv : box_pos[1] - off * cv;
    }
}
static inline float l2_dist2(const float a[2], const float b[2]) {
    float c1 = a[0] - b[0], c2 = a[1] - b[1];
    return sqrtf(c1 * c1 + c2 * c2);
}
/* ten_ant.py:951-973 with (x_goal, y_goal, z_goal) */
static float box_quat_dist(const float q[4], float xg, float yg, float zg) {
    float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    float x = 2.f * (qx * qy + qw * qz);
    float y = 1.f - 2.f * (qx * qx + qz * qz);
    float z = 2.f * (qy * qz - qw * qx);
    float x1 = x * xg, y1 = y * yg, z1 = z * zg;
    return (x1 + y1 + z1) / sqrtf(x * x + y * y + z * z) / sqrtf(xg * xg + yg * yg + zg * zg);
}
MO_EXPORT void mo_tenant_goals_batch(int64_t n, const float* box_root, float* box_pos, float* box_quat,
                                     float* goals, float* angle, float* quat_dist) {
    for (int64_t i = 0; i < n; i++) {
        mo_tenant_goals(box_root + 13 * i, 10, box_pos + 2 * i, box_quat + 4 * i, goals + 20 * i);
        angle[i] = box_angle(box_root + 13 * i + 3);
        quat_dist[i] = box_quat_dist(box_root + 13 * i + 3, 0.f, 1.f, 0.f);
    }
}

typedef struct {
    float up_weight, heading_weight, actions_cost, energy_cost, joints_at_limit_cost;
    float termination_height, death_cost;
    float quat_reward_scale, ant_dist_reward_scale, goal_dist_reward_scale;
    int32_t max_episode_length;
} reward_params;

/* ten_ant.py:1067-1301 for A ants.  obs: [A][38]; pos_before/goal_before/goals: [A][2]. */
static void tenant_reward(const reward_params* rp, int A, const float* obs, const float* actions,
                          const float* pos_before, const float* goal_before, const float* goals,
                          const float box_quat[4], int64_t reset_in, int64_t progress,
                          float* rew_out, int64_t* reset_out) {
    float quat_dist = box_quat_dist(box_quat, 0.f, 1.f, 0.f);
    float quat_reward = rp->quat_reward_scale * quat_dist;
    float ant_dist_reward = 0.f, goal_dist_reward = 0.f, goal_arrive_reward = 0.f, up_reward = 0.f;
    float electricity_cost = 0.f;
    int dof_at_limit = 0;
    int all_arrive = 1, fallen = 0;
    for (int k = 0; k < A; k++) {
        const float* o = obs + 38 * k;
        const float* goal = goals + 2 * k;
        float off = 1.5f + 3.0f * (float)(k / 2);
        float box_target[2] = {0.f, (k % 2 == 0) ? -off : off};   /* ten_ant.py:172-181 */
        float d_now = l2_dist2(o, goal);
        float ant_push = (d_now < 1.5f) ? 0.f : 1.f;               /* intended meaning of abs(bool-1) */
        float ant_dist = l2_dist2(pos_before + 2 * k, goal_before + 2 * k) - d_now;
        float adr = rp->ant_dist_reward_scale * ant_dist * ant_push;
        float gd_before = l2_dist2(box_target, goal_before + 2 * k);
        float gd = l2_dist2(box_target, goal);
        int arrive = gd < 0.5f;
        float gdr = rp->goal_dist_reward_scale * (gd_before - gd);
        float gar = arrive ? 2.f : 0.f;
        ant_dist_reward = (k == 0) ? adr : ant_dist_reward + adr;  /* left-to-right sums, :1173-1180 */
        goal_dist_reward = (k == 0) ? gdr : goal_dist_reward + gdr;
        goal_arrive_reward = (k == 0) ? gar : goal_arrive_reward + gar;
        all_arrive = all_arrive && arrive;
        float ur = (o[12] > 0.93f) ? (0.f + rp->up_weight) : 0.f;
        up_reward = (k == 0) ? ur : up_reward + ur;
        float ec = 0.f;
        int lim = 0;
        for (int j = 0; j < 8; j++) {
            ec += fabsf(actions[8 * k + j] * o[22 + j]);
            lim += (o[14 + j] > 0.99f) ? 1 : 0;
        }
        electricity_cost = (k == 0) ? ec : electricity_cost + ec;
        dof_at_limit += lim;
        fallen = fallen || (o[2] < rp->termination_height);
    }
    up_reward = up_reward * 10.f;                                  /* :1240-1241 */
    float success_reward = ((quat_dist > 0.9f) && all_arrive) ? 100.f : 0.f;
    float actions_cost = 0.f;
    for (int j = 0; j < 8 * A; j++) actions_cost += actions[j] * actions[j];
    float alive = 5.f;
    float total = alive + up_reward + quat_reward + ant_dist_reward + goal_dist_reward + goal_arrive_reward +
                  success_reward - rp->actions_cost * actions_cost - rp->energy_cost * electricity_cost -
                  (float)dof_at_limit * rp->joints_at_limit_cost;
    if (fallen) total = rp->death_cost;
    int64_t reset = fallen ? 1 : reset_in;
    if (progress >= (int64_t)rp->max_episode_length - 1) reset = 1;
    *rew_out = total;
    *reset_out = reset;
}
MO_EXPORT void mo_tenant_reward_batch(int64_t n, const float* obs /*[n][10][38]*/, const int64_t* reset_in,
                                      const int64_t* progress, const float* actions, const float* pos_before,
                                      const float* goal_before, const float* goals, const float* box_quat,
                                      const float* scal /*reward_params as 10 floats + max_len*/, float* rew,
                                      int64_t* reset) {
    reward_params rp = {scal[0], scal[1], scal[2], scal[3], scal[4], scal[5], scal[6], scal[7], scal[8], scal[9],
                        (int32_t)scal[10]};
    for (int64_t i = 0; i < n; i++)
        tenant_reward(&rp, 10, obs + 380 * i, actions + 80 * i, pos_before + 20 * i, goal_before + 20 * i,
                      goals + 20 * i, box_quat + 4 * i, reset_in[i], progress[i], rew + i, reset + i);
}

/* ------------------------------------------------------------------------------------------ */
/* MultiAntCircle (multi_ant_circle.py:385-502), INTENDED semantics: the reference's functions   */
/* cannot run (numpy on tensors, bool arithmetic); pinned by tests/golden/circle_reward.npz,      */
/* generated from a patched temp copy (tests/golden/make_circle_fixture.py lists the patches).    */
/* ------------------------------------------------------------------------------------------ */
/* compute_angle (:385-398): degrees in [0, 360), measured from +x through +y */
static float circle_angle(float a, float b) {
    float deg = fabsf(atan2f(b, a) * 180.f / 3.141592653589793f);
    return (b < 0.f) ? 360.f + (-1.f) * deg : deg;
}
/* one env: obs [2][38], actions [16], pos_before [2][2] (the caches: obs[k][0:2] of the previous step) */
static void circle_reward(const reward_params* rp, const float* obs, const float* actions, const float* pos_before,
                          int64_t reset_in, int64_t progress, float* rew_out, int64_t* reset_out) {
    float rew = 0.f, up_reward = 0.f, actions_cost = 0.f, electricity = 0.f;
    int lim = 0, fallen = 0;
    for (int k = 0; k < 2; k++) {
        const float* o = obs + 38 * k;
        float sgn = (k == 0) ? 1.f : -1.f;                          /* pos_2 = -obs_buf_2[:, :2] (:428); pos_before_2 is NOT negated (:431) */
        float px = sgn * o[0], py = sgn * o[1];
        float dist = sqrtf(px * px + py * py);                      /* per-env norm: the reading of np.linalg.norm(pos) */
        float ang = circle_angle(px, py), ang_before = circle_angle(pos_before[2 * k], pos_before[2 * k + 1]);
        int on = (ang - ang_before > 0.f) && (dist >= 2.7f) && (dist <= 3.3f);
        float rk = (on ? 2.f : 0.f) + ((on ? 1.f : 0.f) - 1.f);      /* (b) * 2 + ((b) - 1) */
        rew = (k == 0) ? rk : rew + rk;
        float ur = (o[12] > 0.93f) ? (0.f + rp->up_weight) : 0.f;
        up_reward = (k == 0) ? ur : up_reward + ur;
        float ac = 0.f, ec = 0.f;
        for (int j = 0; j < 8; j++) {
            ac += actions[8 * k + j] * actions[8 * k + j];
            ec += fabsf(actions[8 * k + j] * o[22 + j]);
            lim += (o[14 + j] > 0.99f) ? 1 : 0;
        }
        actions_cost = (k == 0) ? ac : actions_cost + ac;
        electricity = (k == 0) ? ec : electricity + ec;
        fallen = fallen || (o[2] < rp->termination_height);
    }
    float total = up_reward + rew - rp->actions_cost * actions_cost - rp->energy_cost * electricity - (float)lim * rp->joints_at_limit_cost;
    if (fallen) total = rp->death_cost;
    int64_t reset = fallen ? 1 : reset_in;
    if (progress >= (int64_t)rp->max_episode_length - 1) reset = 1;
    *rew_out = total;
    *reset_out = reset;
}
MO_EXPORT void mo_circle_reward_batch(int64_t n, const float* obs /*[n][2][38]*/, const int64_t* reset_in, const int64_t* progress,
                                      const float* actions /*[n][16]*/, const float* pos_before /*[n][2][2]*/,
                                      const float* scal /*up_weight, heading_weight, actions_cost, energy_cost, joints_at_limit_cost, termination_height, death_cost, max_len*/,
                                      float* rew, int64_t* reset, float* angle_1) {
    reward_params rp = {scal[0], scal[1], scal[2], scal[3], scal[4], scal[5], scal[6], 0.f, 0.f, 0.f, (int32_t)scal[7]};
    for (int64_t i = 0; i < n; i++) {
        circle_reward(&rp, obs + 76 * i, actions + 16 * i, pos_before + 4 * i, reset_in[i], progress[i], rew + i, reset + i);
        if (angle_1) angle_1[i] = circle_angle(obs[76 * i], obs[76 * i + 1]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* OneAnt (one_ant.py:563-627 obs, :465-560 reward)                                            */
/* ------------------------------------------------------------------------------------------ */
MO_EXPORT void mo_oneant_obs(const float root[13], const float box_root[13], const float dof_pos[8],
                             const float dof_vel[8], const float lower[8], const float upper[8],
                             float dof_vel_scale, const float sensors[24], float contact_force_scale,
                             const float actions[8], float dt, float potentials_in, float obs[60],
                             float* potentials, float* prev_potentials) {
    ant_obs_core c;
    ant_obs_core_compute(root, &c);
    float tb[3] = {0.f - box_root[0], 0.f - box_root[1], 0.0f};
    *prev_potentials = potentials_in;
    *potentials = -sqrtf(tb[0] * tb[0] + tb[1] * tb[1] + tb[2] * tb[2]) / dt;
    obs[0] = root[2];
    for (int i = 0; i < 3; i++) { obs[1 + i] = c.vel_loc[i]; obs[4 + i] = c.angvel_loc[i]; }
    obs[7] = c.yaw; obs[8] = c.roll; obs[9] = c.angle_to_target; obs[10] = c.up_proj; obs[11] = c.heading_proj;
    for (int j = 0; j < 8; j++) {
        obs[12 + j] = unscale1(dof_pos[j], lower[j], upper[j]);
        obs[20 + j] = dof_vel[j] * dof_vel_scale;
        obs[52 + j] = actions[j];
    }
    for (int j = 0; j < 24; j++) obs[28 + j] = sensors[j] * contact_force_scale;
}
static void oneant_reward(const reward_params* rp, const float obs[60], const float actions[8],
                          const float pos_before[2], const float box_before[2], const float ant_pos[2],
                          const float box_pos[2], const float box_quat[4], int64_t reset_in, int64_t progress,
                          float* rew_out, int64_t* reset_out) {
    float quat_dist = box_quat_dist(box_quat, 0.f, 1.f, 0.f);
    float quat_reward = rp->quat_reward_scale * quat_dist;
    float d_now = l2_dist2(ant_pos, box_pos);
    float ant_push = (d_now < 1.5f) ? 0.f : 1.f;
    float ant_dist = l2_dist2(pos_before, box_before) - d_now;
    float ant_dist_reward = rp->ant_dist_reward_scale * ant_dist * ant_push;
    const float box_targets[2] = {0.f, 0.f};
    float gd_before = l2_dist2(box_targets, box_before);
    float gd = l2_dist2(box_targets, box_pos);
    int arrive = gd < 0.5f;
    float goal_dist_reward = rp->goal_dist_reward_scale * (gd_before - gd);
    float goal_arrive_reward = arrive ? 2.f : 0.f;
    float success = ((quat_dist > 0.9f) && arrive) ? 10.f : 0.f;
    float up_reward = (obs[10] > 0.93f) ? (0.f + rp->up_weight) : 0.f;
    float actions_cost = 0.f, electricity = 0.f;
    int lim = 0;
    for (int j = 0; j < 8; j++) {
        actions_cost += actions[j] * actions[j];
        electricity += fabsf(actions[j] * obs[20 + j]);
        lim += (obs[12 + j] > 0.99f) ? 1 : 0;
    }
    float total = 0.5f + up_reward + quat_reward + ant_dist_reward + goal_dist_reward + goal_arrive_reward + success -
                  rp->actions_cost * actions_cost - rp->energy_cost * electricity - (float)lim * rp->joints_at_limit_cost;
    int fallen = obs[0] < rp->termination_height;
    if (fallen) total = rp->death_cost;
    int64_t reset = fallen ? 1 : reset_in;
    if (progress >= (int64_t)rp->max_episode_length - 1) reset = 1;
    *rew_out = total;
    *reset_out = reset;
}
MO_EXPORT void mo_oneant_obs_batch(int64_t n, const float* root, const float* box_root, const float* dof_pos,
                                   const float* dof_vel, const float* lower, const float* upper,
                                   const float* sensors, const float* actions, const float* potentials_in,
                                   float* obs, float* potentials, float* prev_potentials) {
    for (int64_t i = 0; i < n; i++)
        mo_oneant_obs(root + 13 * i, box_root + 13 * i, dof_pos + 8 * i, dof_vel + 8 * i, lower, upper, 0.2f,
                      sensors + 24 * i, 0.1f, actions + 8 * i, 0.0166f, potentials_in[i], obs + 60 * i,
                      potentials + i, prev_potentials + i);
}
MO_EXPORT void mo_oneant_reward_batch(int64_t n, const float* obs, const int64_t* reset_in, const int64_t* progress,
                                      const float* actions, const float* pos_before, const float* box_before,
                                      const float* ant_pos, const float* box_pos, const float* box_quat,
                                      const float* scal, float* rew, int64_t* reset) {
    reward_params rp = {scal[0], scal[1], scal[2], scal[3], scal[4], scal[5], scal[6], scal[7], scal[8], scal[9],
                        (int32_t)scal[10]};
    for (int64_t i = 0; i < n; i++)
        oneant_reward(&rp, obs + 60 * i, actions + 8 * i, pos_before + 2 * i, box_before + 2 * i, ant_pos + 2 * i,
                      box_pos + 2 * i, box_quat + 4 * i, reset_in[i], progress[i], rew + i, reset + i);
}

/* ------------------------------------------------------------------------------------------ */
/* MultiIngenuity (multi_ingenuity.py:268-339 thrust, :381-453 reward)                         */
/* ------------------------------------------------------------------------------------------ */
/* thrusts[8][3] for one env from actions[24]: rotor (2h + r) uses actions[6h + 3r .. 6h + 3r + 2] */
MO_EXPORT void mo_ingenuity_thrust(const float actions[24], float dt, float thrusts[24]) { ingenuity_thrust_real(actions, dt, thrusts); }
MO_EXPORT void mo_ingenuity_thrust_batch(int64_t n, const float* actions, float dt, float* thrusts) {
    for (int64_t i = 0; i < n; i++) mo_ingenuity_thrust(actions + 24 * i, dt, thrusts + 24 * i);
}
static void ingenuity_reward(const float* roots /*[4][13] global*/, int32_t max_len, int64_t progress,
                             float* rew_out, int64_t* reset_out) {
    static const float goals[4][3] = {{4, 2, 1}, {4, -2, 1}, {4, 6, 1}, {4, -6, 1}}; /* multi_ingenuity.py:103-106 */
    float pos_reward = 0.f, up_reward = 0.f, spin_reward = 0.f;
    int die = 0, low = 0;
    for (int k = 0; k < 4; k++) {
        const float* r = roots + 13 * k;
        float dx = goals[k][0] - r[0], dy = goals[k][1] - r[1], dz = goals[k][2] - r[2];
        float td = sqrtf(dx * dx + dy * dy + dz * dz);
        float pr = 1.0f / (1.0f + td * td);
        pos_reward = (k == 0) ? pr : pos_reward + pr;
        const float e2[3] = {0, 0, 1};
        float ups[3];
        mo_quat_rotate(r + 3, e2, ups);
        float tilt = fabsf(1.f - ups[2]);
        float ur = 5.0f / (1.0f + tilt * tilt);
        up_reward = (k == 0) ? ur : up_reward + ur;
        float spin = fabsf(r[12]);
        float sr = 1.0f / (1.0f + spin * spin);
        spin_reward = (k == 0) ? sr : spin_reward + sr;
        die = die || (td > 8.0f);
        low = low || (r[2] < 0.5f);
    }
    *rew_out = pos_reward + pos_reward * (up_reward + spin_reward);
    int64_t d = (die || low) ? 1 : 0;
    *reset_out = (progress >= (int64_t)max_len - 1) ? 1 : d;
}
MO_EXPORT void mo_ingenuity_reward_batch(int64_t n, const float* roots, const int64_t* progress, float* rew, int64_t* reset) {
    for (int64_t i = 0; i < n; i++) ingenuity_reward(roots + 52 * i, 1000, progress[i], rew + i, reset + i);
}

/* ------------------------------------------------------------------------------------------ */
/* wrappers: multi_vec_task.py:105-142                                                         */
/* ------------------------------------------------------------------------------------------ */
MO_EXPORT void mo_marl_views(int64_t n, int agents, int per_agent, int shared, float clip, const float* obs_buf,
                             float* obs_all /*[n][agents][per_agent+shared]*/) {
    int row = agents * per_agent + shared, w = per_agent + shared;
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < agents; k++) {
            for (int j = 0; j < per_agent; j++)
                obs_all[(i * agents + k) * w + j] = clampf(obs_buf[i * row + k * per_agent + j], -clip, clip);
            for (int j = 0; j < shared; j++)
                obs_all[(i * agents + k) * w + per_agent + j] = clampf(obs_buf[i * row + agents * per_agent + j], -clip, clip);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* GAE: storage.py:51-65 and separated_buffer.py:153-164                                       */
/* ------------------------------------------------------------------------------------------ */
MO_EXPORT void mo_gae_ppo(int T, int64_t N, const float* rewards, const uint8_t* dones, const float* values,
                          const float* last_values, float gamma, float lam, float* returns, float* advantages,
                          int normalize) {
    for (int64_t i = 0; i < N; i++) {
        float adv = 0.f;
        for (int t = T - 1; t >= 0; t--) {
            float next_v = (t == T - 1) ? last_values[i] : values[(t + 1) * N + i];
            float nt = 1.0f - (float)dones[t * N + i];
            float delta = rewards[t * N + i] + nt * gamma * next_v - values[t * N + i];
            adv = delta + nt * gamma * lam * adv;
            returns[t * N + i] = adv + values[t * N + i];
        }
    }
    int64_t cnt = (int64_t)T * N;
    double s = 0.0, s2 = 0.0;
    for (int64_t j = 0; j < cnt; j++) {
        advantages[j] = returns[j] - values[j];
        s += advantages[j];
    }
    if (!normalize) return;
    double mean = s / (double)cnt;
    for (int64_t j = 0; j < cnt; j++) { double d = advantages[j] - mean; s2 += d * d; }
    double std = sqrt(s2 / (double)(cnt - 1));                 /* torch.std: unbiased */
    for (int64_t j = 0; j < cnt; j++) advantages[j] = (float)(((double)advantages[j] - mean) / (std + 1e-8));
}
MO_EXPORT void mo_gae_marl(int T, int64_t N, const float* rewards, const float* value_preds /*[T+1][N]*/,
                           const float* masks /*[T+1][N]*/, float gamma, float lam, int use_norm, float mean,
                           float var, float* returns /*[T+1][N]*/) {
    float sd = sqrtf(var);
    for (int64_t i = 0; i < N; i++) {
        float gae = 0.f;
        for (int t = T - 1; t >= 0; t--) {
            float v1 = value_preds[(t + 1) * N + i], v0 = value_preds[t * N + i];
            if (use_norm) { v1 = v1 * sd + mean; v0 = v0 * sd + mean; }
            float delta = rewards[t * N + i] + gamma * v1 * masks[(t + 1) * N + i] - v0;
            gae = delta + gamma * lam * masks[(t + 1) * N + i] * gae;
            returns[t * N + i] = gae + v0;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* counter-based RNG for reset noise (replaces the two torch.rand draws of ten_ant.py:822-823;  */
/* shard-invariant: keyed by the GLOBAL env index)                                             */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
MO_EXPORT float mo_rand_uniform(uint64_t seed, uint64_t env_global, uint64_t step, uint32_t k) { /* step = the env's reset count */
    uint32_t x = mix32((uint32_t)seed ^ 0x9E3779B9U);
    x = mix32(x ^ (uint32_t)(seed >> 32));
    x = mix32(x ^ (uint32_t)env_global);
    x = mix32(x ^ (uint32_t)(env_global >> 32) ^ 0x85EBCA6BU);
    x = mix32(x ^ (uint32_t)step);
    x = mix32(x ^ (uint32_t)(step >> 32) ^ (k * 0xC2B2AE35U));
    return (float)(x >> 8) * (1.0f / 16777216.0f);             /* [0,1), 24 bits like torch.rand */
}

/* standard normal by Box-Muller from two counter-based uniforms; the first is moved off zero */
MO_EXPORT float mo_rand_normal(uint64_t seed, uint64_t row, uint64_t counter, uint32_t k) {
    float u1 = mo_rand_uniform(seed, row, counter, 2u * k) + (0.5f / 16777216.0f);
    float u2 = mo_rand_uniform(seed, row, counter, 2u * k + 1u);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

/* The sampling tail of ActorCritic.act (agents/algorithms/rl/ppo/module.py:73-87):
 *   covariance = diag(exp(log_std) * exp(log_std)); MultivariateNormal(mean, scale_tril=covariance)   (:76-77)
 *   actions = sample(); log_prob(actions); sigma returned = log_std.repeat(N, 1)                      (:79-87)
 * scale_tril = diag(s) gives x = mean + s * eps and log_prob = sum(-0.5 eps^2 - log s - 0.5 log 2 pi).
 * reference_scale = 0 uses s = exp(log_std) (the conventional reading); the noise is this build's counter-based stream. */
MO_EXPORT void mo_ppo_act(int64_t n, int A, const float* mean, const float* log_std, uint64_t seed, int64_t* counters, int64_t row_offset,
                          int reference_scale, float* actions, float* logp, float* sigma) {
    for (int64_t i = 0; i < n; i++) {
        float lp = 0.f, part[64];
        for (int l = 0; l < 64; l++) part[l] = 0.f;
        for (int j = 0; j < A; j++) {
            float ls = log_std[j], scale, lscale;
            if (reference_scale) { float sd = expf(ls); scale = sd * sd; lscale = logf(scale); }
            else { scale = expf(ls); lscale = ls; }
            float z = mo_rand_normal(seed, (uint64_t)(row_offset + i), (uint64_t)counters[i], (uint32_t)j);
            actions[i * A + j] = mean[i * A + j] + scale * z;
            part[j & 63] += -0.5f * z * z - lscale - 0.9189385332046727f;
            sigma[i * A + j] = ls;
        }
        for (int m = 32; m >= 1; m >>= 1)                    /* the kernel's butterfly order */
            for (int l = 0; l < 64; l++) if ((l & m) == 0) { float t = part[l] + part[l | m]; part[l] = t; part[l | m] = t; }
        lp = part[0];
        logp[i] = lp;
        counters[i] += 1;
    }
}

/* log_prob and entropy of given actions under the same distribution (ActorCritic.evaluate, module.py:93-109):
 * entropy of N(mean, diag(s^2)) = sum(0.5 + 0.5 log(2 pi) + log s), the same for every row. */
MO_EXPORT void mo_ppo_log_prob(int64_t n, int A, const float* mean, const float* log_std, const float* actions, int reference_scale,
                               float* logp, float* entropy) {
    for (int64_t i = 0; i < n; i++) {
        float lp = 0.f, ent = 0.f;
        for (int j = 0; j < A; j++) {
            float ls = log_std[j], scale, lscale;
            if (reference_scale) { float sd = expf(ls); scale = sd * sd; lscale = logf(scale); }
            else { scale = expf(ls); lscale = ls; }
            float z = (actions[i * A + j] - mean[i * A + j]) / scale;
            lp += -0.5f * z * z - lscale - 0.9189385332046727f;
            ent += 0.5f + 0.9189385332046727f + lscale;
        }
        logp[i] = lp;
        if (entropy) entropy[i] = ent;
    }
}

#endif /* !MO_F64 */
/* ------------------------------------------------------------------------------------------ */
/* PHYSICS (this build's model; parity unpinned against Isaac Gym -- see header)               */
/* ------------------------------------------------------------------------------------------ */
typedef real m66[6][6];

static void m66_zero(m66 M) { memset(M, 0, sizeof(m66)); }
static void m66_add(m66 A, const m66 B) { for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) A[i][j] += B[i][j]; }
static void m66_mulv(const m66 M, const real v[6], real o[6]) {
    real t[6];
    for (int i = 0; i < 6; i++) { real s = 0.f; for (int j = 0; j < 6; j++) s += M[i][j] * v[j]; t[i] = s; }
    memcpy(o, t, sizeof(t));
}
/* spatial inertia about the frame origin for mass m, COM offset c, rotational inertia Ic about the COM */
static void spatial_inertia(real m, const real c[3], const real Ic[3][3], m66 I) {
    real cc = dot3(c, c);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            I[i][j] = Ic[i][j] + m * ((i == j ? cc : 0.f) - c[i] * c[j]);
            I[3 + i][3 + j] = (i == j) ? m : 0.f;
        }
    real cx[3][3] = {{0.f, -c[2], c[1]}, {c[2], 0.f, -c[0]}, {-c[1], c[0], 0.f}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { I[i][3 + j] = m * cx[i][j]; I[3 + j][i] = m * cx[i][j]; }
}
/* capsule / axisymmetric body: Ic = It*1 + (Ia - It) u u^T */
static void axisym_inertia(real Ia, real It, const real u[3], real Ic[3][3]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ic[i][j] = (i == j ? It : 0.f) + (Ia - It) * u[i] * u[j];
}
/* bias force p = v x* (I v) - (c x m g, m g) with g = (0,0,-grav) */
static void bias_force(const m66 I, const real v[6], real m, const real c[3], real grav, real p[6]) {
    real h[6];
    m66_mulv(I, v, h);
    real t1[3], t2[3], t3[3];
    cross3(v, h, t1);          /* w x n */
    cross3(v + 3, h + 3, t2);  /* v x f */
    cross3(v, h + 3, t3);      /* w x f */
    real fg[3] = {0.f, 0.f, -m * grav}, ng[3];
    cross3(c, fg, ng);
    for (int i = 0; i < 3; i++) { p[i] = t1[i] + t2[i] - ng[i]; p[3 + i] = t3[i] - fg[i]; }
}
/* motion cross product: (w,v) xm (sw,sv) */
static void cross_motion(const real a[6], const real b[6], real o[6]) {
    real t1[3], t2[3], t3[3];
    cross3(a, b, t1);
    cross3(a, b + 3, t2);
    cross3(a + 3, b, t3);
    for (int i = 0; i < 3; i++) { o[i] = t1[i]; o[3 + i] = t2[i] + t3[i]; }
}
/* symmetric positive definite 6x6 solve (LDL^T, no pivoting) */
static void solve6(const m66 A, const real b[6], real x[6]) {
    real L[6][6], D[6], y[6];
    for (int j = 0; j < 6; j++) {
        real d = A[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
        D[j] = d;
        real inv = 1.f / d;
        for (int i = j + 1; i < 6; i++) {
            real s = A[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k] * D[k];
            L[i][j] = s * inv;
        }
    }
    for (int i = 0; i < 6; i++) { real s = b[i]; for (int k = 0; k < i; k++) s -= L[i][k] * y[k]; y[i] = s; }
    for (int i = 5; i >= 0; i--) {
        real s = y[i] / D[i];
        for (int k = i + 1; k < 6; k++) s -= L[k][i] * x[k];
        x[i] = s;
    }
}

/* one active contact on a body of the articulation */
typedef struct {
    int active;
    real xc[3];     /* contact point relative to the spatial origin O */
    real n[3];      /* unit normal, from the other object into this body */
    real kd;        /* k * penetration */
    real gn, ct;    /* normal gain h*k + c; tangential (friction) damper */
    real vrel[3];   /* point velocity relative to the other object's point */
    real f[3];      /* resulting force on this body (filled after the solve) */
} contact_t;

/* I^A += h P^T G P;  p^A -= P^T (kd n - G vrel);  P = [-[xc]x | 1], G = (gn-ct) n n^T + ct 1 */
static void contact_fold(const contact_t* c, real h, m66 IA, real pA[6]) {
    if (!c->active) return;
    real G[3][3], P[3][6];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) G[i][j] = (c->gn - c->ct) * c->n[i] * c->n[j] + (i == j ? c->ct : 0.f);
    const real* x = c->xc;
    real ncx[3][3] = {{0.f, x[2], -x[1]}, {-x[2], 0.f, x[0]}, {x[1], -x[0], 0.f}};   /* -[xc]x */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { P[i][j] = ncx[i][j]; P[i][3 + j] = (i == j) ? 1.f : 0.f; }
    real GP[3][6];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 6; j++) GP[i][j] = G[i][0] * P[0][j] + G[i][1] * P[1][j] + G[i][2] * P[2][j];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) IA[i][j] += h * (P[0][i] * GP[0][j] + P[1][i] * GP[1][j] + P[2][i] * GP[2][j]);
    real Gv[3], f0[3], m0[3];
    matvec3(G, c->vrel, Gv);
    for (int i = 0; i < 3; i++) f0[i] = c->kd * c->n[i] - Gv[i];
    cross3(x, f0, m0);
    for (int i = 0; i < 3; i++) { pA[i] -= m0[i]; pA[3 + i] -= f0[i]; }
}
/* f = kd n - G (vrel + h (a_lin + alpha x xc)) */
static void contact_force(contact_t* c, real h, const real a[6]) {
    if (!c->active) { c->f[0] = c->f[1] = c->f[2] = 0.f; return; }
    real ax[3], u[3];
    cross3(a, c->xc, ax);
    for (int i = 0; i < 3; i++) u[i] = c->vrel[i] + h * (a[3 + i] + ax[i]);
    real un = dot3(c->n, u);
    for (int i = 0; i < 3; i++) c->f[i] = c->kd * c->n[i] - ((c->gn - c->ct) * un * c->n[i] + c->ct * u[i]);
}

typedef struct { real pos[3]; real R[3][3]; real v[3]; real w[3]; real half[3]; } box_pose;

/* A contact is ACTIVE while the pair penetrates (d > 0) or would penetrate by the end of the substep at
 * the current normal velocity (d - h v_n > 0); pairs farther apart than CONTACT_MARGIN are not examined.
 * While active the force k d - (h k + c) v_n(new) is applied two-sidedly: no chatter at rest, no rebound;
 * the price is a viscous adhesion that lasts only while the shapes still overlap. */
#define CONTACT_MARGIN 0.1f
/* Activation weight w = clamp(max(d, d - h v_n) / ramp, 0, 1) scales the whole contact law (spring and damper):
 * the force is then a CONTINUOUS function of the state -- zero at the activation boundary, full strength once
 * the pair overlaps by `ramp` -- so rounding-level differences cannot flip a contact on or off with a force
 * jump.  Joint limits use the same construction. */
static inline real ramp01(real d, real r) { return r_fmin(r_fmax(d / r, 0.f), 1.f); }

/* Sphere (centre xs relative to O, O at world position Ow, radius rad) on a body with spatial velocity vb:
 * contact against the ground plane z = 0 -> cg, and against the box -> cb. */
static void sphere_contacts(const mms_model* M, real h, const real Ow[3], const real xs[3], real rad,
                            const real vb[6], const box_pose* box, contact_t* cg, contact_t* cb) {
    /* ground */
    cg->active = 0;
    {
        real zc = Ow[2] + xs[2];
        real d = rad - zc;
        if (d > -CONTACT_MARGIN) {
            real xc[3] = {xs[0], xs[1], xs[2] - rad};
            real vp[3], wx[3];
            cross3(vb, xc, wx);
            for (int i = 0; i < 3; i++) vp[i] = vb[3 + i] + wx[i];
            real w = ramp01(r_fmax(d, d - h * vp[2]), M->pen_ramp);   /* penetrating now or at the end of the step */
            real gn = w * (h * M->gnd_k + M->gnd_c);
            real fn = r_fmax(w * M->gnd_k * d - gn * vp[2], 0.f);     /* explicit estimate: friction bound only */
            if (w > 0.f) {
                cg->active = 1;
                memcpy(cg->xc, xc, sizeof(xc));
                cg->n[0] = 0.f; cg->n[1] = 0.f; cg->n[2] = 1.f;
                cg->kd = w * M->gnd_k * d;
                cg->gn = gn;
                real vt = r_sqrt(vp[0] * vp[0] + vp[1] * vp[1]);
                cg->ct = M->gnd_mu * fn / r_fmax(vt, M->slip_eps);
                memcpy(cg->vrel, vp, sizeof(vp));
            }
        }
    }
    /* box (frictionless unless model.antbox_mu > 0) */
    cb->active = 0;
    if (box) {
        real rel[3] = {Ow[0] + xs[0] - box->pos[0], Ow[1] + xs[1] - box->pos[1], Ow[2] + xs[2] - box->pos[2]};
        real xb[3], q[3];
        matTvec3(box->R, rel, xb);
        int inside = 1;
        for (int i = 0; i < 3; i++) {
            q[i] = r_clamp(xb[i], -box->half[i], box->half[i]);
            if (q[i] != xb[i]) inside = 0;
        }
        real nb[3] = {0.f, 0.f, 0.f}, d;
        const int inside_far = 0;
        if (!inside) {
            real dl[3] = {xb[0] - q[0], xb[1] - q[1], xb[2] - q[2]};
            real dist = r_sqrt(dot3(dl, dl));
            d = rad - dist;
            if (d > -CONTACT_MARGIN) { nb[0] = dl[0] / dist; nb[1] = dl[1] / dist; nb[2] = dl[2] / dist; }
        } else {
            int ax = 0;
            real best = box->half[0] - r_fabs(xb[0]);
            for (int i = 1; i < 3; i++) {
                real m = box->half[i] - r_fabs(xb[i]);
                if (m < best) { best = m; ax = i; }
            }
            nb[ax] = (xb[ax] >= 0.f) ? 1.f : -1.f;
            d = rad + best;
        }
        if (d > -CONTACT_MARGIN && !inside_far) {
            real n[3];
            matvec3(box->R, nb, n);
            real xc[3] = {xs[0] - rad * n[0], xs[1] - rad * n[1], xs[2] - rad * n[2]};
            real vp[3], wx[3], rb[3], vbx[3];
            cross3(vb, xc, wx);
            for (int i = 0; i < 3; i++) { vp[i] = vb[3 + i] + wx[i]; rb[i] = Ow[i] + xc[i] - box->pos[i]; }
            cross3(box->w, rb, vbx);
            real vrel[3] = {vp[0] - box->v[0] - vbx[0], vp[1] - box->v[1] - vbx[1], vp[2] - box->v[2] - vbx[2]};
            real w = ramp01(r_fmax(d, d - h * dot3(n, vrel)), M->pen_ramp);
            real gn = w * (h * M->antbox_k + M->antbox_c);
            if (w > 0.f) {
                cb->active = 1;
                memcpy(cb->xc, xc, sizeof(xc));
                memcpy(cb->n, n, sizeof(n));
                cb->kd = w * M->antbox_k * d;
                cb->gn = gn;
                cb->ct = 0.f;
                if (M->antbox_mu > 0.f) {                            /* Coulomb friction, regularised like the ground's */
                    real vn = dot3(n, vrel);
                    real fn = r_fmax(w * M->antbox_k * d - gn * vn, 0.f);
                    real vt[3] = {vrel[0] - vn * n[0], vrel[1] - vn * n[1], vrel[2] - vn * n[2]};
                    cb->ct = M->antbox_mu * fn / r_fmax(r_sqrt(dot3(vt, vt)), M->slip_eps);
                }
                memcpy(cb->vrel, vrel, sizeof(vrel));
            }
        }
    }
}

/* One substep of one ant.  root[13]: env-local pos, quat, linvel, angvel (world frame).
 * dof[8][2]: (pos, vel).  tau_motor[8].  box may be NULL.  box_wrench[6] (torque about the box COM,
 * force) is ACCUMULATED.  sensors[4][6]: net contact (force, torque about the foot origin) in the foot frame. */
/* dr: per-ant physical domain randomisation (cfg/TenAnt.yaml:97-122, applied by base_task.py:343-395 through
 * set_actor_rigid_body_properties / set_actor_dof_properties), MMS_DR_FLOATS values or NULL for the nominal ant:
 *   [0] torso, [1..4] leg, [5..8] foot mass scale (inertia scales with the mass: recomputeInertia = True is the setter's
 *   default argument), [9..16] joint damping scale, [17..24] lower-limit offset, [25..32] upper-limit offset (rad). */
static void ant_substep(const mms_model* M, real h, real root[13], real dof[8][2], const real tau_motor[8],
                        const box_pose* box, real box_wrench[6], real sensors[4][6], const float* dr) {
    real Rt[3][3];
    quat_to_mat(root + 3, Rt);
    const real* Ow = root;
    real v0[6] = {root[10], root[11], root[12], root[7], root[8], root[9]};  /* (w, vO) */
    const real zero3[3] = {0.f, 0.f, 0.f};

    /* ---- torso ---- */
    m66 IA0;
    real pA0[6];
    {
        real ez[3] = {Rt[0][2], Rt[1][2], Rt[2][2]};
        real Ic[3][3];
        const real mt = dr ? dr[0] : 1.f;
        axisym_inertia(M->torso_izz * mt, M->torso_ixx * mt, ez, Ic);
        spatial_inertia(M->torso_mass * mt, zero3, Ic, IA0);
        bias_force(IA0, v0, M->torso_mass * mt, zero3, M->gravity, pA0);
    }
    contact_t ct_g, ct_b;
    sphere_contacts(M, h, Ow, zero3, M->torso_radius, v0, box, &ct_g, &ct_b);
    contact_fold(&ct_g, h, IA0, pA0);
    contact_fold(&ct_b, h, IA0, pA0);

    /* per-leg data kept for the outward pass */
    real s1[4][6], s2[4][6], c1[4][6], c2[4][6], U1[4][6], U2[4][6], D1[4], D2[4], u1[4], u2[4];
    real Rf[4][3][3], J2[4][3];
    contact_t chip_g[4], chip_b[4], cknee_g[4], cknee_b[4], ctip_g[4], ctip_b[4];

    for (int l = 0; l < 4; l++) {
        real q1 = dof[2 * l][0], qd1 = dof[2 * l][1], q2 = dof[2 * l + 1][0], qd2 = dof[2 * l + 1][1];
        /* kinematics */
        real J1[3], a1[3] = {Rt[0][2], Rt[1][2], Rt[2][2]};
        R3(hip_pos, M->hip_pos[l]); R3(limb_dir, M->limb_dir[l]); R3(ankle_axis, M->ankle_axis[l]);
        matvec3(Rt, hip_pos, J1);
        real Rz[3][3], Rl[3][3], Ra[3][3];
        const real zax[3] = {0.f, 0.f, 1.f};
        axis_angle_to_mat(zax, q1, Rz);
        matmul3(Rt, Rz, Rl);
        real ul[3], a2[3], uf[3];
        matvec3(Rl, limb_dir, ul);
        matvec3(Rl, ankle_axis, a2);
        for (int i = 0; i < 3; i++) J2[l][i] = J1[i] + M->leg_len * ul[i];
        axis_angle_to_mat(ankle_axis, q2, Ra);
        matmul3(Rl, Ra, Rf[l]);
        matvec3(Rf[l], limb_dir, uf);
        real cl[3], cf[3], tip[3];
        for (int i = 0; i < 3; i++) {
            cl[i] = J1[i] + 0.5f * M->leg_len * ul[i];
            cf[i] = J2[l][i] + 0.5f * M->foot_len * uf[i];
            tip[i] = J2[l][i] + M->foot_len * uf[i];
        }
        /* motion subspaces and velocities */
        real t3[3];
        cross3(J1, a1, t3);
        for (int i = 0; i < 3; i++) { s1[l][i] = a1[i]; s1[l][3 + i] = t3[i]; }
        cross3(J2[l], a2, t3);
        for (int i = 0; i < 3; i++) { s2[l][i] = a2[i]; s2[l][3 + i] = t3[i]; }
        real vl[6], vf[6], sq[6];
        for (int i = 0; i < 6; i++) { sq[i] = s1[l][i] * qd1; vl[i] = v0[i] + sq[i]; }
        cross_motion(v0, sq, c1[l]);
        for (int i = 0; i < 6; i++) { sq[i] = s2[l][i] * qd2; vf[i] = vl[i] + sq[i]; }
        cross_motion(vl, sq, c2[l]);
        /* inertias and bias forces */
        m66 IAl, IAf;
        real pAl[6], pAf[6], Ic[3][3];
        const real ml = dr ? dr[1 + l] : 1.f, mf = dr ? dr[5 + l] : 1.f;
        axisym_inertia(M->leg_ia * ml, M->leg_it * ml, ul, Ic);
        spatial_inertia(M->leg_mass * ml, cl, Ic, IAl);
        bias_force(IAl, vl, M->leg_mass * ml, cl, M->gravity, pAl);
        axisym_inertia(M->foot_ia * mf, M->foot_it * mf, uf, Ic);
        spatial_inertia(M->foot_mass * mf, cf, Ic, IAf);
        bias_force(IAf, vf, M->foot_mass * mf, cf, M->gravity, pAf);
        /* contacts: hip and knee spheres on the leg body, tip sphere on the foot body */
        sphere_contacts(M, h, Ow, J1, M->limb_radius, vl, box, &chip_g[l], &chip_b[l]);
        sphere_contacts(M, h, Ow, J2[l], M->limb_radius, vl, box, &cknee_g[l], &cknee_b[l]);
        sphere_contacts(M, h, Ow, tip, M->limb_radius, vf, box, &ctip_g[l], &ctip_b[l]);
        contact_fold(&chip_g[l], h, IAl, pAl);  contact_fold(&chip_b[l], h, IAl, pAl);
        contact_fold(&cknee_g[l], h, IAl, pAl); contact_fold(&cknee_b[l], h, IAl, pAl);
        contact_fold(&ctip_g[l], h, IAf, pAf);  contact_fold(&ctip_b[l], h, IAf, pAf);

        /* joint torques with linearly-implicit damping and limits */
        real tau[2], Dextra[2];
        for (int j = 0; j < 2; j++) {
            int d = 2 * l + j;
            real q = dof[d][0], qd = dof[d][1];
            const real damping = dr ? M->joint_damping * dr[9 + d] : M->joint_damping;
            const real upper = dr ? M->dof_upper[d] + dr[25 + d] : M->dof_upper[d];
            const real lower = dr ? M->dof_lower[d] + dr[17 + d] : M->dof_lower[d];
            real t = tau_motor[d] - damping * qd;
            real De = M->armature + h * damping;
            real ehi = q - upper, elo = lower - q;
            real whi = ramp01(r_fmax(ehi, ehi + h * qd), M->limit_ramp), wlo = ramp01(r_fmax(elo, elo - h * qd), M->limit_ramp);
            if (whi > 0.f) {
                real gl = whi * (h * M->limit_k + M->limit_c);
                t += -whi * M->limit_k * ehi - gl * qd; De += h * gl;
            } else if (wlo > 0.f) {
                real gl = wlo * (h * M->limit_k + M->limit_c);
                t += wlo * M->limit_k * elo - gl * qd; De += h * gl;
            }
            tau[j] = t; Dextra[j] = De;
        }
        /* inward: foot */
        m66_mulv(IAf, s2[l], U2[l]);
        D2[l] = Dextra[1];
        for (int i = 0; i < 6; i++) D2[l] += s2[l][i] * U2[l][i];
        u2[l] = tau[1];
        for (int i = 0; i < 6; i++) u2[l] -= s2[l][i] * pAf[i];
        {
            real invD = 1.f / D2[l];
            m66 Ia;
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Ia[i][j] = IAf[i][j] - U2[l][i] * U2[l][j] * invD;
            real Iac[6];
            m66_mulv(Ia, c2[l], Iac);
            m66_add(IAl, Ia);
            for (int i = 0; i < 6; i++) pAl[i] += pAf[i] + Iac[i] + U2[l][i] * (u2[l] * invD);
        }
        /* inward: leg */
        m66_mulv(IAl, s1[l], U1[l]);
        D1[l] = Dextra[0];
        for (int i = 0; i < 6; i++) D1[l] += s1[l][i] * U1[l][i];
        u1[l] = tau[0];
        for (int i = 0; i < 6; i++) u1[l] -= s1[l][i] * pAl[i];
        {
            real invD = 1.f / D1[l];
            m66 Ia;
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Ia[i][j] = IAl[i][j] - U1[l][i] * U1[l][j] * invD;
            real Iac[6];
            m66_mulv(Ia, c1[l], Iac);
            m66_add(IA0, Ia);
            for (int i = 0; i < 6; i++) pA0[i] += pAl[i] + Iac[i] + U1[l][i] * (u1[l] * invD);
        }
    }
    /* root */
    real a0[6], rhs[6];
    for (int i = 0; i < 6; i++) rhs[i] = -pA0[i];
    solve6(IA0, rhs, a0);

    /* outward + contact forces */
    contact_force(&ct_g, h, a0);
    contact_force(&ct_b, h, a0);
    real fbox[3] = {0, 0, 0}, tbox[3] = {0, 0, 0};
#define ACC_BOX(c)                                                                          \
    if ((c).active && box) {                                                                \
        real rb_[3] = {Ow[0] + (c).xc[0] - box->pos[0], Ow[1] + (c).xc[1] - box->pos[1],   \
                        Ow[2] + (c).xc[2] - box->pos[2]};                                   \
        real nf_[3] = {-(c).f[0], -(c).f[1], -(c).f[2]}, tq_[3];                           \
        cross3(rb_, nf_, tq_);                                                              \
        for (int i_ = 0; i_ < 3; i_++) { fbox[i_] += nf_[i_]; tbox[i_] += tq_[i_]; }         \
    }
    ACC_BOX(ct_b);
    real qdd[8];
    for (int l = 0; l < 4; l++) {
        real al[6], af[6];
        for (int i = 0; i < 6; i++) al[i] = a0[i] + c1[l][i];
        real t = u1[l];
        for (int i = 0; i < 6; i++) t -= U1[l][i] * al[i];
        qdd[2 * l] = t / D1[l];
        for (int i = 0; i < 6; i++) al[i] += s1[l][i] * qdd[2 * l];
        for (int i = 0; i < 6; i++) af[i] = al[i] + c2[l][i];
        t = u2[l];
        for (int i = 0; i < 6; i++) t -= U2[l][i] * af[i];
        qdd[2 * l + 1] = t / D2[l];
        for (int i = 0; i < 6; i++) af[i] += s2[l][i] * qdd[2 * l + 1];
        contact_force(&chip_g[l], h, al);  contact_force(&chip_b[l], h, al);
        contact_force(&cknee_g[l], h, al); contact_force(&cknee_b[l], h, al);
        contact_force(&ctip_g[l], h, af);  contact_force(&ctip_b[l], h, af);
        ACC_BOX(chip_b[l]); ACC_BOX(cknee_b[l]); ACC_BOX(ctip_b[l]);
        if (sensors) {
            real F[3] = {0, 0, 0}, T[3] = {0, 0, 0};
            contact_t* cs[2] = {&ctip_g[l], &ctip_b[l]};
            for (int k = 0; k < 2; k++)
                if (cs[k]->active) {
                    real r[3] = {cs[k]->xc[0] - J2[l][0], cs[k]->xc[1] - J2[l][1], cs[k]->xc[2] - J2[l][2]}, tq[3];
                    cross3(r, cs[k]->f, tq);
                    for (int i = 0; i < 3; i++) { F[i] += cs[k]->f[i]; T[i] += tq[i]; }
                }
            matTvec3(Rf[l], F, sensors[l]);
            matTvec3(Rf[l], T, sensors[l] + 3);
        }
    }
#undef ACC_BOX
    if (box_wrench) for (int i = 0; i < 3; i++) { box_wrench[i] += tbox[i]; box_wrench[3 + i] += fbox[i]; }

    /* integrate (semi-implicit Euler) */
    for (int d = 0; d < 8; d++) {
        dof[d][1] += h * qdd[d];
        dof[d][0] += h * dof[d][1];
    }
    real wxv[3];
    cross3(v0, v0 + 3, wxv);                                     /* classical accel = a_lin + w x v */
    for (int i = 0; i < 3; i++) {
        root[7 + i] += h * (a0[3 + i] + wxv[i]);
        root[10 + i] += h * a0[i];
    }
    real wn = r_sqrt(root[10] * root[10] + root[11] * root[11] + root[12] * root[12]);
    if (wn > 64.f) { real s = 64.f / wn; root[10] *= s; root[11] *= s; root[12] *= s; }  /* PhysX default max angular velocity */
    for (int i = 0; i < 3; i++) root[i] += h * root[7 + i];
    quat_integrate(root + 3, root + 10, h);
}

/* One substep of the free box: gravity, explicit ant reaction wrench, implicit frictionless corner contacts. */
static void box_substep(const mms_model* M, real h, real root[13], const real wrench[6]) {
    real R[3][3];
    quat_to_mat(root + 3, R);
    real Iw[3][3], RI[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RI[i][j] = R[i][j] * M->box_inertia[j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Iw[i][j] = RI[i][0] * R[j][0] + RI[i][1] * R[j][1] + RI[i][2] * R[j][2];
    m66 A;
    m66_zero(A);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A[i][j] = Iw[i][j]; A[3 + i][3 + i] = M->box_mass; }
    real w[3] = {root[10], root[11], root[12]}, v[3] = {root[7], root[8], root[9]};
    real Iww[3], gyro[3];
    matvec3(Iw, w, Iww);
    cross3(w, Iww, gyro);
    real b[6] = {wrench[0] - gyro[0], wrench[1] - gyro[1], wrench[2] - gyro[2],
                  wrench[3], wrench[4], wrench[5] - M->box_mass * M->gravity};
    real vb[6] = {w[0], w[1], w[2], v[0], v[1], v[2]};
    for (int c = 0; c < 8; c++) {
        real loc[3] = {(c & 1 ? 1.f : -1.f) * M->box_half[0], (c & 2 ? 1.f : -1.f) * M->box_half[1],
                        (c & 4 ? 1.f : -1.f) * M->box_half[2]};
        real xc[3];
        matvec3(R, loc, xc);
        real d = -(root[2] + xc[2]);
        if (d <= -CONTACT_MARGIN) continue;
        real wx[3], vp[3];
        cross3(w, xc, wx);
        for (int i = 0; i < 3; i++) vp[i] = v[i] + wx[i];
        real w = ramp01(r_fmax(d, d - h * vp[2]), M->pen_ramp);
        real gn = w * (h * M->boxgnd_k + M->boxgnd_c);
        if (!(w > 0.f)) continue;
        contact_t ct;
        ct.active = 1;
        memcpy(ct.xc, xc, sizeof(xc));
        ct.n[0] = 0.f; ct.n[1] = 0.f; ct.n[2] = 1.f;
        ct.kd = w * M->boxgnd_k * d; ct.gn = gn; ct.ct = 0.f;
        if (M->boxgnd_mu > 0.f) {                                    /* optional Coulomb friction, regularised like the ants' */
            real fn = r_fmax(w * M->boxgnd_k * d - gn * vp[2], 0.f);
            real vt = r_sqrt(vp[0] * vp[0] + vp[1] * vp[1]);
            ct.ct = M->boxgnd_mu * fn / r_fmax(vt, M->slip_eps);
        }
        memcpy(ct.vrel, vp, sizeof(vp));
        real p[6] = {0, 0, 0, 0, 0, 0};
        contact_fold(&ct, h, A, p);
        for (int i = 0; i < 6; i++) b[i] -= p[i];
    }
    (void)vb;
    real a[6];
    solve6(A, b, a);
    for (int i = 0; i < 3; i++) { root[10 + i] += h * a[i]; root[7 + i] += h * a[3 + i]; }
    real wn = r_sqrt(root[10] * root[10] + root[11] * root[11] + root[12] * root[12]);
    if (wn > 64.f) { real s = 64.f / wn; root[10] *= s; root[11] *= s; root[12] *= s; }
    for (int i = 0; i < 3; i++) root[i] += h * root[7 + i];
    quat_integrate(root + 3, root + 10, h);
}

/* One substep of one helicopter (single rigid body).  thrust[2][3]: rotor forces in the body frame. */
static void heli_substep(const mms_model* M, real h, real root[13], const real thrust[2][3]) {
    real R[3][3];
    quat_to_mat(root + 3, R);
    /* reference point = body origin; COM at (0,0,com_z) in the body frame */
    real cb[3] = {0.f, 0.f, M->heli_com_z}, c[3];
    matvec3(R, cb, c);
    real Iw[3][3], RI[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) RI[i][j] = R[i][j] * M->heli_inertia[j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Iw[i][j] = RI[i][0] * R[j][0] + RI[i][1] * R[j][1] + RI[i][2] * R[j][2];
    m66 A;
    spatial_inertia(M->heli_mass, c, Iw, A);
    real v0[6] = {root[10], root[11], root[12], root[7], root[8], root[9]};
    real p[6];
    bias_force(A, v0, M->heli_mass, c, M->gravity, p);
    for (int r = 0; r < 2; r++) {
        real xb[3] = {0.f, 0.f, M->heli_rotor_z[r]}, x[3], f[3], m[3];
        matvec3(R, xb, x);
        matvec3(R, thrust[r], f);
        cross3(x, f, m);
        for (int i = 0; i < 3; i++) { p[i] -= m[i]; p[3 + i] -= f[i]; }
    }
    for (int k = 0; k < 8; k++) {                      /* chassis corners against the ground */
        real loc[3] = {(k & 1 ? 1.f : -1.f) * M->heli_half, (k & 2 ? 1.f : -1.f) * M->heli_half,
                        (k & 4 ? 1.f : -1.f) * M->heli_half};
        real xc[3];
        matvec3(R, loc, xc);
        real d = -(root[2] + xc[2]);
        if (d <= -CONTACT_MARGIN) continue;
        real wx[3], vp[3];
        cross3(v0, xc, wx);
        for (int i = 0; i < 3; i++) vp[i] = v0[3 + i] + wx[i];
        real w = ramp01(r_fmax(d, d - h * vp[2]), M->pen_ramp);
        real gn = w * (h * M->heli_gnd_k + M->heli_gnd_c);
        real fn = r_fmax(w * M->heli_gnd_k * d - gn * vp[2], 0.f);
        if (!(w > 0.f)) continue;
        contact_t ct;
        ct.active = 1;
        memcpy(ct.xc, xc, sizeof(xc));
        ct.n[0] = 0.f; ct.n[1] = 0.f; ct.n[2] = 1.f;
        ct.kd = w * M->heli_gnd_k * d; ct.gn = gn;
        real vt = r_sqrt(vp[0] * vp[0] + vp[1] * vp[1]);
        ct.ct = M->gnd_mu * fn / r_fmax(vt, M->slip_eps);
        memcpy(ct.vrel, vp, sizeof(vp));
        contact_fold(&ct, h, A, p);
    }
    real a[6], rhs[6];
    for (int i = 0; i < 6; i++) rhs[i] = -p[i];
    solve6(A, rhs, a);
    real wxv[3];
    cross3(v0, v0 + 3, wxv);
    for (int i = 0; i < 3; i++) { root[7 + i] += h * (a[3 + i] + wxv[i]); root[10 + i] += h * a[i]; }
    real wn = r_sqrt(root[10] * root[10] + root[11] * root[11] + root[12] * root[12]);
    if (wn > M->heli_max_angvel) { real s = M->heli_max_angvel / wn; root[10] *= s; root[11] *= s; root[12] *= s; }
    for (int i = 0; i < 3; i++) root[i] += h * root[7 + i];
    quat_integrate(root + 3, root + 10, h);
}

/* The physics of one env for one control step (all substeps), in place on `real` state: roots [actors][13] (env-local frame),
 * dofs [dofs][2], sensors [A][24] (ant tasks; may be NULL).  act: the env's raw actions; dr: [A][MMS_DR_FLOATS] or NULL. */
static void physics_core(const mms_config* c, int A, const float* act_in, real* roots, real* dofs, real* sensors, const float* dr) {
    const mms_model* M = &c->model;
    const real h = (real)c->dt / (real)c->substeps;
    if (c->task == MMS_TASK_TEN_ANT || c->task == MMS_TASK_ONE_ANT || c->task == MMS_TASK_MULTI_ANT_CIRCLE) {
        for (int s = 0; s < c->substeps; s++) {
            real* br = roots + 13 * A;
            box_pose box;
            for (int j = 0; j < 3; j++) { box.pos[j] = br[j]; box.v[j] = br[7 + j]; box.w[j] = br[10 + j]; box.half[j] = M->box_half[j]; }
            quat_to_mat(br + 3, box.R);
            real wrench[6] = {0, 0, 0, 0, 0, 0};
            for (int k = 0; k < A; k++) {
                real tau[8];
                for (int j = 0; j < 8; j++) {
                    float a = fmaxf(fminf(act_in[8 * k + j], c->clip_actions), -c->clip_actions);
                    tau[j] = (real)a * M->gear[j] * c->power_scale;            /* ten_ant.py:889 */
                }
                ant_substep(M, h, roots + 13 * k, (real(*)[2])(dofs + 16 * k), tau, &box, wrench,
                            sensors ? (real(*)[6])(sensors + (size_t)k * 24) : NULL, dr ? dr + (size_t)k * MMS_DR_FLOATS : NULL);
            }
            box_substep(M, h, br, wrench);
        }
    } else {
        real thr[24], act[24];
        for (int j = 0; j < 24; j++) act[j] = fmaxf(fminf(act_in[j], c->clip_actions), -c->clip_actions);
        ingenuity_thrust_real(act, c->dt, thr);
        for (int s = 0; s < c->substeps; s++)
            for (int k = 0; k < A; k++) {
                heli_substep(M, h, roots + 13 * k, (const real(*)[3])(thr + 6 * k));
                for (int j = 0; j < 4; j++) dofs[2 * (4 * k + j)] += h * dofs[2 * (4 * k + j) + 1];  /* visual rotors: kinematic */
            }
    }
}

#ifdef MO_F64
/* The double-precision yardstick: physics of n envs from fp32 state (as the fp32 engines hold it), results in double.
 * Envs with reset[i] != 0 skip the physics, as in mo_step.  sens_in / sens_out [n][A][24] may be NULL (not an ant task). */
MO_EXPORT void mo_physics_f64(const mms_config* c, int64_t n, const float* actions, const float* root_in, const float* dof_in,
                              const float* sens_in, const float* dr, const int64_t* reset, double* root_out, double* dof_out,
                              double* sens_out) {
    int A = c->num_agents;
    int ant = (c->task == MMS_TASK_TEN_ANT || c->task == MMS_TASK_ONE_ANT || c->task == MMS_TASK_MULTI_ANT_CIRCLE);
    int actors = ant ? A + 1 : A, dofs = ant ? 8 * A : 4 * A, nact = ant ? 8 * A : 6 * A;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double* r = root_out + (size_t)i * actors * 13;
        double* d = dof_out + (size_t)i * dofs * 2;
        double* sn = (sens_out && ant) ? sens_out + (size_t)i * A * 24 : NULL;
        for (int j = 0; j < actors * 13; j++) r[j] = root_in[(size_t)i * actors * 13 + j];
        for (int j = 0; j < dofs * 2; j++) d[j] = dof_in[(size_t)i * dofs * 2 + j];
        if (sn) for (int j = 0; j < A * 24; j++) sn[j] = sens_in ? sens_in[(size_t)i * A * 24 + j] : 0.0;
        if (reset[i] == 0) physics_core(c, A, actions + (size_t)i * nact, r, d, sn, dr ? dr + (size_t)i * A * MMS_DR_FLOATS : NULL);
    }
}
MO_EXPORT int mo_abi_version(void) { return MMS_ABI_VERSION; }
MO_EXPORT int mo_sizeof_config(void) { return (int)sizeof(mms_config); }
#endif

#ifndef MO_F64   /* engine + unit-level entry points: fp32 build only */
/* ------------------------------------------------------------------------------------------ */
/* engine: same buffers and step protocol as the product (include/mms.h)                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct mo_engine {
    mms_config cfg;
    int N, A, actors, dofs_per_env, num_actions, obs_dim, prev_dim;
    float *actions, *obs, *obs_clipped, *rew, *root_states, *initial_root_states, *dof_state, *env_origin, *prev,
        *reset_noise, *foot_sensors;
    int64_t *reset, *progress, *reset_count;
    float* dr_params;          /* [N][A][MMS_DR_FLOATS] */
    int dr_enabled;
} mo_engine;

static int is_ant_task(int task) { return task == MMS_TASK_TEN_ANT || task == MMS_TASK_ONE_ANT || task == MMS_TASK_MULTI_ANT_CIRCLE; }

/* global-frame copy of a root row: position + env origin, rest unchanged */
static void to_global(const float* root_local, const float* origin, float out[13]) {
    memcpy(out, root_local, 13 * 4);
    out[0] = root_local[0] + origin[0]; out[1] = root_local[1] + origin[1]; out[2] = root_local[2] + origin[2];
}

/* first-step cache initialisation = what reset_idx reads from the stale wrapped tensors
 * (ten_ant.py:870-882; one_ant.py:410-411): construction-time poses. */
static void init_prev_from_initial(mo_engine* e, int i) {
    const mms_config* c = &e->cfg;
    const float* r0 = e->initial_root_states + (size_t)i * e->actors * 13;
    const float* org = e->env_origin + 3 * i;
    float* pv = e->prev + (size_t)i * e->prev_dim;
    if (c->task == MMS_TASK_TEN_ANT) {
        int A = e->A;
        float g[13], bp[2], bq[4];
        for (int k = 0; k < A; k++) { to_global(r0 + 13 * k, org, g); pv[2 * k] = g[0]; pv[2 * k + 1] = g[1]; }
        to_global(r0 + 13 * A, org, g);
        mo_tenant_goals(g, A, bp, bq, pv + 2 * A);
        pv[4 * A] = bp[0]; pv[4 * A + 1] = bp[1];
    } else if (c->task == MMS_TASK_ONE_ANT) {
        float g[13];
        to_global(r0, org, g); pv[0] = g[0]; pv[1] = g[1];
        to_global(r0 + 13, org, g); pv[2] = g[0]; pv[3] = g[1];
        pv[4] = -4.f / c->dt; pv[5] = -4.f / c->dt;                       /* one_ant.py:143-144 */
    } else if (c->task == MMS_TASK_MULTI_ANT_CIRCLE) {
        float g[13];                                                      /* multi_ant_circle.py:367-368 on the first step */
        for (int k = 0; k < e->A; k++) { to_global(r0 + 13 * k, org, g); pv[2 * k] = g[0]; pv[2 * k + 1] = g[1]; }
    }
}

MO_EXPORT mo_engine* mo_create(const mms_config* cfg) {
    if (cfg->abi_version != MMS_ABI_VERSION) return NULL;
    mo_engine* e = (mo_engine*)calloc(1, sizeof(mo_engine));
    e->cfg = *cfg;
    int N = e->N = cfg->num_envs, A = e->A = cfg->num_agents;
    if (is_ant_task(cfg->task)) {
        e->actors = A + 1; e->dofs_per_env = 8 * A; e->num_actions = 8 * A;
        e->obs_dim = (cfg->task == MMS_TASK_TEN_ANT) ? 38 * A + 8 : (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE ? 38 * A : 60);
        e->prev_dim = (cfg->task == MMS_TASK_TEN_ANT) ? 4 * A + 2 : (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE ? 2 * A : 6);  /* OneAnt: pos_before, box_before, potentials, prev_potentials */
    } else {
        e->actors = A; e->dofs_per_env = 4 * A; e->num_actions = 6 * A; e->obs_dim = 13 * A; e->prev_dim = 3 * A;
    }
    e->actions = (float*)calloc((size_t)N * e->num_actions, 4);
    e->obs = (float*)calloc((size_t)N * e->obs_dim, 4);
    e->obs_clipped = (float*)calloc((size_t)N * e->obs_dim, 4);
    e->rew = (float*)calloc(N, 4);
    e->root_states = (float*)calloc((size_t)N * e->actors * 13, 4);
    e->initial_root_states = (float*)calloc((size_t)N * e->actors * 13, 4);
    e->dof_state = (float*)calloc((size_t)N * e->dofs_per_env * 2, 4);
    e->env_origin = (float*)calloc((size_t)N * 3, 4);
    e->prev = (float*)calloc((size_t)N * e->prev_dim, 4);
    e->reset_noise = (float*)calloc((size_t)N * 16, 4);
    e->foot_sensors = (float*)calloc((size_t)N * A * 24, 4);
    e->reset = (int64_t*)calloc(N, 8);
    e->progress = (int64_t*)calloc(N, 8);
    e->reset_count = (int64_t*)calloc(N, 8);
    e->dr_params = (float*)calloc((size_t)N * A * MMS_DR_FLOATS, 4);
    for (size_t k = 0; k < (size_t)N * A; k++)
        for (int j = 0; j < 17; j++) e->dr_params[k * MMS_DR_FLOATS + j] = 1.f;      /* scales 1, offsets 0 */
    int64_t npr = (int64_t)sqrt((double)cfg->total_envs);
    if (npr < 1) npr = 1;
    for (int i = 0; i < N; i++) {
        int64_t gi = cfg->env_offset + i;
        e->env_origin[3 * i + 0] = (float)(gi % npr) * 2.f * cfg->env_spacing;  /* SURVEY B.2 grid convention */
        e->env_origin[3 * i + 1] = (float)(gi / npr) * 2.f * cfg->env_spacing;
        e->env_origin[3 * i + 2] = 0.f;
        e->reset[i] = 1;                                                      /* base_task.py:62-63 */
        float* r = e->initial_root_states + (size_t)i * e->actors * 13;
        for (int k = 0; k < e->actors; k++) r[13 * k + 6] = 1.f;
        if (is_ant_task(cfg->task)) {
            for (int k = 0; k < A; k++) {
                float off = (A == 1) ? 0.f : (1.5f + 3.f * (float)(k / 2)) * ((k % 2 == 0) ? -1.f : 1.f);
                r[13 * k + 0] = cfg->ant_start_x; r[13 * k + 1] = off; r[13 * k + 2] = cfg->ant_start_z;
                if (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE) {             /* multi_ant_circle.py:216-219: (3, 0, 1) and (-3, 0, 1) */
                    r[13 * k + 0] = (k % 2 == 0) ? cfg->ant_start_x : -cfg->ant_start_x; r[13 * k + 1] = 0.f;
                }
            }
            for (int j = 0; j < 3; j++) r[13 * A + j] = cfg->box_start[j];
        } else {
            static const float hy[4] = {2.f, -2.f, 6.f, -6.f};               /* multi_ingenuity.py:157-164 */
            for (int k = 0; k < A; k++) { r[13 * k + 0] = 0.f; r[13 * k + 1] = hy[k % 4]; r[13 * k + 2] = 1.f; }
        }
    }
    memcpy(e->root_states, e->initial_root_states, (size_t)N * e->actors * 13 * 4);
    /* caches start as the construction-time positions (ten_ant.py:870-882 on the first step) */
    for (int i = 0; i < N; i++) init_prev_from_initial(e, i);
    return e;
}
MO_EXPORT void mo_destroy(mo_engine* e) {
    if (!e) return;
    free(e->actions); free(e->obs); free(e->obs_clipped); free(e->rew); free(e->root_states);
    free(e->initial_root_states); free(e->dof_state); free(e->env_origin); free(e->prev); free(e->reset_noise);
    free(e->foot_sensors); free(e->reset); free(e->progress); free(e->reset_count); free(e->dr_params); free(e);
}
MO_EXPORT void* mo_tensor(mo_engine* e, const char* name, int64_t* numel) {
#define T(nm, p, cnt) if (!strcmp(name, nm)) { *numel = (int64_t)(cnt); return (void*)(p); }
    T("actions", e->actions, (size_t)e->N * e->num_actions)
    T("obs", e->obs, (size_t)e->N * e->obs_dim)
    T("obs_clipped", e->obs_clipped, (size_t)e->N * e->obs_dim)
    T("rew", e->rew, e->N)
    T("reset", e->reset, e->N)
    T("progress", e->progress, e->N)
    T("reset_count", e->reset_count, e->N)
    T("root_states", e->root_states, (size_t)e->N * e->actors * 13)
    T("initial_root_states", e->initial_root_states, (size_t)e->N * e->actors * 13)
    T("dof_state", e->dof_state, (size_t)e->N * e->dofs_per_env * 2)
    T("env_origin", e->env_origin, (size_t)e->N * 3)
    T("prev", e->prev, (size_t)e->N * e->prev_dim)
    T("reset_noise", e->reset_noise, (size_t)e->N * 16)
    T("foot_sensors", e->foot_sensors, (size_t)e->N * e->A * 24)
    T("dr_params", e->dr_params, (size_t)e->N * e->A * MMS_DR_FLOATS)
#undef T
    *numel = 0;
    return NULL;
}
MO_EXPORT void mo_set_dr(mo_engine* e, int enable) { e->dr_enabled = enable != 0; }
MO_EXPORT void mo_dims(mo_engine* e, int32_t out[6]) {
    out[0] = e->actors; out[1] = e->dofs_per_env; out[2] = e->num_actions; out[3] = e->obs_dim; out[4] = e->prev_dim; out[5] = e->A;
}

static void reward_params_from_cfg(const mms_config* c, reward_params* rp) {
    rp->up_weight = c->up_weight; rp->heading_weight = c->heading_weight; rp->actions_cost = c->actions_cost;
    rp->energy_cost = c->energy_cost; rp->joints_at_limit_cost = c->joints_at_limit_cost;
    rp->termination_height = c->termination_height; rp->death_cost = c->death_cost;
    rp->quat_reward_scale = c->quat_reward_scale; rp->ant_dist_reward_scale = c->ant_dist_reward_scale;
    rp->goal_dist_reward_scale = c->goal_dist_reward_scale; rp->max_episode_length = c->max_episode_length;
}

/* post-physics glue for one env: progress, reset, obs, reward, caches. */
static void post_step_env(mo_engine* e, int i, int first_step) {
    const mms_config* c = &e->cfg;
    const mms_model* M = &c->model;
    int A = e->A;
    float* roots = e->root_states + (size_t)i * e->actors * 13;
    float* dofs = e->dof_state + (size_t)i * e->dofs_per_env * 2;
    const float* org = e->env_origin + 3 * i;
    float* pv = e->prev + (size_t)i * e->prev_dim;
    float act[1024];
    for (int j = 0; j < e->num_actions; j++) act[j] = clampf(e->actions[(size_t)i * e->num_actions + j], -c->clip_actions, c->clip_actions);
    (void)first_step;

    e->progress[i] += 1;
    if (e->reset[i] != 0) {
        memcpy(roots, e->initial_root_states + (size_t)i * e->actors * 13, (size_t)e->actors * 13 * 4);
        if (is_ant_task(c->task)) {
            float npos[8], nvel[8];
            for (int j = 0; j < 8; j++) {
                if (c->external_noise) { npos[j] = e->reset_noise[16 * i + j]; nvel[j] = e->reset_noise[16 * i + 8 + j]; }
                else {
                    uint64_t gi = (uint64_t)(c->env_offset + i);
                    npos[j] = 0.4f * mo_rand_uniform(c->seed, gi, (uint64_t)e->reset_count[i], (uint32_t)j) - 0.2f;
                    nvel[j] = 0.2f * mo_rand_uniform(c->seed, gi, (uint64_t)e->reset_count[i], (uint32_t)(8 + j)) - 0.1f;
                }
            }
            for (int k = 0; k < A; k++)                                    /* same noise for every ant: ten_ant.py:822-854 */
                for (int j = 0; j < 8; j++) {
                    dofs[2 * (8 * k + j) + 0] = clampf(M->dof_init[j] + npos[j], M->dof_lower[j], M->dof_upper[j]);
                    dofs[2 * (8 * k + j) + 1] = nvel[j];
                }
        } else {
            for (int k = 0; k < A; k++)                                    /* multi_ingenuity.py:231-238 */
                for (int j = 0; j < 4; j++) { dofs[2 * (4 * k + j)] = 0.f; dofs[2 * (4 * k + j) + 1] = (j == 1) ? -50.f : (j == 3 ? 50.f : 0.f); }
        }
        e->progress[i] = 0;
        e->reset[i] = 0;
        e->reset_count[i] += 1;
    }
    float* obs = e->obs + (size_t)i * e->obs_dim;
    reward_params rp;
    reward_params_from_cfg(c, &rp);
    if (c->task == MMS_TASK_TEN_ANT) {
        float g[13], bp[2], bq[4], goals[2 * 128];
        for (int k = 0; k < A; k++) {
            to_global(roots + 13 * k, org, g);
            float dp[8], dv[8];
            for (int j = 0; j < 8; j++) { dp[j] = dofs[2 * (8 * k + j)]; dv[j] = dofs[2 * (8 * k + j) + 1]; }
            mo_tenant_ant_obs(g, dp, dv, M->dof_lower, M->dof_upper, c->dof_vel_scale, act + 8 * k, obs + 38 * k);
        }
        to_global(roots + 13 * A, org, g);
        mo_tenant_goals(g, A, bp, bq, goals);
        obs[38 * A + 0] = bp[0]; obs[38 * A + 1] = bp[1];
        for (int j = 0; j < 4; j++) obs[38 * A + 2 + j] = bq[j];
        obs[38 * A + 6] = 0.f; obs[38 * A + 7] = 0.f;                        /* box_targets = (0,0), ten_ant.py:171,808 */
        tenant_reward(&rp, A, obs, act, pv, pv + 2 * A, goals, bq, e->reset[i], e->progress[i], e->rew + i, e->reset + i);
        for (int k = 0; k < A; k++) { pv[2 * k] = obs[38 * k]; pv[2 * k + 1] = obs[38 * k + 1]; }   /* :906-926 */
        memcpy(pv + 2 * A, goals, (size_t)2 * A * 4);
        pv[4 * A] = bp[0]; pv[4 * A + 1] = bp[1];
    } else if (c->task == MMS_TASK_MULTI_ANT_CIRCLE) {
        float g[13];
        for (int k = 0; k < A; k++) {                                      /* :322-341: the 38 entries of TenAnt's ants, target = origin */
            to_global(roots + 13 * k, org, g);
            float dp[8], dv[8];
            for (int j = 0; j < 8; j++) { dp[j] = dofs[2 * (8 * k + j)]; dv[j] = dofs[2 * (8 * k + j) + 1]; }
            mo_tenant_ant_obs(g, dp, dv, M->dof_lower, M->dof_upper, c->dof_vel_scale, act + 8 * k, obs + 38 * k);
        }
        circle_reward(&rp, obs, act, pv, e->reset[i], e->progress[i], e->rew + i, e->reset + i);
        for (int k = 0; k < A; k++) { pv[2 * k] = obs[38 * k]; pv[2 * k + 1] = obs[38 * k + 1]; }   /* :382-383 */
    } else if (c->task == MMS_TASK_ONE_ANT) {
        float g[13], gb[13], dp[8], dv[8];
        to_global(roots, org, g);
        to_global(roots + 13, org, gb);
        for (int j = 0; j < 8; j++) { dp[j] = dofs[2 * j]; dv[j] = dofs[2 * j + 1]; }
        float pot, prevpot;
        mo_oneant_obs(g, gb, dp, dv, M->dof_lower, M->dof_upper, c->dof_vel_scale, e->foot_sensors + 24 * i,
                      c->contact_force_scale, act, c->dt, pv[4], obs, &pot, &prevpot);
        pv[4] = pot; pv[5] = prevpot;
        float ant_pos[2] = {g[0], g[1]}, box_pos[2] = {gb[0], gb[1]};
        oneant_reward(&rp, obs, act, pv, pv + 2, ant_pos, box_pos, gb + 3, e->reset[i], e->progress[i], e->rew + i, e->reset + i);
        pv[0] = ant_pos[0]; pv[1] = ant_pos[1]; pv[2] = box_pos[0]; pv[3] = box_pos[1];
    } else {
        for (int k = 0; k < A; k++) to_global(roots + 13 * k, org, obs + 13 * k);
        ingenuity_reward(obs, c->max_episode_length, e->progress[i], e->rew + i, e->reset + i);
    }
    float* oc = e->obs_clipped + (size_t)i * e->obs_dim;
    for (int j = 0; j < e->obs_dim; j++) oc[j] = clampf(obs[j], -c->clip_obs, c->clip_obs);
}

static void physics_env(mo_engine* e, int i) {
    physics_core(&e->cfg, e->A, e->actions + (size_t)i * e->num_actions, e->root_states + (size_t)i * e->actors * 13,
                 e->dof_state + (size_t)i * e->dofs_per_env * 2, e->foot_sensors + (size_t)i * e->A * 24,
                 e->dr_enabled ? e->dr_params + (size_t)i * e->A * MMS_DR_FLOATS : NULL);
}

/* BaseTask.step (base_task.py:129-149).  Envs flagged for reset skip the physics: their state is
 * overwritten by reset_idx before anything reads it. */
MO_EXPORT void mo_step(mo_engine* e, int do_physics) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < e->N; i++) {
        if (do_physics && e->reset[i] == 0) physics_env(e, i);
        post_step_env(e, i, 0);
    }
}

/* unit-level entry points for the physics KATs */
MO_EXPORT void mo_ant_substep_dr(const mms_model* M, float h, float* root, float* dof, const float* tau,
                                 const float* box_root /*may be NULL*/, float* box_wrench, float* sensors, const float* dr);
MO_EXPORT void mo_ant_substep(const mms_model* M, float h, float* root, float* dof, const float* tau,
                              const float* box_root /*may be NULL*/, float* box_wrench, float* sensors) {
    mo_ant_substep_dr(M, h, root, dof, tau, box_root, box_wrench, sensors, NULL);
}
MO_EXPORT void mo_ant_substep_dr(const mms_model* M, float h, float* root, float* dof, const float* tau,
                                 const float* box_root /*may be NULL*/, float* box_wrench, float* sensors, const float* dr) {
    box_pose box;
    if (box_root) {
        memcpy(box.pos, box_root, 12);
        quat_to_mat(box_root + 3, box.R);
        memcpy(box.v, box_root + 7, 12);
        memcpy(box.w, box_root + 10, 12);
        memcpy(box.half, M->box_half, 12);
    }
    ant_substep(M, h, root, (float(*)[2])dof, tau, box_root ? &box : NULL, box_wrench, (float(*)[6])sensors, dr);
}
MO_EXPORT void mo_box_substep(const mms_model* M, float h, float* root, const float* wrench) { box_substep(M, h, root, wrench); }
MO_EXPORT void mo_heli_substep(const mms_model* M, float h, float* root, const float* thrust) {
    heli_substep(M, h, root, (const float(*)[3])thrust);
}
/* Diagnostics for the physics invariants: total momentum of one ant about the WORLD origin
 * out[0..2] angular, out[3..5] linear, out[6] kinetic energy (incl. armature), out[7] potential energy. */
MO_EXPORT void mo_ant_momentum(const mms_model* M, const float* root, const float* dofp, float* out) {
    const float(*dof)[2] = (const float(*)[2])dofp;
    float Rt[3][3];
    quat_to_mat(root + 3, Rt);
    float v0[6] = {root[10], root[11], root[12], root[7], root[8], root[9]};
    const float zero3[3] = {0.f, 0.f, 0.f};
    double hsum[6] = {0, 0, 0, 0, 0, 0}, ke = 0.0, pe = 0.0;
    m66 I;
    float Ic[3][3], h[6];
    float ez[3] = {Rt[0][2], Rt[1][2], Rt[2][2]};
    axisym_inertia(M->torso_izz, M->torso_ixx, ez, Ic);
    spatial_inertia(M->torso_mass, zero3, Ic, I);
    m66_mulv(I, v0, h);
    for (int i = 0; i < 6; i++) { hsum[i] += h[i]; ke += 0.5 * h[i] * v0[i]; }
    pe += (double)M->torso_mass * M->gravity * root[2];
    for (int l = 0; l < 4; l++) {
        float q1 = dof[2 * l][0], qd1 = dof[2 * l][1], q2 = dof[2 * l + 1][0], qd2 = dof[2 * l + 1][1];
        float J1[3], a1[3] = {Rt[0][2], Rt[1][2], Rt[2][2]}, J2[3];
        matvec3(Rt, M->hip_pos[l], J1);
        float Rz[3][3], Rl[3][3], Ra[3][3], Rf[3][3];
        const float zax[3] = {0.f, 0.f, 1.f};
        axis_angle_to_mat(zax, q1, Rz);
        matmul3(Rt, Rz, Rl);
        float ul[3], a2[3], uf[3], cl[3], cf[3], t3[3], s1[6], s2[6], vl[6], vf[6];
        matvec3(Rl, M->limb_dir[l], ul);
        matvec3(Rl, M->ankle_axis[l], a2);
        for (int i = 0; i < 3; i++) J2[i] = J1[i] + M->leg_len * ul[i];
        axis_angle_to_mat(M->ankle_axis[l], q2, Ra);
        matmul3(Rl, Ra, Rf);
        matvec3(Rf, M->limb_dir[l], uf);
        for (int i = 0; i < 3; i++) { cl[i] = J1[i] + 0.5f * M->leg_len * ul[i]; cf[i] = J2[i] + 0.5f * M->foot_len * uf[i]; }
        cross3(J1, a1, t3);
        for (int i = 0; i < 3; i++) { s1[i] = a1[i]; s1[3 + i] = t3[i]; }
        cross3(J2, a2, t3);
        for (int i = 0; i < 3; i++) { s2[i] = a2[i]; s2[3 + i] = t3[i]; }
        for (int i = 0; i < 6; i++) { vl[i] = v0[i] + s1[i] * qd1; vf[i] = vl[i] + s2[i] * qd2; }
        axisym_inertia(M->leg_ia, M->leg_it, ul, Ic);
        spatial_inertia(M->leg_mass, cl, Ic, I);
        m66_mulv(I, vl, h);
        for (int i = 0; i < 6; i++) { hsum[i] += h[i]; ke += 0.5 * h[i] * vl[i]; }
        axisym_inertia(M->foot_ia, M->foot_it, uf, Ic);
        spatial_inertia(M->foot_mass, cf, Ic, I);
        m66_mulv(I, vf, h);
        for (int i = 0; i < 6; i++) { hsum[i] += h[i]; ke += 0.5 * h[i] * vf[i]; }
        ke += 0.5 * M->armature * ((double)qd1 * qd1 + (double)qd2 * qd2);
        pe += (double)M->leg_mass * M->gravity * (root[2] + cl[2]) + (double)M->foot_mass * M->gravity * (root[2] + cf[2]);
    }
    /* shift the angular momentum from the torso origin to the world origin: L_w = L_O + r x p */
    double r[3] = {root[0], root[1], root[2]};
    out[0] = (float)(hsum[0] + r[1] * hsum[5] - r[2] * hsum[4]);
    out[1] = (float)(hsum[1] + r[2] * hsum[3] - r[0] * hsum[5]);
    out[2] = (float)(hsum[2] + r[0] * hsum[4] - r[1] * hsum[3]);
    out[3] = (float)hsum[3]; out[4] = (float)hsum[4]; out[5] = (float)hsum[5];
    out[6] = (float)ke; out[7] = (float)pe;
}
MO_EXPORT int mo_abi_version(void) { return MMS_ABI_VERSION; }
MO_EXPORT int mo_sizeof_config(void) { return (int)sizeof(mms_config); }
#endif /* !MO_F64 */
