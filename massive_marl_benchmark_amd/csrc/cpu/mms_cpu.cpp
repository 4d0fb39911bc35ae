// mms_cpu.cpp -- the CPU build of the engine behind the SAME C ABI (include/mms.h): lib/libmms_cpu.so.
//
// What it is for: the reference selects a CPU pipeline with `--sim_device cpu` (agents/tasks/agent_base/base_task.py:27-32,
// vec_task.py:126-139; BASELINE configs[0] "OneAnt num_envs=64 ... sim_device=cpu -- plumbing, no GPU").  This library is that
// pipeline: explicit and opt-in (mms_config.device = -1, `device_type="cpu"` in the task constructors), NEVER a fallback -- the
// HIP library still fails without a GPU and nothing selects this one automatically.
// What it is built from: the per-lane functions of the HIP step kernels (../mms_lane.h through lane_step.h) and the per-column
// functions of the rollout kernels (../rollout_lane.h) -- the product's own math, compiled for the host, OpenMP over envs.
// It shares no source with the oracle (oracle/mms_oracle.c), which stays the checker.
// Streams: the `hip_stream` arguments are ignored; every call has completed when it returns.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../../include/mms.h"
#include "../rollout_lane.h"
#include "lane_step.h"

#define MMS_API extern "C" __attribute__((visibility("default")))

struct mms_buffer {
    const char* name;
    void* ptr;
    int64_t shape[4];
    int ndim;
    int dtype;
    size_t bytes;
    int64_t row_bytes;
};

struct mms_engine {
    mms_config cfg;
    int actors = 0, dofs = 0, num_actions = 0, obs_dim = 0, prev_dim = 0;
    float* obs_out = nullptr;
    void* obs_planes = nullptr;
    float obs_planes_scale = 1.f;
    const float* actions_in = nullptr;
    int head_on = 0;                        // mms_bind_policy_head: consumed by the next mms_step
    mms_policy_head head{};
    int write_raw_obs = 1, write_clipped_obs = 1, dr_enabled = 0;
    float* rew_out = nullptr;
    uint8_t* done_out = nullptr;
    std::vector<mms_buffer> bufs;
    std::string err;
};

static std::string g_error;
static size_t dtype_size(int dt) { return dt == MMS_F32 ? 4 : dt == MMS_I64 ? 8 : dt == MMS_I32 ? 4 : 1; }
static int fail(mms_engine* e, const std::string& msg) {
    if (e) e->err = msg; else g_error = msg;
    return 1;
}
static mms_buffer* find(mms_engine* e, const char* name) {
    for (auto& b : e->bufs)
        if (!strcmp(b.name, name)) return &b;
    return nullptr;
}
static void add_buffer(mms_engine* e, const char* name, int dtype, std::initializer_list<int64_t> shape) {
    mms_buffer b{};
    b.name = name;
    b.dtype = dtype;
    b.ndim = (int)shape.size();
    size_t n = 1;
    int i = 0;
    for (int64_t s : shape) { b.shape[i++] = s; n *= (size_t)s; }
    b.bytes = n * dtype_size(dtype);
    b.row_bytes = (int64_t)(b.bytes / (size_t)e->cfg.num_envs);
    b.ptr = calloc(b.bytes ? b.bytes : 16, 1);
    e->bufs.push_back(b);
}
template <typename T> static T* buf(mms_engine* e, const char* name) { return (T*)find(e, name)->ptr; }

MMS_API int mms_abi_version(void) { return MMS_ABI_VERSION; }
MMS_API const char* mms_last_error(mms_handle h) { return h ? h->err.c_str() : g_error.c_str(); }

MMS_API int mms_destroy(mms_handle h) {
    if (!h) return 0;
    for (auto& b : h->bufs) free(b.ptr);
    delete h;
    return 0;
}

MMS_API int mms_create(const mms_config* cfg, mms_handle* out) {
    if (!cfg || !out) return fail(nullptr, "mms_create: null argument");
    if (cfg->abi_version != MMS_ABI_VERSION) return fail(nullptr, "mms_create: ABI version mismatch");
    if (cfg->device != -1) return fail(nullptr, "mms_create: this is the CPU build of the engine (libmms_cpu.so); device must be -1");
    if (cfg->num_envs <= 0 || cfg->num_agents <= 0) return fail(nullptr, "mms_create: num_envs and num_agents must be positive");
    if (cfg->task == MMS_TASK_MULTI_INGENUITY && cfg->num_agents != 4) return fail(nullptr, "mms_create: MultiIngenuity has 4 helicopters per env");
    if (cfg->task == MMS_TASK_ONE_ANT && cfg->num_agents != 1) return fail(nullptr, "mms_create: OneAnt has one ant per env");
    if (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE && cfg->num_agents != 2) return fail(nullptr, "mms_create: MultiAntCircle has two ants per env");
    if (cfg->task != MMS_TASK_MULTI_INGENUITY && cfg->num_agents > 126) return fail(nullptr, "mms_create: at most 126 ants per env");
    mms_engine* e = new mms_engine();
    e->cfg = *cfg;
    const int N = cfg->num_envs, A = cfg->num_agents;
    if (cfg->task == MMS_TASK_TEN_ANT) { e->actors = A + 1; e->dofs = 8 * A; e->num_actions = 8 * A; e->obs_dim = 38 * A + 8; e->prev_dim = 4 * A + 2; }
    else if (cfg->task == MMS_TASK_ONE_ANT) { e->actors = 2; e->dofs = 8; e->num_actions = 8; e->obs_dim = 60; e->prev_dim = 6; }
    else if (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE) { e->actors = A + 1; e->dofs = 8 * A; e->num_actions = 8 * A; e->obs_dim = 38 * A; e->prev_dim = 2 * A; }
    else if (cfg->task == MMS_TASK_MULTI_INGENUITY) { e->actors = A; e->dofs = 4 * A; e->num_actions = 6 * A; e->obs_dim = 13 * A; e->prev_dim = 3 * A; }
    else { delete e; return fail(nullptr, "mms_create: unknown task"); }
    add_buffer(e, "actions", MMS_F32, {N, e->num_actions});
    add_buffer(e, "obs", MMS_F32, {N, e->obs_dim});
    add_buffer(e, "obs_clipped", MMS_F32, {N, e->obs_dim});
    add_buffer(e, "rew", MMS_F32, {N});
    add_buffer(e, "reset", MMS_I64, {N});
    add_buffer(e, "progress", MMS_I64, {N});
    add_buffer(e, "reset_count", MMS_I64, {N});
    add_buffer(e, "root_states", MMS_F32, {(int64_t)N * e->actors, 13});
    add_buffer(e, "initial_root_states", MMS_F32, {(int64_t)N * e->actors, 13});
    add_buffer(e, "dof_state", MMS_F32, {(int64_t)N * e->dofs, 2});
    add_buffer(e, "env_origin", MMS_F32, {N, 3});
    add_buffer(e, "prev", MMS_F32, {N, e->prev_dim});
    add_buffer(e, "reset_noise", MMS_F32, {N, 16});
    add_buffer(e, "foot_sensors", MMS_F32, {(int64_t)N * A, 24});
    add_buffer(e, "dr_params", MMS_F32, {(int64_t)N * A, MMS_DR_FLOATS});
    // construction-time scene: the same as the HIP build's mms_create (ten_ant.py:339-358,494-495; one_ant.py:234;
    // multi_ingenuity.py:157-164), env grid per SURVEY.md B.2
    float* init = buf<float>(e, "initial_root_states");
    float* origin = buf<float>(e, "env_origin");
    float* prev = buf<float>(e, "prev");
    int64_t npr = (int64_t)sqrt((double)cfg->total_envs);
    if (npr < 1) npr = 1;
    for (int i = 0; i < N; i++) {
        int64_t gi = cfg->env_offset + i;
        origin[3 * (size_t)i + 0] = (float)(gi % npr) * 2.f * cfg->env_spacing;
        origin[3 * (size_t)i + 1] = (float)(gi / npr) * 2.f * cfg->env_spacing;
        float* r = init + (size_t)i * e->actors * 13;
        for (int k = 0; k < e->actors; k++) r[13 * k + 6] = 1.f;
        if (cfg->task != MMS_TASK_MULTI_INGENUITY) {
            for (int k = 0; k < A; k++) {
                float off = (A == 1) ? 0.f : (1.5f + 3.f * (float)(k / 2)) * ((k % 2 == 0) ? -1.f : 1.f);
                r[13 * k + 0] = cfg->ant_start_x; r[13 * k + 1] = off; r[13 * k + 2] = cfg->ant_start_z;
                if (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE) {                       // multi_ant_circle.py:216-219: (3, 0, 1) and (-3, 0, 1)
                    r[13 * k + 0] = (k % 2 == 0) ? cfg->ant_start_x : -cfg->ant_start_x; r[13 * k + 1] = 0.f;
                }
            }
            for (int j = 0; j < 3; j++) r[13 * A + j] = cfg->box_start[j];
        } else {
            static const float hy[4] = {2.f, -2.f, 6.f, -6.f};
            for (int k = 0; k < A; k++) { r[13 * k + 0] = 0.f; r[13 * k + 1] = hy[k % 4]; r[13 * k + 2] = 1.f; }
        }
        // caches start as the construction-time poses in the global frame (ten_ant.py:870-882, one_ant.py:410-411)
        const float* o = origin + 3 * (size_t)i;
        float* pv = prev + (size_t)i * e->prev_dim;
        if (cfg->task == MMS_TASK_TEN_ANT) {
            const float* b = r + 13 * A;
            float bx = b[0] + o[0], by = b[1] + o[1];
            float ang = atanf((2.f * b[6] * b[5]) / (1.f - 2.f * b[5] * b[5]));
            float sv = sinf(ang), cv = -cosf(ang);
            for (int k = 0; k < A; k++) {
                pv[2 * k] = r[13 * k] + o[0]; pv[2 * k + 1] = r[13 * k + 1] + o[1];
                float off = 1.5f + 3.0f * (float)(k / 2);
                pv[2 * A + 2 * k] = (k % 2 == 0) ? bx + off * sv : bx - off * sv;
                pv[2 * A + 2 * k + 1] = (k % 2 == 0) ? by + off * cv : by - off * cv;
            }
            pv[4 * A] = bx; pv[4 * A + 1] = by;
        } else if (cfg->task == MMS_TASK_ONE_ANT) {
            pv[0] = r[0] + o[0]; pv[1] = r[1] + o[1]; pv[2] = r[13] + o[0]; pv[3] = r[14] + o[1];
            pv[4] = -4.f / cfg->dt; pv[5] = -4.f / cfg->dt;
        } else if (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE) {
            for (int k = 0; k < A; k++) { pv[2 * k] = r[13 * k] + o[0]; pv[2 * k + 1] = r[13 * k + 1] + o[1]; }   // multi_ant_circle.py:367-368
        }
    }
    memcpy(buf<float>(e, "root_states"), init, find(e, "root_states")->bytes);
    int64_t* reset = buf<int64_t>(e, "reset");
    for (int i = 0; i < N; i++) reset[i] = 1;                                     // base_task.py:62-63
    float* dr = buf<float>(e, "dr_params");
    for (size_t k = 0; k < (size_t)N * A; k++)
        for (int j = 0; j < 17; j++) dr[k * MMS_DR_FLOATS + j] = 1.f;            // nominal: scales 1, limit offsets 0
    *out = e;
    return 0;
}

MMS_API int mms_get_tensor(mms_handle h, const char* name, mms_tensor* out) {
    if (!h || !name || !out) return fail(h, "mms_get_tensor: null argument");
    mms_buffer* b = find(h, name);
    if (!b) return fail(h, std::string("mms_get_tensor: unknown buffer '") + name + "'");
    memset(out, 0, sizeof(*out));
    out->ptr = b->ptr;
    for (int i = 0; i < b->ndim; i++) out->shape[i] = b->shape[i];
    out->ndim = b->ndim;
    out->dtype = b->dtype;
    out->device = -1;
    return 0;
}

MMS_API int mms_ppo_heads_act(int device, const float* hidden, const float* weight, const float* bias, int32_t H, const float* value,
                              const float* vhidden, const float* vweight, const float* vbias, int32_t VH, const float* log_std, uint64_t seed,
                              int64_t* counters, int64_t row_offset, int32_t reference_scale, float* actions_out, float* act_slot,
                              float* logp_slot, float* value_slot, float* mu_slot, float* sigma_slot, int64_t N, int32_t A, void*);

static int do_step(mms_handle h, int physics) {
    if (!h) return fail(nullptr, "mms_step: null handle");
    const float* head_actions = nullptr;
    if (physics && h->head_on) {
        // the fused policy head (mms_bind_policy_head): on the host the heads operator runs in front of the step, into the action tensor
        // the step then reads -- the same values as the two calls made separately
        const mms_policy_head& p = h->head;
        float* dst = p.actions_out ? p.actions_out : buf<float>(h, "actions");
        h->head_on = 0;
        // (with the tiled copy of the actor's last layer bound, THAT is what is read -- as on the device; element
        //  ((ct (H / 4) + k / 4) 16 + i) 4 + k % 4 = weight[16 ct + i][k] -- so a stale or mis-laid copy shows up in the host tests too)
        std::vector<float> untiled;
        if (p.weight_tiles) {
            untiled.resize((size_t)p.A * p.H);
            for (int j = 0; j < p.A; j++)
                for (int k = 0; k < p.H; k++)
                    untiled[(size_t)j * p.H + k] = p.weight_tiles[((size_t)((j >> 4) * (p.H / 4) + (k >> 2)) * 16 + (j & 15)) * 4 + (k & 3)];
        }
        if (mms_ppo_heads_act(-1, p.hidden, p.weight_tiles ? untiled.data() : p.weight, p.bias, p.H, nullptr, p.vhidden, p.vweight, p.vbias, p.VH, p.log_std, p.seed, p.counters, p.row_offset,
                              p.reference_scale, dst, p.act_slot, p.logp_slot, p.value_slot, p.mu_slot, p.sigma_slot, h->cfg.num_envs, p.A, nullptr))
            return fail(h, "mms_step: the bound policy head failed: " + g_error);
        head_actions = dst;
    }
    mms::HostBufs b{head_actions ? const_cast<float*>(head_actions) : (h->actions_in ? const_cast<float*>(h->actions_in) : buf<float>(h, "actions")), h->write_raw_obs ? buf<float>(h, "obs") : nullptr,
                    h->write_clipped_obs ? buf<float>(h, "obs_clipped") : nullptr, buf<float>(h, "rew"), buf<int64_t>(h, "reset"),
                    buf<int64_t>(h, "progress"), buf<float>(h, "root_states"), buf<float>(h, "initial_root_states"),
                    buf<float>(h, "dof_state"), buf<float>(h, "env_origin"), buf<float>(h, "prev"), buf<float>(h, "reset_noise"),
                    buf<float>(h, "foot_sensors"), buf<int64_t>(h, "reset_count"), h->dr_enabled ? buf<float>(h, "dr_params") : nullptr};
    b.obs_out = h->obs_out; b.rew_out = h->rew_out; b.done_out = h->done_out;
    b.obs_planes = (uint16_t*)h->obs_planes; b.obs_planes_scale = h->obs_planes_scale;
    const mms_config* C = &h->cfg;
#pragma omp parallel for schedule(static)
    for (int env = 0; env < C->num_envs; env++) {
        if (C->task == MMS_TASK_MULTI_INGENUITY) mms::host_heli_env(C, b, env, physics);
        else mms::host_ant_env(C, b, env, physics, h->obs_dim, h->prev_dim);
    }
    return 0;
}
MMS_API int mms_step(mms_handle h, void*) { return do_step(h, 1); }
MMS_API int mms_post_step(mms_handle h, void*) { return do_step(h, 0); }

MMS_API int mms_reset_all(mms_handle h, void*) {
    if (!h) return fail(nullptr, "mms_reset_all: null handle");
    int64_t* r = buf<int64_t>(h, "reset");
    for (int i = 0; i < h->cfg.num_envs; i++) r[i] = 1;
    return 0;
}

MMS_API int mms_set_state(mms_handle h, const char* name, const void* src, int, const int64_t* env_ids, int64_t n, void*) {
    if (!h || !name || !src) return fail(h, "mms_set_state: null argument");
    mms_buffer* b = find(h, name);
    if (!b) return fail(h, std::string("mms_set_state: unknown buffer '") + name + "'");
    if (!env_ids) { memcpy(b->ptr, src, b->bytes); return 0; }
    if (b->row_bytes <= 0) return fail(h, "mms_set_state: buffer is not per-env");
    if (n < 0) return fail(h, "mms_set_state: negative row count");
    for (int64_t i = 0; i < n; i++)                          // all ids are checked before anything is written
        if (env_ids[i] < 0 || env_ids[i] >= h->cfg.num_envs) return fail(h, "mms_set_state: env id out of range");
    for (int64_t i = 0; i < n; i++) memcpy((char*)b->ptr + env_ids[i] * b->row_bytes, (const char*)src + i * b->row_bytes, (size_t)b->row_bytes);
    return 0;
}

MMS_API int mms_bind_obs_out(mms_handle h, void* dst) {
    if (!h) return fail(nullptr, "mms_bind_obs_out: null handle");
    h->obs_out = (float*)dst;
    return 0;
}
MMS_API int mms_bind_obs_planes16(mms_handle h, void* planes, float scale) {
    if (!h) return fail(nullptr, "mms_bind_obs_planes16: null handle");
    if (!planes) { h->obs_planes = nullptr; return 0; }
    if (h->cfg.task == MMS_TASK_MULTI_INGENUITY) return fail(h, "mms_bind_obs_planes16: not for the helicopter task (its policies' layers are 256 wide: exact-fp32 kernel)");
    int e = 0;
    if (!(scale > 0.f) || frexpf(scale, &e) != 0.5f) return fail(h, "mms_bind_obs_planes16: the scale must be a power of two");
    if (!(h->cfg.clip_obs * scale <= 16384.f)) return fail(h, "mms_bind_obs_planes16: clip_observations x scale must not exceed 2^14 (fp16 planes)");
    h->obs_planes = planes;
    h->obs_planes_scale = scale;
    return 0;
}
MMS_API int mms_bind_actions(mms_handle h, const float* src) {
    if (!h) return fail(nullptr, "mms_bind_actions: null handle");
    h->actions_in = src;
    return 0;
}
MMS_API int mms_bind_policy_head(mms_handle h, const mms_policy_head* head) {
    if (!h) return fail(nullptr, "mms_bind_policy_head: null handle");
    if (!head) { h->head_on = 0; return 0; }
    if (h->dr_enabled || h->cfg.task != MMS_TASK_TEN_ANT || h->cfg.num_agents != 10 || h->cfg.num_envs % 16 != 0)
        return fail(h, "mms_bind_policy_head: not available for this engine (needs the 16-envs-per-workgroup TenAnt layout: 10 ants, num_envs a multiple "
                       "of 16 and >= 16 per CU, no physical DR) -- launch mms_ppo_heads_act instead");
    if (!head->hidden || !head->weight || !head->bias || !head->vhidden || !head->vweight || !head->vbias || !head->log_std || !head->counters)
        return fail(h, "mms_bind_policy_head: null pointer (hidden, weight, bias, vhidden, vweight, vbias, log_std, counters are required)");
    if (head->A != 8 * h->cfg.num_agents || head->H <= 0 || head->H % 512 != 0 || head->VH <= 0 || head->VH % 4 != 0)
        return fail(h, "mms_bind_policy_head: A must be 8 x num_agents, H a multiple of 512, VH a multiple of 4");
    uintptr_t bits = reinterpret_cast<uintptr_t>(head->hidden) | reinterpret_cast<uintptr_t>(head->weight) | reinterpret_cast<uintptr_t>(head->vhidden) |
                     reinterpret_cast<uintptr_t>(head->vweight) | reinterpret_cast<uintptr_t>(head->weight_tiles);
    if ((bits & 15) != 0) return fail(h, "mms_bind_policy_head: hidden, weight, weight_tiles, vhidden, vweight must be 16-byte aligned");
    h->head = *head;
    h->head_on = 1;
    return 0;
}
MMS_API int mms_set_dr(mms_handle h, int32_t enable) {
    if (!h) return fail(nullptr, "mms_set_dr: null handle");
    if (enable && h->cfg.task == MMS_TASK_MULTI_INGENUITY) return fail(h, "mms_set_dr: the helicopter task has no randomised physical parameters");
    h->dr_enabled = enable != 0;
    return 0;
}
MMS_API int mms_set_obs_outputs(mms_handle h, int32_t raw, int32_t clipped) {
    if (!h) return fail(nullptr, "mms_set_obs_outputs: null handle");
    h->write_raw_obs = raw != 0;
    h->write_clipped_obs = clipped != 0;
    return 0;
}
MMS_API int mms_bind_rollout_out(mms_handle h, float* rew_out, uint8_t* done_out) {
    if (!h) return fail(nullptr, "mms_bind_rollout_out: null handle");
    h->rew_out = rew_out;
    h->done_out = done_out;
    return 0;
}

// ---- rollout functions (device argument: -1) ---------------------------------------------------------------------------------
static int cpu_only(int device) {
    if (device != -1) { g_error = "libmms_cpu.so: device must be -1"; return 1; }
    return 0;
}
MMS_API int mms_marl_views(int device, const float* obs_clipped, float* obs_all, int64_t n, int32_t agents, int32_t per_agent, int32_t shared, void*) {
    if (cpu_only(device)) return 1;
    const int64_t total = n * agents * (per_agent + shared);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < total; i++) obs_all[i] = obs_clipped[mms::marl_view_source(i, agents, per_agent, shared)];
    return 0;
}
MMS_API int mms_gae_ppo(int device, const float* rewards, const uint8_t* dones, const float* values, const float* last_values, float* returns,
                        float* advantages, double* stats, int32_t T, int64_t N, float gamma, float lam, void*) {
    if (cpu_only(device)) return 1;
    double sum = 0.0, sq = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sum, sq)
    for (int64_t i = 0; i < N; i++) mms::gae_ppo_column(rewards, dones, values, last_values, returns, advantages, T, N, i, gamma, lam, sum, sq);
    stats[0] = sum; stats[1] = sq; stats[2] = (double)T * (double)N;
    return 0;
}
MMS_API int mms_adv_normalize(int device, float* advantages, const double* stats, int64_t count, void*) {
    if (cpu_only(device)) return 1;
    float fm, inv;
    mms::adv_norm_params(stats, fm, inv);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; i++) advantages[i] = (advantages[i] - fm) * inv;
    return 0;
}
MMS_API int mms_layer_clock_probe(int device, uint64_t* out, int32_t slots) {
    (void)device;
    if (out && slots < 1) { g_error = "mms_layer_clock_probe: slots must be >= 1 with an output buffer"; return 1; }
    if (out && (reinterpret_cast<uintptr_t>(out) & 7) != 0) { g_error = "mms_layer_clock_probe: the buffer must be 8-byte aligned"; return 1; }
    return 0;                                     // (no shader clock on this build: nothing is stored)
}

MMS_API int mms_gae_ppo_normalized(int device, const float* rewards, const uint8_t* dones, const float* values, const float* last_values, float* returns,
                                   float* advantages, double* stats, int32_t T, int64_t N, float gamma, float lam, void*) {
    if (cpu_only(device)) return 1;
    if (!rewards || !dones || !values || !last_values || !returns || !advantages || !stats || T < 1 || N < 1) {
        g_error = "mms_gae_ppo_normalized: bad arguments (null pointer, T < 1 or N < 1)";
        return 1;
    }
    if (mms_gae_ppo(device, rewards, dones, values, last_values, returns, advantages, stats, T, N, gamma, lam, nullptr)) return 1;
    return mms_adv_normalize(device, advantages, stats, (int64_t)T * N, nullptr);
}
MMS_API int mms_gae_marl(int device, const float* rewards, const float* value_preds, const float* masks, float* returns, int32_t T, int64_t N,
                         float gamma, float lam, int32_t use_norm, const float* norm_mean, const float* norm_var, void*) {
    if (cpu_only(device)) return 1;
    const float mean = use_norm ? norm_mean[0] : 0.f, var = use_norm ? norm_var[0] : 1.f;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; i++) mms::gae_marl_column(rewards, value_preds, masks, returns, T, N, N, i, i, gamma, lam, use_norm, mean, var);
    return 0;
}
MMS_API int mms_gae_marl_agents(int device, const float* rewards, const float* value_preds, const float* masks, float* returns, int32_t T,
                                int64_t N, int32_t A, float gamma, float lam, int32_t use_norm, const float* norm_mean, const float* norm_var, void*) {
    if (cpu_only(device)) return 1;
    const int64_t cols = N * A;
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < cols; c++) {
        const int64_t i = c / A;
        const int k = (int)(c - i * A);
        mms::gae_marl_column(rewards, value_preds, masks, returns, T, N, cols, c, i, gamma, lam, use_norm, use_norm ? norm_mean[k] : 0.f,
                             use_norm ? norm_var[k] : 1.f);
    }
    return 0;
}

// one row of ActorCritic.act's sampling tail + the add_transitions stores (module.py:73-87, storage.py:33-47)
static void sample_row(const float* mean_row, float value_now, bool have_value, const float* log_std, uint64_t seed, int64_t* counters,
                       int64_t row_offset, int ref_scale, float* actions_out, float* act_slot, float* logp_slot, float* value_slot,
                       float* mu_slot, float* sigma_slot, int64_t row, int A) {
    const int64_t c = counters[row];
    // the wave's butterfly sum of the GPU kernel (lane j holds actions j, j + 64; xor 32, 16, ... 1), reproduced on 64 slots
    float lane[64];
    for (int l = 0; l < 64; l++) lane[l] = 0.f;
    for (int j = 0; j < A; j++) {
        float term;
        const float act = mms::ppo_sample_one(mean_row[j], log_std[j], seed, (uint64_t)(row_offset + row), (uint64_t)c, (uint32_t)j, ref_scale, term);
        lane[j & 63] += term;
        if (actions_out) actions_out[row * A + j] = act;
        if (act_slot) act_slot[row * A + j] = act;
        if (mu_slot) mu_slot[row * A + j] = mean_row[j];
        if (sigma_slot) sigma_slot[row * A + j] = log_std[j];
    }
    for (int m = 32; m >= 1; m >>= 1) {
        float nxt[64];
        for (int l = 0; l < 64; l++) nxt[l] = lane[l] + lane[l ^ m];
        memcpy(lane, nxt, sizeof(lane));
    }
    if (logp_slot) logp_slot[row] = lane[0];
    if (value_slot && have_value) value_slot[row] = value_now;
    counters[row] = c + 1;
}
MMS_API int mms_ppo_act(int device, const float* mean, const float* value, const float* log_std, uint64_t seed, int64_t* counters,
                        int64_t row_offset, int32_t reference_scale, float* actions_out, float* act_slot, float* logp_slot, float* value_slot,
                        float* mu_slot, float* sigma_slot, int64_t N, int32_t A, void*) {
    if (cpu_only(device)) return 1;
    if (!mean || !log_std || !counters || N < 0 || A <= 0 || A > 128) { g_error = "mms_ppo_act: bad arguments (A must be in 1..128)"; return 1; }
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < N; row++)
        sample_row(mean + row * A, value ? value[row] : 0.f, value != nullptr, log_std, seed, counters, row_offset, reference_scale, actions_out,
                   act_slot, logp_slot, value_slot, mu_slot, sigma_slot, row, A);
    return 0;
}
MMS_API int mms_ppo_heads_act(int device, const float* hidden, const float* weight, const float* bias, int32_t H, const float* value,
                              const float* vhidden, const float* vweight, const float* vbias, int32_t VH, const float* log_std, uint64_t seed,
                              int64_t* counters, int64_t row_offset, int32_t reference_scale, float* actions_out, float* act_slot,
                              float* logp_slot, float* value_slot, float* mu_slot, float* sigma_slot, int64_t N, int32_t A, void*) {
    if (cpu_only(device)) return 1;
    if (!hidden || !weight || !bias || !log_std || !counters || N < 0 || A <= 0 || A > 128 || H <= 0 || (H % 64) != 0) {
        g_error = "mms_ppo_heads_act: bad arguments (A must be in 1..128, H a positive multiple of 64)";
        return 1;
    }
    if (vhidden && (!vweight || !vbias || VH <= 0 || (VH % 4) != 0)) {
        g_error = "mms_ppo_heads_act: the value head needs weight, bias and a hidden width that is a multiple of 4";
        return 1;
    }
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < N; row++) {
        float mean[128];
        for (int j = 0; j < A; j++) {
            float s = 0.f;
            for (int k = 0; k < H; k++) s = fmaf(hidden[row * (int64_t)H + k], weight[(int64_t)j * H + k], s);
            mean[j] = s + bias[j];
        }
        float v = value ? value[row] : 0.f;
        if (vhidden) {
            float s = 0.f;
            for (int k = 0; k < VH; k++) s = fmaf(vhidden[row * (int64_t)VH + k], vweight[k], s);
            v = s + vbias[0];
        }
        sample_row(mean, v, vhidden != nullptr || value != nullptr, log_std, seed, counters, row_offset, reference_scale, actions_out, act_slot,
                   logp_slot, value_slot, mu_slot, sigma_slot, row, A);
    }
    return 0;
}
static float act_fn(float v, int act) {
    if (act == 1) return (v > 0.f) ? v : (expf(v) - 1.f);
    if (act == 2) return fmaxf(v, 0.f);
    if (act == 3) return tanhf(v);
    return v;
}
MMS_API int mms_linear2_act(int device, int64_t M, int32_t N, int32_t K, const float* x0, const float* w0, const float* b0, float* y0,
                            const float* x1, const float* w1, const float* b1, float* y1, int32_t act, void*) {
    if (cpu_only(device)) return 1;
    if (!x0 || !w0 || !b0 || !y0 || M < 0 || N <= 0 || K <= 0 || (K % 4) != 0 || act < 0 || act > 3) {
        g_error = "mms_linear2_act: bad arguments (K must be a positive multiple of 4, act 0..3)";
        return 1;
    }
    const bool two = x1 || w1 || b1 || y1;
    if (two && !(x1 && w1 && b1 && y1)) { g_error = "mms_linear2_act: the second problem needs all four pointers"; return 1; }
    const float* xs[2] = {x0, x1};
    const float* ws[2] = {w0, w1};
    const float* bs[2] = {b0, b1};
    float* ys[2] = {y0, y1};
    for (int g = 0; g < (two ? 2 : 1); g++) {
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; m++)
            for (int n = 0; n < N; n++) {
                float s = 0.f;
                for (int k = 0; k < K; k++) s = fmaf(xs[g][m * K + k], ws[g][(int64_t)n * K + k], s);    // (the fp32 MFMA is an fmaf chain too)
                ys[g][m * N + n] = act_fn(s + bs[g][n], act);
            }
    }
    return 0;
}

// ---- split-operand layers (csrc/split_kernels.hip): fp32 carried as three bf16 planes, format P32 = bf16 [rows, KC, 3, 32] ------------
// On the host the planes are summed back (a0 + a1 + a2 IS the fp32 number) and the product is the same fmaf chain as mms_linear2_act.
static inline uint16_t f2bf(float f) {                                    // round to nearest even, as v_cvt_pk_bf16_f32
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf2f(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline void split3(float v, uint16_t* p0, uint16_t* p1, uint16_t* p2) {
    *p0 = f2bf(v); v -= bf2f(*p0);
    *p1 = f2bf(v); v -= bf2f(*p1);
    *p2 = f2bf(v);
}
static inline float join3(const uint16_t* chunk, int j) { return (bf2f(chunk[j]) + bf2f(chunk[32 + j])) + bf2f(chunk[64 + j]); }

MMS_API int mms_split_planes(int device, int64_t rows, int32_t K, int32_t x_pitch, const float* x, void* planes, void*) {
    if (cpu_only(device)) return 1;
    if (x_pitch == 0) x_pitch = K;
    if (!x || !planes || rows < 0 || K <= 0 || x_pitch < K) { g_error = "mms_split_planes: bad arguments"; return 1; }
    const int KC = (K + 31) / 32;
    uint16_t* out = (uint16_t*)planes;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++)
        for (int kc = 0; kc < KC; kc++) {
            uint16_t* c = out + (r * KC + kc) * 96;
            for (int j = 0; j < 32; j++) {
                const int k = kc * 32 + j;
                split3(k < K ? x[r * x_pitch + k] : 0.f, c + j, c + 32 + j, c + 64 + j);
            }
        }
    return 0;
}

MMS_API int mms_split_planes_group(int device, int32_t groups, int64_t rows, int32_t K, int32_t x_pitch, const float* const* x, void* const* planes, void*) {
    if (cpu_only(device)) return 1;
    if (groups < 1 || groups > MMS_MAX_GROUPS) { g_error = "mms_split_planes_group: groups must be 1.." + std::to_string(MMS_MAX_GROUPS); return 1; }
    if (!x || !planes) { g_error = "mms_split_planes_group: bad arguments (x_pitch >= K)"; return 1; }
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !planes[g]) { g_error = "mms_split_planes_group: null or misaligned pointer in a group (planes 16-byte aligned)"; return 1; }
        if (mms_split_planes(device, rows, K, x_pitch, x[g], planes[g], nullptr)) return 1;
    }
    return 0;
}

// The layer on decoded operands, shared by the two plane formats: `wf` [N][KC * 32] and `decode_x(m, xr)` give the operands as the
// floats the planes stand for (times their row scales in the H32 format, undone by `rescale(m, n)` = 1 for P32), `encode_y(m, row)`
// stores the finished row (out_mode 0 / 1).
template <class DecodeX, class Rescale, class EncodeY>
static void split_layer_rows(int64_t M, int32_t N, int KC, const std::vector<float>& wf, const float* b, int32_t act, int32_t out_mode, const float* ln_s,
                             const float* ln_stat_in, float* ln_part_out, const float* head_w, float* head_part, int32_t head_dim, DecodeX decode_x,
                             Rescale rescale, EncodeY encode_y) {
    const int slots = N / 64;
    const bool ln = ln_s != nullptr;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; m++) {
        std::vector<float> xr((size_t)KC * 32), row((size_t)N);
        decode_x(m, xr.data());
        for (int n = 0; n < N; n++) {
            float s = 0.f;
            const float* wr = wf.data() + (size_t)n * KC * 32;
            for (int k = 0; k < KC * 32; k++) s = fmaf(xr[k], wr[k], s);
            s *= rescale(m, n);
            if (ln) s = ln_stat_in[2 * m + 1] * (s - ln_stat_in[2 * m] * ln_s[n]);
            row[n] = act_fn(s + b[n], act);
        }
        if (ln)
            for (int sl = 0; sl < slots; sl++) {                                    // (sum, M2 about the slot mean) per 64 columns
                float sum = 0.f, m2 = 0.f;
                for (int n = 64 * sl; n < 64 * sl + 64; n++) sum += row[n];
                const float mean = sum * (1.f / 64.f);
                for (int n = 64 * sl; n < 64 * sl + 64; n++) m2 += (row[n] - mean) * (row[n] - mean);
                ln_part_out[((size_t)sl * M + m) * 2 + 0] = sum;
                ln_part_out[((size_t)sl * M + m) * 2 + 1] = m2;
            }
        if (out_mode == 2) {
            for (int sl = 0; sl < slots; sl++)
                for (int j = 0; j < head_dim; j++) {
                    float p = 0.f;
                    for (int n = 64 * sl; n < 64 * sl + 64; n++) p += row[n] * head_w[(size_t)j * N + n];
                    head_part[((size_t)sl * M + m) * ((head_dim + 3) & ~3) + j] = p;
                }
        } else {
            encode_y(m, row.data());
        }
    }
}

static int split_layer_check(const char* fn, int32_t groups, const void* x, const void* w, const void* b, int64_t M, int32_t N, int32_t K, int32_t act,
                             int32_t out_mode, const void* y, const void* ln_s, const void* ln_stat_in, const void* ln_part_out, const void* head_w,
                             const void* head_part, const int32_t* head_dims) {
    const std::string f(fn);
    if (groups < 1 || groups > MMS_MAX_GROUPS) { g_error = f + ": groups must be 1.." + std::to_string(MMS_MAX_GROUPS); return 1; }
    if (!x || !w || !b || M < 0 || (M % 128) != 0 || N <= 0 || (N % 128) != 0 || K <= 0 || act < 0 || act > 3 || out_mode < 0 || out_mode > 2 || (out_mode != 2 && !y)) {
        g_error = f + ": bad arguments (M and N multiples of 128, act 0..3, out_mode 0..2)";
        return 1;
    }
    const bool ln = ln_s || ln_stat_in || ln_part_out;
    if (ln && (!ln_s || !ln_stat_in || !ln_part_out || act != 1 || out_mode == 0)) {
        g_error = f + ": the LayerNorm folds come together (ln_s, ln_stat_in, ln_part_out), with act = ELU and out_mode 1 or 2";
        return 1;
    }
    if (out_mode == 2 && (!ln || !head_w || !head_part || !head_dims)) {
        g_error = f + ": out_mode 2 needs the LayerNorm folds, head_w, head_part and head_dims";
        return 1;
    }
    if (out_mode == 2)
        for (int g = 0; g < groups; g++)
            if (head_dims[g] < 1 || head_dims[g] > 16) { g_error = "split layer, out_mode 2: 1 <= head_dims[g] <= 16"; return 1; }
    return 0;
}

MMS_API int mms_linear_group_act_split(int device, int32_t groups, int64_t M, int32_t N, int32_t K, const void* const* x, const void* const* w,
                                       const float* const* b, void* const* y, int32_t act, int32_t out_mode, const float* const* ln_s,
                                       const float* const* ln_stat_in, float* const* ln_part_out, const float* const* head_w, float* const* head_part,
                                       const int32_t* head_dims, void*) {
    if (cpu_only(device)) return 1;
    if (split_layer_check("mms_linear_group_act_split", groups, x, w, b, M, N, K, act, out_mode, y, ln_s, ln_stat_in, ln_part_out, head_w, head_part, head_dims)) return 1;
    const bool ln = ln_s != nullptr;
    const int KC = (K + 31) / 32, NC = N / 32;
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !w[g] || !b[g] || (out_mode != 2 && !y[g]) || (ln && (!ln_s[g] || !ln_stat_in[g] || !ln_part_out[g])) ||
            (out_mode == 2 && (!head_w[g] || !head_part[g]))) {
            g_error = "mms_linear_group_act_split: null pointer in a group";
            return 1;
        }
        const uint16_t* xp = (const uint16_t*)x[g];
        const uint16_t* wp = (const uint16_t*)w[g];
        std::vector<float> wf((size_t)N * KC * 32);
        for (int64_t n = 0; n < N; n++)
            for (int k = 0; k < KC * 32; k++) wf[n * KC * 32 + k] = join3(wp + (n * KC + k / 32) * 96, k % 32);
        void* yg = out_mode != 2 ? y[g] : nullptr;
        split_layer_rows(M, N, KC, wf, b[g], act, out_mode, ln ? ln_s[g] : nullptr, ln ? ln_stat_in[g] : nullptr, ln ? ln_part_out[g] : nullptr,
                         out_mode == 2 ? head_w[g] : nullptr, out_mode == 2 ? head_part[g] : nullptr, out_mode == 2 ? head_dims[g] : 0,
                         [&](int64_t m, float* xr) { for (int k = 0; k < KC * 32; k++) xr[k] = join3(xp + (m * KC + k / 32) * 96, k % 32); },
                         [](int64_t, int) { return 1.f; },
                         [&](int64_t m, const float* row) {
                             if (out_mode == 1) {
                                 for (int n = 0; n < N; n++) {
                                     uint16_t* c = (uint16_t*)yg + (m * NC + n / 32) * 96 + n % 32;
                                     split3(row[n], c, c + 32, c + 64);
                                 }
                             } else {
                                 memcpy((float*)yg + m * N, row, (size_t)N * 4);
                             }
                         });
    }
    return 0;
}

// ---- the same with two scaled fp16 planes per operand (csrc/split16_kernels.hip), format H32 = f16 [rows, KC, 2, 32] -------------------
using mms::f2h;
using mms::h2f;
static inline void pow2_scale(float bound, float& scale, float& inv) {    // as split16_kernels.hip
    int e = 14;
    if (bound > 0.f) (void)frexpf(bound, &e);
    int sh = 14 - e;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    scale = ldexpf(1.f, sh);
    inv = ldexpf(1.f, -sh);
}
static inline void split2(float t, uint16_t* hi, uint16_t* lo) {
    *hi = f2h(t);
    *lo = f2h((t - h2f(*hi)) * 2048.f);
}
static inline float join2(const uint16_t* chunk, int j) { return h2f(chunk[j]) + h2f(chunk[32 + j]) * (1.f / 2048.f); }

MMS_API int mms_split_planes16_group(int device, int32_t groups, int64_t rows, int32_t K, int32_t x_pitch, const float* const* x, void* const* planes,
                                     float* const* scale, float* const* inv, int32_t nchains, int32_t L, const float* const* chain,
                                     float* const* chain_scale, float* const* chain_inv, float* const* stat, float eps, void*) {
    if (cpu_only(device)) return 1;
    if (groups < 1 || groups > MMS_MAX_GROUPS) { g_error = "mms_split_planes16_group: groups must be 1.." + std::to_string(MMS_MAX_GROUPS); return 1; }
    if (x_pitch == 0) x_pitch = K;
    if (!x || !planes || !scale || !inv || rows < 0 || K <= 0 || x_pitch < K || nchains < 0 || L < 0 || (nchains > 0 && (L < 1 || !chain || !chain_scale || !chain_inv))) {
        g_error = "mms_split_planes16_group: bad arguments (x_pitch >= K; nchains > 0 needs L >= 1, chain, chain_scale, chain_inv)";
        return 1;
    }
    const int KC = (K + 31) / 32;
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !planes[g] || (nchains > 0 && (!chain[g] || !chain_scale[g] || !chain_inv[g]))) {
            g_error = "mms_split_planes16_group: null or misaligned pointer in a group (planes 16-byte aligned)";
            return 1;
        }
        if (stat && !stat[g]) { g_error = "mms_split_planes16_group: null or misaligned stat pointer in a group"; return 1; }
        uint16_t* out = (uint16_t*)planes[g];
        const float* xg = x[g];
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < rows; r++) {
            float big = 0.f;
            for (int k = 0; k < K; k++) big = fmaxf(big, fabsf(xg[r * x_pitch + k]));
            float sc, iv;
            pow2_scale(big, sc, iv);
            for (int kc = 0; kc < KC; kc++) {
                uint16_t* c = out + (r * KC + kc) * 64;
                for (int j = 0; j < 32; j++) {
                    const int k = kc * 32 + j;
                    split2(k < K ? xg[r * x_pitch + k] * sc : 0.f, c + j, c + 32 + j);
                }
            }
            if (scale[g]) scale[g][r] = sc;
            if (inv[g]) inv[g][r] = iv;
            if (stat) {                                                   // two-pass LayerNorm statistics of the row
                float sum = 0.f, m2 = 0.f;
                for (int k = 0; k < K; k++) sum += xg[r * x_pitch + k];
                const float mean = sum / (float)K;
                for (int k = 0; k < K; k++) m2 += (xg[r * x_pitch + k] - mean) * (xg[r * x_pitch + k] - mean);
                stat[g][2 * r] = mean;
                stat[g][2 * r + 1] = 1.0f / sqrtf(m2 / (float)K + eps);
            }
            for (int c = 0; c < nchains; c++) {
                float bound = big;
                for (int l = 0; l < L; l++) {
                    bound = (chain[g][((size_t)c * L + l) * 2] * bound + chain[g][((size_t)c * L + l) * 2 + 1]) * 1.001f;
                    float s2, i2;
                    pow2_scale(bound, s2, i2);
                    chain_scale[g][((size_t)c * L + l) * rows + r] = s2;
                    chain_inv[g][((size_t)c * L + l) * rows + r] = i2;
                }
            }
        }
    }
    return 0;
}

// The device-side refresh of the weights' planes and of the bound chain (include/mms.h).  l1 follows the kernel's summation order
// (split16_planes_kernel: P lanes per row walk the row's 8-element pieces p = lane, lane + P, ...; xor butterfly over the lanes), so
// that the bound -- and with it every hidden activation's power-of-two scale -- is the same number on both builds.
MMS_API int mms_weight_planes16_group(int device, int32_t groups, const int64_t* N, const int32_t* K, const float* const* w, void* const* planes,
                                      float* const* scale, float* const* inv, float* const* l1, void*) {
    if (cpu_only(device)) return 1;
    if (groups < 1 || groups > MMS_MAX_GROUPS) { g_error = "mms_weight_planes16_group: groups must be 1.." + std::to_string(MMS_MAX_GROUPS); return 1; }
    if (!N || !K || !w || !planes || !scale || !inv) { g_error = "mms_weight_planes16_group: bad arguments (null array)"; return 1; }
    for (int g = 0; g < groups; g++) {
        if (N[g] < 0 || K[g] <= 0) { g_error = "mms_weight_planes16_group: bad shape in a group (N >= 0, K > 0)"; return 1; }
        if (!w[g] || !planes[g] || !scale[g] || !inv[g]) { g_error = "mms_weight_planes16_group: null or misaligned pointer in a group (planes and inv 16-byte aligned)"; return 1; }
    }
    for (int g = 0; g < groups; g++) {
        if (N[g] == 0) continue;
        if (mms_split_planes16_group(device, 1, N[g], K[g], K[g], w + g, planes + g, scale + g, inv + g, 0, 0, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr)) return 1;
        if (!l1 || !l1[g]) continue;
        const int Kg = K[g], KC = (Kg + 31) / 32, pieces = KC * 4;
        int P = 1;
        while (P < pieces && P < 64) P <<= 1;
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < N[g]; r++) {
            float lane[64];
            for (int sub = 0; sub < P; sub++) {
                float a = 0.f;
                for (int p = sub; p < pieces; p += P)
                    for (int j = 0; j < 8; j++) { const int k = p * 8 + j; a += (k < Kg) ? fabsf(w[g][r * Kg + k]) : 0.f; }
                lane[sub] = a;
            }
            for (int m = P >> 1; m >= 1; m >>= 1) {
                float nxt[64];
                for (int i = 0; i < P; i++) nxt[i] = lane[i] + lane[i ^ m];
                for (int i = 0; i < P; i++) lane[i] = nxt[i];
            }
            l1[g][r] = lane[0];
        }
    }
    return 0;
}

MMS_API int mms_chain_refresh16(int device, int32_t nchains, int32_t L, const float* const* l1, const float* const* bias, const int32_t* n, float* chain,
                                float bound0, int64_t rows, float* chain_scale, float* chain_inv, void*) {
    if (cpu_only(device)) return 1;
    if (nchains < 1 || L < 1 || (int64_t)nchains * L > MMS_MAX_GROUPS || !l1 || !n || !chain || rows < 0 || (rows > 0 && (!chain_scale || !chain_inv || !(bound0 >= 0.f)))) {
        g_error = "mms_chain_refresh16: bad arguments (nchains, L >= 1, nchains * L <= " + std::to_string(MMS_MAX_GROUPS) + "; rows > 0 needs chain_scale, chain_inv, bound0 >= 0)";
        return 1;
    }
    for (int e = 0; e < nchains * L; e++)
        if (!l1[e] || n[e] < 0) { g_error = "mms_chain_refresh16: null pointer or negative count in an entry"; return 1; }
    for (int e = 0; e < nchains * L; e++) {
        float m = 0.f, b = 0.f;
        for (int i = 0; i < n[e]; i++) {
            m = fmaxf(m, l1[e][i]);
            if (bias && bias[e]) b = fmaxf(b, fabsf(bias[e][i]));
        }
        chain[2 * e] = m;
        chain[2 * e + 1] = b;
    }
    for (int c = 0; c < nchains && rows > 0; c++) {
        float bound = bound0;
        for (int l = 0; l < L; l++) {
            bound = (chain[((size_t)c * L + l) * 2] * bound + chain[((size_t)c * L + l) * 2 + 1]) * 1.001f;
            float sc, iv;
            pow2_scale(bound, sc, iv);
            for (int64_t r = 0; r < rows; r++) {
                chain_scale[((size_t)c * L + l) * rows + r] = sc;
                chain_inv[((size_t)c * L + l) * rows + r] = iv;
            }
        }
    }
    return 0;
}

// The folded-LayerNorm layers' weight side (csrc/fold16_kernels.hip; include/mms.h)
MMS_API int mms_fold_planes16_group(int device, int32_t groups, const int64_t* N, const int32_t* K, const float* const* w, const float* const* gamma,
                                    const float* const* beta, const float* const* bias, void* const* planes, float* const* inv, float* const* s_out,
                                    float* const* c_out, float* const* rb, float* const* wt, void*) {
    if (cpu_only(device)) return 1;
    if (groups < 1 || groups > MMS_MAX_GROUPS) { g_error = "mms_fold_planes16_group: groups must be 1.." + std::to_string(MMS_MAX_GROUPS); return 1; }
    if (!N || !K || !w) { g_error = "mms_fold_planes16_group: bad arguments (null array)"; return 1; }
    for (int g = 0; g < groups; g++) {
        if (N[g] < 0 || K[g] <= 0) { g_error = "mms_fold_planes16_group: bad shape in a group (N >= 0, K > 0)"; return 1; }
        if (!w[g] || (planes && planes[g] && (!inv || !inv[g]))) {
            g_error = "mms_fold_planes16_group: null or misaligned pointer in a group (planes 16-byte aligned and with inv)";
            return 1;
        }
    }
    for (int g = 0; g < groups; g++) {
        const int Kg = K[g], KC = (Kg + 31) / 32;
        const float* gm = gamma ? gamma[g] : nullptr;
        const float* bt = beta ? beta[g] : nullptr;
        uint16_t* out = planes ? (uint16_t*)planes[g] : nullptr;
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < N[g]; r++) {
            const float* wr = w[g] + r * Kg;
            float big = 0.f, s = 0.f, c = 0.f, l2 = 0.f;
            for (int k = 0; k < Kg; k++) {
                const float v = gm ? wr[k] * gm[k] : wr[k];
                big = fmaxf(big, fabsf(v));
                s += v;
                l2 += v * v;
                if (bt) c += wr[k] * bt[k];
            }
            if (bias && bias[g]) c += bias[g][r];
            float sc, iv;
            pow2_scale(big, sc, iv);
            for (int kc = 0; kc < KC && out; kc++) {
                uint16_t* ch = out + (r * KC + kc) * 64;
                for (int j = 0; j < 32; j++) {
                    const int k = kc * 32 + j;
                    split2(k < Kg ? (gm ? wr[k] * gm[k] : wr[k]) * sc : 0.f, ch + j, ch + 32 + j);
                }
            }
            if (wt && wt[g])
                for (int k = 0; k < Kg; k++) wt[g][r * Kg + k] = gm ? wr[k] * gm[k] : wr[k];
            if (inv && inv[g]) inv[g][r] = iv;
            if (s_out && s_out[g]) s_out[g][r] = s;
            if (c_out && c_out[g]) c_out[g][r] = c;
            if (rb && rb[g]) rb[g][r] = sqrtf(l2) * sqrtf((float)Kg) + fabsf(c);
        }
    }
    return 0;
}

MMS_API int mms_fold_scales16_group(int device, int32_t groups, const float* const* rb, const int32_t* n, int64_t M, float* const* scale1,
                                    float* const* ysc, float* const* yinv, void*) {
    if (cpu_only(device)) return 1;
    if (groups < 1 || groups > MMS_MAX_GROUPS) { g_error = "mms_fold_scales16_group: groups must be 1.." + std::to_string(MMS_MAX_GROUPS); return 1; }
    if (!rb || !n || M < 0) { g_error = "mms_fold_scales16_group: bad arguments"; return 1; }
    for (int g = 0; g < groups; g++)
        if (!rb[g] || n[g] < 0) { g_error = "mms_fold_scales16_group: null pointer or negative count in a group"; return 1; }
    for (int g = 0; g < groups; g++) {
        float m = 0.f;
        for (int i = 0; i < n[g]; i++) m = fmaxf(m, rb[g][i]);
        float bound = m * 1.001f;
        if (!(bound > 1e-30f)) bound = 1e-30f;
        float sc, iv;
        pow2_scale(bound, sc, iv);
        if (scale1 && scale1[g]) scale1[g][0] = sc;
        for (int64_t r = 0; r < M; r++) {
            if (ysc && ysc[g]) ysc[g][r] = sc;
            if (yinv && yinv[g]) yinv[g][r] = iv;
        }
    }
    return 0;
}

MMS_API int mms_linear_group_act_split16(int device, int32_t groups, int64_t M, int32_t N, int32_t K, const void* const* x, const void* const* w,
                                         const float* const* b, void* const* y, const float* const* x_inv, const float* const* w_inv,
                                         const float* const* y_scale, int32_t act, int32_t out_mode, const float* const* ln_s,
                                         const float* const* ln_stat_in, float* const* ln_part_out, const float* const* head_w, float* const* head_part,
                                         const int32_t* head_dims, void*) {
    if (cpu_only(device)) return 1;
    if (split_layer_check("mms_linear_group_act_split16", groups, x, w, b, M, N, K, act, out_mode, y, ln_s, ln_stat_in, ln_part_out, head_w, head_part, head_dims)) return 1;
    if (!x_inv || !w_inv || (out_mode == 1 && !y_scale)) { g_error = "mms_linear_group_act_split16: bad arguments (x_inv, w_inv, y_scale with out_mode 1)"; return 1; }
    const bool ln = ln_s != nullptr;
    const int KC = (K + 31) / 32, NC = N / 32;
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !w[g] || !b[g] || !x_inv[g] || !w_inv[g] || (out_mode != 2 && !y[g]) || (out_mode == 1 && !y_scale[g]) ||
            (ln && (!ln_s[g] || !ln_stat_in[g] || !ln_part_out[g])) || (out_mode == 2 && (!head_w[g] || !head_part[g]))) {
            g_error = "mms_linear_group_act_split16: null pointer in a group";
            return 1;
        }
        const uint16_t* xp = (const uint16_t*)x[g];
        const uint16_t* wp = (const uint16_t*)w[g];
        std::vector<float> wf((size_t)N * KC * 32);
        for (int64_t n = 0; n < N; n++)
            for (int k = 0; k < KC * 32; k++) wf[n * KC * 32 + k] = join2(wp + (n * KC + k / 32) * 64, k % 32);
        void* yg = out_mode != 2 ? y[g] : nullptr;
        const float* xi = x_inv[g];
        const float* wi = w_inv[g];
        const float* ys = out_mode == 1 ? y_scale[g] : nullptr;
        split_layer_rows(M, N, KC, wf, b[g], act, out_mode, ln ? ln_s[g] : nullptr, ln ? ln_stat_in[g] : nullptr, ln ? ln_part_out[g] : nullptr,
                         out_mode == 2 ? head_w[g] : nullptr, out_mode == 2 ? head_part[g] : nullptr, out_mode == 2 ? head_dims[g] : 0,
                         [&](int64_t m, float* xr) { for (int k = 0; k < KC * 32; k++) xr[k] = join2(xp + (m * KC + k / 32) * 64, k % 32); },
                         [&](int64_t m, int n) { return wi[n] * xi[m]; },
                         [&](int64_t m, const float* row) {
                             if (out_mode == 1) {
                                 for (int n = 0; n < N; n++) {
                                     uint16_t* c = (uint16_t*)yg + (m * NC + n / 32) * 64 + n % 32;
                                     split2(row[n] * ys[m], c, c + 32);
                                 }
                             } else {
                                 memcpy((float*)yg + m * N, row, (size_t)N * 4);
                             }
                         });
    }
    return 0;
}

static void chan_combine(const float* part, int64_t M, int64_t row, int slots, float& mean, float& m2) {
    float sum = 0.f;
    for (int k = 0; k < slots; k++) sum += part[((size_t)k * M + row) * 2];
    mean = sum / (64.f * (float)slots);
    m2 = 0.f;
    for (int k = 0; k < slots; k++) {
        const float d = part[((size_t)k * M + row) * 2] * (1.f / 64.f) - mean;
        m2 += part[((size_t)k * M + row) * 2 + 1] + 64.f * d * d;
    }
}

MMS_API int mms_row_stats_chan_group(int device, int32_t groups, int64_t M, int32_t slots, const float* const* part, float* const* stat, float eps, void*) {
    if (cpu_only(device)) return 1;
    if (groups < 1 || groups > MMS_MAX_GROUPS) { g_error = "mms_row_stats_chan_group: groups must be 1.." + std::to_string(MMS_MAX_GROUPS); return 1; }
    if (!part || !stat || M < 0 || slots < 1) { g_error = "mms_row_stats_chan_group: bad arguments"; return 1; }
    for (int g = 0; g < groups; g++) {
        if (!part[g] || !stat[g]) { g_error = "mms_row_stats_chan_group: null pointer in a group"; return 1; }
        for (int64_t r = 0; r < M; r++) {
            float mean, m2;
            chan_combine(part[g], M, r, slots, mean, m2);
            stat[g][2 * r] = mean;
            stat[g][2 * r + 1] = 1.0f / sqrtf(m2 / (64.f * (float)slots) + eps);
        }
    }
    return 0;
}

MMS_API int mms_marl_heads_finish(int device, int32_t groups, int64_t M, int32_t slots, const float* const* part, const float* const* head_part,
                                  const float* const* hs, const float* const* hc, const int32_t* A, const float* const* std, float* const* out,
                                  float* const* logp, const int32_t* out_pitch, int64_t* const* counters, uint64_t seed, int64_t row_offset, float eps, void*) {
    if (cpu_only(device)) return 1;
    if (groups < 1 || groups > MMS_MAX_GROUPS) { g_error = "mms_marl_heads_finish: groups must be 1.." + std::to_string(MMS_MAX_GROUPS); return 1; }
    if (!part || !head_part || !hs || !hc || !A || !out || M < 0 || slots < 1) { g_error = "mms_marl_heads_finish: bad arguments"; return 1; }
    for (int g = 0; g < groups; g++) {
        if (!part[g] || !head_part[g] || !hs[g] || !hc[g] || !out[g] || A[g] < 1 || A[g] > 16) {
            g_error = "mms_marl_heads_finish: null pointer or output width outside 1..16 in a group";
            return 1;
        }
        const int pitch = out_pitch ? out_pitch[g] : A[g];
        if (pitch < A[g]) { g_error = "mms_marl_heads_finish: out_pitch below the output width"; return 1; }
        const float* sd = std ? std[g] : nullptr;
        int64_t* cnt = (sd && counters) ? counters[g] : nullptr;
        for (int64_t r = 0; r < M; r++) {
            float mean, m2;
            chan_combine(part[g], M, r, slots, mean, m2);
            const float rstd = 1.0f / sqrtf(m2 / (64.f * (float)slots) + eps);
            const int64_t c = cnt ? cnt[r] : 0;
            for (int j = 0; j < A[g]; j++) {
                float dot = 0.f;
                for (int k = 0; k < slots; k++) dot += head_part[g][((size_t)k * M + r) * ((A[g] + 3) & ~3) + j];
                const float mu = rstd * (dot - mean * hs[g][j]) + hc[g][j];
                if (!sd) { out[g][r * pitch + j] = mu; continue; }
                const float z = mms::rand_normal(seed + (uint64_t)g, (uint64_t)(row_offset + r), (uint64_t)c, (uint32_t)j);
                out[g][r * pitch + j] = mu + sd[j] * z;
                if (logp && logp[g]) logp[g][r * pitch + j] = -0.5f * z * z - logf(sd[j]) - 0.9189385332046727f;
            }
            if (cnt) cnt[r] = c + 1;
        }
    }
    return 0;
}

// ---- grouped policy inference (the same three operators as the HIP build; plain loops) -------------------------------------------
static bool bad_groups(int32_t groups, const char* what) {
    if (groups >= 1 && groups <= MMS_MAX_GROUPS) return false;
    g_error = std::string(what) + ": groups must be 1.." + std::to_string(MMS_MAX_GROUPS);
    return true;
}
MMS_API int mms_linear_group_act(int device, int32_t groups, int64_t M, int32_t N, int32_t K, const float* const* x, const float* const* w,
                                 const float* const* b, float* const* y, int32_t act, const float* const* ln_s, const float* const* ln_stat_in,
                                 float* const* ln_part_out, void*) {
    if (cpu_only(device)) return 1;
    if (bad_groups(groups, "mms_linear_group_act")) return 1;
    if (!x || !w || !b || !y || M < 0 || N <= 0 || K <= 0 || (K % 4) != 0 || act < 0 || act > 3) {
        g_error = "mms_linear_group_act: bad arguments (K must be a positive multiple of 4, act 0..3)";
        return 1;
    }
    if ((ln_stat_in != nullptr) != (ln_s != nullptr)) { g_error = "mms_linear_group_act: ln_stat_in and ln_s come together"; return 1; }
    if ((ln_stat_in || ln_part_out) && (act != 1 || M % 128 != 0 || N % 128 != 0 || K < 8)) {
        g_error = "mms_linear_group_act: the LayerNorm folds need act = ELU, M and N multiples of 128";
        return 1;
    }
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !w[g] || !b[g] || !y[g] || (ln_s && (!ln_s[g] || !ln_stat_in[g])) || (ln_part_out && !ln_part_out[g])) {
            g_error = "mms_linear_group_act: null pointer in a group";
            return 1;
        }
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; m++) {
            for (int n = 0; n < N; n++) {
                float s = 0.f;
                for (int k = 0; k < K; k++) s = fmaf(x[g][m * K + k], w[g][(int64_t)n * K + k], s);
                if (ln_s) s = ln_stat_in[g][2 * m + 1] * (s - ln_stat_in[g][2 * m] * ln_s[g][n]);      // rstd (W~ h - mean s)
                y[g][m * N + n] = act_fn(s + b[g][n], act);
            }
            if (ln_part_out)                                                                            // slot = 64 consecutive columns
                for (int slot = 0; slot < N / 64; slot++) {
                    float ps = 0.f, pq = 0.f;
                    for (int n = 64 * slot; n < 64 * slot + 64; n++) { const float v = y[g][m * N + n]; ps += v; pq += v * v; }
                    ln_part_out[g][((int64_t)slot * M + m) * 2] = ps;
                    ln_part_out[g][((int64_t)slot * M + m) * 2 + 1] = pq;
                }
        }
    }
    return 0;
}
MMS_API int mms_row_stats_group(int device, int32_t groups, int64_t M, int32_t slots, int32_t width, const float* const* part, float* const* stat,
                                float eps, void*) {
    if (cpu_only(device)) return 1;
    if (bad_groups(groups, "mms_row_stats_group")) return 1;
    if (!part || !stat || M < 0 || slots < 1 || width < 1) { g_error = "mms_row_stats_group: bad arguments"; return 1; }
    for (int g = 0; g < groups; g++) {
        if (!part[g] || !stat[g]) { g_error = "mms_row_stats_group: null pointer in a group"; return 1; }
        for (int64_t m = 0; m < M; m++) {
            float sum = 0.f, sq = 0.f;
            for (int k = 0; k < slots; k++) { sum += part[g][((int64_t)k * M + m) * 2]; sq += part[g][((int64_t)k * M + m) * 2 + 1]; }
            const float mean = sum / (float)width;
            const float var = fmaxf(sq / (float)width - mean * mean, 0.f);
            stat[g][2 * m] = mean;
            stat[g][2 * m + 1] = 1.0f / sqrtf(var + eps);
        }
    }
    return 0;
}
static void ln_row(const float* x, int K, float eps, float& mean, float& rstd) {
    float s = 0.f;
    for (int k = 0; k < K; k++) s += x[k];
    mean = s / (float)K;
    float q = 0.f;
    for (int k = 0; k < K; k++) q += (x[k] - mean) * (x[k] - mean);
    rstd = 1.0f / sqrtf(q / (float)K + eps);
}
MMS_API int mms_row_moments_group(int device, int32_t groups, int64_t M, int32_t K, int32_t x_pitch, const float* const* x, float* const* stat,
                                  float eps, void*) {
    if (cpu_only(device)) return 1;
    if (bad_groups(groups, "mms_row_moments_group")) return 1;
    if (x_pitch == 0) x_pitch = K;
    if (!x || !stat || M < 0 || K <= 0 || K > 4096 || x_pitch < K) { g_error = "mms_row_moments_group: bad arguments (1 <= K <= 4096)"; return 1; }
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !stat[g]) { g_error = "mms_row_moments_group: null pointer in a group"; return 1; }
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; m++) ln_row(x[g] + m * x_pitch, K, eps, stat[g][2 * m], stat[g][2 * m + 1]);
    }
    return 0;
}
MMS_API int mms_layernorm_group(int device, int32_t groups, int64_t M, int32_t K, int32_t Kp, int32_t x_pitch, const float* const* x,
                                const float* const* gamma, const float* const* beta, float* const* y, float eps, void*) {
    if (cpu_only(device)) return 1;
    if (bad_groups(groups, "mms_layernorm_group")) return 1;
    if (x_pitch == 0) x_pitch = K;
    if (!x || !gamma || !beta || !y || M < 0 || K <= 0 || K > 4096 || Kp < K || x_pitch < K) {
        g_error = "mms_layernorm_group: bad arguments (1 <= K <= 4096, Kp >= K, x_pitch >= K or 0)";
        return 1;
    }
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !gamma[g] || !beta[g] || !y[g]) { g_error = "mms_layernorm_group: null pointer in a group"; return 1; }
        if ((Kp != K || x_pitch != K) && x[g] == y[g]) { g_error = "mms_layernorm_group: in place needs Kp == x_pitch == K"; return 1; }
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; m++) {
            float mean, rstd;
            ln_row(x[g] + m * x_pitch, K, eps, mean, rstd);
            for (int k = 0; k < K; k++) y[g][m * Kp + k] = (x[g][m * x_pitch + k] - mean) * rstd * gamma[g][k] + beta[g][k];
            for (int k = K; k < Kp; k++) y[g][m * Kp + k] = 0.f;
        }
    }
    return 0;
}
MMS_API int mms_marl_heads_act(int device, int32_t groups, int64_t M, int32_t H, const float* const* h, const float* const* gamma,
                               const float* const* beta, const float* const* w, const float* const* b, const int32_t* A, const float* const* std,
                               float* const* out, float* const* logp, const int32_t* out_pitch, int64_t* const* counters, uint64_t seed,
                               int64_t row_offset, float eps, void*) {
    if (cpu_only(device)) return 1;
    if (bad_groups(groups, "mms_marl_heads_act")) return 1;
    if (!h || !gamma || !beta || !w || !b || !A || !out || M < 0 || H <= 0 || H > 1024) {
        g_error = "mms_marl_heads_act: bad arguments (1 <= H <= 1024)";
        return 1;
    }
    for (int g = 0; g < groups; g++) {
        if (!h[g] || !gamma[g] || !beta[g] || !w[g] || !b[g] || !out[g] || A[g] < 1 || A[g] > 16) {
            g_error = "mms_marl_heads_act: null pointer in a group, or outputs outside 1..16";
            return 1;
        }
        const int op = out_pitch ? out_pitch[g] : A[g];
        if (op < A[g]) { g_error = "mms_marl_heads_act: out_pitch below the number of outputs"; return 1; }
        const float* sd = std ? std[g] : nullptr;
        float* lp = logp ? logp[g] : nullptr;
        int64_t* cnt = counters ? counters[g] : nullptr;
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; m++) {
            float mean, rstd, xh[1024];
            if (eps >= 0.f) {
                ln_row(h[g] + m * H, H, eps, mean, rstd);
                for (int k = 0; k < H; k++) xh[k] = (h[g][m * H + k] - mean) * rstd * gamma[g][k] + beta[g][k];
            } else {
                for (int k = 0; k < H; k++) xh[k] = h[g][m * H + k];
            }
            const int64_t c = cnt ? cnt[m] : 0;
            for (int j = 0; j < A[g]; j++) {
                float p = 0.f;
                for (int k = 0; k < H; k++) p += xh[k] * w[g][(int64_t)j * H + k];
                p += b[g][j];
                if (sd) {
                    const float z = mms::rand_normal(seed + (uint64_t)g, (uint64_t)(row_offset + m), (uint64_t)c, (uint32_t)j);
                    p += sd[j] * z;
                    if (lp) lp[m * op + j] = -0.5f * z * z - logf(sd[j]) - 0.9189385332046727f;
                }
                out[g][m * op + j] = p;
            }
            if (sd && cnt) cnt[m] = c + 1;
        }
    }
    return 0;
}
