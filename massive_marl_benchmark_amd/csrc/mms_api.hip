// mms_api.hip -- the C ABI of include/mms.h: engine lifetime, named device buffers, launches.
// Host side only allocates, fills the construction-time scene (what create_sim .. prepare_sim do in the
// reference, agents/tasks/ten_ant.py:205-633) and enqueues kernels on the caller's stream.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mms.h"
#include "policy_args.h"
#include "step_args.h"

namespace mms {
hipError_t launch_step(const StepArgs& a, int task, hipStream_t stream);
bool step_layout_takes_head(int task, int num_envs, int num_agents, int packing);
hipError_t launch_gae_ppo(const float*, const uint8_t*, const float*, const float*, float*, float*, double*, int, int64_t, float, float, hipStream_t);
hipError_t launch_adv_normalize(float*, const double*, int64_t, hipStream_t);
hipError_t launch_gae_ppo_normalized(const float*, const uint8_t*, const float*, const float*, float*, float*, double*, int, int64_t, float, float, hipStream_t);
hipError_t launch_gae_marl(const float*, const float*, const float*, float*, int, int64_t, float, float, int, const float*, const float*, hipStream_t);
hipError_t launch_marl_views(const float*, float*, int64_t, int, int, int, hipStream_t);
hipError_t launch_gae_marl_agents(const float*, const float*, const float*, float*, int, int64_t, int, float, float, int, const float*, const float*, hipStream_t);
hipError_t launch_ppo_act(const float*, const float*, const float*, uint64_t, int64_t*, int64_t, int, float*, float*, float*, float*, float*, float*, int64_t, int, hipStream_t);
hipError_t launch_ppo_head_act(const float*, const float*, const float*, int, const float*, const float*, const float*, const float*, int, const float*, uint64_t,
                               int64_t*, int64_t, int, float*, float*, float*, float*, float*, float*, int64_t, int, hipStream_t);
}  // namespace mms

struct mms_buffer {
    const char* name;
    void* ptr;
    int64_t shape[4];
    int ndim;
    int dtype;
    size_t bytes;
    int64_t row_bytes;   // bytes per env (for indexed set_state); 0 if not per-env
};

struct mms_engine {
    mms_config cfg;
    mms_config* d_cfg = nullptr;
    int actors = 0, dofs = 0, num_actions = 0, obs_dim = 0, prev_dim = 0;
    float* obs_out = nullptr;
    void* obs_planes = nullptr;
    float obs_planes_scale = 1.f;
    const float* actions_in = nullptr;      // mms_bind_actions
    bool head_on = false;                   // mms_bind_policy_head: consumed (and cleared) by the next mms_step
    mms_policy_head head{};
    void* scratch = nullptr;                // staging of mms_set_state(env_ids)
    size_t scratch_bytes = 0;
    int write_raw_obs = 1, write_clipped_obs = 1;
    int dr_enabled = 0;
    float* rew_out = nullptr;
    uint8_t* done_out = nullptr;
    int packing = 1;
    std::vector<mms_buffer> bufs;
    std::string err;
};

static std::string g_create_error;

// Every entry point that touches the GPU runs on ITS device (the engine's, or the `device` argument) and leaves the caller's
// current device as it found it: two engines in one process, or a torch current device other than the engine's, must not make a
// launch pick another device's stream or caches, and torch allocations after a call must not move GPU.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) { err = hipSetDevice(device); changed = (err == hipSuccess); }
    }
    ~DeviceGuard() { if (changed) (void)hipSetDevice(prev); }
};

static size_t dtype_size(int dt) { return dt == MMS_F32 ? 4 : dt == MMS_I64 ? 8 : dt == MMS_I32 ? 4 : 1; }

static int fail(mms_engine* e, const std::string& msg) {
    if (e) e->err = msg; else g_create_error = msg;
    return 1;
}
#define MMS_HIP(e, call)                                                                              \
    do {                                                                                              \
        hipError_t err_ = (call);                                                                     \
        if (err_ != hipSuccess) return fail(e, std::string(#call) + ": " + hipGetErrorString(err_));  \
    } while (0)

static mms_buffer* find(mms_engine* e, const char* name) {
    for (auto& b : e->bufs)
        if (!strcmp(b.name, name)) return &b;
    return nullptr;
}

static int add_buffer(mms_engine* e, const char* name, int dtype, std::initializer_list<int64_t> shape, int64_t rows_per_env) {
    mms_buffer b{};
    b.name = name;
    b.dtype = dtype;
    b.ndim = (int)shape.size();
    size_t n = 1;
    int i = 0;
    for (int64_t s : shape) { b.shape[i++] = s; n *= (size_t)s; }
    b.bytes = n * dtype_size(dtype);
    b.row_bytes = (rows_per_env > 0) ? (int64_t)(b.bytes / (size_t)e->cfg.num_envs) : 0;
    size_t alloc = b.bytes ? b.bytes : 16;
    MMS_HIP(e, hipMalloc(&b.ptr, alloc));
    MMS_HIP(e, hipMemset(b.ptr, 0, alloc));
    e->bufs.push_back(b);
    return 0;
}

// rows of a per-env buffer scattered to their env slots in one launch (mms_set_state with env ids): block = row, 4-byte words
__global__ void __launch_bounds__(256) scatter_rows_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, const int64_t* __restrict__ ids,
                                                           int64_t row_words) {
    const int64_t i = blockIdx.x;
    uint32_t* d = dst + ids[i] * row_words;
    const uint32_t* s = src + i * row_words;
    for (int64_t k = threadIdx.x; k < row_words; k += 256) d[k] = s[k];
}

extern "C" {

__attribute__((visibility("default"))) int mms_abi_version(void) { return MMS_ABI_VERSION; }

__attribute__((visibility("default"))) const char* mms_last_error(mms_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

__attribute__((visibility("default"))) int mms_create(const mms_config* cfg, mms_handle* out) {
    if (!cfg || !out) return fail(nullptr, "mms_create: null argument");
    if (cfg->abi_version != MMS_ABI_VERSION) return fail(nullptr, "mms_create: ABI version mismatch");
    if (cfg->num_envs <= 0 || cfg->num_agents <= 0) return fail(nullptr, "mms_create: num_envs and num_agents must be positive");
    if (cfg->device < 0) return fail(nullptr, "mms_create: this is the HIP build of the engine (no CPU fallback); device must be a HIP ordinal >= 0 -- device -1 is served by libmms_cpu.so, an explicit opt-in");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, "mms_create: no HIP device available (no CPU fallback)");
    if (cfg->device >= ndev) return fail(nullptr, "mms_create: device ordinal out of range");
    if (cfg->task == MMS_TASK_MULTI_INGENUITY && cfg->num_agents != 4) return fail(nullptr, "mms_create: MultiIngenuity has 4 helicopters per env");
    if (cfg->task == MMS_TASK_ONE_ANT && cfg->num_agents != 1) return fail(nullptr, "mms_create: OneAnt has one ant per env");
    if (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE && cfg->num_agents != 2) return fail(nullptr, "mms_create: MultiAntCircle has two ants per env");
    if (cfg->task != MMS_TASK_MULTI_INGENUITY && ((4 * cfg->num_agents + 7) & ~7) + 8 > 512)
        return fail(nullptr, "mms_create: at most 126 ants per env");
    DeviceGuard guard(cfg->device);
    MMS_HIP(nullptr, guard.err);
    mms_engine* e = new mms_engine();
    e->cfg = *cfg;
    if (const char* pk = getenv("MMS_PACKING")) e->packing = atoi(pk);   // A/B switch for profiling
    const int N = cfg->num_envs, A = cfg->num_agents;
    if (cfg->task == MMS_TASK_TEN_ANT) { e->actors = A + 1; e->dofs = 8 * A; e->num_actions = 8 * A; e->obs_dim = 38 * A + 8; e->prev_dim = 4 * A + 2; }
    else if (cfg->task == MMS_TASK_ONE_ANT) { e->actors = 2; e->dofs = 8; e->num_actions = 8; e->obs_dim = 60; e->prev_dim = 6; }
    else if (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE) { e->actors = A + 1; e->dofs = 8 * A; e->num_actions = 8 * A; e->obs_dim = 38 * A; e->prev_dim = 2 * A; }
    else if (cfg->task == MMS_TASK_MULTI_INGENUITY) { e->actors = A; e->dofs = 4 * A; e->num_actions = 6 * A; e->obs_dim = 13 * A; e->prev_dim = 3 * A; }
    else { delete e; return fail(nullptr, "mms_create: unknown task"); }
    int rc = 0;
    rc |= add_buffer(e, "actions", MMS_F32, {N, e->num_actions}, 1);
    rc |= add_buffer(e, "obs", MMS_F32, {N, e->obs_dim}, 1);
    rc |= add_buffer(e, "obs_clipped", MMS_F32, {N, e->obs_dim}, 1);
    rc |= add_buffer(e, "rew", MMS_F32, {N}, 1);
    rc |= add_buffer(e, "reset", MMS_I64, {N}, 1);
    rc |= add_buffer(e, "progress", MMS_I64, {N}, 1);
    rc |= add_buffer(e, "reset_count", MMS_I64, {N}, 1);
    rc |= add_buffer(e, "root_states", MMS_F32, {(int64_t)N * e->actors, 13}, 1);
    rc |= add_buffer(e, "initial_root_states", MMS_F32, {(int64_t)N * e->actors, 13}, 1);
    rc |= add_buffer(e, "dof_state", MMS_F32, {(int64_t)N * e->dofs, 2}, 1);
    rc |= add_buffer(e, "env_origin", MMS_F32, {N, 3}, 1);
    rc |= add_buffer(e, "prev", MMS_F32, {N, e->prev_dim}, 1);
    rc |= add_buffer(e, "reset_noise", MMS_F32, {N, 16}, 1);
    rc |= add_buffer(e, "foot_sensors", MMS_F32, {(int64_t)N * A, 24}, 1);
    rc |= add_buffer(e, "dr_params", MMS_F32, {(int64_t)N * A, MMS_DR_FLOATS}, 1);
    if (rc) { g_create_error = e->err; mms_destroy(e); return 1; }

    // construction-time scene (host), uploaded once
    std::vector<float> init((size_t)N * e->actors * 13, 0.f), origin((size_t)N * 3, 0.f), prev((size_t)N * e->prev_dim, 0.f);
    std::vector<int64_t> ones((size_t)N, 1);
    int64_t npr = (int64_t)sqrt((double)cfg->total_envs);
    if (npr < 1) npr = 1;
    for (int i = 0; i < N; i++) {
        int64_t gi = cfg->env_offset + i;
        origin[3 * (size_t)i + 0] = (float)(gi % npr) * 2.f * cfg->env_spacing;   // env grid: SURVEY.md B.2 convention
        origin[3 * (size_t)i + 1] = (float)(gi / npr) * 2.f * cfg->env_spacing;
        float* r = init.data() + (size_t)i * e->actors * 13;
        for (int k = 0; k < e->actors; k++) r[13 * k + 6] = 1.f;
        if (cfg->task != MMS_TASK_MULTI_INGENUITY) {
            for (int k = 0; k < A; k++) {                                        // ten_ant.py:339-358 / one_ant.py:234
                float off = (A == 1) ? 0.f : (1.5f + 3.f * (float)(k / 2)) * ((k % 2 == 0) ? -1.f : 1.f);
                r[13 * k + 0] = cfg->ant_start_x; r[13 * k + 1] = off; r[13 * k + 2] = cfg->ant_start_z;
                if (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE) {                       // multi_ant_circle.py:216-219: (3, 0, 1) and (-3, 0, 1)
                    r[13 * k + 0] = (k % 2 == 0) ? cfg->ant_start_x : -cfg->ant_start_x; r[13 * k + 1] = 0.f;
                }
            }
            for (int j = 0; j < 3; j++) r[13 * A + j] = cfg->box_start[j];       // ten_ant.py:494-495
        } else {
            static const float hy[4] = {2.f, -2.f, 6.f, -6.f};                   // multi_ingenuity.py:157-164
            for (int k = 0; k < A; k++) { r[13 * k + 0] = 0.f; r[13 * k + 1] = hy[k % 4]; r[13 * k + 2] = 1.f; }
        }
    }
    // caches start as the construction-time poses: what reset_idx reads from the not-yet-refreshed tensors on the
    // first step (ten_ant.py:870-882, one_ant.py:410-411), in the global frame
    for (int i = 0; i < N; i++) {
        const float* r = init.data() + (size_t)i * e->actors * 13;
        const float* o = origin.data() + 3 * (size_t)i;
        float* pv = prev.data() + (size_t)i * e->prev_dim;
        if (cfg->task == MMS_TASK_TEN_ANT) {
            const float* b = r + 13 * A;
            float bx = b[0] + o[0], by = b[1] + o[1];
            float ang = atanf((2.f * b[6] * b[5]) / (1.f - 2.f * b[5] * b[5]));   // ten_ant.py:935-947
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
            pv[4] = -4.f / cfg->dt; pv[5] = -4.f / cfg->dt;                       // one_ant.py:143-144
        } else if (cfg->task == MMS_TASK_MULTI_ANT_CIRCLE) {
            for (int k = 0; k < A; k++) { pv[2 * k] = r[13 * k] + o[0]; pv[2 * k + 1] = r[13 * k + 1] + o[1]; }   // multi_ant_circle.py:367-368
        }
    }
    hipError_t he = hipSuccess;
    auto up = [&](const char* name, const void* src, size_t bytes) {
        if (he == hipSuccess) he = hipMemcpy(find(e, name)->ptr, src, bytes, hipMemcpyHostToDevice);
    };
    up("initial_root_states", init.data(), init.size() * 4);
    up("root_states", init.data(), init.size() * 4);
    up("env_origin", origin.data(), origin.size() * 4);
    up("prev", prev.data(), prev.size() * 4);
    up("reset", ones.data(), ones.size() * 8);                                   // base_task.py:62-63
    {
        std::vector<float> dr((size_t)N * A * MMS_DR_FLOATS, 0.f);               // nominal: scales 1, limit offsets 0
        for (size_t k = 0; k < (size_t)N * A; k++)
            for (int j = 0; j < 17; j++) dr[k * MMS_DR_FLOATS + j] = 1.f;
        up("dr_params", dr.data(), dr.size() * 4);
    }
    if (he == hipSuccess) he = hipMalloc((void**)&e->d_cfg, sizeof(mms_config));
    if (he == hipSuccess) he = hipMemcpy(e->d_cfg, &e->cfg, sizeof(mms_config), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipDeviceSynchronize();
    if (he != hipSuccess) { g_create_error = std::string("mms_create: ") + hipGetErrorString(he); mms_destroy(e); return 1; }
    *out = e;
    return 0;
}

__attribute__((visibility("default"))) int mms_destroy(mms_handle h) {
    if (!h) return 0;
    DeviceGuard guard(h->cfg.device);                  // teardown: nothing useful to do with an error here
    (void)hipDeviceSynchronize();
    for (auto& b : h->bufs)
        if (b.ptr) (void)hipFree(b.ptr);
    if (h->d_cfg) (void)hipFree(h->d_cfg);
    if (h->scratch) (void)hipFree(h->scratch);
    delete h;
    return 0;
}

__attribute__((visibility("default"))) int mms_get_tensor(mms_handle h, const char* name, mms_tensor* out) {
    if (!h || !name || !out) return fail(h, "mms_get_tensor: null argument");
    mms_buffer* b = find(h, name);
    if (!b) return fail(h, std::string("mms_get_tensor: unknown buffer '") + name + "'");
    memset(out, 0, sizeof(*out));
    out->ptr = b->ptr;
    for (int i = 0; i < b->ndim; i++) out->shape[i] = b->shape[i];
    out->ndim = b->ndim;
    out->dtype = b->dtype;
    out->device = h->cfg.device;
    return 0;
}

static mms::StepArgs step_args(mms_handle h, int physics) {
    mms::StepArgs a{};
    a.cfg = h->d_cfg;
    a.actions = h->actions_in ? h->actions_in : (const float*)find(h, "actions")->ptr;
    a.obs = h->write_raw_obs ? (float*)find(h, "obs")->ptr : nullptr;
    a.obs_clipped = h->write_clipped_obs ? (float*)find(h, "obs_clipped")->ptr : nullptr;
    a.obs_out = h->obs_out;
    a.obs_planes = h->obs_planes; a.obs_planes_scale = h->obs_planes_scale;
    a.rew_out = h->rew_out;
    a.done_out = h->done_out;
    a.rew = (float*)find(h, "rew")->ptr;
    a.reset = (int64_t*)find(h, "reset")->ptr;
    a.progress = (int64_t*)find(h, "progress")->ptr;
    a.root_states = (float*)find(h, "root_states")->ptr;
    a.initial_root_states = (const float*)find(h, "initial_root_states")->ptr;
    a.dof_state = (float*)find(h, "dof_state")->ptr;
    a.env_origin = (const float*)find(h, "env_origin")->ptr;
    a.prev = (float*)find(h, "prev")->ptr;
    a.reset_noise = (const float*)find(h, "reset_noise")->ptr;
    a.foot_sensors = (float*)find(h, "foot_sensors")->ptr;
    a.reset_count = (int64_t*)find(h, "reset_count")->ptr;
    a.dr = h->dr_enabled ? (const float*)find(h, "dr_params")->ptr : nullptr;
    a.do_physics = physics;
    a.num_envs = h->cfg.num_envs;
    a.num_agents = h->cfg.num_agents;
    a.obs_dim = h->obs_dim;
    a.prev_dim = h->prev_dim;
    a.packing = h->packing;
    a.head_on = (physics && h->head_on) ? 1 : 0;
    if (a.head_on) a.head = h->head;
    return a;
}

static int do_step(mms_handle h, void* stream, int physics) {
    if (!h) return fail(nullptr, "mms_step: null handle");
    DeviceGuard guard(h->cfg.device);
    MMS_HIP(h, guard.err);
    mms::StepArgs a = step_args(h, physics);
    if (physics) h->head_on = false;                                  // the binding is for one step (the slot pointers move every step)
    MMS_HIP(h, mms::launch_step(a, h->cfg.task, (hipStream_t)stream));
    return 0;
}
__attribute__((visibility("default"))) int mms_step(mms_handle h, void* hip_stream) { return do_step(h, hip_stream, 1); }
__attribute__((visibility("default"))) int mms_post_step(mms_handle h, void* hip_stream) { return do_step(h, hip_stream, 0); }

__attribute__((visibility("default"))) int mms_reset_all(mms_handle h, void* hip_stream) {
    if (!h) return fail(nullptr, "mms_reset_all: null handle");
    DeviceGuard guard(h->cfg.device);
    MMS_HIP(h, guard.err);
    std::vector<int64_t> ones((size_t)h->cfg.num_envs, 1);
    MMS_HIP(h, hipMemcpyAsync(find(h, "reset")->ptr, ones.data(), ones.size() * 8, hipMemcpyHostToDevice, (hipStream_t)hip_stream));
    MMS_HIP(h, hipStreamSynchronize((hipStream_t)hip_stream));   // `ones` is a temporary
    return 0;
}

__attribute__((visibility("default"))) int mms_set_state(mms_handle h, const char* name, const void* src, int src_is_host, const int64_t* env_ids, int64_t n, void* hip_stream) {
    if (!h || !name || !src) return fail(h, "mms_set_state: null argument");
    mms_buffer* b = find(h, name);
    if (!b) return fail(h, std::string("mms_set_state: unknown buffer '") + name + "'");
    DeviceGuard guard(h->cfg.device);
    MMS_HIP(h, guard.err);
    hipStream_t s = (hipStream_t)hip_stream;
    hipMemcpyKind kind = src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (!env_ids) {
        MMS_HIP(h, hipMemcpyAsync(b->ptr, src, b->bytes, kind, s));
    } else {
        if (b->row_bytes <= 0) return fail(h, "mms_set_state: buffer is not per-env");
        if (n < 0) return fail(h, "mms_set_state: negative row count");
        for (int64_t i = 0; i < n; i++)                      // all ids are checked before anything is written
            if (env_ids[i] < 0 || env_ids[i] >= h->cfg.num_envs) return fail(h, "mms_set_state: env id out of range");
        if (n <= 16 || (b->row_bytes & 3) != 0) {
            for (int64_t i = 0; i < n; i++)
                MMS_HIP(h, hipMemcpyAsync((char*)b->ptr + env_ids[i] * b->row_bytes, (const char*)src + i * b->row_bytes, (size_t)b->row_bytes, kind, s));
        } else {
            // the indexed setter of the reference (gym.set_*_tensor_indexed, ten_ant.py:867-875) takes thousands of ids: one scatter
            // launch instead of one copy per env.  Staging (ids, and the rows when they come from the host) lives in engine scratch.
            const size_t id_bytes = (size_t)n * 8, row_total = (size_t)n * (size_t)b->row_bytes;
            const size_t need = id_bytes + (src_is_host ? row_total : 0);
            if (need > h->scratch_bytes) {
                if (h->scratch) MMS_HIP(h, hipFree(h->scratch));
                h->scratch = nullptr; h->scratch_bytes = 0;
                MMS_HIP(h, hipMalloc(&h->scratch, need));
                h->scratch_bytes = need;
            }
            MMS_HIP(h, hipMemcpyAsync(h->scratch, env_ids, id_bytes, hipMemcpyHostToDevice, s));
            const void* rows = src;
            if (src_is_host) {
                MMS_HIP(h, hipMemcpyAsync((char*)h->scratch + id_bytes, src, row_total, hipMemcpyHostToDevice, s));
                rows = (char*)h->scratch + id_bytes;
            }
            hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)n), dim3(256), 0, s, (uint32_t*)b->ptr, (const uint32_t*)rows,
                               (const int64_t*)h->scratch, b->row_bytes / 4);
            MMS_HIP(h, hipGetLastError());
            MMS_HIP(h, hipStreamSynchronize(s));                 // env_ids is the caller's host array; the scratch is reused by the next call
        }
    }
    if (src_is_host) MMS_HIP(h, hipStreamSynchronize(s));
    return 0;
}

__attribute__((visibility("default"))) int mms_bind_obs_out(mms_handle h, void* dst) {
    if (!h) return fail(nullptr, "mms_bind_obs_out: null handle");
    h->obs_out = (float*)dst;
    return 0;
}

__attribute__((visibility("default"))) int mms_bind_obs_planes16(mms_handle h, void* planes, float scale) {
    if (!h) return fail(nullptr, "mms_bind_obs_planes16: null handle");
    if (!planes) { h->obs_planes = nullptr; return 0; }
    if (h->cfg.task == MMS_TASK_MULTI_INGENUITY) return fail(h, "mms_bind_obs_planes16: not for the helicopter task (its policies' layers are 256 wide: exact-fp32 kernel)");
    int e = 0;
    if (!(scale > 0.f) || frexpf(scale, &e) != 0.5f) return fail(h, "mms_bind_obs_planes16: the scale must be a power of two");
    if (!(h->cfg.clip_obs * scale <= 16384.f)) return fail(h, "mms_bind_obs_planes16: clip_observations x scale must not exceed 2^14 (fp16 planes)");
    if ((reinterpret_cast<uintptr_t>(planes) & 15) != 0) return fail(h, "mms_bind_obs_planes16: the planes must be 16-byte aligned");
    h->obs_planes = planes;
    h->obs_planes_scale = scale;
    return 0;
}

__attribute__((visibility("default"))) int mms_bind_actions(mms_handle h, const float* src) {
    if (!h) return fail(nullptr, "mms_bind_actions: null handle");
    if (src && (reinterpret_cast<uintptr_t>(src) & 7) != 0) return fail(h, "mms_bind_actions: the action tensor must be 8-byte aligned");
    h->actions_in = src;
    return 0;
}

__attribute__((visibility("default"))) int mms_bind_policy_head(mms_handle h, const mms_policy_head* head) {
    if (!h) return fail(nullptr, "mms_bind_policy_head: null handle");
    if (!head) { h->head_on = false; return 0; }
    DeviceGuard guard(h->cfg.device);
    MMS_HIP(h, guard.err);
    if (h->dr_enabled || !mms::step_layout_takes_head(h->cfg.task, h->cfg.num_envs, h->cfg.num_agents, h->packing))
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
    h->head_on = true;
    return 0;
}

__attribute__((visibility("default"))) int mms_set_dr(mms_handle h, int32_t enable) {
    if (!h) return fail(nullptr, "mms_set_dr: null handle");
    if (enable && h->cfg.task == MMS_TASK_MULTI_INGENUITY) return fail(h, "mms_set_dr: the helicopter task has no randomised physical parameters");
    h->dr_enabled = enable != 0;
    return 0;
}

__attribute__((visibility("default"))) int mms_set_obs_outputs(mms_handle h, int32_t raw, int32_t clipped) {
    if (!h) return fail(nullptr, "mms_set_obs_outputs: null handle");
    h->write_raw_obs = raw != 0;
    h->write_clipped_obs = clipped != 0;
    return 0;
}

__attribute__((visibility("default"))) int mms_bind_rollout_out(mms_handle h, float* rew_out, uint8_t* done_out) {
    if (!h) return fail(nullptr, "mms_bind_rollout_out: null handle");
    h->rew_out = rew_out;
    h->done_out = done_out;
    return 0;
}

#define MMS_DEV(device)                                                                                \
    if ((device) < 0) { g_create_error = "no CPU path in this library: device must be a HIP ordinal (the CPU build is libmms_cpu.so)"; return 1; } \
    DeviceGuard guard_(device);                                                                        \
    if (guard_.err != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(guard_.err); return 1; }
#define MMS_FREE(call)                                                                                 \
    do {                                                                                               \
        hipError_t err_ = (call);                                                                      \
        if (err_ != hipSuccess) { g_create_error = std::string(#call) + ": " + hipGetErrorString(err_); return 1; } \
    } while (0)

__attribute__((visibility("default"))) int mms_marl_views(int device, const float* obs_clipped, float* obs_all, int64_t n, int32_t agents, int32_t per_agent, int32_t shared, void* s) {
    MMS_DEV(device)
    MMS_FREE(mms::launch_marl_views(obs_clipped, obs_all, n, agents, per_agent, shared, (hipStream_t)s));
    return 0;
}
__attribute__((visibility("default"))) int mms_gae_ppo(int device, const float* rewards, const uint8_t* dones, const float* values, const float* last_values, float* returns,
                float* advantages, double* stats, int32_t T, int64_t N, float gamma, float lam, void* s) {
    MMS_DEV(device)
    MMS_FREE(mms::launch_gae_ppo(rewards, dones, values, last_values, returns, advantages, stats, T, N, gamma, lam, (hipStream_t)s));
    return 0;
}
__attribute__((visibility("default"))) int mms_gae_ppo_normalized(int device, const float* rewards, const uint8_t* dones, const float* values, const float* last_values,
                                                                  float* returns, float* advantages, double* stats, int32_t T, int64_t N, float gamma, float lam, void* s) {
    MMS_DEV(device)
    if (!rewards || !dones || !values || !last_values || !returns || !advantages || !stats || T < 1 || N < 1) {
        g_create_error = "mms_gae_ppo_normalized: bad arguments (null pointer, T < 1 or N < 1)";
        return 1;
    }
    MMS_FREE(mms::launch_gae_ppo_normalized(rewards, dones, values, last_values, returns, advantages, stats, T, N, gamma, lam, (hipStream_t)s));
    return 0;
}
__attribute__((visibility("default"))) int mms_adv_normalize(int device, float* advantages, const double* stats, int64_t count, void* s) {
    MMS_DEV(device)
    MMS_FREE(mms::launch_adv_normalize(advantages, stats, count, (hipStream_t)s));
    return 0;
}
__attribute__((visibility("default"))) int mms_gae_marl(int device, const float* rewards, const float* value_preds, const float* masks, float* returns, int32_t T, int64_t N,
                 float gamma, float lam, int32_t use_norm, const float* norm_mean, const float* norm_var, void* s) {
    MMS_DEV(device)
    MMS_FREE(mms::launch_gae_marl(rewards, value_preds, masks, returns, T, N, gamma, lam, use_norm, norm_mean, norm_var, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_gae_marl_agents(int device, const float* rewards, const float* value_preds, const float* masks,
                                                               float* returns, int32_t T, int64_t N, int32_t A, float gamma, float lam,
                                                               int32_t use_norm, const float* norm_mean, const float* norm_var, void* s) {
    MMS_DEV(device)
    MMS_FREE(mms::launch_gae_marl_agents(rewards, value_preds, masks, returns, T, N, A, gamma, lam, use_norm, norm_mean, norm_var, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_ppo_act(int device, const float* mean, const float* value, const float* log_std, uint64_t seed,
                                                       int64_t* counters, int64_t row_offset, int32_t reference_scale, float* actions_out,
                                                       float* act_slot, float* logp_slot, float* value_slot, float* mu_slot, float* sigma_slot,
                                                       int64_t N, int32_t A, void* s) {
    MMS_DEV(device)
    if (!mean || !log_std || !counters || N < 0 || A <= 0 || A > 128) { g_create_error = "mms_ppo_act: bad arguments (A must be in 1..128)"; return 1; }
    MMS_FREE(mms::launch_ppo_act(mean, value, log_std, seed, counters, row_offset, reference_scale, actions_out, act_slot, logp_slot, value_slot,
                                 mu_slot, sigma_slot, N, A, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_ppo_heads_act(int device, const float* hidden, const float* weight, const float* bias, int32_t H,
                                                             const float* value, const float* vhidden, const float* vweight, const float* vbias,
                                                             int32_t VH, const float* log_std, uint64_t seed, int64_t* counters,
                                                             int64_t row_offset, int32_t reference_scale, float* actions_out, float* act_slot,
                                                             float* logp_slot, float* value_slot, float* mu_slot, float* sigma_slot, int64_t N,
                                                             int32_t A, void* s) {
    MMS_DEV(device)
    if (!hidden || !weight || !bias || !log_std || !counters || N < 0 || A <= 0 || A > 128 || H <= 0 || (H % 64) != 0) {
        g_create_error = "mms_ppo_heads_act: bad arguments (A must be in 1..128, H a positive multiple of 64)";
        return 1;
    }
    if (vhidden && (!vweight || !vbias || VH <= 0 || (VH % 4) != 0)) {
        g_create_error = "mms_ppo_heads_act: the value head needs weight, bias and a hidden width that is a multiple of 4";
        return 1;
    }
    MMS_FREE(mms::launch_ppo_head_act(hidden, weight, bias, H, value, vhidden, vweight, vbias, VH, log_std, seed, counters, row_offset, reference_scale,
                                      actions_out, act_slot, logp_slot, value_slot, mu_slot, sigma_slot, N, A, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_linear2_act(int device, int64_t M, int32_t N, int32_t K, const float* x0, const float* w0, const float* b0,
                                                           float* y0, const float* x1, const float* w1, const float* b1, float* y1, int32_t act,
                                                           void* s) {
    MMS_DEV(device)
    if (!x0 || !w0 || !b0 || !y0 || M < 0 || M > 0x7fffffff || N <= 0 || K <= 0 || (K % 4) != 0 || act < 0 || act > 3) {
        g_create_error = "mms_linear2_act: bad arguments (K must be a positive multiple of 4, act 0..3)";
        return 1;
    }
    const bool two = x1 || w1 || b1 || y1;
    if (two && !(x1 && w1 && b1 && y1)) { g_create_error = "mms_linear2_act: the second problem needs all four pointers"; return 1; }
    mms::LinearArgs a = {};
    a.x[0] = x0; a.x[1] = x1; a.w[0] = w0; a.w[1] = w1; a.b[0] = b0; a.b[1] = b1; a.y[0] = y0; a.y[1] = y1;
    a.M = (int)M; a.N = N; a.K = K; a.act = act;
    MMS_FREE(mms::launch_linear_act(a, two ? 2 : 1, (hipStream_t)s));
    return 0;
}

// ---- split-operand layers (split_kernels.hip) ---------------------------------------------------------------------------------
__attribute__((visibility("default"))) int mms_split_planes(int device, int64_t rows, int32_t K, int32_t x_pitch, const float* x, void* planes, void* s) {
    MMS_DEV(device)
    if (x_pitch == 0) x_pitch = K;
    if (!x || !planes || rows < 0 || K <= 0 || x_pitch < K || (x_pitch % 4) != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0 ||
        (reinterpret_cast<uintptr_t>(planes) & 15) != 0) {
        g_create_error = "mms_split_planes: bad arguments (x and planes 16-byte aligned, x_pitch >= K and a multiple of 4)";
        return 1;
    }
    MMS_FREE(mms::launch_split_planes(x, planes, rows, K, x_pitch, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_split_planes_group(int device, int32_t groups, int64_t rows, int32_t K, int32_t x_pitch,
                                                                  const float* const* x, void* const* planes, void* s) {
    MMS_DEV(device)
    if (groups < 1 || groups > mms::kMaxGroups) { g_create_error = "mms_split_planes_group: groups must be 1.." + std::to_string(mms::kMaxGroups); return 1; }
    if (x_pitch == 0) x_pitch = K;
    if (!x || !planes || rows < 0 || K <= 0 || x_pitch < K) { g_create_error = "mms_split_planes_group: bad arguments (x_pitch >= K)"; return 1; }
    mms::SplitPlanesArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !planes[g] || (reinterpret_cast<uintptr_t>(planes[g]) & 15) != 0 || (reinterpret_cast<uintptr_t>(x[g]) & 3) != 0) {
            g_create_error = "mms_split_planes_group: null or misaligned pointer in a group (planes 16-byte aligned)";
            return 1;
        }
        a.x[g] = x[g]; a.planes[g] = planes[g];
    }
    a.rows = rows; a.K = K; a.x_pitch = x_pitch;
    MMS_FREE(mms::launch_split_planes_group(a, groups, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_linear_group_act_split(int device, int32_t groups, int64_t M, int32_t N, int32_t K, const void* const* x,
                                                                      const void* const* w, const float* const* b, void* const* y, int32_t act,
                                                                      int32_t out_mode, const float* const* ln_s, const float* const* ln_stat_in,
                                                                      float* const* ln_part_out, const float* const* head_w, float* const* head_part,
                                                                      const int32_t* head_dims, void* s) {
    MMS_DEV(device)
    if (groups < 1 || groups > mms::kMaxGroups) { g_create_error = "mms_linear_group_act_split: groups must be 1.." + std::to_string(mms::kMaxGroups); return 1; }
    if (!x || !w || !b || M < 0 || M > 0x7fffffff || (M % 128) != 0 || N <= 0 || (N % 128) != 0 || K <= 0 || act < 0 || act > 3 || out_mode < 0 || out_mode > 2 ||
        (out_mode != 2 && !y)) {
        g_create_error = "mms_linear_group_act_split: bad arguments (M and N multiples of 128, act 0..3, out_mode 0..2)";
        return 1;
    }
    const bool ln = ln_s || ln_stat_in || ln_part_out;
    if (ln && (!ln_s || !ln_stat_in || !ln_part_out || act != 1 || out_mode == 0)) {
        g_create_error = "mms_linear_group_act_split: the LayerNorm folds come together (ln_s, ln_stat_in, ln_part_out), with act = ELU and out_mode 1 or 2";
        return 1;
    }
    if (out_mode == 2 && (!ln || !head_w || !head_part || !head_dims)) {
        g_create_error = "mms_linear_group_act_split: out_mode 2 needs the LayerNorm folds, head_w, head_part and head_dims";
        return 1;
    }
    mms::SplitLinearArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !w[g] || !b[g] || (out_mode != 2 && !y[g]) || (ln && (!ln_s[g] || !ln_stat_in[g] || !ln_part_out[g])) ||
            (out_mode == 2 && (!head_w[g] || !head_part[g]))) {
            g_create_error = "mms_linear_group_act_split: null pointer in a group";
            return 1;
        }
        uintptr_t bits = reinterpret_cast<uintptr_t>(x[g]) | reinterpret_cast<uintptr_t>(w[g]) | reinterpret_cast<uintptr_t>(b[g]);
        if (out_mode != 2) bits |= reinterpret_cast<uintptr_t>(y[g]);
        if (ln) bits |= reinterpret_cast<uintptr_t>(ln_s[g]) | (reinterpret_cast<uintptr_t>(ln_stat_in[g]) << 1) | (reinterpret_cast<uintptr_t>(ln_part_out[g]) << 1);
        if ((bits & 15) != 0) { g_create_error = "mms_linear_group_act_split: operands must be 16-byte aligned"; return 1; }
        a.x[g] = x[g]; a.w[g] = w[g]; a.b[g] = b[g]; a.y[g] = out_mode != 2 ? y[g] : nullptr;
        if (ln) { a.s[g] = ln_s[g]; a.stat_in[g] = ln_stat_in[g]; a.part_out[g] = ln_part_out[g]; }
        if (out_mode == 2) {
            if (head_dims[g] < 1 || head_dims[g] > 16) { g_create_error = "split layer, out_mode 2: 1 <= head_dims[g] <= 16"; return 1; }
            a.head_w[g] = head_w[g]; a.head_part[g] = head_part[g]; a.hdims[g] = head_dims[g];
        }
    }
    a.M = (int)M; a.N = N; a.KC = (K + 31) / 32; a.act = act; a.out_mode = out_mode; /* head dims: per group, below */
    MMS_FREE(mms::launch_linear_split(a, groups, (hipStream_t)s));
    return 0;
}

// ---- the same with two scaled fp16 planes per operand (split16_kernels.hip) -----------------------------------------------------
__attribute__((visibility("default"))) int mms_split_planes16_group(int device, int32_t groups, int64_t rows, int32_t K, int32_t x_pitch,
                                                                    const float* const* x, void* const* planes, float* const* scale, float* const* inv,
                                                                    int32_t nchains, int32_t L, const float* const* chain, float* const* chain_scale,
                                                                    float* const* chain_inv, float* const* stat, float eps, void* s) {
    MMS_DEV(device)
    if (groups < 1 || groups > mms::kMaxGroups) { g_create_error = "mms_split_planes16_group: groups must be 1.." + std::to_string(mms::kMaxGroups); return 1; }
    if (x_pitch == 0) x_pitch = K;
    if (!x || !planes || !scale || !inv || rows < 0 || K <= 0 || x_pitch < K || nchains < 0 || L < 0 || (nchains > 0 && (L < 1 || !chain || !chain_scale || !chain_inv))) {
        g_create_error = "mms_split_planes16_group: bad arguments (x_pitch >= K; nchains > 0 needs L >= 1, chain, chain_scale, chain_inv)";
        return 1;
    }
    mms::Split16PlanesArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !planes[g] || (reinterpret_cast<uintptr_t>(planes[g]) & 15) != 0 || (reinterpret_cast<uintptr_t>(x[g]) & 3) != 0 ||
            (nchains > 0 && (!chain[g] || !chain_scale[g] || !chain_inv[g]))) {
            g_create_error = "mms_split_planes16_group: null or misaligned pointer in a group (planes 16-byte aligned)";
            return 1;
        }
        a.x[g] = x[g]; a.planes[g] = planes[g]; a.scale[g] = scale[g]; a.inv[g] = inv[g];
        if (nchains > 0) { a.chain[g] = chain[g]; a.chain_scale[g] = chain_scale[g]; a.chain_inv[g] = chain_inv[g]; }
        if (stat) {
            if (!stat[g] || (reinterpret_cast<uintptr_t>(stat[g]) & 7) != 0) { g_create_error = "mms_split_planes16_group: null or misaligned stat pointer in a group"; return 1; }
            a.stat[g] = stat[g];
        }
    }
    a.rows = rows; a.K = K; a.x_pitch = x_pitch; a.nchains = nchains; a.L = nchains > 0 ? L : 0; a.eps = eps;
    MMS_FREE(mms::launch_split16_planes_group(a, groups, (hipStream_t)s));
    return 0;
}

// The weights' side of the split16 layers, refreshed on the device after every parameter update: planes, row scales and row 1-norms of
// `groups` weight matrices of ANY shapes in one launch ...
__attribute__((visibility("default"))) int mms_weight_planes16_group(int device, int32_t groups, const int64_t* N, const int32_t* K, const float* const* w,
                                                                     void* const* planes, float* const* scale, float* const* inv,
                                                                     float* const* l1, void* s) {
    MMS_DEV(device)
    if (groups < 1 || groups > mms::kMaxGroups) { g_create_error = "mms_weight_planes16_group: groups must be 1.." + std::to_string(mms::kMaxGroups); return 1; }
    if (!N || !K || !w || !planes || !scale || !inv) { g_create_error = "mms_weight_planes16_group: bad arguments (null array)"; return 1; }
    mms::Split16PlanesArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (N[g] < 0 || K[g] <= 0) { g_create_error = "mms_weight_planes16_group: bad shape in a group (N >= 0, K > 0)"; return 1; }
        if (!w[g] || !planes[g] || !scale[g] || !inv[g] || (reinterpret_cast<uintptr_t>(planes[g]) & 15) != 0 || (reinterpret_cast<uintptr_t>(w[g]) & 3) != 0 ||
            (reinterpret_cast<uintptr_t>(inv[g]) & 15) != 0) {
            g_create_error = "mms_weight_planes16_group: null or misaligned pointer in a group (planes and inv 16-byte aligned)";
            return 1;
        }
        a.x[g] = w[g]; a.planes[g] = planes[g]; a.scale[g] = scale[g]; a.inv[g] = inv[g];
        a.l1[g] = l1 ? l1[g] : nullptr;
        a.rows_g[g] = N[g]; a.K_g[g] = K[g];
    }
    a.per_group = 1;
    MMS_FREE(mms::launch_split16_planes_group(a, groups, (hipStream_t)s));
    return 0;
}

// ... and the bound chain (+ the scales of a constant-bound input) from them, one launch.
__attribute__((visibility("default"))) int mms_chain_refresh16(int device, int32_t nchains, int32_t L, const float* const* l1, const float* const* bias,
                                                               const int32_t* n, float* chain, float bound0, int64_t rows, float* chain_scale,
                                                               float* chain_inv, void* s) {
    MMS_DEV(device)
    if (nchains < 1 || L < 1 || (int64_t)nchains * L > mms::kMaxGroups || !l1 || !n || !chain || rows < 0 || (rows > 0 && (!chain_scale || !chain_inv || !(bound0 >= 0.f)))) {
        g_create_error = "mms_chain_refresh16: bad arguments (nchains, L >= 1, nchains * L <= " + std::to_string(mms::kMaxGroups) + "; rows > 0 needs chain_scale, chain_inv, bound0 >= 0)";
        return 1;
    }
    mms::ChainRefreshArgs a = {};
    for (int e = 0; e < nchains * L; e++) {
        if (!l1[e] || n[e] < 0) { g_create_error = "mms_chain_refresh16: null pointer or negative count in an entry"; return 1; }
        a.l1[e] = l1[e]; a.bias[e] = bias ? bias[e] : nullptr; a.n[e] = n[e];
    }
    a.chain = chain; a.nchains = nchains; a.L = L; a.bound0 = bound0; a.rows = rows; a.chain_scale = chain_scale; a.chain_inv = chain_inv;
    MMS_FREE(mms::launch_chain_refresh16(a, (hipStream_t)s));
    return 0;
}

// ---- the folded-LayerNorm layers' weight side, refreshed on the device (fold16_kernels.hip) ------------------------------------
__attribute__((visibility("default"))) int mms_fold_planes16_group(int device, int32_t groups, const int64_t* N, const int32_t* K, const float* const* w,
                                                                   const float* const* gamma, const float* const* beta, const float* const* bias,
                                                                   void* const* planes, float* const* inv, float* const* s_out, float* const* c_out,
                                                                   float* const* rb, float* const* wt, void* s) {
    MMS_DEV(device)
    if (groups < 1 || groups > mms::kMaxGroups) { g_create_error = "mms_fold_planes16_group: groups must be 1.." + std::to_string(mms::kMaxGroups); return 1; }
    if (!N || !K || !w) { g_create_error = "mms_fold_planes16_group: bad arguments (null array)"; return 1; }
    mms::FoldPlanesArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (N[g] < 0 || K[g] <= 0) { g_create_error = "mms_fold_planes16_group: bad shape in a group (N >= 0, K > 0)"; return 1; }
        if (!w[g] || (planes && planes[g] && (!inv || !inv[g])) || (planes && (reinterpret_cast<uintptr_t>(planes[g]) & 15) != 0)) {
            g_create_error = "mms_fold_planes16_group: null or misaligned pointer in a group (planes 16-byte aligned and with inv)";
            return 1;
        }
        a.w[g] = w[g];
        a.gamma[g] = gamma ? gamma[g] : nullptr; a.beta[g] = beta ? beta[g] : nullptr; a.bias[g] = bias ? bias[g] : nullptr;
        a.planes[g] = planes ? planes[g] : nullptr; a.inv[g] = inv ? inv[g] : nullptr;
        a.s[g] = s_out ? s_out[g] : nullptr; a.c[g] = c_out ? c_out[g] : nullptr; a.rb[g] = rb ? rb[g] : nullptr; a.wt[g] = wt ? wt[g] : nullptr;
        a.rows_g[g] = N[g]; a.K_g[g] = K[g];
    }
    MMS_FREE(mms::launch_fold_planes16(a, groups, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_fold_scales16_group(int device, int32_t groups, const float* const* rb, const int32_t* n, int64_t M,
                                                                   float* const* scale1, float* const* ysc, float* const* yinv, void* s) {
    MMS_DEV(device)
    if (groups < 1 || groups > mms::kMaxGroups) { g_create_error = "mms_fold_scales16_group: groups must be 1.." + std::to_string(mms::kMaxGroups); return 1; }
    if (!rb || !n || M < 0) { g_create_error = "mms_fold_scales16_group: bad arguments"; return 1; }
    mms::FoldScalesArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!rb[g] || n[g] < 0) { g_create_error = "mms_fold_scales16_group: null pointer or negative count in a group"; return 1; }
        a.rb[g] = rb[g]; a.n[g] = n[g];
        a.scale1[g] = scale1 ? scale1[g] : nullptr; a.ysc[g] = ysc ? ysc[g] : nullptr; a.yinv[g] = yinv ? yinv[g] : nullptr;
    }
    a.M = M;
    MMS_FREE(mms::launch_fold_scales16(a, groups, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_linear_group_act_split16(int device, int32_t groups, int64_t M, int32_t N, int32_t K, const void* const* x,
                                                                        const void* const* w, const float* const* b, void* const* y,
                                                                        const float* const* x_inv, const float* const* w_inv, const float* const* y_scale,
                                                                        int32_t act, int32_t out_mode, const float* const* ln_s,
                                                                        const float* const* ln_stat_in, float* const* ln_part_out,
                                                                        const float* const* head_w, float* const* head_part, const int32_t* head_dims, void* s) {
    MMS_DEV(device)
    if (groups < 1 || groups > mms::kMaxGroups) { g_create_error = "mms_linear_group_act_split16: groups must be 1.." + std::to_string(mms::kMaxGroups); return 1; }
    if (!x || !w || !b || !x_inv || !w_inv || M < 0 || M > 0x7fffffff || (M % 128) != 0 || N <= 0 || (N % 128) != 0 || K <= 0 || act < 0 || act > 3 ||
        out_mode < 0 || out_mode > 2 || (out_mode != 2 && !y) || (out_mode == 1 && !y_scale)) {
        g_create_error = "mms_linear_group_act_split16: bad arguments (M and N multiples of 128, act 0..3, out_mode 0..2, x_inv, w_inv, y_scale with out_mode 1)";
        return 1;
    }
    const bool ln = ln_s || ln_stat_in || ln_part_out;
    if (ln && (!ln_s || !ln_stat_in || !ln_part_out || act != 1 || out_mode == 0)) {
        g_create_error = "mms_linear_group_act_split16: the LayerNorm folds come together (ln_s, ln_stat_in, ln_part_out), with act = ELU and out_mode 1 or 2";
        return 1;
    }
    if (out_mode == 2 && (!ln || !head_w || !head_part || !head_dims)) {
        g_create_error = "mms_linear_group_act_split16: out_mode 2 needs the LayerNorm folds, head_w, head_part and head_dims";
        return 1;
    }
    mms::Split16LinearArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !w[g] || !b[g] || !x_inv[g] || !w_inv[g] || (out_mode != 2 && !y[g]) || (out_mode == 1 && !y_scale[g]) ||
            (ln && (!ln_s[g] || !ln_stat_in[g] || !ln_part_out[g])) || (out_mode == 2 && (!head_w[g] || !head_part[g]))) {
            g_create_error = "mms_linear_group_act_split16: null pointer in a group";
            return 1;
        }
        uintptr_t bits = reinterpret_cast<uintptr_t>(x[g]) | reinterpret_cast<uintptr_t>(w[g]) | reinterpret_cast<uintptr_t>(b[g]) | reinterpret_cast<uintptr_t>(w_inv[g]);
        if (out_mode != 2) bits |= reinterpret_cast<uintptr_t>(y[g]);
        if (ln) bits |= reinterpret_cast<uintptr_t>(ln_s[g]) | (reinterpret_cast<uintptr_t>(ln_stat_in[g]) << 1) | (reinterpret_cast<uintptr_t>(ln_part_out[g]) << 1);
        if ((bits & 15) != 0) { g_create_error = "mms_linear_group_act_split16: operands must be 16-byte aligned"; return 1; }
        a.x[g] = x[g]; a.w[g] = w[g]; a.b[g] = b[g]; a.y[g] = out_mode != 2 ? y[g] : nullptr;
        a.xinv[g] = x_inv[g]; a.winv[g] = w_inv[g]; a.yscale[g] = out_mode == 1 ? y_scale[g] : nullptr;
        if (ln) { a.s[g] = ln_s[g]; a.stat_in[g] = ln_stat_in[g]; a.part_out[g] = ln_part_out[g]; }
        if (out_mode == 2) {
            if (head_dims[g] < 1 || head_dims[g] > 16) { g_create_error = "split layer, out_mode 2: 1 <= head_dims[g] <= 16"; return 1; }
            a.head_w[g] = head_w[g]; a.head_part[g] = head_part[g]; a.hdims[g] = head_dims[g];
        }
    }
    a.M = (int)M; a.N = N; a.KC = (K + 31) / 32; a.act = act; a.out_mode = out_mode; /* head dims: per group, below */
    MMS_FREE(mms::launch_linear_split16(a, groups, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_layer_clock_probe(int device, uint64_t* out, int32_t slots) {
    (void)device;
    if (out && slots < 1) { g_create_error = "mms_layer_clock_probe: slots must be >= 1 with an output buffer"; return 1; }
    if (out && (reinterpret_cast<uintptr_t>(out) & 7) != 0) { g_create_error = "mms_layer_clock_probe: the buffer must be 8-byte aligned"; return 1; }
    mms::set_split16_clock_probe(out, out ? slots : 0);
    return 0;
}

__attribute__((visibility("default"))) int mms_row_stats_chan_group(int device, int32_t groups, int64_t M, int32_t slots, const float* const* part,
                                                                    float* const* stat, float eps, void* s) {
    MMS_DEV(device)
    if (groups < 1 || groups > mms::kMaxGroups) { g_create_error = "mms_row_stats_chan_group: groups must be 1.." + std::to_string(mms::kMaxGroups); return 1; }
    if (!part || !stat || M < 0 || slots < 1) { g_create_error = "mms_row_stats_chan_group: bad arguments"; return 1; }
    mms::RowStatsArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!part[g] || !stat[g]) { g_create_error = "mms_row_stats_chan_group: null pointer in a group"; return 1; }
        a.part[g] = part[g]; a.stat[g] = stat[g];
    }
    a.M = M; a.slots = slots; a.width = 64 * slots; a.eps = eps;
    MMS_FREE(mms::launch_row_stats_chan(a, groups, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_marl_heads_finish(int device, int32_t groups, int64_t M, int32_t slots, const float* const* part,
                                                                 const float* const* head_part, const float* const* hs, const float* const* hc,
                                                                 const int32_t* A, const float* const* std, float* const* out, float* const* logp,
                                                                 const int32_t* out_pitch, int64_t* const* counters, uint64_t seed, int64_t row_offset,
                                                                 float eps, void* s) {
    MMS_DEV(device)
    if (groups < 1 || groups > mms::kMaxGroups) { g_create_error = "mms_marl_heads_finish: groups must be 1.." + std::to_string(mms::kMaxGroups); return 1; }
    if (!part || !head_part || !hs || !hc || !A || !out || M < 0 || slots < 1) { g_create_error = "mms_marl_heads_finish: bad arguments"; return 1; }
    mms::HeadsFinishArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!part[g] || !head_part[g] || !hs[g] || !hc[g] || !out[g] || A[g] < 1 || A[g] > 16) {
            g_create_error = "mms_marl_heads_finish: null pointer or output width outside 1..16 in a group";
            return 1;
        }
        if ((reinterpret_cast<uintptr_t>(head_part[g]) & 15) != 0 || (reinterpret_cast<uintptr_t>(part[g]) & 7) != 0) {
            g_create_error = "mms_marl_heads_finish: head_part must be 16-byte aligned and part 8-byte aligned (read as float4 / float2)";
            return 1;
        }
        a.part[g] = part[g]; a.head_part[g] = head_part[g]; a.hs[g] = hs[g]; a.hc[g] = hc[g]; a.out[g] = out[g];
        a.std[g] = std ? std[g] : nullptr;
        a.logp[g] = logp ? logp[g] : nullptr;
        a.counters[g] = counters ? counters[g] : nullptr;
        a.A[g] = A[g];
        a.out_pitch[g] = out_pitch ? out_pitch[g] : A[g];
        if (a.out_pitch[g] < A[g]) { g_create_error = "mms_marl_heads_finish: out_pitch below the output width"; return 1; }
    }
    a.seed = seed; a.M = M; a.row_offset = row_offset; a.slots = slots; a.width = 64 * slots; a.eps = eps;
    MMS_FREE(mms::launch_marl_heads_finish(a, groups, (hipStream_t)s));
    return 0;
}

// ---- grouped policy inference (MAPPO / HAPPO: all agents' networks per launch) -----------------------------------------------
static bool bad_group_count(int32_t groups, const char* what) {
    if (groups >= 1 && groups <= mms::kMaxGroups) return false;
    g_create_error = std::string(what) + ": groups must be 1.." + std::to_string(mms::kMaxGroups);
    return true;
}

__attribute__((visibility("default"))) int mms_linear_group_act(int device, int32_t groups, int64_t M, int32_t N, int32_t K, const float* const* x,
                                                                const float* const* w, const float* const* b, float* const* y, int32_t act,
                                                                const float* const* ln_s, const float* const* ln_stat_in, float* const* ln_part_out,
                                                                void* s) {
    MMS_DEV(device)
    if (bad_group_count(groups, "mms_linear_group_act")) return 1;
    if (!x || !w || !b || !y || M < 0 || M > 0x7fffffff || N <= 0 || K <= 0 || (K % 4) != 0 || act < 0 || act > 3) {
        g_create_error = "mms_linear_group_act: bad arguments (K must be a positive multiple of 4, act 0..3)";
        return 1;
    }
    if ((ln_stat_in != nullptr) != (ln_s != nullptr)) { g_create_error = "mms_linear_group_act: ln_stat_in and ln_s come together"; return 1; }
    if ((ln_stat_in || ln_part_out) && (act != 1 || M % 128 != 0 || N % 128 != 0 || K < 8)) {
        g_create_error = "mms_linear_group_act: the LayerNorm folds need act = ELU, M and N multiples of 128";
        return 1;
    }
    mms::LinearArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !w[g] || !b[g] || !y[g] || (ln_s && (!ln_s[g] || !ln_stat_in[g])) || (ln_part_out && !ln_part_out[g])) {
            g_create_error = "mms_linear_group_act: null pointer in a group";
            return 1;
        }
        a.x[g] = x[g]; a.w[g] = w[g]; a.b[g] = b[g]; a.y[g] = y[g];
        if (ln_s) { a.s[g] = ln_s[g]; a.stat_in[g] = ln_stat_in[g]; }
        if (ln_part_out) a.part_out[g] = ln_part_out[g];
    }
    a.M = (int)M; a.N = N; a.K = K; a.act = act;
    MMS_FREE(mms::launch_linear_act(a, groups, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_row_stats_group(int device, int32_t groups, int64_t M, int32_t slots, int32_t width, const float* const* part,
                                                               float* const* stat, float eps, void* s) {
    MMS_DEV(device)
    if (bad_group_count(groups, "mms_row_stats_group")) return 1;
    if (!part || !stat || M < 0 || slots < 1 || width < 1) { g_create_error = "mms_row_stats_group: bad arguments"; return 1; }
    mms::RowStatsArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!part[g] || !stat[g]) { g_create_error = "mms_row_stats_group: null pointer in a group"; return 1; }
        a.part[g] = part[g]; a.stat[g] = stat[g];
    }
    a.M = M; a.slots = slots; a.width = width; a.eps = eps;
    MMS_FREE(mms::launch_row_stats(a, groups, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_row_moments_group(int device, int32_t groups, int64_t M, int32_t K, int32_t x_pitch, const float* const* x,
                                                                 float* const* stat, float eps, void* s) {
    MMS_DEV(device)
    if (bad_group_count(groups, "mms_row_moments_group")) return 1;
    if (x_pitch == 0) x_pitch = K;
    if (!x || !stat || M < 0 || K <= 0 || K > 4096 || x_pitch < K) { g_create_error = "mms_row_moments_group: bad arguments (1 <= K <= 4096)"; return 1; }
    mms::LayerNormArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !stat[g]) { g_create_error = "mms_row_moments_group: null pointer in a group"; return 1; }
        a.x[g] = x[g]; a.y[g] = stat[g];
    }
    a.M = M; a.K = K; a.Kp = K; a.x_pitch = x_pitch; a.eps = eps; a.stats_only = 1;
    MMS_FREE(mms::launch_layernorm(a, groups, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_layernorm_group(int device, int32_t groups, int64_t M, int32_t K, int32_t Kp, int32_t x_pitch,
                                                               const float* const* x, const float* const* gamma, const float* const* beta, float* const* y,
                                                               float eps, void* s) {
    MMS_DEV(device)
    if (bad_group_count(groups, "mms_layernorm_group")) return 1;
    if (x_pitch == 0) x_pitch = K;
    if (!x || !gamma || !beta || !y || M < 0 || K <= 0 || K > 4096 || Kp < K || x_pitch < K) {
        g_create_error = "mms_layernorm_group: bad arguments (1 <= K <= 4096, Kp >= K, x_pitch >= K or 0)";
        return 1;
    }
    mms::LayerNormArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!x[g] || !gamma[g] || !beta[g] || !y[g]) { g_create_error = "mms_layernorm_group: null pointer in a group"; return 1; }
        if ((Kp != K || x_pitch != K) && x[g] == y[g]) { g_create_error = "mms_layernorm_group: in place needs Kp == x_pitch == K"; return 1; }
        a.x[g] = x[g]; a.gamma[g] = gamma[g]; a.beta[g] = beta[g]; a.y[g] = y[g];
    }
    a.M = M; a.K = K; a.Kp = Kp; a.x_pitch = x_pitch; a.eps = eps;
    MMS_FREE(mms::launch_layernorm(a, groups, (hipStream_t)s));
    return 0;
}

__attribute__((visibility("default"))) int mms_marl_heads_act(int device, int32_t groups, int64_t M, int32_t H, const float* const* h,
                                                              const float* const* gamma, const float* const* beta, const float* const* w,
                                                              const float* const* b, const int32_t* A, const float* const* std, float* const* out,
                                                              float* const* logp, const int32_t* out_pitch, int64_t* const* counters, uint64_t seed,
                                                              int64_t row_offset, float eps, void* s) {
    MMS_DEV(device)
    if (bad_group_count(groups, "mms_marl_heads_act")) return 1;
    if (!h || !gamma || !beta || !w || !b || !A || !out || M < 0 || H <= 0 || H > 1024) {
        g_create_error = "mms_marl_heads_act: bad arguments (1 <= H <= 1024)";
        return 1;
    }
    mms::HeadsArgs a = {};
    for (int g = 0; g < groups; g++) {
        if (!h[g] || !gamma[g] || !beta[g] || !w[g] || !b[g] || !out[g] || A[g] < 1 || A[g] > 16) {
            g_create_error = "mms_marl_heads_act: null pointer in a group, or outputs outside 1..16";
            return 1;
        }
        a.h[g] = h[g]; a.gamma[g] = gamma[g]; a.beta[g] = beta[g]; a.w[g] = w[g]; a.b[g] = b[g]; a.A[g] = A[g]; a.out[g] = out[g];
        a.out_pitch[g] = out_pitch ? out_pitch[g] : A[g];
        if (a.out_pitch[g] < A[g]) { g_create_error = "mms_marl_heads_act: out_pitch below the number of outputs"; return 1; }
        a.std[g] = std ? std[g] : nullptr;
        a.logp[g] = logp ? logp[g] : nullptr;
        a.counters[g] = counters ? counters[g] : nullptr;
    }
    a.seed = seed; a.M = M; a.row_offset = row_offset; a.H = H; a.eps = eps;
    MMS_FREE(mms::launch_marl_heads(a, groups, (hipStream_t)s));
    return 0;
}

}  // extern "C"
