// rollout_kernels.hip -- GAE scans and MARL wrapper views for gfx950.
//
//   gae_ppo_kernel       RolloutStorage.compute_returns   (agents/algorithms/rl/ppo/storage.py:51-65)
//   adv_normalize_kernel the normalisation at storage.py:64-65, split off so that data-parallel ranks can
//                        all-reduce {sum, sum of squares, count} between the two launches
//   gae_marl_kernel      SeparatedReplayBuffer.compute_returns, use_gae branch without proper time limits
//                        (agents/algorithms/marl/utils/separated_buffer.py:153-164)
//   marl_views_kernel    MultiVecTaskPython.step slicing  (agents/tasks/agent_base/multi_vec_task.py:105-142)
//   ppo_act_kernel       the sampling tail of ActorCritic.act (agents/algorithms/rl/ppo/module.py:73-87) fused with the
//                        stores of RolloutStorage.add_transitions (storage.py:33-47)
//   ppo_head_act_kernel  the same with the actor's last Linear layer (module.py:29-30) on the matrix cores in front of it
//
// All are HBM-bound streaming kernels: thread = env column, T serial steps, every load of a [T,N] plane is a
// coalesced 256 B wave transaction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include "head_block.h"
#include "mms_lane.h"
#include "rollout_lane.h"

#ifndef MMS_HEAD_RT_DEFAULT
#define MMS_HEAD_RT_DEFAULT 1
#endif

namespace mms {

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

__global__ void __launch_bounds__(256) gae_ppo_kernel(const float* __restrict__ rewards, const uint8_t* __restrict__ dones,
                                                      const float* __restrict__ values, const float* __restrict__ last_values,
                                                      float* __restrict__ returns, float* __restrict__ advantages,
                                                      double* __restrict__ stats, int T, int64_t N, float gamma, float lam) {
    __shared__ double s_sum[4], s_sq[4];
    double lsum = 0.0, lsq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        gae_ppo_column(rewards, dones, values, last_values, returns, advantages, T, N, i, gamma, lam, lsum, lsq);
    lsum = wave_sum(lsum);
    lsq = wave_sum(lsq);
    int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[wave] = lsum; s_sq[wave] = lsq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) { a += s_sum[w]; b += s_sq[w]; }
        atomicAdd(&stats[0], a);
        atomicAdd(&stats[1], b);
        if (blockIdx.x == 0) stats[2] = (double)T * (double)N;
    }
}

// The one-rank form of the two steps (nothing to all-reduce in between), bit-reproducible and with nothing to zero beforehand: the scan
// leaves per-block partial sums in a scratch area (no atomics; the atomics of gae_ppo_kernel need a 24-byte memset that the runtime
// turns into two fill kernels, ~9.5 us inside a captured rollout), and EVERY block of the normalisation sums the partials in the same
// fixed order before it normalises its share.  stats = f64 [3 + 2 * 2048] (grid_for caps the blocks at 2048): {sum, sum of squares, count}, then the partials.
__global__ void __launch_bounds__(256) gae_ppo_partials_kernel(const float* __restrict__ rewards, const uint8_t* __restrict__ dones,
                                                               const float* __restrict__ values, const float* __restrict__ last_values,
                                                               float* __restrict__ returns, float* __restrict__ advantages,
                                                               double* __restrict__ stats, int T, int64_t N, float gamma, float lam) {
    __shared__ double s_sum[4], s_sq[4];
    double lsum = 0.0, lsq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        gae_ppo_column(rewards, dones, values, last_values, returns, advantages, T, N, i, gamma, lam, lsum, lsq);
    lsum = wave_sum(lsum);
    lsq = wave_sum(lsq);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[wave] = lsum; s_sq[wave] = lsq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[3 + 2 * blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        stats[4 + 2 * blockIdx.x] = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]);
    }
}
__global__ void __launch_bounds__(256) adv_normalize_partials_kernel(float* __restrict__ advantages, double* __restrict__ stats, int nparts,
                                                                     double count_d, int64_t count) {
    __shared__ double s_sum[4], s_sq[4], s_tot[3];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) { a += stats[3 + 2 * i]; b += stats[4 + 2 * i]; }
    a = wave_sum(a);
    b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = a; s_sq[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_tot[0] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        s_tot[1] = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]);
        s_tot[2] = count_d;
        if (blockIdx.x == 0) { stats[0] = s_tot[0]; stats[1] = s_tot[1]; stats[2] = s_tot[2]; }
    }
    __syncthreads();
    float fm, inv;
    adv_norm_params(s_tot, fm, inv);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        advantages[i] = (advantages[i] - fm) * inv;
}

// advantages := (advantages - mean) / (std + 1e-8), std unbiased (torch.std default)
__global__ void __launch_bounds__(256) adv_normalize_kernel(float* __restrict__ advantages, const double* __restrict__ stats, int64_t count) {
    float fm, inv;
    adv_norm_params(stats, fm, inv);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        advantages[i] = (advantages[i] - fm) * inv;
}

__global__ void __launch_bounds__(256) gae_marl_kernel(const float* __restrict__ rewards, const float* __restrict__ value_preds,
                                                       const float* __restrict__ masks, float* __restrict__ returns, int T, int64_t N,
                                                       float gamma, float lam, int use_norm, const float* __restrict__ norm_mean,
                                                       const float* __restrict__ norm_var) {
    const float mean = use_norm ? norm_mean[0] : 0.f, var = use_norm ? norm_var[0] : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        gae_marl_column(rewards, value_preds, masks, returns, T, N, N, i, i, gamma, lam, use_norm, mean, var);
}

// All agents of all envs in one launch: value_preds / returns are [T+1, N, A] (agent fastest), rewards and
// masks are shared per env ([T, N], [T+1, N]: the reference stores the same reward / mask once per agent buffer), the
// normaliser statistics are per agent.  Column (i, k) = env i, agent k.
__global__ void __launch_bounds__(256) gae_marl_agents_kernel(const float* __restrict__ rewards, const float* __restrict__ value_preds,
                                                              const float* __restrict__ masks, float* __restrict__ returns, int T, int64_t N,
                                                              int A, float gamma, float lam, int use_norm,
                                                              const float* __restrict__ norm_mean, const float* __restrict__ norm_var) {
    const int64_t cols = N * A;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = c / A;
        const int k = (int)(c - i * A);
        gae_marl_column(rewards, value_preds, masks, returns, T, N, cols, c, i, gamma, lam, use_norm, use_norm ? norm_mean[k] : 0.f, use_norm ? norm_var[k] : 1.f);
    }
}

// obs_all[n][k][0:per] = obs[n][k*per : (k+1)*per], obs_all[n][k][per:] = obs[n][agents*per:]; input already clamped
__global__ void __launch_bounds__(256) marl_views_kernel(const float* __restrict__ obs, float* __restrict__ obs_all, int64_t n,
                                                         int agents, int per, int shared) {
    const int64_t total = n * agents * (per + shared);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        obs_all[i] = obs[marl_view_source(i, agents, per, shared)];
}

__global__ void __launch_bounds__(256) ppo_act_kernel(const float* __restrict__ mean, const float* __restrict__ value,
                                                      const float* __restrict__ log_std, uint64_t seed, int64_t* __restrict__ counters,
                                                      int64_t row_offset, int ref_scale, PpoActOut o, int64_t N, int A) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    ppo_sample_row(mean + row * A, value, log_std, seed, counters, row_offset, ref_scale, o, row, A, threadIdx.x & 63);
}

// The same with the actor's last Linear layer folded in (head_block.h: the body is shared with the step kernel's fused prologue)
template <int NCT, int WAVES, int RT>
__global__ void __launch_bounds__(64 * WAVES) ppo_head_act_kernel(const float* __restrict__ hidden, const float* __restrict__ weight,
                                                           const float* __restrict__ bias, int H, const float* __restrict__ value,
                                                           const float* __restrict__ vhidden, const float* __restrict__ vweight,
                                                           const float* __restrict__ vbias, int VH,
                                                           const float* __restrict__ log_std, uint64_t seed, int64_t* __restrict__ counters,
                                                           int64_t row_offset, int ref_scale, PpoActOut o, int64_t N, int A) {
    extern __shared__ __attribute__((aligned(16))) float s_part[];      // [WAVES][16 RT rows][AP], then [16 RT][AP] means
    ppo_head_block<NCT, WAVES, RT>(s_part, (int)threadIdx.x, true, (int64_t)blockIdx.x * 16 * RT, hidden, weight, bias, H, value, vhidden, vweight, vbias, VH,
                                   log_std, seed, counters, row_offset, ref_scale, o, N, A);
}

hipError_t launch_adv_normalize(float* advantages, const double* stats, int64_t count, hipStream_t s);
static int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

hipError_t launch_gae_ppo(const float* rewards, const uint8_t* dones, const float* values, const float* last_values, float* returns,
                          float* advantages, double* stats, int T, int64_t N, float gamma, float lam, hipStream_t s) {
    hipError_t e = hipMemsetAsync(stats, 0, 3 * sizeof(double), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gae_ppo_kernel, dim3(grid_for(N)), dim3(256), 0, s, rewards, dones, values, last_values, returns, advantages, stats, T, N, gamma, lam);
    return hipGetLastError();
}
hipError_t launch_gae_ppo_normalized(const float* rewards, const uint8_t* dones, const float* values, const float* last_values, float* returns,
                                     float* advantages, double* stats, int T, int64_t N, float gamma, float lam, hipStream_t s) {
    const int blocks = grid_for(N);                                        // <= 2048 partial pairs
    hipLaunchKernelGGL(gae_ppo_partials_kernel, dim3(blocks), dim3(256), 0, s, rewards, dones, values, last_values, returns, advantages, stats, T, N, gamma, lam);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    const int64_t count = (int64_t)T * N;
    hipLaunchKernelGGL(adv_normalize_partials_kernel, dim3(grid_for(count)), dim3(256), 0, s, advantages, stats, blocks, (double)T * (double)N, count);
    return hipGetLastError();
}
hipError_t launch_adv_normalize(float* advantages, const double* stats, int64_t count, hipStream_t s) {
    hipLaunchKernelGGL(adv_normalize_kernel, dim3(grid_for(count)), dim3(256), 0, s, advantages, stats, count);
    return hipGetLastError();
}
hipError_t launch_gae_marl(const float* rewards, const float* value_preds, const float* masks, float* returns, int T, int64_t N, float gamma,
                           float lam, int use_norm, const float* mean, const float* var, hipStream_t s) {
    hipLaunchKernelGGL(gae_marl_kernel, dim3(grid_for(N)), dim3(256), 0, s, rewards, value_preds, masks, returns, T, N, gamma, lam, use_norm, mean, var);
    return hipGetLastError();
}
hipError_t launch_gae_marl_agents(const float* rewards, const float* value_preds, const float* masks, float* returns, int T, int64_t N, int A,
                                  float gamma, float lam, int use_norm, const float* mean, const float* var, hipStream_t s) {
    hipLaunchKernelGGL(gae_marl_agents_kernel, dim3(grid_for(N * A)), dim3(256), 0, s, rewards, value_preds, masks, returns, T, N, A, gamma, lam, use_norm, mean, var);
    return hipGetLastError();
}
hipError_t launch_ppo_act(const float* mean, const float* value, const float* log_std, uint64_t seed, int64_t* counters, int64_t row_offset,
                          int ref_scale, float* actions_out, float* act_slot, float* logp_slot, float* value_slot, float* mu_slot,
                          float* sigma_slot, int64_t N, int A, hipStream_t s) {
    if (N == 0) return hipSuccess;
    PpoActOut o{actions_out, act_slot, logp_slot, value_slot, mu_slot, sigma_slot};
    hipLaunchKernelGGL(ppo_act_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, mean, value, log_std, seed, counters, row_offset, ref_scale, o, N, A);
    return hipGetLastError();
}
hipError_t launch_ppo_head_act(const float* hidden, const float* weight, const float* bias, int H, const float* value, const float* vhidden,
                               const float* vweight, const float* vbias, int VH, const float* log_std,
                               uint64_t seed, int64_t* counters, int64_t row_offset, int ref_scale, float* actions_out, float* act_slot,
                               float* logp_slot, float* value_slot, float* mu_slot, float* sigma_slot, int64_t N, int A, hipStream_t s) {
    if (N == 0) return hipSuccess;
    PpoActOut o{actions_out, act_slot, logp_slot, value_slot, mu_slot, sigma_slot};
    const int nct = (A + 15) / 16;
    int waves = (H % 512 == 0) ? 8 : (H % 256 == 0) ? 4 : (H % 128 == 0) ? 2 : 1;            // H / waves is a multiple of 64
    // 32 rows per block (MMS_HEAD_RT=2) when eight waves split K: see the kernel's note; the default is what measured faster
    static const int rt_env = getenv("MMS_HEAD_RT") ? atoi(getenv("MMS_HEAD_RT")) : MMS_HEAD_RT_DEFAULT;
    const int rt = (waves == 8 && rt_env == 2 && N >= 32) ? 2 : 1;
    size_t lds = (size_t)(waves + 1) * 16 * rt * (nct * 16) * sizeof(float);
    if (rt == 1 && lds > 64 * 1024) {                                                        // (8 waves x 8 column tiles: 73.7 KB) stay within the default 64 KB
        waves = 4;
        lds = (size_t)(waves + 1) * 16 * (nct * 16) * sizeof(float);
    }
    const dim3 grid((unsigned)((N + 16 * rt - 1) / (16 * rt)));
#define MMS_HEAD_W(NCT, W)                                                                                                                        \
    hipLaunchKernelGGL((ppo_head_act_kernel<NCT, W, 1>), grid, dim3(64 * W), lds, s, hidden, weight, bias, H, value, vhidden, vweight, vbias, VH, \
                       log_std, seed, counters, row_offset, ref_scale, o, N, A)
#define MMS_HEAD2(NCT)                                                                                                                             \
    case NCT: {                                                                                                                                    \
        auto kern = ppo_head_act_kernel<NCT, 8, 2>;                                                                                                \
        if (lds > 64 * 1024) {                                                                                                                     \
            static bool done[64] = {};                                                                                                             \
            int dev = 0;                                                                                                                           \
            if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;                                                                      \
            if (dev < 0 || dev >= 64 || !done[dev]) {                                                                                              \
                if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e != hipSuccess) return e; \
                if (dev >= 0 && dev < 64) done[dev] = true;                                                                                        \
            }                                                                                                                                      \
        }                                                                                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, hidden, weight, bias, H, value, vhidden, vweight, vbias, VH, log_std, seed, counters,    \
                           row_offset, ref_scale, o, N, A);                                                                                        \
        break;                                                                                                                                     \
    }
#define MMS_HEAD(NCT)                                                                                                                              \
    case NCT:                                                                                                                                      \
        if (waves == 8) MMS_HEAD_W(NCT, 8);                                                                                                        \
        else if (waves == 4) MMS_HEAD_W(NCT, 4);                                                                                                   \
        else if (waves == 2) MMS_HEAD_W(NCT, 2);                                                                                                   \
        else MMS_HEAD_W(NCT, 1);                                                                                                                   \
        break;
    if (rt == 2) {
        switch (nct) {
            MMS_HEAD2(1) MMS_HEAD2(2) MMS_HEAD2(3) MMS_HEAD2(4) MMS_HEAD2(5) MMS_HEAD2(6) MMS_HEAD2(7) MMS_HEAD2(8)
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (nct) {
        MMS_HEAD(1) MMS_HEAD(2) MMS_HEAD(3) MMS_HEAD(4) MMS_HEAD(5) MMS_HEAD(6) MMS_HEAD(7) MMS_HEAD(8)
        default: return hipErrorInvalidValue;
    }
#undef MMS_HEAD
#undef MMS_HEAD2
#undef MMS_HEAD_W
    return hipGetLastError();
}
hipError_t launch_marl_views(const float* obs, float* obs_all, int64_t n, int agents, int per, int shared, hipStream_t s) {
    hipLaunchKernelGGL(marl_views_kernel, dim3(grid_for(n * agents * (per + shared))), dim3(256), 0, s, obs, obs_all, n, agents, per, shared);
    return hipGetLastError();
}

}  // namespace mms
