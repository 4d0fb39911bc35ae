// rollout_lane.h -- the per-column / per-row math of the rollout kernels (GAE scans, advantage normalisation, MARL views, the
// Gaussian sampling of one action), written once for the HIP kernels (rollout_kernels.hip) and the CPU build of the engine
// (cpu/mms_cpu.cpp).  Reference call sites:
//   gae_ppo_column      RolloutStorage.compute_returns            agents/algorithms/rl/ppo/storage.py:51-65
//   gae_marl_column     SeparatedReplayBuffer.compute_returns     agents/algorithms/marl/utils/separated_buffer.py:153-164
//   marl_view_source    MultiVecTaskPython.step slicing           agents/tasks/agent_base/multi_vec_task.py:105-142
//   ppo_sample_one      ActorCritic.act                           agents/algorithms/rl/ppo/module.py:73-87
#pragma once
#include "mms_lane.h"

namespace mms {

// One env column of the PPO GAE scan: writes returns and UN-normalised advantages (= returns - values, storage.py:64), adds the
// column's sum and sum of squares of the advantages to (sum, sq).  Planes are [T, N].
MMS_HD void gae_ppo_column(const float* rewards, const uint8_t* dones, const float* values, const float* last_values, float* returns,
                           float* advantages, int T, int64_t N, int64_t i, float gamma, float lam, double& sum, double& sq) {
    float adv = 0.f;
    float next_v = last_values[i];
    for (int t = T - 1; t >= 0; t--) {
        float v = values[t * N + i];
        float nt = 1.0f - (float)dones[t * N + i];
        float delta = rewards[t * N + i] + nt * gamma * next_v - v;
        adv = delta + nt * gamma * lam * adv;
        float ret = adv + v;
        returns[t * N + i] = ret;
        float a = ret - v;
        advantages[t * N + i] = a;
        sum += (double)a;
        sq += (double)a * (double)a;
        next_v = v;
    }
}
// (mean, 1 / (std + 1e-8)) of the advantage normalisation from stats = {sum, sum of squares, count}; std unbiased (torch.std)
MMS_HD void adv_norm_params(const double* stats, float& mean, float& inv) {
    double n = stats[2];
    double m = stats[0] / n;
    double var = (stats[1] - n * m * m) / (n - 1.0);
    inv = (float)(1.0 / (sqrt(var > 0.0 ? var : 0.0) + 1e-8));
    mean = (float)m;
}
// One column of the MARL GAE scan.  value_preds / returns: element (t, c) at t * cols + c; rewards / masks: (t, i) at t * N + i
// (stored once per env; cols = N for one agent buffer, N * A with the agents fastest for the shared buffers).
MMS_HD void gae_marl_column(const float* rewards, const float* value_preds, const float* masks, float* returns, int T, int64_t N, int64_t cols,
                            int64_t c, int64_t i, float gamma, float lam, int use_norm, float mean, float var) {
    float sd = 1.f;
    if (use_norm) sd = sqrtf(var); else mean = 0.f;
    float gae = 0.f;
    float v1 = value_preds[(int64_t)T * cols + c];
    if (use_norm) v1 = v1 * sd + mean;
    for (int t = T - 1; t >= 0; t--) {
        float v0 = value_preds[(int64_t)t * cols + c];
        if (use_norm) v0 = v0 * sd + mean;
        float m = masks[(int64_t)(t + 1) * N + i];
        float delta = rewards[(int64_t)t * N + i] + gamma * v1 * m - v0;
        gae = delta + gamma * lam * m * gae;
        returns[(int64_t)t * cols + c] = gae + v0;
        v1 = v0;
    }
}
// obs_all[n][k][0:per] = obs[n][k*per : (k+1)*per], obs_all[n][k][per:] = obs[n][agents*per:]: source index of flat output element i
MMS_HD int64_t marl_view_source(int64_t i, int agents, int per, int shared) {
    const int w = per + shared, row = agents * per + shared;
    int64_t e = i / ((int64_t)agents * w);
    int r = (int)(i - e * agents * w);
    int k = r / w, j = r - k * w;
    int src = (j < per) ? k * per + j : agents * per + (j - per);
    return e * row + src;
}
// One action of one row: noise ~ N(0,1) from the counter-based stream, action = mean + scale * noise, the action's term of the
// log-probability.  ref_scale: module.py:76-77 hands diag(sigma^2) to scale_tril.
MMS_HD float ppo_sample_one(float mean, float log_std, uint64_t seed, uint64_t row_global, uint64_t counter, uint32_t j, int ref_scale, float& logp_term) {
    float scale, lscale;
    if (ref_scale) { float sd = expf(log_std); scale = sd * sd; lscale = logf(scale); }
    else { scale = expf(log_std); lscale = log_std; }
    const float z = rand_normal(seed, row_global, counter, j);
    logp_term = -0.5f * z * z - lscale - 0.9189385332046727f;
    return mean + scale * z;
}

}  // namespace mms
