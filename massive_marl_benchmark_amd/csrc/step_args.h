// step_args.h -- kernel argument block of the step kernels (passed by value as kernarg).
#pragma once
#include <stdint.h>

#include "../../include/mms.h"

namespace mms {

struct StepArgs {
    const mms_config* cfg;            // device copy of the configuration (uniform loads)
    const float* actions;             // [N, num_actions]
    float* obs;                       // [N, obs_dim] raw
    float* obs_clipped;               // [N, obs_dim] clamped to +-clip_obs
    float* obs_out;                   // optional bound destination for the clamped row (may be null)
    void* obs_planes;                 // optional: the clamped row as H32 planes [N, KC, 2, 32] f16 (mms_bind_obs_planes16), or null
    float obs_planes_scale;           // ... times this power of two
    float* rew;                       // [N]
    float* rew_out;                   // optional bound destination for the reward (may be null)
    uint8_t* done_out;                // optional bound destination for the done flag (may be null)
    int64_t* reset;                   // [N]
    int64_t* progress;                // [N]
    float* root_states;               // [N * actors, 13]  env-local frame
    const float* initial_root_states; // [N * actors, 13]
    float* dof_state;                 // [N * dofs, 2]
    const float* env_origin;          // [N, 3]
    float* prev;                      // [N, prev_dim]
    const float* reset_noise;         // [N, 16]
    float* foot_sensors;              // [N * A, 24]
    int64_t* reset_count;             // [N] resets so far (RNG counter)
    const float* dr;                  // [N * A, MMS_DR_FLOATS] physical domain randomisation, or null (nominal model)
    int32_t do_physics;
    int32_t num_envs, num_agents, obs_dim, prev_dim;
    int32_t packing;                  // 1: several envs per workgroup where the lane counts allow (default), 0: one env per workgroup
    int32_t head_on;                  // 1: the policy's output heads + sampling run in the prologue (mms_bind_policy_head), `head` is valid
    mms_policy_head head;
};

}  // namespace mms
