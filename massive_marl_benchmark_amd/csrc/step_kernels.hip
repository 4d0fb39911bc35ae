// step_kernels.hip -- fused VecTask step for gfx950 (MI355X): physics substeps + reset_idx +
// compute_observations + compute_reward + cache update in ONE launch per control step.
//
// Replaces BaseTask.step (agents/tasks/agent_base/base_task.py:129-149) together with the task hooks
// it calls (agents/tasks/ten_ant.py:635-926, one_ant.py:314-436, multi_ingenuity.py:228-379) and the
// `gym.simulate` call inside it.  The per-lane math lives in mms_lane.h (shared with the CPU lane
// emulation test); this file holds only what needs the GPU: the lane -> work mapping, DPP / LDS
// reductions, and the HBM traffic.
//
// Mapping.  A workgroup holds EPB environments.  For the ant tasks the ant lanes come first,
//   (env e, ant a, leg l) = e * 4A + 4a + l; the quad of an ant shares the torso state,
// followed (8-aligned) by the box-corner lanes, 8 per env.  TenAnt at A = 10 packs 4 envs into 192 threads: 160 ant lanes
// and 32 box lanes; 4096 envs = 1024 blocks = 3072 waves = 3 per SIMD, all resident at once.
// HBM traffic per env-step is the env's own contiguous blocks: root_states 572 B, dof_state 640 B,
// actions 320 B, caches 168 B in; the same state plus the 1552 B observation row(s) out.
#include <hip/hip_runtime.h>

#include "head_block.h"
#include "mms_lane.h"
#include "step_args.h"

namespace mms {

// ---- cross-lane primitives -------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// all-reduce over the 4 lanes of a quad: (x0 + x1) + (x2 + x3) in every lane (bit-identical across the quad)
__device__ __forceinline__ float quad_sum(float x) {
    x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
    return x;
}
// all-reduce over 8 consecutive, 8-aligned lanes
__device__ __forceinline__ float oct_sum(float x) {
    x = quad_sum(x);
    x += dpp_mov<0x141>(x);  // row_half_mirror: lane i <-> 7 - i within each 8
    return x;
}
__device__ __forceinline__ void quad_sum(Sym6& A, S6& p) {
#pragma unroll
    for (int k = 0; k < 21; k++) A.m[k] = quad_sum(A.m[k]);
    p.a.x = quad_sum(p.a.x); p.a.y = quad_sum(p.a.y); p.a.z = quad_sum(p.a.z);
    p.l.x = quad_sum(p.l.x); p.l.y = quad_sum(p.l.y); p.l.z = quad_sum(p.l.z);
}
__device__ __forceinline__ void oct_sum(BoxCorner& c) {
#pragma unroll
    for (int k = 0; k < 9; k++) c.t[k] = oct_sum(c.t[k]);
}
// all-reduce over the 64 lanes of a wave: DPP inside rows of 16, then two cross-row exchanges
__device__ __forceinline__ float wave_sum(float x) {
    x = oct_sum(x);
    x += dpp_mov<0x140>(x);                       // row_mirror: lane i <-> 15 - i within each 16
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store(float4* dst, float4 v) {
    __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_t*>(dst));
}

__device__ __forceinline__ RigidState load_rigid(const float* r) {
    RigidState B;
    B.pos = V3{r[0], r[1], r[2]};
    B.qx = r[3]; B.qy = r[4]; B.qz = r[5]; B.qw = r[6];
    B.vel = V3{r[7], r[8], r[9]};
    B.ang = V3{r[10], r[11], r[12]};
    return B;
}
__device__ __forceinline__ void store_rigid(float* r, const RigidState& B) {
    r[0] = B.pos.x; r[1] = B.pos.y; r[2] = B.pos.z;
    r[3] = B.qx; r[4] = B.qy; r[5] = B.qz; r[6] = B.qw;
    r[7] = B.vel.x; r[8] = B.vel.y; r[9] = B.vel.z;
    r[10] = B.ang.x; r[11] = B.ang.y; r[12] = B.ang.z;
}

// ---------------------------------------------------------------------------------------------
// ant tasks.  TASK: MMS_TASK_TEN_ANT or MMS_TASK_ONE_ANT.
// A block of BLOCK threads holds EPB environments.  Thread layout: the ant lanes of all EPB envs first (4A per env,
// lane = (ant, leg), a quad per ant), then -- 8-aligned -- the box-corner lanes (8 per env):
//   <64, 1>    one env per wave (any A <= 14)
//   <192, 4>   TenAnt, A = 10: 160 ant lanes (two full waves + half of the third) + 32 box lanes: all 64 lanes of the three
//              waves are live, and the box phase and the reward finish issue on one wave in three instead of on every wave.
//              (<384, 8> -- five full ant waves + one box wave -- measured 42.6 us against 30.5 us: the six-wave blocks do not
//              all become resident at three waves per SIMD; profiles/r01_v6_layout_ab.txt)
//   <64, 4>    OneAnt: 16 ant lanes + 32 box lanes in one wave
//   <512, 1>   up to 126 ants per env (the 100-agent swarm)
// Quads (one ant) and 8-lane box groups never straddle a wave; everything else that crosses lanes goes through LDS and
// __syncthreads.  dynamic LDS: LegConst[4], then per env the block described by ant_env_lds_floats().
// ---------------------------------------------------------------------------------------------
#ifndef MMS_WAVES_PER_EU
#define MMS_WAVES_PER_EU 2      // one-wave envs: 2 waves per SIMD; 3 and 4 need spills and measured slower (profiles/r01_v3_bench_wpe*.json)
#endif
__host__ __device__ inline size_t ant_env_lds_floats(int obs_dim, int A) {
    return (size_t)((obs_dim + 3) & ~3) + 16 + (sizeof(BoxPose) + 15) / 16 * 4 + 28 + 8 + (size_t)(((RP_STRIDE + 6) * A + 3) & ~3) +
           (size_t)((4 * A + 2 + 7 + 3) & ~3);     // + staging of the epilogue's inputs (prev_dim <= 4A+2, origin, progress, reset count)
}
// Residency target: TenAnt at 4096 envs is 1024 blocks x 3 waves = 3072 waves = exactly 3 per SIMD on 256 CUs, so the whole
// grid is resident at once when the kernel fits in 168 VGPRs (the unpacked layout needs 4096 waves = 4 per SIMD = 128 VGPRs).
#ifndef MMS_WAVES_PER_EU_PACKED
#define MMS_WAVES_PER_EU_PACKED 3
#endif
// AT > 0 fixes the number of ants at compile time (LDS offsets become immediates, the per-ant loops unroll); 0 = runtime.
// DR: per-env physical domain randomisation (mass / damping scales, limit offsets) read from a.dr; a separate instantiation so
// that the nominal kernel carries none of it.
template <int TASK, int BLOCK, int EPB, int AT, bool DR, bool HEAD = false>
// The 512-thread layout (the runtime-sized one: the 100-ant swarm) is held to 168 VGPRs as well.  Its eight waves are two per SIMD
// and a second block never fits beside them, so the register count looks free -- it is not: at 182-194 VGPRs (what the compiler takes
// when allowed 256) the same instruction stream, counter for counter (SQ_INSTS_*, SQ_WAVE_CYCLES, I-cache misses), took 214-232 us
// against 162-166 us at <= 168 for 2048 x 100 ants (profiles/r02_swarm_occupancy.txt).
#ifndef MMS_WAVES_PER_EU_WIDE
#define MMS_WAVES_PER_EU_WIDE 3
#endif
__global__ void __launch_bounds__(BLOCK, BLOCK == 64 ? MMS_WAVES_PER_EU : ((BLOCK == 192 || BLOCK == 768) ? MMS_WAVES_PER_EU_PACKED : MMS_WAVES_PER_EU_WIDE)) ant_step_kernel(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // The config block (~0.6 KB, read all over the kernel) is copied into LDS by the prologue and read from there: the kernel
    // writes global memory, so the compiler may not use scalar loads for it, and a uniform VECTOR load costs an L2 round trip at
    // every point of use (53 of them per substep) where an LDS broadcast read costs tens of cycles.  Cg: the prologue's view.
    const mms_config* __restrict__ Cg = a.cfg;
    constexpr int kCfgFloats = (int)((sizeof(mms_config) + 15) / 16 * 4);
    static_assert(sizeof(mms_config) % 4 == 0, "copied word by word");
    const mms_config* C = reinterpret_cast<const mms_config*>(lds);
    const mms_model* M = &C->model;
    float2 ac_head = make_float2(0.f, 0.f);          // (HEAD instantiation: this lane's two sampled actions)
    constexpr int kObsAT = TASK == MMS_TASK_TEN_ANT ? 38 * AT + 8 : (TASK == MMS_TASK_MULTI_ANT_CIRCLE ? 38 * AT : 60);
    constexpr int kPrevAT = TASK == MMS_TASK_TEN_ANT ? 4 * AT + 2 : (TASK == MMS_TASK_MULTI_ANT_CIRCLE ? 2 * AT : 6);
    const int obs_dim = AT > 0 ? kObsAT : a.obs_dim;
    const int prev_dim = AT > 0 ? kPrevAT : a.prev_dim;
    const int obs_pad = (obs_dim + 3) & ~3;

    const int A = AT > 0 ? AT : a.num_agents;
    const int LA = 4 * A;                                     // ant lanes per env
    const int ant_region = EPB * LA;
    // The box-corner lanes follow the ant lanes, 8-aligned -- in the 512-thread layout on a WAVE OF THEIR OWN (lane 448 on) whenever
    // the ants leave the last wave free (up to 112 ants): the corner terms, their reduction and the factorisation of the box's matrix
    // then run beside the ant waves instead of lengthening the one wave that would hold both (one block per CU: nothing else hides it).
    const int box_region = (BLOCK == 512 && EPB == 1 && ant_region <= 448) ? 448 : ((ant_region + 7) & ~7);
    const int bt = (int)threadIdx.x - box_region;             // index among the box lanes
    const bool is_ant = (int)threadIdx.x < ant_region;
    const bool is_box = bt >= 0 && bt < 8 * EPB;
    const int e_loc = EPB == 1 ? 0 : (is_ant ? (int)threadIdx.x / LA : (is_box ? bt >> 3 : 0));
    const int tid = is_ant ? (int)threadIdx.x - e_loc * LA : 0;   // ant lane within its env
    const int corner = bt & 7;
    const bool box_lead = is_box && corner == 0;
    const int env_raw = blockIdx.x * EPB + e_loc;
    const bool live = env_raw < a.num_envs;                  // a partial last block still runs every barrier
    const int env = live ? env_raw : a.num_envs - 1;
    const int ant = tid >> 2, leg = tid & 3;

    // per-leg constants and the box pose live in LDS and are read at the point of use: keeping them in registers
    // costs ~55 VGPRs per lane for values that are wave-uniform or 4-periodic
    LegConst* s_leg = reinterpret_cast<LegConst*>(lds + kCfgFloats);               // [4], shared by the block
    // per-env LDS block: the fixed-size parts first, at compile-time offsets from one lane-varying base
    constexpr int kBoxOff = 0, kBpOff = 16, kFacOff = kBpOff + (int)((sizeof(BoxPose) + 15) / 16 * 4), kEpiOff = kFacOff + 28, kRedOff = kEpiOff + 8;
    const size_t env_stride = (ant_env_lds_floats(obs_dim, A) + 3) & ~(size_t)3;
    float* lds_envs = lds + kCfgFloats + (4 * sizeof(LegConst) + 15) / 16 * 4;
    // behind the env blocks: the block's slice of root_states, staged.  The EPB envs of a block are CONTIGUOUS in HBM (572 B per TenAnt
    // env): the slice comes in as coalesced 16-B loads (nine 1-KB wave transactions for 16 envs) and every lane takes its torso --
    // and the box lead its box -- from LDS, instead of 13 scalar dword loads per lane with four lanes of a quad fetching the same 52
    // bytes (each such wave load touched ~26 cache lines).  The way back is the same: final poses into LDS, coalesced stores out.
    float* s_root = lds_envs + (size_t)EPB * ((ant_env_lds_floats(AT > 0 ? kObsAT : a.obs_dim, A) + 3) & ~(size_t)3);
    // (compile-time layouts whose slice starts and ends on 16 B only; the runtime-sized layouts load and store per lane)
    constexpr bool kStage = AT > 0 && ((EPB * (AT + 1) * 13) % 4) == 0;
    const int root_floats = kStage ? EPB * (A + 1) * 13 : 0;
    // behind it: six 16-B words per lane where a leg lane parks its joint axes between the two passes of a substep
    float* lds_lanes = s_root + ((root_floats + 3) & ~3);
    // ---- the policy's output heads + sampling, fused (mms_bind_policy_head; HEAD instantiation of the 16-envs-per-block layout only) ----
    // The block's 16 envs are the 16 rows of one heads-kernel block: waves 0-7 run head_block.h's body (the same instruction sequence as
    // mms_ppo_heads_act: K = 512 split over eight waves, v_mfma_f32_16x16x4_f32, partial sums in wave order, sampling by row) in the
    // kinematics parking space (partials [8][16][80], means [16][80], the block's sampled actions [16][80]: free until the inward pass).
    // Waves 8-11 have no part in it but its barriers: they stage the block's inputs meanwhile -- the root slice, the config block, the
    // leg constants, the caches -- which the other instantiations do with all twelve waves further down; the HBM round trip of the
    // staging then lies under the head's.  Every ant lane takes its two actions into registers before the space is reused.
    if constexpr (HEAD) {
        static_assert(BLOCK == 768 && EPB == 16 && AT == 10 && TASK == MMS_TASK_TEN_ANT && kStage, "the fused head assumes 16 rows x 80 actions per block");
        float* s_head = lds_lanes;
        float* s_head_actions = s_head + 9 * 16 * 80;
        static_assert((9 + 1) * 16 * 80 <= 6 * 4 * BLOCK, "the head's LDS must fit the parking space");
        if (threadIdx.x >= 512) {
            const int t = (int)threadIdx.x - 512;
            const float4* src = reinterpret_cast<const float4*>(a.root_states + (size_t)blockIdx.x * root_floats);   // (num_envs % 16 == 0: no partial block)
            for (int i = t; i < root_floats / 4; i += 256) reinterpret_cast<float4*>(s_root)[i] = src[i];
            for (int i = t; i < (int)(sizeof(mms_config) / 4); i += 256) lds[i] = reinterpret_cast<const float*>(Cg)[i];
            if (t < 4) s_leg[t] = load_leg_const(&Cg->model, t);
            constexpr int kStageOff = ((kRedOff + (RP_STRIDE + 6) * AT + 3) & ~3) + ((kObsAT + 3) & ~3);          // s_stage - env_lds
            for (int i = t; i < EPB * prev_dim; i += 256) {
                const int e = i / prev_dim, k = i - e * prev_dim;
                (lds_envs + (size_t)e * env_stride + kStageOff)[k] = a.prev[((size_t)blockIdx.x * EPB + e) * prev_dim + k];
            }
        }
        PpoActOut o{a.head.actions_out, a.head.act_slot, a.head.logp_slot, a.head.value_slot, a.head.mu_slot, a.head.sigma_slot};
        o.lds_actions = s_head_actions;
        o.lds_row0 = (int64_t)blockIdx.x * 16;
        if (a.head.weight_tiles)                                     // (uniform: the tiled copy of the head's weights, head_block.h)
            ppo_head_block<5, 8, 1, 4, true>(s_head, (int)threadIdx.x, threadIdx.x < 512, (int64_t)blockIdx.x * 16, a.head.hidden, a.head.weight, a.head.bias, a.head.H, nullptr,
                                             a.head.vhidden, a.head.vweight, a.head.vbias, a.head.VH, a.head.log_std, a.head.seed, a.head.counters, a.head.row_offset,
                                             a.head.reference_scale, o, a.num_envs, 8 * AT, a.head.weight_tiles);
        else
            ppo_head_block<5, 8, 1, 4>(s_head, (int)threadIdx.x, threadIdx.x < 512, (int64_t)blockIdx.x * 16, a.head.hidden, a.head.weight, a.head.bias, a.head.H, nullptr,
                                       a.head.vhidden, a.head.vweight, a.head.vbias, a.head.VH, a.head.log_std, a.head.seed, a.head.counters, a.head.row_offset,
                                       a.head.reference_scale, o, a.num_envs, 8 * AT);
        __syncthreads();                                             // the sampled actions are in LDS
        if (threadIdx.x < 16 * 4 * AT) ac_head = reinterpret_cast<const float2*>(s_head_actions)[threadIdx.x];   // lane (env e, ant lane t) = thread 40 e + t: actions 80 e + 2 t
        // (no barrier behind the reads: nothing writes the parking space before the first substep's inward pass, which sits behind the
        //  prologue's own barriers)
    }
    const KinPark park{lds_lanes + 4 * threadIdx.x, 4 * BLOCK};
    LegDR* s_dr = reinterpret_cast<LegDR*>(lds_lanes + 6 * 4 * BLOCK) + threadIdx.x;     // (DR kernels only) this lane's slice
    float* env_lds = lds_envs + (size_t)e_loc * env_stride;
    float* s_box = env_lds + kBoxOff;              // [16] box rigid state (home of the box between phases)
    BoxPose* s_bp = reinterpret_cast<BoxPose*>(env_lds + kBpOff);
    float4* s_fac = reinterpret_cast<float4*>(env_lds + kFacOff);   // [7] the factored box matrix + its right-hand side, parked across the barrier
    float* s_epi = env_lds + kEpiOff;              // [8] box values the ant lanes need in the epilogue: global x, y, sin, -cos of the yaw
    float* s_red = env_lds + kRedOff;              // [A][RP_STRIDE] reward partials
    float* s_wr = s_red + RP_STRIDE * A;           // [6][A] per-ant reactions on the box
    float* s_obs = env_lds + ((kRedOff + (RP_STRIDE + 6) * A + 3) & ~3);   // [obs_pad] observation row (16-B aligned)
    // The epilogue's inputs (caches, env origin, progress, reset count) are fetched together with the state and parked
    // behind the observation row so that the kernel pays ONE HBM round trip at the top.
    float* s_stage = s_obs + obs_pad;              // [prev_dim][3 origin][2 progress][2 reset_count]
    // The clamped actions go straight into their slots of the observation row (ten_ant.py:1346, one_ant.py:615) and are
    // read back from there by the substeps: no registers held across the physics loop for them.
    float* s_act = s_obs + (TASK != MMS_TASK_ONE_ANT ? 38 * ant + 30 + 2 * leg : 52 + 2 * leg);

    const int actors = A + 1;
    const bool reset_now = a.reset[env] != 0;

    // ---- load state -------------------------------------------------------------------------
    // (per-env addresses are lane-varying in the packed layouts: they are formed where they are used, not kept alive
    // across the physics loop -- see the fence before the epilogue)
    // Everything the kernel needs from HBM is requested before the first barrier, the lane's own state first: all waves of the
    // single residency round are in their prologue at the same time, so a second round trip would not be hidden by anything.
    AntLane S = {};
    float2 ac = make_float2(0.f, 0.f);
    if (kStage && !HEAD) {                                 // (HEAD: staged by waves 8-11 beside the head prologue, above)
        const size_t g0 = (size_t)blockIdx.x * root_floats, gtot = (size_t)a.num_envs * actors * 13;
        if (g0 + root_floats <= gtot) {
            const float4* src = reinterpret_cast<const float4*>(a.root_states + g0);
            for (int i = threadIdx.x; i < root_floats / 4; i += BLOCK) reinterpret_cast<float4*>(s_root)[i] = src[i];
        } else {                                                                    // the partial last block:
            for (int i = threadIdx.x; i < root_floats; i += BLOCK) {                // envs past the end read the last env (never stored)
                size_t f = g0 + i;
                if (f >= gtot) f = gtot - (size_t)actors * 13 + (f % ((size_t)actors * 13));
                s_root[i] = a.root_states[f];
            }
        }
    } else if (is_ant) {
        const float* r = a.root_states + ((size_t)env * actors + ant) * 13;
        S.pos = V3{r[0], r[1], r[2]};
        S.qx = r[3]; S.qy = r[4]; S.qz = r[5]; S.qw = r[6];
        S.vel = V3{r[7], r[8], r[9]};
        S.ang = V3{r[10], r[11], r[12]};
    }
    if (is_ant) {
        float4 d = reinterpret_cast<const float4*>(a.dof_state + (size_t)env * A * 16)[tid];   // (q1, qd1, q2, qd2): coalesced 16 B / lane
        S.q[0] = d.x; S.qd[0] = d.y; S.q[1] = d.z; S.qd[1] = d.w;
        if constexpr (HEAD) ac = ac_head;                                                        // ... sampled by the head prologue above
        else ac = reinterpret_cast<const float2*>(a.actions + (size_t)env * A * 8)[tid];        // this lane's two actions
        if (DR) *s_dr = load_leg_dr(a.dr + ((size_t)env * A + ant) * MMS_DR_FLOATS, leg);       // read back by this lane only
    }
    // (HEAD: staged by waves 8-11 above)
    for (int i = threadIdx.x; i < (HEAD ? 0 : (int)(sizeof(mms_config) / 4)); i += BLOCK) lds[i] = reinterpret_cast<const float*>(Cg)[i];
    if (!HEAD && threadIdx.x < 4) s_leg[threadIdx.x] = load_leg_const(&Cg->model, threadIdx.x);

    for (int i = threadIdx.x; i < (HEAD ? 0 : EPB * prev_dim); i += BLOCK) {       // the caches of the block's envs are contiguous in HBM
        const int e = i / prev_dim, k = i - e * prev_dim;
        const int en = min((int)(blockIdx.x * EPB + e), a.num_envs - 1);
        (lds_envs + (size_t)e * env_stride + (s_stage - env_lds))[k] = a.prev[(size_t)en * prev_dim + k];
    }
    if (is_box && corner == 1) {
        s_stage[prev_dim + 0] = a.env_origin[3 * env]; s_stage[prev_dim + 1] = a.env_origin[3 * env + 1];
        s_stage[prev_dim + 2] = a.env_origin[3 * env + 2];
        int64_t pr = a.progress[env], rc = a.reset_count[env];
        int* si = reinterpret_cast<int*>(s_stage + prev_dim + 3);
        si[0] = (int)(pr & 0xffffffff); si[1] = (int)(pr >> 32); si[2] = (int)(rc & 0xffffffff); si[3] = (int)(rc >> 32);
    }
    if (is_ant) {
        s_act[0] = clampf(ac.x, -Cg->clip_actions, Cg->clip_actions);     // vec_task.py:127
        s_act[1] = clampf(ac.y, -Cg->clip_actions, Cg->clip_actions);
    }
    if (kStage) __syncthreads();
    if (kStage && is_ant) {                                          // this lane's torso from the staged slice
        const float* r = s_root + (e_loc * actors + ant) * 13;
        S.pos = V3{r[0], r[1], r[2]};
        S.qx = r[3]; S.qy = r[4]; S.qz = r[5]; S.qw = r[6];
        S.vel = V3{r[7], r[8], r[9]};
        S.ang = V3{r[10], r[11], r[12]};
    }
    if (box_lead) {
        RigidState B0 = load_rigid(kStage ? s_root + (e_loc * actors + A) * 13 : a.root_states + ((size_t)env * actors + A) * 13);
        store_rigid(s_box, B0);
        s_bp->pos = B0.pos; s_bp->R = quat_to_mat(B0.qx, B0.qy, B0.qz, B0.qw); s_bp->v = B0.vel; s_bp->w = B0.ang;
        const mms_model* Mb = kStage ? &C->model : &Cg->model;          // (the LDS copy is valid behind the first barrier only)
        s_bp->half = V3{Mb->box_half[0], Mb->box_half[1], Mb->box_half[2]};
    }
    __syncthreads();                                                 // the box pose is read by the ant lanes of the first substep
    const LegConst& L = s_leg[leg];
    float sens[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (TASK == MMS_TASK_ONE_ANT && is_ant) {      // sensors of the last simulated substep persist across a skipped step
        const float* fs = a.foot_sensors + ((size_t)env * A + ant) * 24 + 6 * leg;
#pragma unroll
        for (int i = 0; i < 6; i++) sens[i] = fs[i];
    }

    // ---- physics: skipped per env for envs flagged for reset (their state is overwritten below) ---------------
    const bool simulate = a.do_physics && !reset_now;
    if (a.do_physics) {
        const float h = C->dt / (float)C->substeps;
        constexpr bool kSensors = (TASK == MMS_TASK_ONE_ANT);
        constexpr bool kOneWave = (BLOCK == 64 && EPB == 1);
        // inward pass, quad reduction, root solve, outward pass; returns this lane's reaction on the box
        auto ant_phase = [&](bool wait_for_box) -> S6 {
            S6 wr = S6{V3{0, 0, 0}, V3{0, 0, 0}};
            Sym6 IA0;
            S6 pA0;
            LegPass P;
            SensorPass SP;
            LegInward W;
            // the opening of the inward pass (kinematics, foot body, the tip's ground contact) does not need the box: the ant waves
            // work through it while the box lanes finish the previous substep's serial tail; the barrier behind which this
            // substep's box pose is valid sits HERE, not at the end of the previous substep
            // (one wave per block -- the 64-thread layouts: nothing to overlap; the pass stays in one piece behind the barrier)
            constexpr bool kSplit = BLOCK > 64;
            if (kSplit && is_ant && simulate) leg_inward_open<kSensors, DR>(M, L, h, S, P, &SP, W, park, s_dr);
            if (kSplit ? wait_for_box : false) __syncthreads();
            if (is_ant && simulate) {
                const float tau1 = s_act[0] * L.gear[0] * C->power_scale;    // ten_ant.py:889
                const float tau2 = s_act[1] * L.gear[1] * C->power_scale;
                if (!kSplit) leg_inward<kSensors, DR>(M, L, h, S, leg, tau1, tau2, true, *s_bp, P, &SP, IA0, pA0, park, s_dr);
                else leg_inward_close<kSensors, DR>(M, L, h, S, leg, tau1, tau2, true, *s_bp, P, &SP, IA0, pA0, W, s_dr);
            }
            // (quads never mix ant and box lanes, and `simulate` is per env: a quad that skipped the inward pass skips the outward
            // pass too, so what the reduction leaves in its lanes is never read -- no zero fill)
            quad_sum(IA0, pA0);
            if (is_ant && simulate) leg_outward<kSensors>(M, L, h, S, leg, *s_bp, P, &SP, IA0, pA0, wr, sens, park);
            return wr;
        };
        // envs straddle waves: per-ant sums of the reactions (DPP) into LDS
        auto stage_wrench = [&](S6 wr) {
            wr.a.x = quad_sum(wr.a.x); wr.a.y = quad_sum(wr.a.y); wr.a.z = quad_sum(wr.a.z);
            wr.l.x = quad_sum(wr.l.x); wr.l.y = quad_sum(wr.l.y); wr.l.z = quad_sum(wr.l.z);
            if (is_ant && leg == 0) {
                s_wr[0 * A + ant] = wr.a.x; s_wr[1 * A + ant] = wr.a.y; s_wr[2 * A + ant] = wr.a.z;
                s_wr[3 * A + ant] = wr.l.x; s_wr[4 * A + ant] = wr.l.y; s_wr[5 * A + ant] = wr.l.z;
            }
        };
        // total reaction on the box: column c of the staged per-ant sums is added up in ant order by box lane c; the eight
        // box lanes of an env sit in one wave, 8-aligned, and lane c hands its column to all of them
        auto gather_wrench = [&]() -> S6 {
            float t = 0.f;
            if (corner < 6) {
                const float* col = s_wr + corner * A;
                for (int i = 0; i < A; i++) t += col[i];
            }
            // lane c of each 8-group to the whole group: ds_swizzle in bit mode (lane' = (lane & 0x18) | c within each 32), the
            // pattern an immediate -- six bpermute address registers hoisted out of the substep loop were being spilled
#define MMS_FROM(c) __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(t), 0x18 | ((c) << 5)))
            return S6{V3{MMS_FROM(0), MMS_FROM(1), MMS_FROM(2)}, V3{MMS_FROM(3), MMS_FROM(4), MMS_FROM(5)}};
#undef MMS_FROM
        };
        auto box_store = [&](const RigidState& B) {
            store_rigid(s_box, B);
            s_bp->pos = B.pos; s_bp->R = quat_to_mat(B.qx, B.qy, B.qz, B.qw); s_bp->v = B.vel; s_bp->w = B.ang;
        };
        // (the loop itself is block-uniform so that every lane reaches every barrier)
        const bool box_friction = M->boxgnd_mu > 0.f;                // uniform: box-ground friction couples all six accelerations
        for (int s = 0; s < C->substeps; s++) {
            S6 wr = ant_phase(s > 0);
            // The box's own ground contacts (with or without friction) do not depend on this substep's ant reactions: the corner
            // lanes evaluate and reduce them BEFORE the barrier -- in the layouts whose box lanes fill waves of their own that is time in which those waves would
            // only wait for the ant lanes; after the barrier the box needs the wrench, one small solve and the integration.
            BoxCorner bc = {};
            if (is_box && !box_friction) {
                bc = box_corner(M, h, load_rigid(s_box), s_bp->R, corner);
                oct_sum(bc);
            }
            if (is_box && box_friction) {
                // with friction: the corner terms, their reduction AND the factorisation of the box's 6x6 matrix (it does not depend
                // on the wrench) before the barrier; behind it only the two substitutions and the integration are serial
                const RigidState B0 = load_rigid(s_box);
                BoxCornerF bf = box_corner_friction(M, h, B0, s_bp->R, corner);
#pragma unroll
                for (int k = 0; k < 21; k++) bf.IA.m[k] = oct_sum(bf.IA.m[k]);
                bf.pA.a.x = oct_sum(bf.pA.a.x); bf.pA.a.y = oct_sum(bf.pA.a.y); bf.pA.a.z = oct_sum(bf.pA.a.z);
                bf.pA.l.x = oct_sum(bf.pA.l.x); bf.pA.l.y = oct_sum(bf.pA.l.y); bf.pA.l.z = oct_sum(bf.pA.l.z);
                static_assert(sizeof(BoxFactorF) == 27 * 4, "parked as seven 16-B words");
                union { BoxFactorF f; float4 q[7]; } u;              // 27 floats: parked in LDS, not carried across the barrier (on top of
                u.q[6] = make_float4(0.f, 0.f, 0.f, 0.f);            // the ant lanes' live state they were being spilled to scratch)
                u.f = box_factor_friction(M, B0, s_bp->R, bf);
                if (corner == 0) {
#pragma unroll
                    for (int k = 0; k < 7; k++) s_fac[k] = u.q[k];
                }
            }
            S6 w = S6{V3{0, 0, 0}, V3{0, 0, 0}};
            if (kOneWave) {                                          // one wave per env: the reaction wrench by DPP / permute
                w = S6{V3{wave_sum(wr.a.x), wave_sum(wr.a.y), wave_sum(wr.a.z)}, V3{wave_sum(wr.l.x), wave_sum(wr.l.y), wave_sum(wr.l.z)}};
            } else {
                stage_wrench(wr);
                __syncthreads();
            }
            if (is_box) {                                            // waves without box lanes skip the whole box phase
                if (!kOneWave) w = gather_wrench();
                RigidState B = load_rigid(s_box);                    // (again: cheaper than carrying 22 values across the barrier)
                M3 R = s_bp->R;
                if (box_friction) {
                    union { BoxFactorF f; float4 q[7]; } u;
#pragma unroll
                    for (int k = 0; k < 7; k++) u.q[k] = s_fac[k];
                    if (simulate) box_solve_friction(h, B, u.f, w);
                } else if (simulate) box_finish(M, h, B, R, bc, w);
                if (corner == 0) box_store(B);
            }
            // (no barrier here: the next substep's ant phase waits for the box pose after its box-independent opening, and behind
            // the last substep only the box lead reads what it wrote itself until the epilogue's own barrier)
            if (BLOCK == 64) __syncthreads();
        }
    }

    // ---- post_physics_step: progress, reset_idx (ten_ant.py:894-901) --------------------------
    int env_e = env;
    asm volatile("" : "+v"(env_e));                 // opaque copy: per-env addresses are re-formed here, not carried through the loop
    const float* init_env = a.initial_root_states + (size_t)env_e * actors * 13;
    float* dof_env = a.dof_state + (size_t)env_e * A * 16;
    float* prev_env = a.prev + (size_t)env_e * prev_dim;
    const V3 origin = V3{s_stage[prev_dim + 0], s_stage[prev_dim + 1], s_stage[prev_dim + 2]};
    const unsigned* si = reinterpret_cast<const unsigned*>(s_stage + prev_dim + 3);
    int64_t progress = (int64_t)(((uint64_t)si[1] << 32) | si[0]);
    const int64_t reset_count = (int64_t)(((uint64_t)si[3] << 32) | si[2]);   // bumped by the box lead after the last barrier
    const uint64_t env_global = (uint64_t)(C->env_offset + env_e);
    progress += 1;
    if (reset_now) progress = 0;
    if (reset_now && is_ant) ant_reset_lane(C, L, S, init_env + 13 * ant, leg, a.reset_noise + 16 * (size_t)env_e, env_global, (uint64_t)reset_count);
    // ---- write the state back ---------------------------------------------------------------
    if (is_ant && live) {
        reinterpret_cast<float4*>(dof_env)[tid] = make_float4(S.q[0], S.qd[0], S.q[1], S.qd[1]);
        if (leg == 0) {
            // (into the staged slice, which leaves with the observation rows; the runtime-sized layouts store per lane)
            float* r = kStage ? s_root + (e_loc * actors + ant) * 13 : a.root_states + ((size_t)env_e * actors + ant) * 13;
            r[0] = S.pos.x; r[1] = S.pos.y; r[2] = S.pos.z; r[3] = S.qx; r[4] = S.qy; r[5] = S.qz; r[6] = S.qw;
            r[7] = S.vel.x; r[8] = S.vel.y; r[9] = S.vel.z; r[10] = S.ang.x; r[11] = S.ang.y; r[12] = S.ang.z;
        }
    }
    // The box lead owns the box in the epilogue: state write-back, its observation entries, and the values the ant lanes
    // need (global position, yaw direction), handed over through LDS.
    RigidState B = {};
    if (box_lead) {
        B = reset_now ? load_rigid(init_env + 13 * A) : load_rigid(s_box);
        if (live) store_rigid(kStage ? s_root + (e_loc * actors + A) * 13 : a.root_states + ((size_t)env_e * actors + A) * 13, B);
        const float bgx = B.pos.x + origin.x, bgy = B.pos.y + origin.y;      // global frame
        s_epi[0] = bgx; s_epi[1] = bgy;
        if (TASK == MMS_TASK_TEN_ANT) {
            box_yaw_dir(B.qz, B.qw, s_epi[2], s_epi[3]);
            float* t = s_obs + 38 * A;
            t[0] = bgx; t[1] = bgy; t[2] = B.qx; t[3] = B.qy; t[4] = B.qz; t[5] = B.qw; t[6] = 0.f; t[7] = 0.f;
        }
    }
    __syncthreads();

    // ---- observations + reward --------------------------------------------------------------
    const float bgx = s_epi[0], bgy = s_epi[1];
    if (TASK == MMS_TASK_TEN_ANT) {
        if (is_ant) {
            const float sv = s_epi[2], cv = s_epi[3];
            // caches: previous step's values; mms_create fills them from the construction-time poses (ten_ant.py:870-882)
            const float pbx = s_stage[2 * ant], pby = s_stage[2 * ant + 1];
            const float gbx = s_stage[2 * A + 2 * ant], gby = s_stage[2 * A + 2 * ant + 1];
            const float act0 = s_act[0], act1 = s_act[1];
            TenAntLaneOut o = tenant_obs_reward_lane(C, L, S, ant, leg, origin, act0, act1, bgx, bgy, sv, cv, pbx, pby, gbx, gby, s_obs);
            float ec = quad_sum(o.ec), lim = quad_sum(o.lim), ac = quad_sum(o.acost);
            if (leg == 0) {
                float* r = s_red + RP_STRIDE * ant;
                r[RP_ADR] = o.adr; r[RP_GDR] = o.gdr; r[RP_GAR] = o.gar; r[RP_UP] = o.up; r[RP_EC] = ec; r[RP_LIM] = lim;
                r[RP_FALLEN] = o.fallen; r[RP_ACOST] = ac;
            }
            if (leg == 0 && live) {
                prev_env[2 * ant] = o.px; prev_env[2 * ant + 1] = o.py;                 // ten_ant.py:906-926
                prev_env[2 * A + 2 * ant] = o.gx; prev_env[2 * A + 2 * ant + 1] = o.gy;
            }
        }
        __syncthreads();
        if (box_lead && live) {
            prev_env[4 * A] = bgx; prev_env[4 * A + 1] = bgy;
            float rew;
            int64_t rs;
            tenant_reward_finish(C, A, s_red, B.qx, B.qy, B.qz, B.qw, progress, rew, rs);
            a.rew[env_e] = rew;
            a.reset[env_e] = rs;
            a.progress[env_e] = progress;
            if (a.rew_out) a.rew_out[env_e] = rew;
            if (a.done_out) a.done_out[env_e] = (uint8_t)rs;
        }
    } else if (TASK == MMS_TASK_MULTI_ANT_CIRCLE) {   // two ants on the ring, no box in the scene (the box lead only finishes the reward)
        if (is_ant) {
            const float pbx = s_stage[2 * ant], pby = s_stage[2 * ant + 1];
            const float act0 = s_act[0], act1 = s_act[1];
            CircleLaneOut o = circle_obs_reward_lane(C, L, S, ant, leg, origin, act0, act1, pbx, pby, s_obs);
            float ec = quad_sum(o.ec), lim = quad_sum(o.lim), ac = quad_sum(o.acost);
            if (leg == 0) {
                float* r = s_red + RP_STRIDE * ant;
                r[RP_ADR] = o.rk; r[RP_UP] = o.up; r[RP_EC] = ec; r[RP_LIM] = lim; r[RP_FALLEN] = o.fallen; r[RP_ACOST] = ac;
            }
            if (leg == 0 && live) { prev_env[2 * ant] = o.px; prev_env[2 * ant + 1] = o.py; }   // multi_ant_circle.py:382-383
        }
        __syncthreads();
        if (box_lead && live) {
            float rew;
            int64_t rs;
            circle_reward_finish(C, A, s_red, progress, rew, rs);
            a.rew[env_e] = rew;
            a.reset[env_e] = rs;
            a.progress[env_e] = progress;
            if (a.rew_out) a.rew_out[env_e] = rew;
            if (a.done_out) a.done_out[env_e] = (uint8_t)rs;
        }
    } else {  // OneAnt: the four leg lanes stage their reward partials, the box lead finishes
        if (is_ant) {
            const float act0 = s_act[0], act1 = s_act[1];
            AntObsCore core;
            V3 pg;
            OneAntLaneOut o = oneant_obs_lane(C, L, S, leg, origin, act0, act1, sens, s_obs, core, pg);
            float ec = quad_sum(o.ec), lim = quad_sum(o.lim), ac = quad_sum(o.acost);
            if (simulate && live) {
                float* fs = a.foot_sensors + ((size_t)env_e * A + ant) * 24 + 6 * leg;
#pragma unroll
                for (int i = 0; i < 6; i++) fs[i] = sens[i];
            }
            if (leg == 0) {
                float* r = s_red;
                r[0] = pg.x; r[1] = pg.y; r[2] = pg.z; r[3] = core.up_proj; r[4] = ec; r[5] = lim; r[6] = ac;
            }
        }
        __syncthreads();
        if (box_lead && live) {
            const float pot_in = s_stage[4];
            const float pbx = s_stage[0], pby = s_stage[1], bbx = s_stage[2], bby = s_stage[3];
            const float* r = s_red;
            float tbx = 0.f - bgx, tby = 0.f - bgy;
            float pot = -sqrtf(tbx * tbx + tby * tby + 0.f * 0.f) / C->dt;          // one_ant.py:583-587
            float rew;
            int64_t rs;
            oneant_reward(C, r[2], r[3], r[4], r[5], r[6], pbx, pby, bbx, bby, r[0], r[1], bgx, bgy, B.qx, B.qy, B.qz, B.qw, progress, rew, rs);
            a.rew[env_e] = rew;
            a.reset[env_e] = rs;
            a.progress[env_e] = progress;
            if (a.rew_out) a.rew_out[env_e] = rew;
            if (a.done_out) a.done_out[env_e] = (uint8_t)rs;
            prev_env[0] = r[0]; prev_env[1] = r[1]; prev_env[2] = bgx; prev_env[3] = bgy;   // one_ant.py:432-433
            prev_env[4] = pot; prev_env[5] = pot_in;
        }
    }
    if (reset_now && box_lead && live) a.reset_count[env_e] = reset_count + 1;   // every lane of the env read it before the barriers above
    // ---- the staged root slice back to HBM: coalesced 16-B stores (live envs only) ----------------------------------
    if (kStage) {
        const size_t g0 = (size_t)blockIdx.x * root_floats, gtot = (size_t)a.num_envs * actors * 13;
        if (g0 + root_floats <= gtot) {
            float4* dst = reinterpret_cast<float4*>(a.root_states + g0);
            for (int i = threadIdx.x; i < root_floats / 4; i += BLOCK) dst[i] = reinterpret_cast<const float4*>(s_root)[i];
        } else {
            for (int i = threadIdx.x; i < root_floats; i += BLOCK)
                if (g0 + i < gtot) a.root_states[g0 + i] = s_root[i];
        }
    }
    // ---- coalesced observation row(s): the block's EPB rows are contiguous in HBM ---------------------------------
    {
        const int e0 = blockIdx.x * EPB;
        const int n_live = min(EPB, a.num_envs - e0);
        const float clip = C->clip_obs;
        // each of the three destinations is optional (mms_set_obs_outputs / mms_bind_obs_out)
        float* obs = a.obs ? a.obs + (size_t)e0 * obs_dim : nullptr;
        float* obs_clip = a.obs_clipped ? a.obs_clipped + (size_t)e0 * obs_dim : nullptr;
        float* obs_out = a.obs_out ? a.obs_out + (size_t)e0 * obs_dim : nullptr;
        const float* rows = lds_envs + (s_obs - env_lds);
        if ((obs_dim & 3) == 0) {
            const int q = obs_dim >> 2;
            for (int i = threadIdx.x; i < n_live * q; i += BLOCK) {
                const int e = EPB == 1 ? 0 : i / q, k = i - e * q;
                float4 v = reinterpret_cast<const float4*>(rows + (size_t)e * env_stride)[k];
                // streaming stores: the rows are not read again by this kernel, and allocating 19 MB of them in L2 would first
                // have to write back whatever the previous kernel (a policy GEMM, in a rollout) left dirty there
                if (obs) stream_store(reinterpret_cast<float4*>(obs) + i, v);
                float4 c = make_float4(clampf(v.x, -clip, clip), clampf(v.y, -clip, clip), clampf(v.z, -clip, clip), clampf(v.w, -clip, clip));
                if (obs_clip) stream_store(reinterpret_cast<float4*>(obs_clip) + i, c);
                if (obs_out) stream_store(reinterpret_cast<float4*>(obs_out) + i, c);
            }
        } else {
            for (int i = threadIdx.x; i < n_live * obs_dim; i += BLOCK) {
                const int e = EPB == 1 ? 0 : i / obs_dim, k = i - e * obs_dim;
                float v = rows[(size_t)e * env_stride + k];
                if (obs) obs[i] = v;
                float c = clampf(v, -clip, clip);
                if (obs_clip) obs_clip[i] = c;
                if (obs_out) obs_out[i] = c;
            }
        }
        // ... and, when bound, the clamped row once more as the policy layers' operand planes (format H32 of split16_kernels.hip: per
        // 32 k the fp16 hi and lo planes, 128 bytes): the rows are bounded by clip_obs, so their scale is a constant and the separate
        // split pass over the observation (6 us, one launch) is not needed.  One 8-k piece per thread and trip.
        if (a.obs_planes) {
            typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
            const int KC = (obs_dim + 31) >> 5, pieces = KC * 4;
            const float sc = a.obs_planes_scale;
            uint8_t* out = reinterpret_cast<uint8_t*>(a.obs_planes) + (size_t)e0 * KC * 128;
            int tix = threadIdx.x;
            asm volatile("" : "+v"(tix));           // (opaque: nothing of this loop's index math is hoisted across the physics -- one spill otherwise)
            for (int i = tix; i < n_live * pieces; i += BLOCK) {
                const int e = EPB == 1 ? 0 : i / pieces, p = i - e * pieces, k0 = p * 8;
                const float* r = rows + (size_t)e * env_stride + k0;
                f16x8_t hi, lo;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float t = (k0 + j < obs_dim) ? clampf(r[j], -clip, clip) * sc : 0.f;
                    hi[j] = (_Float16)t;
                    lo[j] = (_Float16)((t - (float)hi[j]) * 2048.f);
                }
                uint8_t* dst = out + ((size_t)e * KC + (p >> 2)) * 128 + (p & 3) * 16;
                *reinterpret_cast<f16x8_t*>(dst) = hi;
                *reinterpret_cast<f16x8_t*>(dst + 64) = lo;
            }
        }
    }
}

#ifdef MMS_STEP_HEAD_TU
// ---- the translation unit of the <..., HEAD = true> instantiation (step_head_kernels.hip) --------------------------------------------
// A TU of its own: the register allocation of the 768-thread kernels sits AT the 168-VGPR limit of three waves per SIMD, and whether a
// dword ends up in scratch turned out to depend on what ELSE the module holds (the plain instantiation picked up a spill from an edit
// confined to the discarded `if constexpr (HEAD)` branch, and lost it again when the HEAD instantiation was not emitted).  Compiled
// apart, an edit to the head prologue cannot move the allocation of the kernels every other caller launches.
hipError_t launch_ten_ant_with_head(const StepArgs& a, size_t lds, int grid, hipStream_t stream) {
    if (a.dr || a.num_envs % 16 != 0 || !a.head_on) return hipErrorInvalidValue;
    static bool head_allowed[64] = {};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const void* k = reinterpret_cast<const void*>(ant_step_kernel<MMS_TASK_TEN_ANT, 768, 16, 10, false, true>);
    if (dev >= 0 && dev < 64 && !head_allowed[dev]) {
        if (hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e != hipSuccess) return e;
        head_allowed[dev] = true;
    }
    hipLaunchKernelGGL((ant_step_kernel<MMS_TASK_TEN_ANT, 768, 16, 10, false, true>), dim3(grid), dim3(768), lds, stream, a);
    return hipGetLastError();
}
#else
hipError_t launch_ten_ant_with_head(const StepArgs& a, size_t lds, int grid, hipStream_t stream);       // step_head_kernels.hip

// ---------------------------------------------------------------------------------------------
// MultiIngenuity: one lane per helicopter, 4 lanes per env, 16 envs per wave64.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ingenuity_step_kernel(StepArgs a) {
    const mms_config* __restrict__ C = a.cfg;
    const mms_model* __restrict__ M = &C->model;
    const int A = a.num_agents;                    // 4
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int env = gid / A, k = gid - env * A;
    const bool live = env < a.num_envs;
    const int e = live ? env : 0;
    float* root = a.root_states + ((size_t)e * A + k) * 13;
    const float* init = a.initial_root_states + ((size_t)e * A + k) * 13;
    float* dof = a.dof_state + ((size_t)e * A + k) * 8;
    const float* act = a.actions + (size_t)e * 6 * A + 6 * k;
    const int64_t reset_flag = a.reset[e];
    int64_t progress = a.progress[e];
    RigidState B = load_rigid(root);
    float dq[8];
#pragma unroll
    for (int j = 0; j < 8; j++) dq[j] = dof[j];
    if (a.do_physics && reset_flag == 0) {
        // multi_ingenuity.py:268-327: thrust = dt * clamp(2000 a) vertical, lateral fraction clamp(+-0.2)
        V3 thr[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            float a0 = clampf(act[3 * r + 0], -C->clip_actions, C->clip_actions);
            float a1 = clampf(act[3 * r + 1], -C->clip_actions, C->clip_actions);
            float a2 = clampf(act[3 * r + 2], -C->clip_actions, C->clip_actions);
            float vert = clampf(a2 * 2000.f, -2000.f, 2000.f);
            float tz = C->dt * vert;
            thr[r] = V3{tz * clampf(a0, -0.2f, 0.2f), tz * clampf(a1, -0.2f, 0.2f), tz};
        }
        const float h = C->dt / (float)C->substeps;
        for (int s = 0; s < C->substeps; s++) {
            heli_substep(M, h, B, thr[0], thr[1]);
#pragma unroll
            for (int j = 0; j < 4; j++) dq[2 * j] += h * dq[2 * j + 1];   // visual rotors: kinematic
        }
    }
    progress += 1;
    if (reset_flag != 0) {
        B = load_rigid(init);
#pragma unroll
        for (int j = 0; j < 4; j++) { dq[2 * j] = 0.f; dq[2 * j + 1] = (j == 1) ? -50.f : (j == 3 ? 50.f : 0.f); }   // :231-238
        progress = 0;
        if (live && k == 0) a.reset_count[env] += 1;
    }
    if (live) {
        store_rigid(root, B);
#pragma unroll
        for (int j = 0; j < 8; j++) dof[j] = dq[j];
        // obs = raw root states in the global frame (multi_ingenuity.py:351-357)
        float* o = a.obs ? a.obs + (size_t)env * 13 * A + 13 * k : nullptr;
        float* oc = a.obs_clipped ? a.obs_clipped + (size_t)env * 13 * A + 13 * k : nullptr;
        float* oo = a.obs_out ? a.obs_out + (size_t)env * 13 * A + 13 * k : nullptr;
        float row[13];
        store_rigid(row, B);
        row[0] += a.env_origin[3 * env]; row[1] += a.env_origin[3 * env + 1]; row[2] += a.env_origin[3 * env + 2];
#pragma unroll
        for (int j = 0; j < 13; j++) {
            if (o) o[j] = row[j];
            float c = clampf(row[j], -C->clip_obs, C->clip_obs);
            if (oc) oc[j] = c;
            if (oo) oo[j] = c;
        }
    }
    // the 4 helicopters of an env are the 4 lanes of a quad: every lane receives all four rows by DPP
    // quad broadcasts and lane 0 evaluates the team reward.
    float rows[4][13];
    {
        float mine[13];
        store_rigid(mine, B);
        const float ox = a.env_origin[3 * e], oy = a.env_origin[3 * e + 1], oz = a.env_origin[3 * e + 2];
        mine[0] += ox; mine[1] += oy; mine[2] += oz;
#pragma unroll
        for (int j = 0; j < 13; j++) {
            rows[0][j] = dpp_mov<0x00>(mine[j]);   // quad_perm [0,0,0,0]
            rows[1][j] = dpp_mov<0x55>(mine[j]);   // [1,1,1,1]
            rows[2][j] = dpp_mov<0xAA>(mine[j]);   // [2,2,2,2]
            rows[3][j] = dpp_mov<0xFF>(mine[j]);   // [3,3,3,3]
        }
    }
    if (live && k == 0) {
        float rew;
        int64_t rs;
        ingenuity_reward(&rows[0][0], C->max_episode_length, progress, rew, rs);
        a.rew[env] = rew;
        a.reset[env] = rs;
        a.progress[env] = progress;
        if (a.rew_out) a.rew_out[env] = rew;
        if (a.done_out) a.done_out[env] = (uint8_t)rs;
    }
}

// ---- launchers ---------------------------------------------------------------------------------
template <int TASK, int BLOCK, int EPB, int AT>
static hipError_t launch_ant(const StepArgs& a, hipStream_t stream) {
    size_t lds = (sizeof(mms_config) + 15) / 16 * 16 + (4 * sizeof(LegConst) + 15) / 16 * 16 + (size_t)EPB * ((ant_env_lds_floats(a.obs_dim, a.num_agents) + 3) & ~(size_t)3) * sizeof(float) +
                 ((AT > 0 && (EPB * (AT + 1) * 13) % 4 == 0) ? (size_t)(EPB * (AT + 1) * 13) * sizeof(float) : 0) +   // + the staged root slice (kStage)
                 (size_t)6 * 4 * BLOCK * sizeof(float);          // + the kinematics parking space
    int grid = (a.num_envs + EPB - 1) / EPB;
    if (a.dr) lds += (size_t)BLOCK * sizeof(LegDR);
    if (lds > 64 * 1024) {                                       // dynamic LDS beyond 64 KB has to be allowed per kernel and device
        static bool allowed[2][64] = {};
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && !allowed[a.dr ? 1 : 0][dev]) {
            const void* k = a.dr ? reinterpret_cast<const void*>(ant_step_kernel<TASK, BLOCK, EPB, AT, true>)
                                 : reinterpret_cast<const void*>(ant_step_kernel<TASK, BLOCK, EPB, AT, false>);
            if (hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e != hipSuccess) return e;
            allowed[a.dr ? 1 : 0][dev] = true;
        }
    }
    if (a.head_on) {
        // the fused policy head exists for the 16-envs-per-block TenAnt layout without physical DR only (mms_bind_policy_head checks)
        if constexpr (TASK == MMS_TASK_TEN_ANT && BLOCK == 768 && EPB == 16 && AT == 10) {
            return launch_ten_ant_with_head(a, lds, grid, stream);
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (a.dr) hipLaunchKernelGGL((ant_step_kernel<TASK, BLOCK, EPB, AT, true>), dim3(grid), dim3(BLOCK), lds, stream, a);
    else hipLaunchKernelGGL((ant_step_kernel<TASK, BLOCK, EPB, AT, false>), dim3(grid), dim3(BLOCK), lds, stream, a);
    return hipGetLastError();
}

// would mms_step launch the layout the fused policy head exists for? (mms_bind_policy_head)
bool step_layout_takes_head(int task, int num_envs, int num_agents, int packing) {
    if (task != MMS_TASK_TEN_ANT || num_agents != 10 || packing == 0 || num_envs % 16 != 0) return false;
    const char* force16 = getenv("MMS_STEP_BLOCK16");
    if (force16) return force16[0] != '0';
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return num_envs >= 16 * cus;
}

hipError_t launch_step(const StepArgs& a, int task, hipStream_t stream) {
    if (task == MMS_TASK_MULTI_INGENUITY) {
        if (a.num_agents != 4) return hipErrorInvalidValue;
        int total = a.num_envs * a.num_agents;
        hipLaunchKernelGGL(ingenuity_step_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    const int lpe = ((4 * a.num_agents + 7) & ~7) + 8;        // lanes one env needs
    if (task == MMS_TASK_ONE_ANT) return (a.packing != 0) ? launch_ant<MMS_TASK_ONE_ANT, 64, 4, 1>(a, stream) : launch_ant<MMS_TASK_ONE_ANT, 64, 1, 1>(a, stream);
    // MultiAntCircle: 8 ant + 8 box lanes per env; four envs fill a wave (the packed form OneAnt uses), one env per wave when unpacked
    if (task == MMS_TASK_MULTI_ANT_CIRCLE) return (a.packing != 0) ? launch_ant<MMS_TASK_MULTI_ANT_CIRCLE, 64, 4, 2>(a, stream) : launch_ant<MMS_TASK_MULTI_ANT_CIRCLE, 64, 1, 0>(a, stream);
    if (task != MMS_TASK_TEN_ANT) return hipErrorInvalidValue;
    // TenAnt, two packed layouts.  <192,4>: 4 envs per block, the third wave half ant lanes / half box lanes -- many small blocks,
    // best while there are fewer than 16 envs per CU.  <768,16>: 16 envs per block = ten pure ant waves + two pure box waves, one
    // block per CU and exactly 3 waves per SIMD: no wave issues the ant phase for 32 idle lanes (26.3 against 28.0 us at 4096
    // envs, and 31 against 38 us right after the policy GEMMs).  MMS_STEP_BLOCK16=0 / 1 forces one of them (A/B).
    const char* force16 = getenv("MMS_STEP_BLOCK16");                 // read per launch: the tests switch it
    bool block16;
    if (force16) block16 = force16[0] != '0';
    else {
        static int cus[64] = {};
        int dev = 0;
        if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (cus[dev] == 0 && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return hipErrorInvalidDevice;
        block16 = a.num_envs >= 16 * cus[dev];
    }
    if (a.num_agents == 10 && a.packing != 0 && block16) return launch_ant<MMS_TASK_TEN_ANT, 768, 16, 10>(a, stream);
    if (a.num_agents == 10 && a.packing != 0) return launch_ant<MMS_TASK_TEN_ANT, 192, 4, 10>(a, stream);
    if (lpe <= 64) return launch_ant<MMS_TASK_TEN_ANT, 64, 1, 0>(a, stream);
    if (lpe <= 512) return launch_ant<MMS_TASK_TEN_ANT, 512, 1, 0>(a, stream);
    return hipErrorInvalidValue;
}
#endif  // MMS_STEP_HEAD_TU

}  // namespace mms
