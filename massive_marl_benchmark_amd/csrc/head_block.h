// head_block.h -- the PPO policy's output heads + sampling tail for one block of 16 RT rows, shared by the stand-alone heads kernel
// (rollout_kernels.hip: mms_ppo_heads_act) and the step kernel's fused prologue (step_kernels.hip: mms_bind_policy_head).  One body, so
// that both evaluate the same instruction sequence on the same operands.
//   ppo_sample_row   the sampling tail of ActorCritic.act (agents/algorithms/rl/ppo/module.py:73-87) for one row by one wave
//   ppo_head_block   actor's last Linear layer on the matrix cores (module.py:29-30), critic's last layer as a dot product (:49), sampling
#pragma once
#include <hip/hip_runtime.h>
#ifndef MMS_HEAD_STAMP
#define MMS_HEAD_STAMP 0
#endif
#include <stdint.h>

#include "mms_lane.h"
#include "rollout_lane.h"

namespace mms {

// Sampling of one row by one wave: lane j draws the noise of action j (and j + 64), the row's log-probability is a wave
// reduction.  The per-row draw counter lives in device memory so that a replayed hipGraph sees fresh noise; the store of
// c + 1 depends on the load of c, which orders the two.  `mean_row` may point to global memory or LDS.
struct PpoActOut {
    float* actions_out; float* act_slot; float* logp_slot; float* value_slot; float* mu_slot; float* sigma_slot;
    float* lds_actions = nullptr;     // optional: the block's actions also into LDS, [rows of the block][A] (the step kernel's fused prologue)
    int64_t lds_row0 = 0;             // ... first row of the block
};
__device__ __forceinline__ void ppo_sample_row(const float* mean_row, const float* __restrict__ value, const float* __restrict__ log_std,
                                               uint64_t seed, int64_t* __restrict__ counters, int64_t row_offset, int ref_scale,
                                               const PpoActOut& o, int64_t row, int A, int lane, bool have_value = false, float value_now = 0.f) {
    const int64_t c = counters[row];
    float lp = 0.f;
    for (int j = lane; j < A; j += 64) {
        const float ls = log_std[j];
        const float m = mean_row[j];
        float term;
        const float act = ppo_sample_one(m, ls, seed, (uint64_t)(row_offset + row), (uint64_t)c, (uint32_t)j, ref_scale, term);
        lp += term;
        if (o.actions_out) o.actions_out[row * A + j] = act;
        if (o.lds_actions) o.lds_actions[(row - o.lds_row0) * A + j] = act;
        if (o.act_slot) o.act_slot[row * A + j] = act;
        if (o.mu_slot) o.mu_slot[row * A + j] = m;
        if (o.sigma_slot) o.sigma_slot[row * A + j] = ls;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lp += __shfl_xor(lp, m, 64);
    if (lane == 0) {
        if (o.logp_slot) o.logp_slot[row] = lp;
        if (o.value_slot && (have_value || value)) o.value_slot[row] = have_value ? value_now : value[row];
        counters[row] = c + 1;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The actor's last Linear layer folded in: mean = hidden @ W^T + b on the matrix cores, then the sampling.
// A block owns 16 RT rows; its WAVES (8 when H is a multiple of 512, else 4 / 2 / 1) waves split K = H evenly and each accumulates the 16 x A tile of its share with
// v_mfma_f32_16x16x4_f32 (exact fp32 products and sums).  Operand lane map: lane l supplies A[l & 15][k = l >> 4] and
// B[k = l >> 4][l & 15]; a lane loads 4 consecutive k of its row as one float4 and feeds four MFMAs from it, which only
// permutes the order in which k is summed.  The partial sums meet in LDS, in wave order.
// NCT = number of 16-column tiles (compile time: the accumulators must be plain registers), A <= 16 NCT.
// Rows past N and columns past A are computed from clamped (valid) addresses and never read back.
// RT = 16-row tiles per block: with RT = 2 the head's weight fragments feed two row tiles (weight traffic halved, half as many blocks:
// measured slower at 4096 rows, 19.5 against 13.7 us -- the launch is latency bound).
// tid / active: the thread's index among the 64 WAVES head threads; threads of a larger block (the step kernel's 768) pass active =
// false and only join the two block barriers.  s_part: LDS, (WAVES + 1) * 16 RT * 16 NCT floats.
// UB = float4 operand groups loaded ahead of their MFMAs per trip (4: all of a 64-k trip, the stand-alone kernel; 2: half a trip at a
// time -- 48 fewer live registers for the step kernel's 168-VGPR budget; every accumulator sees the same sequence of products either way).
// TILED: the head's weights are read from `weight_tiles` = the same matrix stored [column tile ct][k / 4][16 outputs][4 consecutive k]
// (zero rows for outputs >= A): the 64 lanes of one operand load -- lane (i, g) wants output 16 ct + i, k-group g -- then read 1 KB of
// CONTIGUOUS memory.  From the row-major [A][H] matrix the same load touches 16 rows x 64 B with the four lanes of a row 16 lanes apart:
// 64 separate cache-line accesses per instruction, and the 160 KB every workgroup needs of it arrive at ~40 GB/s (measured: the fused
// step kernel 38.4-39.2 us against 40.4-41.4, the rollout step 0.1391 against 0.1410 ms).  The values and their order are the same.
template <int NCT, int WAVES, int RT, int UB = 4, bool TILED = false>
__device__ __forceinline__ void ppo_head_block(float* s_part, const int tid, const bool active, const int64_t r0,
                                               const float* __restrict__ hidden, const float* __restrict__ weight,
                                               const float* __restrict__ bias, int H, const float* __restrict__ value,
                                               const float* __restrict__ vhidden, const float* __restrict__ vweight,
                                               const float* __restrict__ vbias, int VH,
                                               const float* __restrict__ log_std, uint64_t seed, int64_t* __restrict__ counters,
                                               int64_t row_offset, int ref_scale, const PpoActOut& o, int64_t N, int A,
                                               const float* __restrict__ weight_tiles = nullptr) {
    constexpr int AP = NCT * 16, ROWS = 16 * RT;
    constexpr int RPW = ROWS / WAVES;                                        // rows sampled per wave (WAVES in 1, 2, 4, 8)
#if MMS_HEAD_STAMP   // phase probe (timing experiments only): 100-MHz ticks of block 0 / wave 0 into sigma_slot[0..5]
    uint64_t st[6];
    st[0] = __builtin_amdgcn_s_memrealtime();
#define MMS_HEAD_ST(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MMS_HEAD_ST(i) do { } while (0)
#endif
    const int lane = tid & 63, wave = active ? (tid >> 6) : 0;
    const int i = lane & 15, g = lane >> 4;
    float v_rows[RPW];
#pragma unroll
    for (int q = 0; q < RPW; q++) v_rows[q] = 0.f;
    if (active) {
        const float* hrow[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int64_t row_a = r0 + 16 * rt + i < N ? r0 + 16 * rt + i : N - 1;
            hrow[rt] = hidden + row_a * (int64_t)H + 4 * g;
        }
        const float* wrow[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) {
            const int j = ct * 16 + i;
            wrow[ct] = weight + (int64_t)(j < A ? j : A - 1) * H + 4 * g;
        }
        f32x4 acc[RT][NCT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The critic's last layer (module.py:49: nn.Linear(hidden, 1)) for the rows this wave samples below: one dot product per row,
        // evaluated here so that its loads travel with the actor head's operands instead of costing a round trip after the barriers.
#pragma unroll
        for (int q = 0; q < RPW; q++) {
            const int64_t row = r0 + wave * RPW + q;
            if (vhidden && row < N) {
                const float* hv = vhidden + row * (int64_t)VH;
                float part = 0.f;
                for (int k = lane * 4; k < VH; k += 256) {
                    const float4 h4 = *reinterpret_cast<const float4*>(hv + k), w4 = *reinterpret_cast<const float4*>(vweight + k);
                    part += h4.x * w4.x + h4.y * w4.y + h4.z * w4.z + h4.w * w4.w;
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
                v_rows[q] = part + vbias[0];
            }
        }
        MMS_HEAD_ST(1);                                                      // the critic's dot products are done
        const int kq = H / WAVES;
        const int kbeg = wave * kq;
        // 64 k per trip (the launcher picks WAVES so that H / WAVES is a multiple of 64): the 4 x (RT + NCT) float4 loads of a
        // trip are issued together, then its 16 RT NCT MFMAs
        for (int kc = kbeg; kc < kbeg + kq; kc += 64) {
#pragma unroll
            for (int u0 = 0; u0 < 4; u0 += UB) {
                float4 a[RT][UB], b[UB][NCT];
#pragma unroll
                for (int u = 0; u < UB; u++) {
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) a[rt][u] = *reinterpret_cast<const float4*>(hrow[rt] + kc + 16 * (u0 + u));
#pragma unroll
                    for (int ct = 0; ct < NCT; ct++)
                        b[u][ct] = TILED ? *reinterpret_cast<const float4*>(weight_tiles + ((size_t)(ct * (H / 4) + (kc / 4 + 4 * (u0 + u) + g)) * 16 + i) * 4)
                                         : *reinterpret_cast<const float4*>(wrow[ct] + kc + 16 * (u0 + u));
                }
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int u = 0; u < UB; u++) {
#pragma unroll
                        for (int ct = 0; ct < NCT; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][u].x, b[u][ct].x, acc[rt][ct], 0, 0, 0);
#pragma unroll
                        for (int ct = 0; ct < NCT; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][u].y, b[u][ct].y, acc[rt][ct], 0, 0, 0);
#pragma unroll
                        for (int ct = 0; ct < NCT; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][u].z, b[u][ct].z, acc[rt][ct], 0, 0, 0);
#pragma unroll
                        for (int ct = 0; ct < NCT; ct++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][u].w, b[u][ct].w, acc[rt][ct], 0, 0, 0);
                    }
            }
        }
        MMS_HEAD_ST(2);                                                      // operands loaded, MFMAs issued
        // C/D map: col = lane & 15, row = 4 (lane >> 4) + reg
        float* mine = s_part + (size_t)wave * ROWS * AP;
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) {
#pragma unroll
                for (int r = 0; r < 4; r++) mine[(16 * rt + 4 * g + r) * AP + ct * 16 + i] = acc[rt][ct][r];
            }
    }
    __syncthreads();
    MMS_HEAD_ST(3);
    float* s_mean = s_part + (size_t)WAVES * ROWS * AP;
    if (active)
        for (int e = tid; e < ROWS * AP; e += 64 * WAVES) {
            const int j = e % AP;
            float sum = s_part[e];
#pragma unroll
            for (int w = 1; w < WAVES; w++) sum += s_part[w * ROWS * AP + e];    // wave order
            s_mean[e] = sum + bias[j < A ? j : 0];
        }
    __syncthreads();
    MMS_HEAD_ST(4);
    if (active) {
#pragma unroll
        for (int q = 0; q < RPW; q++) {
            const int r = wave * RPW + q;
            const int64_t row = r0 + r;
            if (row >= N) continue;
            ppo_sample_row(s_mean + r * AP, value, log_std, seed, counters, row_offset, ref_scale, o, row, A, lane, vhidden != nullptr, v_rows[q]);
        }
    }
#if MMS_HEAD_STAMP
    MMS_HEAD_ST(5);
    if (r0 == 0 && tid == 0 && o.sigma_slot)
        for (int i = 1; i < 6; i++) o.sigma_slot[i] = 0.01f * (float)(st[i] - st[0]);
#endif
}

}  // namespace mms
