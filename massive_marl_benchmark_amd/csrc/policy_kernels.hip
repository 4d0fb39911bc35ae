// policy_kernels.hip -- the hidden layers of the PPO policy (agents/algorithms/rl/ppo/module.py:27-52: nn.Linear + activation,
// actor and critic of the same shape) as ONE launch per layer for both networks: Y = ELU(X W^T + b) on the fp32 matrix cores
// with bias and activation in the epilogue.
//
// This IS a dense contraction, so it belongs on MFMA: v_mfma_f32_32x32x2_f32 (exact fp32 products and sums; the reference's
// networks are fp32).  Tiling for gfx950: a 256-thread block owns a 128 x 128 output tile, each of its four waves a 64 x 64
// quarter as 2 x 2 MFMA tiles (64 accumulator registers); K is walked in slices of 32 through LDS, double buffered -- the
// global loads of slice k+1 are in flight while the 64 MFMAs of slice k issue (4096 cycles per wave), one barrier per slice.
// Both networks (blockIdx.z) and all tiles of a layer make 512 blocks at M = 4096, N = 1024: two resident blocks per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mms_lane.h"
#include "policy_args.h"

namespace mms {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTM = 128, kTN = 128;

// epilogue activation (uniform per launch): 0 identity, 1 ELU (alpha = 1, torch.nn.ELU), 2 ReLU, 3 tanh
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return (v > 0.f) ? v : (expf(v) - 1.f);
    if (act == 2) return fmaxf(v, 0.f);
    if (act == 3) return 1.f - 2.f / (__expf(2.f * v) + 1.f);     // tanh, absolute error ~1e-7; a few instructions where it is inlined 64 times
    return v;
}


// LDS layout: both operand tiles row-major, [128 rows][32 k + 4 pad] -- exactly how they arrive from HBM, so staging is a plain
// 16-B store per 16-B load.  The MFMA wants lane (i, h) to supply A[i][k] for ONE k per instruction; which k it supplies in
// which instruction is free as long as the B operand uses the same map, because the k of a dot product may be summed in any
// order.  So lane (i, h) reads the four consecutive k = 8c + 4h .. 8c + 4h + 3 of its row as ONE ds_read_b128 and feeds them
// to four consecutive MFMAs: 4 LDS reads per 16 MFMAs.  Row pitch 36 floats: eight consecutive rows start in eight different
// 16-B bank groups, so the 128-bit reads and writes are conflict free.
constexpr int kBK = 32, kPitch = kBK + 4;
constexpr size_t kLinearLds = (size_t)2 * 2 * 128 * kPitch * sizeof(float);     // two operands, two buffers: 73.7 KB

// MI = 2: 128 x 128 output tile, each of the four waves a 64 x 64 quarter; MI = 1: 64 x 128 tile, each wave 32 x 64 -- twice
// the blocks for problems whose 128-row tiling would leave CUs idle (the 8192 x 256 layers of the DDPG / TD3 actor).
template <int MI>
__global__ void __launch_bounds__(256, 2) linear_act_kernel(LinearArgs a) {
    constexpr int TM = 64 * MI;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.z;
    const float* __restrict__ X = a.x[g];
    const float* __restrict__ W = a.w[g];
    const float* __restrict__ Bv = a.b[g];
    float* __restrict__ Y = a.y[g];
    const int M = a.M, N = a.N, K = a.K;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * kTN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = (wave >> 1) * (32 * MI), wc = (wave & 1) * 64; // this wave's quarter of the tile
    const int li = lane & 31, lh = lane >> 5;
    float* As = smem;                                       // [2][128][kPitch]
    float* Bs = smem + 2 * 128 * kPitch;
    // staging: 128 rows x 8 float4 per operand and slice = 4 float4 per thread: rows sr + 32 j, k offset sk
    const int sr = t >> 3, sk = (t & 7) * 4;

    // two register sets: while slice kt is multiplied out of LDS, slice kt+1 waits in one set and slice kt+2 is in flight into
    // the other -- one slice of MFMAs (1.7 us) is not always enough to cover an HBM round trip under load
    float4 ra[2][4], rb[2][4];
    auto load_slice = [&](int kt, int set) {
        const int k = kt * kBK + sk;                        // K % 4 == 0: k < K implies k + 3 < K
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int row = m0 + sr + 32 * j, col = n0 + sr + 32 * j;
            if (j < 2 * MI) ra[set][j] = (row < M && k < K) ? *reinterpret_cast<const float4*>(X + (size_t)row * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[set][j] = (col < N && k < K) ? *reinterpret_cast<const float4*>(W + (size_t)col * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_slice = [&](int buf, int set) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < 2 * MI) *reinterpret_cast<float4*>(As + ((size_t)buf * 128 + sr + 32 * j) * kPitch + sk) = ra[set][j];
            *reinterpret_cast<float4*>(Bs + ((size_t)buf * 128 + sr + 32 * j) * kPitch + sk) = rb[set][j];
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    auto multiply_slice = [&](int buf) {
        const float* a_base = As + ((size_t)buf * 128 + wr + li) * kPitch + 4 * lh;
        const float* b_base = Bs + ((size_t)buf * 128 + wc + li) * kPitch + 4 * lh;
#pragma unroll
        for (int half = 0; half < 2; half++) {              // 16 k at a time: 8 fragment reads (32 registers), then 32 MFMAs
            float4 af[2][2], bf[2][2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int ko = 16 * half + 8 * c;
                af[c][0] = *reinterpret_cast<const float4*>(a_base + ko);
                if (MI == 2) af[c][1] = *reinterpret_cast<const float4*>(a_base + 32 * kPitch + ko);
                bf[c][0] = *reinterpret_cast<const float4*>(b_base + ko);
                bf[c][1] = *reinterpret_cast<const float4*>(b_base + 32 * kPitch + ko);
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the machine scheduler from sinking the reads back to their uses
#pragma unroll
            for (int c = 0; c < 2; c++) {
#define MMS_STEP(e)                                                                                          \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0].e, bf[c][0].e, acc[0][0], 0, 0, 0);            \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0].e, bf[c][1].e, acc[0][1], 0, 0, 0);            \
    if (MI == 2) {                                                                                           \
        acc[MI - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1].e, bf[c][0].e, acc[MI - 1][0], 0, 0, 0); \
        acc[MI - 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1].e, bf[c][1].e, acc[MI - 1][1], 0, 0, 0); \
    }
                MMS_STEP(x) MMS_STEP(y) MMS_STEP(z) MMS_STEP(w)
#undef MMS_STEP
            }
        }
    };

    const int nkt = (K + kBK - 1) / kBK;
    load_slice(0, 0);
    store_slice(0, 0);
    if (nkt > 1) load_slice(1, 1);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {                   // two slices per trip so that the register sets alternate by name
        if (kt + 2 < nkt) load_slice(kt + 2, 0);
        multiply_slice(0);
        if (kt + 1 < nkt) store_slice(1, 1);
        __syncthreads();
        if (kt + 1 < nkt) {
            if (kt + 3 < nkt) load_slice(kt + 3, 1);
            multiply_slice(1);
            if (kt + 2 < nkt) store_slice(0, 0);
            __syncthreads();
        }
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int col = n0 + wc + 32 * j + li;
        const float bias = (col < N) ? Bv[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r] + bias;
                v = apply_act(v, a.act);
                if (row < M && col < N) Y[(size_t)row * N + col] = v;
            }
        }
    }
}

// Fast path for M % 128 == 0 and N % 128 == 0 (all three hidden layers at 4096 envs): the same tiling with the K
// loop written as a software pipeline in program order.  One wave issues in order, so whatever is not an MFMA has to sit BETWEEN
// MFMAs (each keeps the matrix pipe busy for 64 cycles) instead of in front of them: per slice of 32 k = four groups of 16 MFMAs,
//   group 0: fragment reads of group 1,            16 MFMAs with the 8 global loads of slice kt+2 between them
//   group 1: fragment reads of group 2,            16 MFMAs with the 8 LDS stores of slice kt+1 between them
//   group 2: fragment reads of group 3,            16 MFMAs, then the slice barrier
//   group 3: fragment reads of group 0 of kt+1,    16 MFMAs
// sched_barrier(0) pins that order against the machine scheduler, which otherwise sinks every read next to its first use.
// Each group of MFMAs is issued at raised wave priority (s_setprio): of the two waves a SIMD holds, the one with matrix work ready goes
// first and the other's loads and LDS traffic fill in behind it (rollout step 334 -> 330 us).
#define MMS_MFMA4(S, e)                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S][0].e, fb[S][0].e, acc00, 0, 0, 0);                     \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S][0].e, fb[S][1].e, acc01, 0, 0, 0);                     \
    if (MI == 2) {                                                                                           \
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S][1].e, fb[S][0].e, acc10, 0, 0, 0);                 \
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S][1].e, fb[S][1].e, acc11, 0, 0, 0);                 \
    }                                                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);
#define MMS_FRAGS(S, abase, bbase, ko)                                                                       \
    fa[S][0] = *reinterpret_cast<const float4*>((abase) + (ko));                                             \
    if (MI == 2) fa[S][1] = *reinterpret_cast<const float4*>((abase) + 32 * kPitch + (ko));                  \
    fb[S][0] = *reinterpret_cast<const float4*>((bbase) + (ko));                                             \
    fb[S][1] = *reinterpret_cast<const float4*>((bbase) + 32 * kPitch + (ko));                               \
    __builtin_amdgcn_sched_barrier(0);

// TAIL: K is any multiple of 4 (the 388-wide first layer): K is walked to the next multiple of 32, a float4 whose k lies beyond K
// is read from the last valid float4 of its row instead and multiplied by zero (no branch in the pipeline).
// ELU_ONLY: the PPO policy's activation compiled in (one compare per output element instead of the run-time dispatch).
// MI = 2: 128 x 128 tile per block (64 x 64 per wave); MI = 1: 64 x 128 (32 x 64 per wave), for launches whose 128-row tiling
// would put one block on a CU: two waves per SIMD are what keeps the matrix pipe fed across barriers and fragment reads.
// LN (grouped MARL inference, MI = 2 and ELU only): bit 0 = leave row statistics of the output for the LayerNorm behind this layer
// (LinearArgs::part_out), bit 1 = the LayerNorm in front of this layer is folded into it (LinearArgs::stat_in, s).
template <bool TAIL, bool ELU_ONLY, int MI, int LN = 0>
__global__ void __launch_bounds__(256, 2) linear_act_fast_kernel(LinearArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = a.N, K = a.K;
    // 1-D grid, XCD-aware: workgroups go to the 8 XCDs round robin by their linear id, so id % 8 names the L2 a block sits behind.
    // Each XCD gets a contiguous eighth of the row panels and walks all column tiles of it: per network it reads its own 1/8 of x and
    // all of w (12 MB at K = 1024) instead of one column tile of w and ALL of x (33 MB) as the (n, m, network) grid order would have it.
    const int tiles_n = N / kTN, tiles_m = a.M / (64 * MI), per_net = tiles_n * tiles_m;
    const int g = blockIdx.x / per_net, l = blockIdx.x - g * per_net;
    int tm, tn;
    if ((tiles_m & 7) == 0) {
        const int xcd = l & 7, q = l >> 3;
        tm = xcd * (tiles_m >> 3) + q / tiles_n;
        tn = q % tiles_n;
    } else {
        tm = l / tiles_n;
        tn = l - tm * tiles_n;
    }
    const float* __restrict__ X = a.x[g];
    const float* __restrict__ W = a.w[g];
    const float* __restrict__ Bv = a.b[g];
    float* __restrict__ Y = a.y[g];
    const int m0 = tm * (64 * MI), n0 = tn * kTN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = (wave >> 1) * (32 * MI), wc = (wave & 1) * 64;
    const int li = lane & 31, lh = lane >> 5;
    float* As = smem;                                       // [2][128][kPitch]
    float* Bs = smem + 2 * 128 * kPitch;
    const int sr = t >> 3, sk = (t & 7) * 4;
    const float* xg = X + (size_t)(m0 + sr) * K + sk;       // this thread's staging source: rows + 32 j, advancing by 32 k per slice
    const float* wg = W + (size_t)(n0 + sr) * K + sk;
    const size_t rs = (size_t)32 * K;
    float* as_st = As + (size_t)sr * kPitch + sk;           // ... and its staging destination (buffer 0)
    float* bs_st = Bs + (size_t)sr * kPitch + sk;
    const float* a_fr = As + (size_t)(wr + li) * kPitch + 4 * lh;   // this lane's fragment rows (buffer 0)
    const float* b_fr = Bs + (size_t)(wc + li) * kPitch + 4 * lh;
    constexpr int kBuf = 128 * kPitch;                      // floats per buffer

    f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    float4 ra0_0, ra0_1, ra0_2 = {}, ra0_3 = {}, rb0_0, rb0_1, rb0_2, rb0_3;   // staged slices: set 0 = even slices,
    float4 ra1_0, ra1_1, ra1_2 = {}, ra1_3 = {}, rb1_0, rb1_1, rb1_2, rb1_3;   // set 1 = odd slices (named scalars: arrays of them land in scratch)
    float4 fa[2][2], fb[2][2];                              // fragment sets, alternating per group

    const int nkt = (K + kBK - 1) / kBK;
    const int k_last = K - 4 - sk;                          // (TAIL) offset of this thread's column group clamped to the last valid float4
#define MMS_LD(SET, J, KOFF)                                                                                                    \
    {                                                                                                                            \
        const bool ok = !TAIL || (KOFF) + sk < K;                                                                                \
        const int ko = ok ? (KOFF) : k_last;                                                                                     \
        const float m = ok ? 1.f : 0.f;                                                                                          \
        if ((J) < 2 * MI) ra##SET##_##J = *reinterpret_cast<const float4*>(xg + (J) * rs + ko);                                  \
        rb##SET##_##J = *reinterpret_cast<const float4*>(wg + (J) * rs + ko);                                                    \
        if (TAIL) {                                                                                                              \
            if ((J) < 2 * MI) { ra##SET##_##J.x *= m; ra##SET##_##J.y *= m; ra##SET##_##J.z *= m; ra##SET##_##J.w *= m; }        \
            rb##SET##_##J.x *= m; rb##SET##_##J.y *= m; rb##SET##_##J.z *= m; rb##SET##_##J.w *= m;                              \
        }                                                                                                                        \
    }
#define MMS_ST(SET, J, BOFF) if ((J) < 2 * MI) *reinterpret_cast<float4*>(as_st + (BOFF) + (J) * 32 * kPitch) = ra##SET##_##J; *reinterpret_cast<float4*>(bs_st + (BOFF) + (J) * 32 * kPitch) = rb##SET##_##J;
    MMS_LD(0, 0, 0) MMS_LD(0, 1, 0) MMS_LD(0, 2, 0) MMS_LD(0, 3, 0)
    { const int k1 = nkt > 1 ? kBK : 0; MMS_LD(1, 0, k1) MMS_LD(1, 1, k1) MMS_LD(1, 2, k1) MMS_LD(1, 3, k1) }
    MMS_ST(0, 0, 0) MMS_ST(0, 1, 0) MMS_ST(0, 2, 0) MMS_ST(0, 3, 0)
    __syncthreads();
    MMS_FRAGS(0, a_fr, b_fr, 0)

    // one slice: computes from buffer BUF, stores staged set ST (slice kt+1) into the other buffer, loads slice kt+2 into set LD
#define MMS_SLICE(BUF, ST, LD, KT)                                                                                              \
    {                                                                                                                            \
        const float* ab = a_fr + (BUF) * kBuf;                                                                                   \
        const float* bb = b_fr + (BUF) * kBuf;                                                                                   \
        const int kl = ((KT) + 2 < nkt ? (KT) + 2 : nkt - 1) * kBK;      /* past the end: reload the last slice, never stored */ \
        /* (TAIL) the last slice holds K - 32 (nkt - 1) valid k: only the 8-k groups that contain any are multiplied -- the     \
           388-wide first layer ends with 4 valid k, i.e. 16 MFMAs instead of 64 (the skipped ones would add exact zeros) */    \
        const int ng = (TAIL && (KT) == nkt - 1) ? g_last : 4;                                                                   \
        MMS_FRAGS(1, ab, bb, 8)                                                                                                  \
        MMS_MFMA4(0, x)                                                                                                          \
        MMS_LD(LD, 0, kl)            \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        MMS_MFMA4(0, y)                                                                                                          \
        MMS_LD(LD, 1, kl) \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        MMS_MFMA4(0, z)                                                                                                          \
        MMS_LD(LD, 2, kl) \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        MMS_MFMA4(0, w)                                                                                                          \
        MMS_LD(LD, 3, kl) \
        __builtin_amdgcn_sched_barrier(0);                                                                                       \
        if (ng > 1) {                                                                                                            \
            MMS_FRAGS(0, ab, bb, 16)                                                                                             \
            MMS_MFMA4(1, x)                                                                                                      \
            MMS_ST(ST, 0, (1 - (BUF)) * kBuf) \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            MMS_MFMA4(1, y)                                                                                                      \
            MMS_ST(ST, 1, (1 - (BUF)) * kBuf) \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            MMS_MFMA4(1, z)                                                                                                      \
            MMS_ST(ST, 2, (1 - (BUF)) * kBuf) \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            MMS_MFMA4(1, w)                                                                                                      \
            MMS_ST(ST, 3, (1 - (BUF)) * kBuf) \
            __builtin_amdgcn_sched_barrier(0);                                                                                   \
            if (ng > 2) {                                                                                                        \
                MMS_FRAGS(1, ab, bb, 24)                                                                                         \
                MMS_MFMA4(0, x) MMS_MFMA4(0, y) MMS_MFMA4(0, z) MMS_MFMA4(0, w)                                                  \
                __syncthreads();                  /* (ng is block-uniform: every wave of the block takes the same path) */      \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                MMS_FRAGS(0, a_fr + (1 - (BUF)) * kBuf, b_fr + (1 - (BUF)) * kBuf, 0)                                            \
                if (ng > 3) { MMS_MFMA4(1, x) MMS_MFMA4(1, y) MMS_MFMA4(1, z) MMS_MFMA4(1, w) }                                  \
            }                                                                                                                    \
        }                                                                                                                        \
    }

    const int g_last = (K - (nkt - 1) * kBK + 7) >> 3;      // 8-k groups of the last slice that hold valid k (1..4)
    for (int kt = 0; kt < nkt; kt += 2) {                   // two slices per trip so that the register sets alternate by name
        MMS_SLICE(0, 1, 0, kt)
        if (kt + 1 >= nkt) break;
        MMS_SLICE(1, 0, 1, kt + 1)
    }
#undef MMS_SLICE
#undef MMS_LD
#undef MMS_ST

    // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#define MMS_EPI(ACC, I, J)                                                                                   \
    {                                                                                                        \
        const int col = n0 + wc + 32 * (J) + li;                                                             \
        const float bias = Bv[col];                                                                          \
        float* yp = Y + (size_t)(m0 + wr + 32 * (I) + 4 * lh) * N + col;                                     \
        _Pragma("unroll") for (int r = 0; r < 16; r++) {                                                     \
            float v = ACC[r] + bias;                                                                         \
            v = ELU_ONLY ? ((v > 0.f) ? v : (expf(v) - 1.f)) : apply_act(v, a.act);                          \
            yp[(size_t)((r & 3) + 8 * (r >> 2)) * N] = v;                                                    \
        }                                                                                                    \
    }
    if constexpr (LN == 0) {
        MMS_EPI(acc00, 0, 0) MMS_EPI(acc01, 0, 1)
        if (MI == 2) { MMS_EPI(acc10, 1, 0) MMS_EPI(acc11, 1, 1) }
    } else {
        static_assert(LN == 0 || (MI == 2 && ELU_ONLY), "the LayerNorm folds exist for the 128 x 128 ELU tiling only");
        const int col0 = n0 + wc + li, col1 = col0 + 32;
        const float bias0 = Bv[col0], bias1 = Bv[col1];
        float s0 = 0.f, s1 = 0.f;
        if (LN & 2) { s0 = a.s[g][col0]; s1 = a.s[g][col1]; }
        const float2* stat = reinterpret_cast<const float2*>(a.stat_in[g]);
        float2* part = reinterpret_cast<float2*>(a.part_out[g]) + (size_t)(2 * tn + (wave & 1)) * a.M;
#define MMS_EPI_LN(ACC0, ACC1, I)                                                                                               \
        {                                                                                                                        \
            const int rbase = m0 + wr + 32 * (I) + 4 * lh;                                                                       \
            float ps[16], pq[16];                                                                                                \
            _Pragma("unroll") for (int r = 0; r < 16; r++) {                                                                     \
                const int row = rbase + (r & 3) + 8 * (r >> 2);                                                                  \
                float v0 = ACC0[r], v1 = ACC1[r];                                                                                \
                if (LN & 2) {                                                                                                    \
                    const float2 st = stat[row];                      /* (mean, rstd): the same address across the 32 lanes */  \
                    v0 = st.y * (v0 - st.x * s0);                                                                                \
                    v1 = st.y * (v1 - st.x * s1);                                                                                \
                }                                                                                                                \
                v0 += bias0; v1 += bias1;                                                                                        \
                v0 = (v0 > 0.f) ? v0 : (expf(v0) - 1.f);                                                                         \
                v1 = (v1 > 0.f) ? v1 : (expf(v1) - 1.f);                                                                         \
                Y[(size_t)row * N + col0] = v0;                                                                                  \
                Y[(size_t)row * N + col1] = v1;                                                                                  \
                ps[r] = v0 + v1;                                                                                                 \
                pq[r] = v0 * v0 + v1 * v1;                                                                                       \
            }                                                                                                                    \
            if (LN & 1) {                                                                                                        \
                /* sums over the wave's 64 columns: a halving butterfly over the 32 lanes of a half (ds_swizzle, xor masks as   \
                   immediates): each step a lane keeps half of its values and adds the partner's, so lane li ends with row      \
                   r = li >> 1 of its sixteen */                                                                                 \
                MMS_BFLY(ps, pq, 8, 16) MMS_BFLY(ps, pq, 4, 8) MMS_BFLY(ps, pq, 2, 4) MMS_BFLY(ps, pq, 1, 2)                     \
                ps[0] += MMS_SWZ(ps[0], 1);                                                                                      \
                pq[0] += MMS_SWZ(pq[0], 1);                                                                                      \
                const int r = li >> 1;                                                                                           \
                if ((li & 1) == 0) part[rbase + (r & 3) + 8 * (r >> 2)] = make_float2(ps[0], pq[0]);                              \
            }                                                                                                                    \
        }
#define MMS_SWZ(v, mask) __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x1F | ((mask) << 10)))
#define MMS_BFLY(A, B, HALF, MASK)                                                                                              \
        {                                                                                                                        \
            const bool up = (li & (MASK)) != 0;                                                                                  \
            _Pragma("unroll") for (int i = 0; i < (HALF); i++) {                                                                 \
                const float ka = up ? A[(HALF) + i] : A[i], sa = up ? A[i] : A[(HALF) + i];                                      \
                const float kb = up ? B[(HALF) + i] : B[i], sb = up ? B[i] : B[(HALF) + i];                                      \
                A[i] = ka + MMS_SWZ(sa, MASK);                                                                                   \
                B[i] = kb + MMS_SWZ(sb, MASK);                                                                                   \
            }                                                                                                                    \
        }
        MMS_EPI_LN(acc00, acc01, 0)
        MMS_EPI_LN(acc10, acc11, 1)
#undef MMS_EPI_LN
#undef MMS_BFLY
#undef MMS_SWZ
    }
#undef MMS_EPI
}
#undef MMS_MFMA4
#undef MMS_FRAGS

// More than 64 KB of dynamic LDS needs an opt-in per kernel and per device; remembered so that it is asked for once.
static hipError_t allow_large_lds(const void* kernel, int slot) {
    static bool done[18][64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !done[slot][dev]) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) done[slot][dev] = true;
    }
    return hipSuccess;
}

hipError_t launch_linear_act(const LinearArgs& a, int groups, hipStream_t s) {
    if (a.M == 0 || a.N == 0) return hipSuccess;
    dim3 grid((a.N + kTN - 1) / kTN, (a.M + kTM - 1) / kTM, groups);
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void*>(linear_act_kernel<2>), 2); e != hipSuccess) return e;
    if (hipError_t e = allow_large_lds(reinterpret_cast<const void*>(linear_act_kernel<1>), 5); e != hipSuccess) return e;
    // 128-row tiles that would leave CUs without a block: 64-row tiles, twice the blocks.  (At exactly one block per CU -- the
    // policy's 512-wide layer -- both tilings measure the same, 72-78 us: the fixed cost of a launch dominates, not the slice rate.)
    static const int small_max = getenv("MMS_LINEAR_SMALL_MAX") ? atoi(getenv("MMS_LINEAR_SMALL_MAX")) : 255;
    const bool ln_out = a.part_out[0] != nullptr, ln_in = a.stat_in[0] != nullptr;
    // (the LayerNorm folds exist for the 128-row tiling only: a small grid keeps it when they are asked for)
    const bool small = (size_t)grid.x * grid.y * grid.z <= (size_t)small_max && a.M > 64 && !getenv("MMS_LINEAR_TALL_TILES") && !ln_out && !ln_in;
    const bool fast = a.M % (small ? 64 : kTM) == 0 && a.N % kTN == 0 && a.K >= 8 && !getenv("MMS_LINEAR_GENERIC");
    if (small) grid.y = (a.M + 63) / 64;
#define MMS_LAUNCH_FAST(TAIL, ELU, MI, SLOT)                                                                                   \
    {                                                                                                                          \
        auto kern = linear_act_fast_kernel<TAIL, ELU, MI>;                                                                     \
        if (hipError_t e = allow_large_lds(reinterpret_cast<const void*>(kern), SLOT); e != hipSuccess) return e;              \
        hipLaunchKernelGGL(kern, dim3(grid.x * grid.y * grid.z), dim3(256), kLinearLds, s, a);                                 \
    }
#define MMS_LAUNCH_FAST_LN(TAIL, LNF, SLOT)                                                                                     \
    {                                                                                                                          \
        auto kern = linear_act_fast_kernel<TAIL, true, 2, LNF>;                                                                \
        if (hipError_t e = allow_large_lds(reinterpret_cast<const void*>(kern), SLOT); e != hipSuccess) return e;              \
        hipLaunchKernelGGL(kern, dim3(grid.x * grid.y * grid.z), dim3(256), kLinearLds, s, a);                                 \
    }
    const bool tail = a.K % kBK != 0, elu = a.act == 1;
    if (ln_out || ln_in) {                  // the LayerNorm folds: the callers (mms_linear_group_act) only ask for them where they exist
        if (!(fast && !small && elu)) return hipErrorInvalidValue;
        if (ln_in && ln_out && tail) MMS_LAUNCH_FAST_LN(true, 3, 16)
        else if (ln_in && tail) MMS_LAUNCH_FAST_LN(true, 2, 17)
        else if (ln_in && ln_out) MMS_LAUNCH_FAST_LN(false, 3, 12)
        else if (ln_in) MMS_LAUNCH_FAST_LN(false, 2, 13)
        else if (tail) MMS_LAUNCH_FAST_LN(true, 1, 14)
        else MMS_LAUNCH_FAST_LN(false, 1, 15)
        return hipGetLastError();
    }
    if (fast && !small) {
        if (!tail && elu) MMS_LAUNCH_FAST(false, true, 2, 3)
        else if (tail && elu) MMS_LAUNCH_FAST(true, true, 2, 4)
        else if (!tail) MMS_LAUNCH_FAST(false, false, 2, 6)
        else MMS_LAUNCH_FAST(true, false, 2, 7)
    } else if (fast) {
        if (!tail && elu) MMS_LAUNCH_FAST(false, true, 1, 8)
        else if (tail && elu) MMS_LAUNCH_FAST(true, true, 1, 9)
        else if (!tail) MMS_LAUNCH_FAST(false, false, 1, 10)
        else MMS_LAUNCH_FAST(true, false, 1, 11)
    } else if (small) hipLaunchKernelGGL(linear_act_kernel<1>, grid, dim3(256), kLinearLds, s, a);
    else hipLaunchKernelGGL(linear_act_kernel<2>, grid, dim3(256), kLinearLds, s, a);
#undef MMS_LAUNCH_FAST
#undef MMS_LAUNCH_FAST_LN
    return hipGetLastError();
}

// ---- MAPPO / HAPPO policy inference for all agents' networks --------------------------------------
// The reference's collect step (agents/algorithms/marl/runner.py:186-216) walks the agents and, per agent, an Actor and a Critic
// (actor_critic.py:43-69, 137-155): feature LayerNorm, then layer_N + 1 blocks of Linear + ELU + LayerNorm (utils/mlp.py:5-65),
// then the DiagGaussian head (utils/distributions.py:94-117) / v_out -- ~30 small launches per agent.  Here: every layer of ALL
// networks is one grouped launch of the matrix-core layer kernel above (bias + ELU in its epilogue), the LayerNorms are one
// grouped pass over rows each (HBM / Infinity-Cache bound), and the last LayerNorm, the output layers and the Gaussian sampling
// are one kernel.

__device__ __forceinline__ float wave_all_sum(float x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

constexpr int kLnPerLane = 16;            // row widths up to 1024: element k of a row sits in lane k % 64, slot k / 64

// mean and 1 / sqrt(var + eps) of one row held lane-strided in v[0 : n) (two passes in registers: torch's LayerNorm statistics)
template <int PL = kLnPerLane>
__device__ __forceinline__ void row_stats(const float (&v)[PL], int K, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PL; i++) if (lane + 64 * i < K) s += v[i];
    mean = wave_all_sum(s) / (float)K;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PL; i++) if (lane + 64 * i < K) { const float d = v[i] - mean; q += d * d; }
    rstd = 1.0f / sqrtf(wave_all_sum(q) / (float)K + eps);
}

// one wave per row, four rows per block, blockIdx.y = network.  PL = register slots per lane: 16 for rows up to 1024 wide, 64 for
// rows up to 4096 (the 3808-wide centralised observation of the 100-ant swarm)
template <int PL>
__global__ void __launch_bounds__(256) layernorm_rows_kernel(LayerNormArgs a) {
    const int g = blockIdx.y, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const int K = a.K, Kp = a.Kp;
    const float* __restrict__ x = a.x[g] + row * a.x_pitch;
    const float* __restrict__ gamma = a.gamma[g];
    const float* __restrict__ beta = a.beta[g];
    float* y = a.y[g] + row * Kp;
    float v[PL];
#pragma unroll
    for (int i = 0; i < PL; i++) v[i] = (lane + 64 * i < K) ? x[lane + 64 * i] : 0.f;
    float mean, rstd;
    row_stats<PL>(v, K, lane, a.eps, mean, rstd);
    if (a.stats_only) {
        if (lane == 0) reinterpret_cast<float2*>(a.y[g])[row] = make_float2(mean, rstd);
        return;
    }
#pragma unroll
    for (int i = 0; i < PL; i++) {
        const int k = lane + 64 * i;
        if (k < K) y[k] = (v[i] - mean) * rstd * gamma[k] + beta[k];
        else if (k < Kp) y[k] = 0.f;
    }
}

// kHeadRows rows per wave: the output layer's weights ([A, H], at most 64 KB) are staged in LDS once per block and reused by its
// 4 x kHeadRows rows (read per row straight from L2 they cost 18 KB per row: 202 us per launch against 60 us now, twenty networks).
// (ROWS = 2 for small launches -- the PPO bootstrap value head, one network x 4096 rows: 128 blocks of 32 rows left half the CUs idle,
//  18.9 us; with 8 rows per block every CU has two)
constexpr int kHeadRows = 8;
template <int ROWS>
__global__ void __launch_bounds__(256) marl_heads_kernel(HeadsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];          // [A][H]
    const int g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = a.H, A = a.A[g];
    const float* __restrict__ gamma = a.gamma[g];
    const float* __restrict__ beta = a.beta[g];
    for (int k = threadIdx.x; k < A * H; k += 256) s_w[k] = a.w[g][k];
    __syncthreads();
    float gm[kLnPerLane], bt[kLnPerLane];
#pragma unroll
    for (int i = 0; i < kLnPerLane; i++) {
        const int k = lane + 64 * i;
        gm[i] = (k < H) ? gamma[k] : 0.f;
        bt[i] = (k < H) ? beta[k] : 0.f;
    }
    const float bias = (lane < A) ? a.b[g][lane] : 0.f;
    const float sd = (a.std[g] && lane < A) ? a.std[g][lane] : 0.f;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * ROWS;
    for (int r = 0; r < ROWS; r++) {
        const int64_t row = row0 + r;
        if (row >= a.M) return;
        const float* __restrict__ h = a.h[g] + row * H;
        float v[kLnPerLane];
#pragma unroll
        for (int i = 0; i < kLnPerLane; i++) v[i] = (lane + 64 * i < H) ? h[lane + 64 * i] : 0.f;
        if (a.eps >= 0.f) {                             // (eps < 0: no LayerNorm in front of the output layer)
            float mean, rstd;
            row_stats(v, H, lane, a.eps, mean, rstd);
#pragma unroll
            for (int i = 0; i < kLnPerLane; i++) v[i] = (v[i] - mean) * rstd * gm[i] + bt[i];  // (slots past H: gm = bt = 0)
        }
        float mine = 0.f;                               // lane j keeps output j
        for (int j = 0; j < A; j++) {
            const float* wj = s_w + j * H;
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < kLnPerLane; i++) if (lane + 64 * i < H) p += v[i] * wj[lane + 64 * i];
            p = wave_all_sum(p);
            if (lane == j) mine = p;
        }
        mine += bias;
        float* out = a.out[g] + row * a.out_pitch[g];
        if (a.std[g] == nullptr) {                      // a value head (or a deterministic action): stored as is
            if (lane < A) out[lane] = mine;
            continue;
        }
        const int64_t c = a.counters[g] ? a.counters[g][row] : 0;
        if (lane < A) {
            const float z = rand_normal(a.seed + (uint64_t)g, (uint64_t)(a.row_offset + row), (uint64_t)c, (uint32_t)lane);
            out[lane] = mine + sd * z;                  // Normal.sample
            // FixedNormal.log_probs (distributions.py:31-34) is the PER-DIMENSION log-density: the reference keeps [M, A], no sum
            if (a.logp[g]) a.logp[g][row * a.out_pitch[g] + lane] = -0.5f * z * z - logf(sd) - 0.9189385332046727f;
        }
        if (lane == 0 && a.counters[g]) a.counters[g][row] = c + 1;
    }
}

// (mean, rstd) per row from the slot partials: thread = row, the slots summed in order
__global__ void __launch_bounds__(256) row_stats_kernel(RowStatsArgs a) {
    const int g = blockIdx.y;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.M) return;
    const float2* part = reinterpret_cast<const float2*>(a.part[g]);
    float sum = 0.f, sq = 0.f;
    for (int k = 0; k < a.slots; k++) {
        const float2 p = part[(size_t)k * a.M + row];
        sum += p.x;
        sq += p.y;
    }
    const float inv = 1.0f / (float)a.width;
    const float mean = sum * inv;
    const float var = fmaxf(sq * inv - mean * mean, 0.f);
    reinterpret_cast<float2*>(a.stat[g])[row] = make_float2(mean, 1.0f / sqrtf(var + a.eps));
}

// The same from slot partials in the two-pass form the split layers leave: (sum, M2 about the slot's own mean) of 64 activations per
// slot, combined with Chan's formula -- no E[x^2] - mean^2 cancellation however large the row's mean is against its spread.
__device__ __forceinline__ void chan_combine(const float2* part, int64_t M, int64_t row, int slots, float& mean, float& m2) {
    // (eight slots at a time with all loads issued before the first use: a hidden size of 512 is one round trip, not sixteen)
    float sum = 0.f;
    m2 = 0.f;
    if (slots <= 16) {
        float2 p[16];
#pragma unroll
        for (int k = 0; k < 16; k++) p[k] = (k < slots) ? part[(size_t)k * M + row] : make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 16; k++) sum += p[k].x;                     // (slots past the last add zero: the order of the real ones is unchanged)
        mean = sum / (64.f * (float)slots);
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < slots) {
                const float d = p[k].x * (1.f / 64.f) - mean;
                m2 += p[k].y + 64.f * d * d;
            }
        return;
    }
    for (int k = 0; k < slots; k++) sum += part[(size_t)k * M + row].x;
    mean = sum / (64.f * (float)slots);
    for (int k = 0; k < slots; k++) {
        const float2 p = part[(size_t)k * M + row];
        const float d = p.x * (1.f / 64.f) - mean;
        m2 += p.y + 64.f * d * d;
    }
}

__global__ void __launch_bounds__(256) row_stats_chan_kernel(RowStatsArgs a) {
    const int g = blockIdx.y;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.M) return;
    float mean, m2;
    chan_combine(reinterpret_cast<const float2*>(a.part[g]), a.M, row, a.slots, mean, m2);
    reinterpret_cast<float2*>(a.stat[g])[row] = make_float2(mean, 1.0f / sqrtf(m2 / (float)a.width + a.eps));
}

hipError_t launch_row_stats_chan(const RowStatsArgs& a, int groups, hipStream_t s) {
    if (a.M == 0 || groups == 0) return hipSuccess;
    hipLaunchKernelGGL(row_stats_chan_kernel, dim3((unsigned)((a.M + 255) / 256), groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

// thread = (row, network): finishes the output head of the last split layer (HeadsFinishArgs) and samples as marl_heads_kernel does
__global__ void __launch_bounds__(256) marl_heads_finish_kernel(HeadsFinishArgs a) {
    const int g = blockIdx.y;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.M) return;
    float mean, m2;
    chan_combine(reinterpret_cast<const float2*>(a.part[g]), a.M, row, a.slots, mean, m2);
    const float rstd = 1.0f / sqrtf(m2 / (float)a.width + a.eps);
    const int A = a.A[g];
    float* out = a.out[g] + row * a.out_pitch[g];
    const int64_t c = (a.std[g] && a.counters[g]) ? a.counters[g][row] : 0;
    float dots[16];                                                      // a row's partials of one slot are contiguous: 16-byte loads, coalesced over the rows
#pragma unroll
    for (int j = 0; j < 16; j++) dots[j] = 0.f;
    const float4* hp = reinterpret_cast<const float4*>(a.head_part[g]);
    const int HQ = (A + 3) >> 2;                                         // float4 pieces per row and slot (the partials' stride is A rounded up to 4)
#pragma unroll 8
    for (int k = 0; k < a.slots; k++) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (4 * q < A) {
                const float4 v = hp[((size_t)k * a.M + row) * HQ + q];
                dots[4 * q] += v.x; dots[4 * q + 1] += v.y; dots[4 * q + 2] += v.z; dots[4 * q + 3] += v.w;
            }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (j < A) {
            const float mu = rstd * (dots[j] - mean * a.hs[g][j]) + a.hc[g][j];
            if (a.std[g] == nullptr) {
                out[j] = mu;
            } else {
                const float sd = a.std[g][j];
                const float z = rand_normal(a.seed + (uint64_t)g, (uint64_t)(a.row_offset + row), (uint64_t)c, (uint32_t)j);
                out[j] = mu + sd * z;
                if (a.logp[g]) a.logp[g][row * a.out_pitch[g] + j] = -0.5f * z * z - logf(sd) - 0.9189385332046727f;
            }
        }
    }
    if (a.std[g] && a.counters[g]) a.counters[g][row] = c + 1;
}

hipError_t launch_marl_heads_finish(const HeadsFinishArgs& a, int groups, hipStream_t s) {
    if (a.M == 0 || groups == 0) return hipSuccess;
    hipLaunchKernelGGL(marl_heads_finish_kernel, dim3((unsigned)((a.M + 255) / 256), groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_row_stats(const RowStatsArgs& a, int groups, hipStream_t s) {
    if (a.M == 0 || groups == 0) return hipSuccess;
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((a.M + 255) / 256), groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_layernorm(const LayerNormArgs& a, int groups, hipStream_t s) {
    if (a.M == 0 || groups == 0) return hipSuccess;
    if (a.K <= 64 * kLnPerLane) hipLaunchKernelGGL(layernorm_rows_kernel<kLnPerLane>, dim3((unsigned)((a.M + 3) / 4), groups), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(layernorm_rows_kernel<64>, dim3((unsigned)((a.M + 3) / 4), groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_marl_heads(const HeadsArgs& a, int groups, hipStream_t s) {
    if (a.M == 0 || groups == 0) return hipSuccess;
    int amax = 1;
    for (int g = 0; g < groups; g++) amax = a.A[g] > amax ? a.A[g] : amax;
    const size_t lds = (size_t)amax * a.H * sizeof(float);                       // <= 16 x 1024 x 4 = 64 KB
    const bool small = (a.M + 4 * kHeadRows - 1) / (4 * kHeadRows) * groups < 512;      // fewer than two blocks per CU at 8 rows per wave
    const int64_t per_block = 4 * (small ? 2 : kHeadRows);
    const dim3 grid((unsigned)((a.M + per_block - 1) / per_block), groups);
    if (small) hipLaunchKernelGGL(marl_heads_kernel<2>, grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL(marl_heads_kernel<kHeadRows>, grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace mms
