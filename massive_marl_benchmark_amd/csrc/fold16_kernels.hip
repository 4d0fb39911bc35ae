// fold16_kernels.hip -- the weights' side of the grouped MAPPO / HAPPO inference, refreshed ON THE DEVICE after every update.
//
// Every layer of that pass sits behind an nn.LayerNorm whose affine part is folded into the layer (policy_inference.py):
//     W (LN(h) gamma + beta) + b  =  rstd (W~ h - mean s) + c,    W~ = W diag(gamma),  s = W~ 1,  c = W beta + b
// (reference: agents/algorithms/utils/mlp.py:19-27, 44-60 -- Linear, ELU, LayerNorm blocks behind a feature LayerNorm).  The layer
// kernels read W~ as two scaled fp16 planes (split16_kernels.hip) plus the vectors s and c, and store their outputs under a scale that
// puts the a-priori bound |W~ xhat + c| <= |W~ row|_2 sqrt(K) + |c| at 2^14.  Until round 4 these derived copies were rebuilt by ~100
// small torch launches per refresh (0.6 ms inside a captured rollout, 2.9 ms eagerly); here they are TWO kernels per refresh, writing
// into buffers whose addresses never change, with no host synchronisation -- so GroupedPolicyInference.refresh() can run at step 0 of
// every rollout (a trainer may update through `.data`, which no version counter sees: hatrpo_trainer.py:122) and inside a hipGraph.
//
//   fold_planes16_kernel   per network and row n:  W~[n, :], its planes + inverse row scale, s[n], c[n], and the row's output bound
//   fold_scales16_kernel   per network: the largest row bound -> the power of two of the output rows, written per row of the batch
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "policy_args.h"

namespace mms {

typedef _Float16 fold_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void fold_pow2_scale(float bound, float& scale, float& inv) {      // as pow2_scale of split16_kernels.hip
    int e = 14;
    if (bound > 0.f) (void)frexpf(bound, &e);
    int sh = 14 - e;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    scale = ldexpf(1.f, sh);
    inv = ldexpf(1.f, -sh);
}

// One wave per row (64 lanes walk the row's 8-element pieces, two passes as split16_planes_kernel), four rows per block,
// blockIdx.y = network.  Matrices of different shapes in one launch (rows_g / K_g per network).
template <bool ALIGNED>
__global__ void __launch_bounds__(256) fold_planes16_kernel(FoldPlanesArgs a) {
    const int g = blockIdx.y;
    const int K = a.K_g[g];
    const int64_t rows = a.rows_g[g];
    if (rows == 0) return;
    const int KC = (K + 31) / 32, pieces = KC * 4;
    const int lane = threadIdx.x & 63;
    const int64_t row_raw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = row_raw < rows;
    const int64_t row = live ? row_raw : rows - 1;
    const float* __restrict__ w = a.w[g] + row * (int64_t)K;
    const float* __restrict__ gam = a.gamma[g];
    const float* __restrict__ bet = a.beta[g];
    auto load8 = [&](const float* src, int p, float* v, float fill) {
        if (ALIGNED && p * 8 + 8 <= K) {                                 // (every row and vector 16-byte aligned: K a multiple of 4)
            const float4 q0 = *reinterpret_cast<const float4*>(src + p * 8), q1 = *reinterpret_cast<const float4*>(src + p * 8 + 4);
            v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (p * 8 + j < K) ? src[p * 8 + j] : fill;
        }
    };
    float big = 0.f, s = 0.f, c = 0.f, l2 = 0.f;
    for (int p = lane; p < pieces; p += 64) {
        float v[8], gm[8], bt[8];
        load8(w, p, v, 0.f);
        if (gam) load8(gam, p, gm, 0.f);
        if (bet) load8(bet, p, bt, 0.f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float wt = gam ? v[j] * gm[j] : v[j];
            big = fmaxf(big, fabsf(wt));
            s += wt;
            l2 += wt * wt;
            if (bet) c += v[j] * bt[j];
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        big = fmaxf(big, __shfl_xor(big, m, 64));
        s += __shfl_xor(s, m, 64);
        l2 += __shfl_xor(l2, m, 64);
        c += __shfl_xor(c, m, 64);
    }
    if (a.bias[g]) c += a.bias[g][row];
    float scale, inv;
    fold_pow2_scale(big, scale, inv);
    uint8_t* __restrict__ out = reinterpret_cast<uint8_t*>(a.planes[g]);
    float* __restrict__ wt_out = a.wt[g];
    for (int p = lane; p < pieces; p += 64) {
        float v[8], gm[8];
        load8(w, p, v, 0.f);
        if (gam) load8(gam, p, gm, 0.f);
        fold_f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float wt = gam ? v[j] * gm[j] : v[j];
            const float t = wt * scale;
            hi[j] = (_Float16)t;
            lo[j] = (_Float16)((t - (float)hi[j]) * 2048.f);
            if (!ALIGNED && wt_out && live && p * 8 + j < K) wt_out[row * (int64_t)K + p * 8 + j] = wt;
            v[j] = wt;
        }
        if (ALIGNED && wt_out && live) {
            float* dst = wt_out + row * (int64_t)K + p * 8;
            if (p * 8 + 8 <= K) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) if (p * 8 + j < K) dst[j] = v[j];
            }
        }
        if (out && live) {
            uint8_t* dst = out + (row * KC + (p >> 2)) * (int64_t)128 + (p & 3) * 16;
            *reinterpret_cast<fold_f16x8*>(dst) = hi;
            *reinterpret_cast<fold_f16x8*>(dst + 64) = lo;
        }
    }
    if (lane == 0 && live) {
        if (a.inv[g]) a.inv[g][row] = inv;
        if (a.s[g]) a.s[g][row] = s;
        if (a.c[g]) a.c[g][row] = c;
        if (a.rb[g]) a.rb[g][row] = sqrtf(l2) * sqrtf((float)K) + fabsf(c);
    }
}

hipError_t launch_fold_planes16(const FoldPlanesArgs& a, int groups, hipStream_t s) {
    if (groups == 0) return hipSuccess;
    unsigned blocks = 0;
    for (int g = 0; g < groups; g++) {
        const unsigned need = (unsigned)((a.rows_g[g] + 3) / 4);
        blocks = need > blocks ? need : blocks;
    }
    if (blocks == 0) return hipSuccess;
    bool aligned = true;
    for (int g = 0; g < groups; g++) {
        uintptr_t bits = reinterpret_cast<uintptr_t>(a.w[g]) | reinterpret_cast<uintptr_t>(a.gamma[g]) | reinterpret_cast<uintptr_t>(a.beta[g]) |
                         reinterpret_cast<uintptr_t>(a.wt[g]);
        aligned = aligned && (a.K_g[g] % 4) == 0 && (bits & 15) == 0;
    }
    if (aligned) hipLaunchKernelGGL(fold_planes16_kernel<true>, dim3(blocks, groups), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(fold_planes16_kernel<false>, dim3(blocks, groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

// scale_g = 2^(14 - e) with 1.001 max_n rb_g[n] <= 2^e (an all-zero network keeps 1): stored once (scale1[g], f32 [1], optional) and per
// row of the batch (ysc[g], yinv[g]: f32 [M] each -- what the layer kernel reads).  Every block re-reduces its network's row bounds
// (a few hundred floats) instead of waiting for a separate reduction launch.
__global__ void __launch_bounds__(256) fold_scales16_kernel(FoldScalesArgs a) {
    __shared__ float s_m[4];
    const int g = blockIdx.y;
    float m = 0.f;
    for (int i = threadIdx.x; i < a.n[g]; i += 256) m = fmaxf(m, a.rb[g][i]);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    float bound = m * 1.001f;
    if (!(bound > 1e-30f)) bound = 1e-30f;
    float sc, iv;
    fold_pow2_scale(bound, sc, iv);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.scale1[g]) a.scale1[g][0] = sc;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row < a.M) {
        if (a.ysc[g]) a.ysc[g][row] = sc;
        if (a.yinv[g]) a.yinv[g][row] = iv;
    }
}

hipError_t launch_fold_scales16(const FoldScalesArgs& a, int groups, hipStream_t s) {
    if (groups == 0) return hipSuccess;
    const unsigned blocks = a.M > 0 ? (unsigned)((a.M + 255) / 256) : 1u;
    hipLaunchKernelGGL(fold_scales16_kernel, dim3(blocks, groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace mms
