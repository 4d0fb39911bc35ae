// split_kernels.hip -- the policy layers Y = act(X W^T + b) with fp32 operands carried as THREE bf16 planes each.
//
// Why: the layers are fp32 (the reference's networks are, agents/algorithms/rl/ppo/module.py:27-52) and gfx950's fp32-input
// MFMA runs at the vector rate, 1/16 of the bf16 MFMA rate.  An fp32 number is EXACTLY the sum of three bf16 numbers
// (a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1): 3 x 8 significant bits and a sign each), so
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0) + [a1 b2 + a2 b1 + a2 b2 <= 2^-25 |a b|, dropped]
// is six v_mfma_f32_16x16x32_bf16 products accumulated in fp32: 16 / 6 of the fp32 MFMA rate, and FEWER roundings per output
// than the fp32 MFMA's k-ordered fma chain (the leading accumulator rounds once per 32 k instead of 32 times).  Measured against
// the float64 product (tools/microbench/split_bf16.hip, profiles/r03_split_bf16_microbench.txt; tests/test_gpu_parity.py
// ::test_split_layers_error): rms error 0.32 x and worst error 0.35-0.42 x the exact-fp32 MFMA kernel's on the same inputs, no
// mean error.  This is not narrower arithmetic; it is the same fp32 product evaluated on the bf16 pipe.
//
// Plane format "P32" (how split operands live in HBM): [rows][KC][3][32] bf16, KC = ceil(K / 32): the three planes of 32
// consecutive k of a row are 192 contiguous bytes, so the slice a block stages per k-step is one contiguous run per row.
// Columns past K are zero.  Weights are split once per optimizer step (mms_split_planes), activations by the epilogue of the
// layer that produces them, the observation by mms_split_planes.
//
// Tiling for gfx950: 512-thread block = 8 waves (two per SIMD: one's fragment reads and waits hide behind the other's MFMAs),
// ONE block per CU; output tile (64 MT) x 128 with MT = 4 (256 x 128, the 1024-wide layers at 4096 rows: 256 tiles = one per
// CU) or MT = 2 (128 x 128, when the larger tile would leave CUs idle); waves 4 (m) x 2 (n), each (16 MT) x 64 as MT x 4 MFMA
// tiles of 16 x 16.  The product is evaluated TRANSPOSED (W fragments as the A operand, X fragments as B), so a lane ends up
// with four consecutive n of one row m: the epilogue packs them into 8-byte plane pieces / one 16-byte fp32 piece.
// K walks in steps of 32 through LDS (double buffered, 72 KB per buffer at MT = 4), the next step's operands prefetched into
// registers at the top of a step and stored to LDS at its end; one barrier per step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "policy_args.h"

namespace mms {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kChunkBytes = 192;                 // one row's three planes of 32 k

// 16-byte slot swizzle of the LDS operand image: row r of a 16-row MFMA tile keeps k-group g (8 k = 16 B) at slot g ^ f(r).
// With 64-byte rows the four lanes groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) then hit 16 distinct 16-byte bank slots.
__device__ __forceinline__ int swz(int row, int g) { return g ^ ((4 - ((row >> 2) & 3)) & 3); }

// ---- fp32 [rows, K] -> P32 planes ------------------------------------------------------------------------------------------
// thread = (row, chunk, k-group of 8): reads 32 bytes, writes one 16-byte piece per plane
__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int64_t rows, int K,
                                                           int x_pitch, int KC) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)(id & 3);
    const int64_t rc = id >> 2;
    const int kc = (int)(rc % KC);
    const int64_t row = rc / KC;
    if (row >= rows) return;
    const int k0 = kc * 32 + g * 8;
    float v[8];
    const float* src = x + row * (int64_t)x_pitch + k0;
    if (k0 + 8 <= K) {
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (k0 + j < K) ? src[j] : 0.f;
    }
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float r = v[j];
        p0[j] = (__bf16)r; r -= (float)p0[j];
        p1[j] = (__bf16)r; r -= (float)p1[j];
        p2[j] = (__bf16)r;
    }
    uint8_t* dst = out + (row * KC + kc) * (int64_t)kChunkBytes + g * 16;
    *reinterpret_cast<bf16x8*>(dst) = p0;
    *reinterpret_cast<bf16x8*>(dst + 64) = p1;
    *reinterpret_cast<bf16x8*>(dst + 128) = p2;
}

hipError_t launch_split_planes(const float* x, void* planes, int64_t rows, int K, int x_pitch, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    const int KC = (K + 31) / 32;
    const int64_t threads = rows * KC * 4;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, reinterpret_cast<uint8_t*>(planes), rows,
                       K, x_pitch, KC);
    return hipGetLastError();
}

// ---- the layer -------------------------------------------------------------------------------------------------------------
template <int MT>
struct SplitGeom {
    static constexpr int TM = 64 * MT, TN = 128;
    static constexpr int XBYTES = TM * kChunkBytes, WBYTES = TN * kChunkBytes, BUF = XBYTES + WBYTES;     // rows of 192 bytes, X tile then W tile
    static constexpr int NDMA = BUF / 1024 / 8;                         // 1-KB LDS-DMA instructions per wave and k-step: 9 (MT = 4), 6 (MT = 2)
    static constexpr size_t LDS = 2 * (size_t)BUF;
};

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return (v > 0.f) ? v : (expf(v) - 1.f);
    if (act == 2) return fmaxf(v, 0.f);
    if (act == 3) return 1.f - 2.f / (__expf(2.f * v) + 1.f);
    return v;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MT, bool OUT_PLANES>
__global__ void __launch_bounds__(512, 2) linear_split_kernel(SplitLinearArgs a) {
    using G = SplitGeom<MT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int KC = a.KC, N = a.N;

    // tile of this block: XCD-aware (id % 8 = the XCD a block lands on): each XCD walks a contiguous run of (network, row panel, column tile)
    const int tiles_n = N / G::TN, tiles_m = a.M / G::TM;
    const int total = (int)gridDim.x;
    int L = blockIdx.x;
    if ((total & 7) == 0) L = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int tn = L % tiles_n;
    const int rest = L / tiles_n;
    const int tm = rest % tiles_m, gi = rest / tiles_m;
    const int m0 = tm * G::TM, n0 = tn * G::TN;

    const uint8_t* __restrict__ X = reinterpret_cast<const uint8_t*>(a.x[gi]);
    const uint8_t* __restrict__ W = reinterpret_cast<const uint8_t*>(a.w[gi]);
    const size_t pitch = (size_t)KC * kChunkBytes;

    // Operands go HBM -> LDS directly (global_load_lds_dwordx4: 64 lanes x 16 bytes land in 1 KB of contiguous LDS, the source address
    // is per lane).  The LDS image of a k-step is the tile's rows at pitch 192 (X rows, then W rows), each row [plane][k-group] with the
    // k-group slot swizzled by the row (swz): instruction c of a step fills image bytes [1024 c, 1024 c + 1024); lane s of it owns the
    // 16-byte slot S = 64 c + s = (row S / 12, plane (S % 12) / 4, slot S % 4) and fetches the k-group that belongs there.  In HBM that
    // is a run of 192 contiguous bytes per row.  Wave w issues c = w, w + 8, ...
    const uint8_t* gsrc[G::NDMA];
#pragma unroll
    for (int i = 0; i < G::NDMA; i++) {
        const int S = 64 * (wave + 8 * i) + lane;
        const int row = S / 12, q = S - row * 12, pl = q >> 2, kg = swz(row, q & 3);
        gsrc[i] = (row < G::TM ? X + (size_t)(m0 + row) * pitch : W + (size_t)(n0 + row - G::TM) * pitch) + pl * 64 + kg * 16;
    }
    auto dma_slice = [&](int kc, int buf) {
#pragma unroll
        for (int i = 0; i < G::NDMA; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[i] + (size_t)kc * kChunkBytes), (lptr_t)(lds + buf * G::BUF + (wave + 8 * i) * 1024), 16, 0, 0);
    };

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this lane's fragment addresses inside a buffer: row r16 of a 16-row tile, k-group g4 at its swizzled slot
    const int frag = r16 * kChunkBytes + swz(r16, g4) * 16;
    const int xfrag = (wm * 16 * MT) * kChunkBytes + frag;               // + plane * 64 + mt * 16 * 192
    const int wfrag = G::XBYTES + (wn * 64) * kChunkBytes + frag;        // + plane * 64 + nt * 16 * 192

    // Two accumulators per output: `acc` takes the leading products a0 b0, `lo` the five small ones (<= 2^-8 of the leading term), and
    // they are added once at the end.  The small products then meet an accumulator of their own size instead of being aligned to
    // (and truncated against) the large one at every step: error against float64 0.32 x the exact-fp32 MFMA chain's (a single
    // accumulator: 0.85 x, with a mean error of -5e-8 rms(Y) that this removes); no extra MFMAs, 64 more registers.
    constexpr bool kDmaBehindFirstTile = MT == 4;    // measured: 256 x 128 tiles 80.5 -> 78.5 us (K = 1024), 128 x 128 tiles 43.3 -> 46.6 us
    f32x4 lo[MT][4];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one k-step: W fragments of the wave's four n-tiles (kept for the step), then per m-tile its three X planes and 24 MFMAs;
    // the DMA of the next slice is issued behind the first m-tile's MFMAs (kDmaBehindFirstTile) or in front of everything
    auto step = [&](int buf, bool more, int kc_next) {
        const uint8_t* base = lds + buf * G::BUF;
        if (!kDmaBehindFirstTile && more) dma_slice(kc_next, buf ^ 1);
        bf16x8 wf[4][3];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) wf[nt][pl] = *reinterpret_cast<const bf16x8*>(base + wfrag + pl * 64 + nt * 16 * kChunkBytes);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            bf16x8 xf[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) xf[pl] = *reinterpret_cast<const bf16x8*>(base + xfrag + pl * 64 + mt * 16 * kChunkBytes);
            // small terms first; product-major so that consecutive MFMAs go to different accumulators
#define MMS_P(PW, PX, ACC)                                                                                                     \
    _Pragma("unroll") for (int nt = 0; nt < 4; nt++)                                                                            \
        ACC[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][PW], xf[PX], ACC[mt][nt], 0, 0, 0);
            MMS_P(2, 0, lo) MMS_P(0, 2, lo) MMS_P(1, 1, lo) MMS_P(1, 0, lo) MMS_P(0, 1, lo) MMS_P(0, 0, acc)
#undef MMS_P
            if (kDmaBehindFirstTile && mt == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) dma_slice(kc_next, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    dma_slice(0, 0);
    for (int kt = 0; kt < KC; kt++) {
        __syncthreads();                                                // slice kt has landed (vmcnt(0) + barrier); everyone is done with slice kt - 1
        step(kt & 1, kt + 1 < KC, kt + 1);
    }
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] += lo[i][j];
    __syncthreads();                                                    // the operand buffers become the waves' epilogue scratch

    // epilogue: lane holds, for tile (mt, nt), rows n = nbase + 16 nt + 4 g4 + reg (reg = 0..3) of column m = mbase + 16 mt + r16
    const float* __restrict__ Bv = a.b[gi];
    const int mbase = m0 + wm * 16 * MT, nbase = n0 + wn * 64;
    float4 bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) bias[nt] = *reinterpret_cast<const float4*>(Bv + nbase + 16 * nt + 4 * g4);
    const int act = a.act;
    if constexpr (OUT_PLANES) {
        constexpr int RS = 2 * kChunkBytes + 16;                         // scratch row: this wave's two chunks (64 n) of one m, padded
        uint8_t* scr = lds + wave * (16 * RS);
        uint8_t* __restrict__ Y = reinterpret_cast<uint8_t*>(a.y[gi]);
        const size_t ypitch = (size_t)(N / 32) * kChunkBytes;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const float bb[4] = {bias[nt].x, bias[nt].y, bias[nt].z, bias[nt].w};
                bf16x4 p0, p1, p2;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = act_apply(acc[mt][nt][r] + bb[r], act);
                    p0[r] = (__bf16)v; v -= (float)p0[r];
                    p1[r] = (__bf16)v; v -= (float)p1[r];
                    p2[r] = (__bf16)v;
                }
                uint8_t* d = scr + r16 * RS + (nt >> 1) * kChunkBytes + (nt & 1) * 32 + g4 * 8;
                *reinterpret_cast<bf16x4*>(d) = p0;
                *reinterpret_cast<bf16x4*>(d + 64) = p1;
                *reinterpret_cast<bf16x4*>(d + 128) = p2;
            }
            // 16 rows x 384 bytes back out as 16-byte pieces: 24 per row, contiguous in HBM
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int idx = lane + 64 * j, row = idx / 24, off = (idx - row * 24) * 16;
                const uint4 d = *reinterpret_cast<const uint4*>(scr + row * RS + off);
                *reinterpret_cast<uint4*>(Y + (size_t)(mbase + 16 * mt + row) * ypitch + (size_t)(nbase / 32) * kChunkBytes + off) = d;
            }
        }
    } else {
        constexpr int RS = 256 + 16;
        uint8_t* scr = lds + wave * (16 * RS);
        float* __restrict__ Y = reinterpret_cast<float*>(a.y[gi]);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                float4 v;
                v.x = act_apply(acc[mt][nt][0] + bias[nt].x, act);
                v.y = act_apply(acc[mt][nt][1] + bias[nt].y, act);
                v.z = act_apply(acc[mt][nt][2] + bias[nt].z, act);
                v.w = act_apply(acc[mt][nt][3] + bias[nt].w, act);
                *reinterpret_cast<float4*>(scr + r16 * RS + (16 * nt + 4 * g4) * 4) = v;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int idx = lane + 64 * j, row = idx >> 4, off = (idx & 15) * 16;
                const uint4 d = *reinterpret_cast<const uint4*>(scr + row * RS + off);
                *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(Y + (size_t)(mbase + 16 * mt + row) * N + nbase) + off) = d;
            }
        }
    }
}

static hipError_t allow_lds(const void* kernel, int slot, size_t bytes) {
    static bool done[4][64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !done[slot][dev]) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) done[slot][dev] = true;
    }
    return hipSuccess;
}

// M a multiple of 128, N of 128 (checked by the caller).  256-row tiles when they still give every CU a block.
hipError_t launch_linear_split(const SplitLinearArgs& a, int groups, hipStream_t s) {
    if (a.M == 0 || a.N == 0 || groups == 0) return hipSuccess;
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    static const int force_mt = getenv("MMS_SPLIT_MT") ? atoi(getenv("MMS_SPLIT_MT")) : 0;
    const int64_t tiles256 = (a.M % 256 == 0) ? (int64_t)groups * (a.M / 256) * (a.N / 128) : 0;
    const bool big = force_mt ? (force_mt == 4 && tiles256 > 0) : tiles256 >= cus;
#define MMS_LAUNCH_SPLIT(MT, OP, SLOT)                                                                                         \
    {                                                                                                                          \
        auto kern = linear_split_kernel<MT, OP>;                                                                               \
        if (hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), SLOT, SplitGeom<MT>::LDS); e != hipSuccess) return e; \
        const unsigned grid = (unsigned)((int64_t)groups * (a.M / (64 * MT)) * (a.N / 128));                                   \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SplitGeom<MT>::LDS, s, a);                                             \
    }
    if (big && a.out_planes) MMS_LAUNCH_SPLIT(4, true, 0)
    else if (big) MMS_LAUNCH_SPLIT(4, false, 1)
    else if (a.out_planes) MMS_LAUNCH_SPLIT(2, true, 2)
    else MMS_LAUNCH_SPLIT(2, false, 3)
#undef MMS_LAUNCH_SPLIT
    return hipGetLastError();
}

}  // namespace mms
