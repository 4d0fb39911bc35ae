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
// thread = (row, chunk, k-group of 8): reads 32 bytes, writes one 16-byte piece per plane; blockIdx.y = matrix.  ALIGNED: the rows
// start on 16-byte boundaries (float4 loads); otherwise scalar loads (agent k's rows of an [M, agents, 46] observation block).
template <bool ALIGNED>
__global__ void __launch_bounds__(256) split_planes_kernel(SplitPlanesArgs a, int KC) {
    const float* __restrict__ x = a.x[blockIdx.y];
    uint8_t* __restrict__ out = reinterpret_cast<uint8_t*>(a.planes[blockIdx.y]);
    const int K = a.K;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)(id & 3);
    const int64_t rc = id >> 2;
    const int kc = (int)(rc % KC);
    const int64_t row = rc / KC;
    if (row >= a.rows) return;
    const int k0 = kc * 32 + g * 8;
    float v[8];
    const float* src = x + row * (int64_t)a.x_pitch + k0;
    if (ALIGNED && k0 + 8 <= K) {
        const float4 p = *reinterpret_cast<const float4*>(src), q = *reinterpret_cast<const float4*>(src + 4);
        v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
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

hipError_t launch_split_planes_group(const SplitPlanesArgs& a, int groups, hipStream_t s) {
    if (a.rows == 0 || groups == 0) return hipSuccess;
    const int KC = (a.K + 31) / 32;
    const int64_t threads = a.rows * KC * 4;
    bool aligned = (a.x_pitch % 4) == 0;
    for (int g = 0; g < groups; g++) aligned = aligned && (reinterpret_cast<uintptr_t>(a.x[g]) & 15) == 0;
    const dim3 grid((unsigned)((threads + 255) / 256), groups);
    if (aligned) hipLaunchKernelGGL(split_planes_kernel<true>, grid, dim3(256), 0, s, a, KC);
    else hipLaunchKernelGGL(split_planes_kernel<false>, grid, dim3(256), 0, s, a, KC);
    return hipGetLastError();
}

hipError_t launch_split_planes(const float* x, void* planes, int64_t rows, int K, int x_pitch, hipStream_t s) {
    SplitPlanesArgs a = {};
    a.x[0] = x; a.planes[0] = planes; a.rows = rows; a.K = K; a.x_pitch = x_pitch;
    return launch_split_planes_group(a, 1, s);
}

// ---- the layer -------------------------------------------------------------------------------------------------------------
template <int MT>
struct SplitGeom {
    static constexpr int TM = 64 * MT, TN = 128;
    static constexpr int XBYTES = TM * kChunkBytes, WBYTES = TN * kChunkBytes, BUF = XBYTES + WBYTES;     // rows of 192 bytes, X tile then W tile
    static constexpr int NDMA = BUF / 1024 / 8;                         // 1-KB LDS-DMA instructions per wave and k-step: 9 (MT = 4), 6 (MT = 2)
    // epilogue scratch (8 waves x 16 rows x 400 bytes): inside the operand buffer the tile's last k-step has just consumed when that
    // is large enough (MT = 4: 72 KB), else behind the two buffers; the output head's weights [16][128] f32 in a region of their own
    static constexpr int SCRATCH = 8 * 16 * (2 * kChunkBytes + 16);
    static constexpr bool SCRATCH_IN_BUF = BUF >= SCRATCH + 0;
    static constexpr int HW_OFF = 2 * BUF + (SCRATCH_IN_BUF ? 0 : SCRATCH);
    static constexpr size_t LDS = (size_t)HW_OFF + 16 * 128 * 4;
};

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return (v > 0.f) ? v : (expf(v) - 1.f);
    if (act == 2) return fmaxf(v, 0.f);
    if (act == 3) return 1.f - 2.f / (__expf(2.f * v) + 1.f);
    return v;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// OUT: 0 = y as fp32 [M, N]; 1 = y as P32 planes (the next split layer's input); 2 = no y at all -- the layer feeds only an output
//      head whose dot products are finished here (head_part) -- the last hidden layer of a MAPPO / HAPPO network.
// LN:  0 = plain layer; 3 = the LayerNorms on both sides folded in (grouped MARL inference, ELU only): the input is the PRE-norm
//      activation and the layer is evaluated as rstd (W~ h - mean s) + c (stat_in, s; c passed as b), and the epilogue leaves, per
//      output row and 64-column slot, the sum of its activations and their squared deviations from the slot mean (part_out), which
//      mms_row_stats_chan_group / mms_marl_heads_finish combine into the next LayerNorm's (mean, rstd).
//
// PERSISTENT: the grid is at most one block per CU and a block walks tiles v = blockIdx.x, + gridDim.x, ... (a.tiles in all).  The
// first slice of the NEXT tile is requested (LDS-DMA into the operand buffer that is free at that point) before the epilogue of the
// current one, and the next tile's first barrier waits only for that DMA, not for the epilogue's stores (a counted vmcnt: the DMA
// is older than the stores) -- the write-out of a tile drains under the next tile's k-loop.  With 20 networks x 512 x 512 (five
// tiles per CU) that hides the ~13 us a tile otherwise pays besides its k-steps.
template <int MT, int OUT, int LN>
__global__ void __launch_bounds__(512, 2) linear_split_kernel(SplitLinearArgs a) {
    constexpr bool OUT_PLANES = OUT == 1;
    using G = SplitGeom<MT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, g4 = lane >> 4;
    const int KC = a.KC, N = a.N;
    const int tiles_n = N / G::TN, tiles_m = a.M / G::TM, total = a.tiles;
    const size_t pitch = (size_t)KC * kChunkBytes;

    // tile v of the launch: XCD-aware (v % 8 = the XCD the block sits on, the grid being a multiple of 8): each XCD walks a
    // contiguous run of (network, row panel, column tile), so that the blocks behind one L2 share X rows and W rows
    // Operands go HBM -> LDS directly (global_load_lds_dwordx4: 64 lanes x 16 bytes land in 1 KB of contiguous LDS, the source address
    // is per lane).  The LDS image of a k-step is the tile's rows at pitch 192 (X rows, then W rows), each row [plane][k-group] with the
    // k-group slot swizzled by the row (swz): instruction c of a step fills image bytes [1024 c, + 1024); lane s of it owns the 16-byte
    // slot S = 64 c + s = (row S / 12, plane (S % 12) / 4, slot S % 4) and fetches the k-group that belongs there.  In HBM that is a run
    // of 192 contiguous bytes per row.  Wave w issues c = w, w + 8, ...: its first NX instructions lie in the X rows, the rest in the
    // W rows (TM * 12 slots = a whole number of instructions), so an instruction's source is a UNIFORM base (the tile's first X or W
    // row at the current k: scalar registers, the only thing that changes from tile to tile) + a per-lane 32-bit offset that does not
    // depend on the tile at all.
    constexpr int NX = G::TM * 12 / 64 / 8;
    static_assert(G::TM * 12 % (64 * 8) == 0, "the X rows must end on an instruction boundary of every wave");
    uint32_t goff[G::NDMA];
#pragma unroll
    for (int i = 0; i < G::NDMA; i++) {
        const int S = 64 * (wave + 8 * i) + lane;
        const int row = S / 12, q = S - row * 12, pl = q >> 2, kg = swz(row, q & 3);
        goff[i] = (uint32_t)((i < NX ? row : row - G::TM) * (int)pitch + pl * 64 + kg * 16);
    }
    int gi, m0, n0, tn;
    const uint8_t* xb;                                                  // first X row / first W row of the current tile
    const uint8_t* wb;
    auto setup_tile = [&](int v) {
        int L = v;
        if ((total & 7) == 0 && (gridDim.x & 7) == 0) L = (v & 7) * (total >> 3) + (v >> 3);
        tn = L % tiles_n;
        const int rest = L / tiles_n;
        const int tm = rest % tiles_m;
        gi = rest / tiles_m;
        m0 = tm * G::TM;
        n0 = tn * G::TN;
        xb = reinterpret_cast<const uint8_t*>(a.x[gi]) + (size_t)m0 * pitch;
        wb = reinterpret_cast<const uint8_t*>(a.w[gi]) + (size_t)n0 * pitch;
    };
    auto dma_slice = [&](int kc, int buf) {
        const uint8_t* xs = xb + (size_t)kc * kChunkBytes;
        const uint8_t* ws = wb + (size_t)kc * kChunkBytes;
#pragma unroll
        for (int i = 0; i < G::NDMA; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)((i < NX ? xs : ws) + goff[i]), (lptr_t)(lds + buf * G::BUF + (wave + 8 * i) * 1024), 16, 0, 0);
    };

    // this lane's fragment addresses inside a buffer: row r16 of a 16-row tile, k-group g4 at its swizzled slot
    const int frag = r16 * kChunkBytes + swz(r16, g4) * 16;
    const int xfrag = (wm * 16 * MT) * kChunkBytes + frag;               // + plane * 64 + mt * 16 * 192
    const int wfrag = G::XBYTES + (wn * 64) * kChunkBytes + frag;        // + plane * 64 + nt * 16 * 192

    // Two accumulators per output: `acc` takes the leading products a0 b0, `lo` the five small ones (<= 2^-8 of the leading term), and
    // they are added once at the end.  The small products then meet an accumulator of their own size instead of being aligned to
    // (and truncated against) the large one at every step: error against float64 0.32 x the exact-fp32 MFMA chain's (a single
    // accumulator: 0.85 x, with a mean error of -5e-8 rms(Y) that this removes); no extra MFMAs, 64 more registers.
    constexpr bool kDmaBehindFirstTile = MT == 4;    // measured: 256 x 128 tiles 80.5 -> 78.5 us (K = 1024), 128 x 128 tiles 43.3 -> 46.6 us
    f32x4 acc[MT][4], lo[MT][4];

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

    if constexpr (OUT == 2) {                         // the output heads' weights for every column of the layer: staged once per block
        float* hw = reinterpret_cast<float*>(lds + G::HW_OFF);
        (void)hw;
    }

    int v = blockIdx.x;
    setup_tile(v);
    int par = 0;                                      // the operand buffer slice 0 of the current tile lives in
    bool stores_in_flight = false;                    // the previous tile's epilogue stores may still be draining
    dma_slice(0, par);
    while (true) {
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int kt = 0; kt < KC; kt++) {
            if (kt == 0 && stores_in_flight) {
                // slice 0 was requested BEFORE the previous tile's output stores, and vmcnt retires in order: once at most as many
                // operations are outstanding as that epilogue issued 16-byte output stores, the DMA has landed
                constexpr int kStores = OUT == 1 ? MT * 6 : (OUT == 0 ? MT * 4 : 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStores) : "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();                                        // slice kt has landed (vmcnt(0) + barrier); everyone is done with slice kt - 1
            }
            step((kt + par) & 1, kt + 1 < KC, kt + 1);
        }
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] += lo[i][j];
        const int blast = (KC - 1 + par) & 1;                            // the buffer the last k-step read
        __syncthreads();                                                // ... which now becomes the waves' epilogue scratch

        // epilogue: lane holds, for tile (mt, nt), rows n = nbase + 16 nt + 4 g4 + reg (reg = 0..3) of column m = mbase + 16 mt + r16
        // (the lane indices are re-derived from an opaque copy: otherwise the epilogue's dozens of tile-independent address terms are
        //  hoisted out of the tile loop and live -- spilled -- across the k-loop)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int r16 = lane_e & 15, g4 = lane_e >> 4, lane = lane_e;
        const int e_gi = gi, e_tn = tn;
        const int mbase = m0 + wm * 16 * MT, nbase = n0 + wn * 64, e_n0 = n0;
        const int vnext = v + (int)gridDim.x;
        const bool has_next = vnext < total;
        if (has_next) {                                                 // the next tile's first slice into the other buffer, ahead of everything below
            setup_tile(vnext);
            dma_slice(0, blast ^ 1);
        }
        uint8_t* scr_base = lds + (G::SCRATCH_IN_BUF ? blast * G::BUF : 2 * G::BUF);
        const float* __restrict__ Bv = a.b[e_gi];
        float4 bias[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) bias[nt] = *reinterpret_cast<const float4*>(Bv + nbase + 16 * nt + 4 * g4);
        const int act = a.act;
        // activations in place of the accumulators
        if constexpr (LN == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    acc[mt][nt][0] = act_apply(acc[mt][nt][0] + bias[nt].x, act);
                    acc[mt][nt][1] = act_apply(acc[mt][nt][1] + bias[nt].y, act);
                    acc[mt][nt][2] = act_apply(acc[mt][nt][2] + bias[nt].z, act);
                    acc[mt][nt][3] = act_apply(acc[mt][nt][3] + bias[nt].w, act);
                }
        } else {
            const float* __restrict__ Sv = a.s[e_gi];
            const float2* __restrict__ stat = reinterpret_cast<const float2*>(a.stat_in[e_gi]);
            float2* __restrict__ part = reinterpret_cast<float2*>(a.part_out[e_gi]) + (size_t)(2 * e_tn + wn) * a.M;
            float4 sv[4];
#pragma unroll
            for (int nt = 0; nt < 4; nt++) sv[nt] = *reinterpret_cast<const float4*>(Sv + nbase + 16 * nt + 4 * g4);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int m = mbase + 16 * mt + r16;
                const float2 st = stat[m];                              // (mean, rstd) of the input row
                float sum = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    const float ss[4] = {sv[nt].x, sv[nt].y, sv[nt].z, sv[nt].w}, bb[4] = {bias[nt].x, bias[nt].y, bias[nt].z, bias[nt].w};
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float x = st.y * (acc[mt][nt][r] - st.x * ss[r]) + bb[r];
                        x = (x > 0.f) ? x : (expf(x) - 1.f);
                        acc[mt][nt][r] = x;
                        sum += x;
                    }
                }
                // the row's 64 activations of this slot sit in the four lanes r16, r16 + 16, r16 + 32, r16 + 48: sum, then squared
                // deviations from the slot mean (two passes in registers: no cancellation, whatever the mean of the row)
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float mean = sum * (1.f / 64.f);
                float m2 = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) { const float d = acc[mt][nt][r] - mean; m2 += d * d; }
                m2 += __shfl_xor(m2, 16, 64);
                m2 += __shfl_xor(m2, 32, 64);
                if (g4 == 0) part[m] = make_float2(sum, m2);
            }
        }
        if constexpr (OUT == 2) {
            // the output head on this layer's activations, as far as this block's 128 columns go: head_part[slot][m][j] = sum over
            // the slot's 64 columns of h[m][n] head_w[j][n]  (head_w = the head's weight with the last LayerNorm's gamma folded in)
            const int A = a.hdims[e_gi], HS = (A + 3) & ~3;           // this network's head rows; its partials' row stride
            float* hw = reinterpret_cast<float*>(lds + G::HW_OFF);                           // [A][128], a region of its own
            const float* __restrict__ HW = a.head_w[e_gi];
            for (int i = t; i < A * 128; i += 512) hw[i] = HW[(size_t)(i >> 7) * N + e_n0 + (i & 127)];
            __syncthreads();
            float keep[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int i = 0; i < 4; i++) keep[mt][i] = 0.f;
            for (int j = 0; j < A; j++) {
                float4 hj[4];
#pragma unroll
                for (int nt = 0; nt < 4; nt++) hj[nt] = *reinterpret_cast<const float4*>(hw + j * 128 + wn * 64 + 16 * nt + 4 * g4);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    float p = 0.f;
#pragma unroll
                    for (int nt = 0; nt < 4; nt++)
                        p += acc[mt][nt][0] * hj[nt].x + acc[mt][nt][1] * hj[nt].y + acc[mt][nt][2] * hj[nt].z + acc[mt][nt][3] * hj[nt].w;
                    p += __shfl_xor(p, 16, 64);
                    p += __shfl_xor(p, 32, 64);
                    if ((j & 3) == g4) {                                // lane group g4 keeps outputs j = g4, g4 + 4, ...
#pragma unroll
                        for (int i = 0; i < 4; i++) if ((j >> 2) == i) keep[mt][i] = p;
                    }
                }
            }
            float* __restrict__ hp = a.head_part[e_gi] + (size_t)(2 * e_tn + wn) * a.M * HS;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (4 * i + g4 < A) hp[(size_t)(mbase + 16 * mt + r16) * HS + 4 * i + g4] = keep[mt][i];
            __syncthreads();                                            // (hw is restaged by the next tile)
        } else if constexpr (OUT_PLANES) {
            constexpr int RS = 2 * kChunkBytes + 16;                     // scratch row: this wave's two chunks (64 n) of one m, padded
            uint8_t* scr = scr_base + wave * (16 * RS);
            uint8_t* __restrict__ Y = reinterpret_cast<uint8_t*>(a.y[e_gi]);
            const size_t ypitch = (size_t)(N / 32) * kChunkBytes;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    bf16x4 p0, p1, p2;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float x = acc[mt][nt][r];
                        p0[r] = (__bf16)x; x -= (float)p0[r];
                        p1[r] = (__bf16)x; x -= (float)p1[r];
                        p2[r] = (__bf16)x;
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
            uint8_t* scr = scr_base + wave * (16 * RS);
            float* __restrict__ Y = reinterpret_cast<float*>(a.y[e_gi]);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
                    *reinterpret_cast<float4*>(scr + r16 * RS + (16 * nt + 4 * g4) * 4) = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int idx = lane + 64 * j, row = idx >> 4, off = (idx & 15) * 16;
                    const uint4 d = *reinterpret_cast<const uint4*>(scr + row * RS + off);
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(Y + (size_t)(mbase + 16 * mt + row) * N + nbase) + off) = d;
                }
            }
        }
        if (!has_next) break;
        v = vnext;
        par = blast ^ 1;
        stores_in_flight = true;
    }
}

static hipError_t allow_lds(const void* kernel, int slot, size_t bytes) {
    static bool done[8][64] = {};
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
        static int cached[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (cached[dev] == 0) {
                int n = 0;
                cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
            }
            cus = cached[dev];
        }
    }
    static const int force_mt = getenv("MMS_SPLIT_MT") ? atoi(getenv("MMS_SPLIT_MT")) : 0;
    const int64_t tiles256 = (a.M % 256 == 0) ? (int64_t)groups * (a.M / 256) * (a.N / 128) : 0;
    const bool big = force_mt ? (force_mt == 4 && tiles256 > 0) : tiles256 >= cus;
    const bool ln = a.stat_in[0] != nullptr;
    if (a.out_mode < 0 || a.out_mode > 2 || (ln && a.out_mode == 0) || (!ln && a.out_mode == 2)) return hipErrorInvalidValue;
#define MMS_LAUNCH_SPLIT(MT, OUT, LNF, SLOT)                                                                                   \
    {                                                                                                                          \
        auto kern = linear_split_kernel<MT, OUT, LNF>;                                                                         \
        if (hipError_t e = allow_lds(reinterpret_cast<const void*>(kern), SLOT, SplitGeom<MT>::LDS); e != hipSuccess) return e; \
        SplitLinearArgs b = a;                                                                                                 \
        b.tiles = (int)((int64_t)groups * (a.M / (64 * MT)) * (a.N / 128));                                                    \
        const unsigned grid = (unsigned)(b.tiles < cus ? b.tiles : cus);        /* persistent: at most one block per CU */      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SplitGeom<MT>::LDS, s, b);                                             \
    }
    if (ln) {
        if (big && a.out_mode == 1) MMS_LAUNCH_SPLIT(4, 1, 3, 4)
        else if (big) MMS_LAUNCH_SPLIT(4, 2, 3, 5)
        else if (a.out_mode == 1) MMS_LAUNCH_SPLIT(2, 1, 3, 6)
        else MMS_LAUNCH_SPLIT(2, 2, 3, 7)
    } else if (big && a.out_mode == 1) MMS_LAUNCH_SPLIT(4, 1, 0, 0)
    else if (big) MMS_LAUNCH_SPLIT(4, 0, 0, 1)
    else if (a.out_mode == 1) MMS_LAUNCH_SPLIT(2, 1, 0, 2)
    else MMS_LAUNCH_SPLIT(2, 0, 0, 3)
#undef MMS_LAUNCH_SPLIT
    return hipGetLastError();
}

}  // namespace mms
