// split16_kernels.hip -- the policy layers Y = act(X W^T + b) with fp32 operands carried as TWO scaled fp16 planes each.
//
// The sibling of split_kernels.hip (three bf16 planes, every operand exact).  Here a number is x s = hi + lo 2^-11 with s a power of
// two per ROW, hi = f16(x s), lo = f16((x s - hi) 2^11): 11 + 11 significant bits and the residual's sign, i.e. the operand is kept
// to 2^-22 |x| worst case (4e-8 rms) instead of exactly, in 4 bytes instead of 6, and
//     x w = hi_x hi_w + 2^-11 (hi_x lo_w + lo_x hi_w) + [2^-22 lo_x lo_w <= 2^-24 |x w|, dropped]
// is THREE v_mfma_f32_16x16x32_f16 products instead of six bf16 ones, two thirds of the operand bytes, and at 48 KB per 256 x 128
// k-step THREE LDS stages fit: a slice's DMA (buffer loads into LDS) is in flight across two barriers (counted vmcnt, raw s_barrier)
// instead of every step draining the queue.  Since round 4 the k-loop keeps a slice's fragments in registers and issues the next
// slice's ds_reads between the MFMAs (MMS_S16_ROLL, below).  What bounds it, measured with timing builds and an in-kernel clock probe
// (MMS_S16_EXP / MMS_S16_STAMP; profiles/r04_split16_kloop_experiments.txt): the chip is power-limited in this kernel (1.55 GHz) and
// the matrix pipe is busy 85 % of the k-loop's cycles.  Error against the float64 product, measured like the bf16 kernel's
// (tests/test_gpu_parity.py::test_split16_layers_error): still below the exact-fp32 MFMA kernel's, whose k-ordered fma chain
// rounds 16 times as often.
//
// Why a scale, and why it cannot overflow: fp16 ends at 65504.  Every plane row carries a power of two that puts a BOUND of the
// row's magnitudes at 2^14: for an input (observation, weight row) the bound is the row's own largest magnitude; for a hidden
// activation it is a-priori, from the chain |act(W x + b)| <= (max row 1-norm of W) max|x| + max|b| (ELU, ReLU, tanh and the
// identity all satisfy |act(y)| <= |y|), evaluated per row by the kernel that splits the network's input (split16_planes_kernel:
// chain / chain_scale).  The products of powers of two are undone exactly in the epilogue.  Elements far below the bound lose
// nothing that matters: hi goes subnormal 2^-29 below the bound and lo still carries the residual (measured: the error of a layer
// does not change when the scale is lowered by 2^18).
//
// Plane format "H32": [rows][KC][2][32] f16, KC = ceil(K / 32): hi and lo of 32 consecutive k of a row are 128 contiguous bytes
// (one cache line).  Columns past K are zero.
//
// Tiling as split_kernels.hip: 512-thread block = 8 waves (4 (m) x 2 (n)), output tile (64 MT) x 128, the product evaluated
// transposed (W fragments as the A operand), persistent grid of at most one block per CU.  LDS image of a k-step: the tile's rows
// at pitch 128 (X rows, then W rows), each row eight 16-byte slots [plane][k-group] with the slot index XORed with (row >> 1) & 7:
// the sixteen lanes of a ds_read_b128 group (rows 0-3 / 12-15 with k-group g, rows 4-11 with k-group g +- 1) then cover the sixteen
// slots of the 256-byte bank row exactly once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "policy_args.h"

namespace mms {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_16 __attribute__((ext_vector_type(4)));

// cache policy of the operand DMA (aux of global_load_lds: 0 default, 2 = nt); A/B knobs, the defaults are what measured fastest
#ifndef MMS_S16_AUX_X
#define MMS_S16_AUX_X 0
#endif
#ifndef MMS_S16_AUX_W
#define MMS_S16_AUX_W 0
#endif
// the k-loop's form: 1 = rolling fragments (the ds_reads of slice k + 1 issued between the MFMAs of slice k), 0 = round 3's (every step
// reads its fragments behind its barrier and waits for them); an A/B knob, bit-identical results
// timing experiments (results INVALID; never in a shipped build): MMS_S16_EXP bit 0 = no operand DMA inside the k-loop, bit 1 = no MFMA
// (the fragments are kept alive), bit 2 = no fragment reads inside the k-loop
#ifndef MMS_S16_STAMP
#define MMS_S16_STAMP 0
#endif
#ifndef MMS_S16_EXP
#define MMS_S16_EXP 0
#endif
#if MMS_S16_EXP & 2
#define MMS_S16_MFMA(a, b, c, x, y, z) ([&] { asm volatile("" ::"v"(a), "v"(b)); return c; }())
#else
#define MMS_S16_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#endif
#ifndef MMS_S16_ROLL
#define MMS_S16_ROLL 1
#endif
constexpr int kChunk16 = 128;                    // one row's two planes of 32 k
constexpr float kLoScale = 2048.f;               // the lo plane holds the residual times 2^11
constexpr int kTopExp = 14;                      // a row's bound sits at 2^14 (fp16 ends just below 2^16)

// scale = 2^(14 - e) with bound <= 2^e; an all-zero row keeps scale 1
__device__ __forceinline__ void pow2_scale(float bound, float& scale, float& inv) {
    int e = kTopExp;
    if (bound > 0.f) (void)frexpf(bound, &e);    // bound = f 2^e, 0.5 <= f < 1
    int sh = kTopExp - e;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    scale = ldexpf(1.f, sh);
    inv = ldexpf(1.f, -sh);
}

// ---- fp32 [rows, K] -> H32 planes + per-row scales -----------------------------------------------------------------------------
// P lanes per row (a power of two, 64 / P rows per wave: a 46-wide row is 8 pieces of 8 k, eight rows share a wave), two passes over
// the row (largest magnitude and sum, then the planes and the squared deviations): the second pass finds the row in L1 / L2.
// stat (optional): (mean, 1 / sqrt(var + eps)) of the row, the statistics of a LayerNorm over it (two-pass form) -- the grouped MARL
// inference folds its feature LayerNorm into the first layer and needs exactly these; one read of the rows serves both.
template <bool ALIGNED>
__global__ void __launch_bounds__(256) split16_planes_kernel(Split16PlanesArgs a, int KC, int P) {
    const int g = blockIdx.y;
    const float* __restrict__ x = a.x[g];
    uint8_t* __restrict__ out = reinterpret_cast<uint8_t*>(a.planes[g]);
    const int lane = threadIdx.x & 63;
    int K = a.K, pitch = a.x_pitch;
    int64_t rows = a.rows;
    if (a.per_group) {                                                  // this group's own shape (uniform per block)
        K = a.K_g[g];
        pitch = K;
        rows = a.rows_g[g];
        KC = (K + 31) / 32;
        P = 1;
        while (P < KC * 4 && P < 64) P <<= 1;
        if (rows == 0) return;
    }
    const int sub = lane & (P - 1), rpw = 64 / P;
    const int64_t row_raw = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + lane / P;
    const bool live = row_raw < rows;
    const int64_t row = live ? row_raw : rows - 1;                      // (idle lanes stay in the shuffles, on a valid row)
    const float* src = x + row * (int64_t)pitch;
    const int pieces = KC * 4;                                          // 8-element pieces of the row
    auto load8 = [&](int p, float* v) {
        const int k0 = p * 8;
        if (ALIGNED && k0 + 8 <= K) {
            const float4 q0 = *reinterpret_cast<const float4*>(src + k0), q1 = *reinterpret_cast<const float4*>(src + k0 + 4);
            v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (k0 + j < K) ? src[k0 + j] : 0.f;
        }
    };
    float big = 0.f, sum = 0.f, asum = 0.f;
    for (int p = sub; p < pieces; p += P) {
        float v[8];
        load8(p, v);
#pragma unroll
        for (int j = 0; j < 8; j++) { big = fmaxf(big, fabsf(v[j])); sum += v[j]; asum += fabsf(v[j]); }
    }
    for (int m = P >> 1; m >= 1; m >>= 1) {
        big = fmaxf(big, __shfl_xor(big, m, 64));
        sum += __shfl_xor(sum, m, 64);
        asum += __shfl_xor(asum, m, 64);
    }
    const float mean = sum / (float)K;
    float scale, inv;
    pow2_scale(big, scale, inv);
    float m2 = 0.f;
    for (int p = sub; p < pieces; p += P) {
        float v[8];
        load8(p, v);
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float t = v[j] * scale;
            hi[j] = (_Float16)t;
            lo[j] = (_Float16)((t - (float)hi[j]) * kLoScale);
            const float d = (p * 8 + j < K) ? v[j] - mean : 0.f;
            m2 += d * d;
        }
        if (live) {
            uint8_t* dst = out + (row * KC + (p >> 2)) * (int64_t)kChunk16 + (p & 3) * 16;
            *reinterpret_cast<f16x8*>(dst) = hi;
            *reinterpret_cast<f16x8*>(dst + 64) = lo;
        }
    }
    if (a.stat[g])
        for (int m = P >> 1; m >= 1; m >>= 1) m2 += __shfl_xor(m2, m, 64);
    if (sub == 0 && live) {
        if (a.scale[g]) a.scale[g][row] = scale;
        if (a.inv[g]) a.inv[g][row] = inv;
        if (a.stat[g]) reinterpret_cast<float2*>(a.stat[g])[row] = make_float2(mean, 1.0f / sqrtf(m2 / (float)K + a.eps));
        if (a.l1[g]) a.l1[g][row] = asum;
    }
    // the scales of the layers behind this input: one chain per lane of the row
    if (live)
        for (int c = sub; c < a.nchains; c += P) {
            const float* ch = a.chain[g] + (size_t)c * a.L * 2;
            float bound = big;
            for (int l = 0; l < a.L; l++) {
                bound = (ch[2 * l] * bound + ch[2 * l + 1]) * 1.001f;   // (rounding of the bound itself; 2^14 leaves a factor 4 besides)
                float sc, iv;
                pow2_scale(bound, sc, iv);
                a.chain_scale[g][((size_t)c * a.L + l) * rows + row] = sc;
                a.chain_inv[g][((size_t)c * a.L + l) * rows + row] = iv;
            }
        }
}

hipError_t launch_split16_planes_group(const Split16PlanesArgs& a, int groups, hipStream_t s) {
    if (groups == 0) return hipSuccess;
    if (a.per_group) {
        // matrices of different shapes: dense rows (16-byte aligned when K is a multiple of 4), the grid sized for the group with the
        // most blocks; the other groups' surplus blocks redo their last row without storing
        bool aligned = true;
        unsigned blocks = 0;
        for (int g = 0; g < groups; g++) {
            aligned = aligned && (a.K_g[g] % 4) == 0 && (reinterpret_cast<uintptr_t>(a.x[g]) & 15) == 0;
            const int KCg = (a.K_g[g] + 31) / 32;
            int Pg = 1;
            while (Pg < KCg * 4 && Pg < 64) Pg <<= 1;
            const int64_t rpb = 4 * (64 / Pg);
            const unsigned need = (unsigned)((a.rows_g[g] + rpb - 1) / rpb);
            blocks = need > blocks ? need : blocks;
        }
        if (blocks == 0) return hipSuccess;
        if (aligned) hipLaunchKernelGGL(split16_planes_kernel<true>, dim3(blocks, groups), dim3(256), 0, s, a, 0, 1);
        else hipLaunchKernelGGL(split16_planes_kernel<false>, dim3(blocks, groups), dim3(256), 0, s, a, 0, 1);
        return hipGetLastError();
    }
    if (a.rows == 0) return hipSuccess;
    const int KC = (a.K + 31) / 32;
    bool aligned = (a.x_pitch % 4) == 0;
    for (int g = 0; g < groups; g++) aligned = aligned && (reinterpret_cast<uintptr_t>(a.x[g]) & 15) == 0;
    int P = 1;
    while (P < KC * 4 && P < 64) P <<= 1;                               // lanes per row
    const int64_t rows_per_block = 4 * (64 / P);
    const dim3 grid((unsigned)((a.rows + rows_per_block - 1) / rows_per_block), groups);
    if (aligned) hipLaunchKernelGGL(split16_planes_kernel<true>, grid, dim3(256), 0, s, a, KC, P);
    else hipLaunchKernelGGL(split16_planes_kernel<false>, grid, dim3(256), 0, s, a, KC, P);
    return hipGetLastError();
}

// ---- the bound chain, refreshed on the device ----------------------------------------------------------------------------------
// Entry e = c L + l of the chain = (largest weight-row 1-norm, largest |bias|) of layer l of chain c, from the row norms the weight
// split left: every block reduces all entries (a few thousand floats: cheaper than a second launch and a dependency), block 0 stores
// them; then the chain's scales for rows whose input bound is a CONSTANT (observation rows clamped to clip_obs, whose planes the step
// kernel writes) -- the recurrence of split16_planes_kernel with b_0 = bound0, the same value for every row.
__global__ void __launch_bounds__(256) chain_refresh16_kernel(ChainRefreshArgs a) {
    __shared__ float s_chain[2 * kMaxGroups];
    const int E = a.nchains * a.L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = wave; e < E; e += 4) {                                  // one wave per entry: the four entries of a PPO policy at once
        float m = 0.f, b = 0.f;
        for (int i = lane; i < a.n[e]; i += 64) {
            m = fmaxf(m, a.l1[e][i]);
            if (a.bias[e]) b = fmaxf(b, fabsf(a.bias[e][i]));
        }
        for (int k = 32; k >= 1; k >>= 1) {
            m = fmaxf(m, __shfl_xor(m, k, 64));
            b = fmaxf(b, __shfl_xor(b, k, 64));
        }
        if (lane == 0) { s_chain[2 * e] = m; s_chain[2 * e + 1] = b; }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 2 * E) a.chain[threadIdx.x] = s_chain[threadIdx.x];
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.rows) return;
    for (int c = 0; c < a.nchains; c++) {
        float bound = a.bound0;
        for (int l = 0; l < a.L; l++) {
            bound = (s_chain[2 * (c * a.L + l)] * bound + s_chain[2 * (c * a.L + l) + 1]) * 1.001f;
            float sc, iv;
            pow2_scale(bound, sc, iv);
            a.chain_scale[((size_t)c * a.L + l) * a.rows + row] = sc;
            a.chain_inv[((size_t)c * a.L + l) * a.rows + row] = iv;
        }
    }
}

hipError_t launch_chain_refresh16(const ChainRefreshArgs& a, hipStream_t s) {
    if (a.nchains * a.L == 0) return hipSuccess;
    const unsigned blocks = a.rows > 0 ? (unsigned)((a.rows + 255) / 256) : 1u;
    hipLaunchKernelGGL(chain_refresh16_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- the layer -------------------------------------------------------------------------------------------------------------
template <int MT>
struct Geom16 {
    static constexpr int TM = 64 * MT, TN = 128;
    static constexpr int XBYTES = TM * kChunk16, WBYTES = TN * kChunk16, BUF = XBYTES + WBYTES;     // 48 KB (MT = 4) / 32 KB (MT = 2)
    static constexpr int NDMA = BUF / 1024 / 8;                         // 1-KB LDS-DMA instructions per wave and k-step: 6 / 4
    static constexpr int NX = TM / 64;                                  // ... of which in the X rows (TM / 8 instructions over 8 waves)
    static constexpr int SCRATCH = 8 * 16 * (2 * kChunk16 + 16);        // epilogue scratch: 8 waves x 16 rows x 272 bytes
    static constexpr bool SCRATCH_IN_BUF = BUF >= SCRATCH;              // inside the operand buffer the last k-step has just consumed
    static constexpr int HW_OFF = 3 * BUF + (SCRATCH_IN_BUF ? 0 : SCRATCH);
    static constexpr size_t LDS = (size_t)HW_OFF + 16 * 128 * 4;        // + the output head's weights [16][128] f32 (out_mode 2)
};

__device__ __forceinline__ float act16_apply(float v, int act) {
    if (act == 1) return (v > 0.f) ? v : (expf(v) - 1.f);
    if (act == 2) return fmaxf(v, 0.f);
    if (act == 3) return 1.f - 2.f / (__expf(2.f * v) + 1.f);
    return v;
}

// ELU on four accumulator values with as few vector instructions as the ISA allows (the epilogue is VALU-issue bound: 64 outputs per
// lane, two waves per SIMD): x log2(e) as a packed multiply (the factor in a register: a packed instruction takes no literal), v_exp_f32
// per element, the -1 as a packed add, and the select as ONE v_med3_f32 -- exp(x) - 1 >= x for every x, so x > 0 orders 0 < x <= e - 1 and
// x <= 0 orders x <= e - 1 <= 0: the median of (x, 0, e - 1) is ELU(x).  (Where fp32 rounding puts e - 1 below a tiny negative x, |x| <
// 6e-8, the median returns x itself, which is the more accurate value -- torch's expm1-based ELU returns x there too.)
__device__ __forceinline__ f32x4 elu16_4(f32x4 x) {
    float l2e = 1.44269504088896340736f;
    asm volatile("" : "+s"(l2e));
    const f32x4 t = x * l2e;
    f32x4 e;
#pragma unroll
    for (int r = 0; r < 4; r++) e[r] = __builtin_amdgcn_exp2f(t[r]);
    e = e - 1.f;
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = __builtin_amdgcn_fmed3f(x[r], 0.f, e[r]);
    return x;
}

// Four outputs -> their two planes: hi = f16(x), lo = f16((x - hi) 2^11) with x = v ys.  x - hi is exact in fp32 and so is every
// scaling by a power of two, hence lo = f16(fma(hi, -2^11, v (ys 2^11))) is the same number -- ONE v_fma_mix{lo,hi}_f16 per element
// (f16 source read in place, f32 result rounded to f16 into one half of the destination) instead of convert back + subtract + multiply +
// convert.  Returns {hi[0..1], hi[2..3], lo[0..1], lo[2..3]} as four dwords.
__device__ __forceinline__ void planes16_4(f32x4 v, float ys, float ys2k, uint32_t& h01, uint32_t& h23, uint32_t& l01, uint32_t& l23) {
    const f32x4 x = v * ys, x2 = v * ys2k;
    const f16x4 hi = __builtin_convertvector(x, f16x4);
    h01 = reinterpret_cast<const uint32_t*>(&hi)[0];
    h23 = reinterpret_cast<const uint32_t*>(&hi)[1];
    float c = -kLoScale;
    asm volatile("" : "+s"(c));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=&v"(l01) : "v"(h01), "s"(c), "v"(x2[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "s"(c), "v"(x2[1]));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=&v"(l23) : "v"(h23), "s"(c), "v"(x2[2]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "s"(c), "v"(x2[3]));
}

typedef __attribute__((address_space(3))) void* lptr16_t;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// OUT / LN as linear_split_kernel (split_kernels.hip).  The k-loop runs on THREE operand buffers: at the top of step k every wave
// waits until at most the DMA of slice k + 1 (its own NDMA instructions, the youngest) is outstanding -- slice k has landed --, the
// raw barrier makes that true for all waves and says that everyone is done reading slice k - 1, whose buffer then takes the DMA of
// slice k + 2.  Nothing in the loop waits for vmcnt(0) except a tile's last step.
// ELU: the activation is ELU (compile time: straight-line epilogue); false = a.act at run time (ReLU, tanh, identity: not on a hot path).
template <int MT, int OUT, int LN, bool ELU>
__global__ void __launch_bounds__(512, 2) linear_split16_kernel(Split16LinearArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass only needs the launch stub; it has no __amdgpu_buffer_rsrc_t)
    using G = Geom16<MT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // mms_layer_clock_probe: block 0 reports the clock the chip holds while this launch runs (shader cycles and 100-MHz ticks of its life)
    // (both counters read unconditionally: a branch on a.clock_probe here would put a kernel-argument round trip in front of everything else)
    const uint64_t probe_c0 = __builtin_readcyclecounter(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#if MMS_S16_STAMP   // phase stamps (timing experiments only: overwrites the first 32 bytes of the first output): shader clocks and 100-MHz ticks of block 0
    const uint64_t stamp_c0 = __builtin_readcyclecounter(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t stamp_r12[2] = {0, 0};                  // ... first slice landed, k-loop done (last tile of the block)
#endif
    const int wm = wave >> 1, wn = wave & 1;
    const int KC = a.KC, N = a.N;
    const int tiles_n = N / G::TN, tiles_m = a.M / G::TM, total = a.tiles;
    // (the grid size rides in the argument block beside `tiles`, and the test is branch-free: as `(total & 7) == 0 && (gridDim.x & 7) == 0`
    //  it cost a dependent scalar load of the hidden grid-size argument behind a branch -- one more round trip in front of the first DMA)
    const bool xcd_aware = (((total | a.grid) & 7) == 0);
    const size_t pitch = (size_t)KC * kChunk16;

    // LDS-DMA: instruction c of a k-step fills image bytes [1024 c, + 1024) = eight rows; lane s of it owns the physical 16-byte
    // slot S = 64 c + s = (row S / 8, slot q = S % 8) and fetches the (plane, k-group) = q ^ ((row >> 1) & 7) that lives there.  Wave w
    // issues c = w, w + 8, ...: the first NX in the X rows, the rest in the W rows; the source is a uniform base (the tile's first
    // X / W row at the current k) + a per-lane 32-bit offset that does not depend on the tile.
    uint32_t goff[G::NDMA];
#pragma unroll
    for (int i = 0; i < G::NDMA; i++) {
        const int S = 64 * (wave + 8 * i) + lane;
        const int row = S >> 3, c8 = (S & 7) ^ ((row >> 1) & 7);
        goff[i] = (uint32_t)((i < G::NX ? row : row - G::TM) * (int)pitch + (c8 >> 2) * 64 + (c8 & 3) * 16);
    }
    int gi, m0, n0, tn;
    uint64_t x01[2] = {reinterpret_cast<uint64_t>(a.x[0]), reinterpret_cast<uint64_t>(a.x[1])};
    uint64_t w01[2] = {reinterpret_cast<uint64_t>(a.w[0]), reinterpret_cast<uint64_t>(a.w[1])};
    asm volatile("" : "+s"(x01[0]), "+s"(x01[1]), "+s"(w01[0]), "+s"(w01[1]));       // (loaded HERE, with the other scalars -- not sunk into the tile setup)
    __amdgpu_buffer_rsrc_t xr, wr;                                      // the current tile's X rows / W rows (raw buffers: [TM or TN rows][pitch] bytes)
    auto setup_tile = [&](int v) {
        int L = v;
        if (xcd_aware) L = (v & 7) * (total >> 3) + (v >> 3);                                      // XCD-aware: as linear_split_kernel
        tn = L % tiles_n;
        const int rest = L / tiles_n;
        const int tm = rest % tiles_m;
        gi = rest / tiles_m;
        m0 = tm * G::TM;
        n0 = tn * G::TN;
        // (the first two networks' operand pointers come with the kernel's first batch of scalar loads -- x01 / w01 above: the PPO pair's
        //  tiles then need no second, tile-dependent argument round trip before their first DMA; further networks load theirs here)
        const uint64_t xg = gi == 0 ? x01[0] : (gi == 1 ? x01[1] : reinterpret_cast<uint64_t>(a.x[gi]));
        const uint64_t wg = gi == 0 ? w01[0] : (gi == 1 ? w01[1] : reinterpret_cast<uint64_t>(a.w[gi]));
        const uint8_t* xb = reinterpret_cast<const uint8_t*>(xg) + (size_t)m0 * pitch;
        const uint8_t* wb = reinterpret_cast<const uint8_t*>(wg) + (size_t)n0 * pitch;
        xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(xb), (short)0, (int)(G::TM * pitch), 0x00020000);
        wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(wb), (short)0, (int)(G::TN * pitch), 0x00020000);
    };
    // The operand DMA as BUFFER loads (`buffer_load_dwordx4 ... offen lds`): the tile's first X / W row in a resource descriptor (four
    // SGPRs, rebuilt per tile by scalar code), the piece's lane offset as the 32-bit VGPR offset, the slice's offset in an SGPR -- no
    // vector arithmetic per piece and no 64-bit lane addresses.  (As global loads the compiler kept base + lane offset as six 64-bit
    // VGPR pairs per operand and added the slice's offset with a v_lshl_add_u64 per piece: 12-24 registers the rolling loop does not have.)
    auto dma_piece = [&](int i, int kc, int buf) {                       // (i: compile time after unrolling)
        const int so = kc * kChunk16;
        if (i < G::NX) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr16_t)(lds + buf * G::BUF + (wave + 8 * i) * 1024), 16, goff[i], so, 0, MMS_S16_AUX_X);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lptr16_t)(lds + buf * G::BUF + (wave + 8 * i) * 1024), 16, goff[i], so, 0, MMS_S16_AUX_W);
    };
    auto dma_slice = [&](int kc, int buf) {
#pragma unroll
        for (int i = 0; i < G::NDMA; i++) dma_piece(i, kc, buf);
    };

    // this lane's fragment address inside a buffer: row r16 of a 16-row tile, plane 0, k-group g4; plane 1 is the same address ^ 64
    const int frag = (lane & 15) * kChunk16 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) * 16);
    const int xfrag = (wm * 16 * MT) * kChunk16 + frag;                  // + mt * 16 * 128
    const int wfrag = G::XBYTES + (wn * 64) * kChunk16 + frag;           // + nt * 16 * 128

    // `acc` takes hi hi, `lo` the two cross products (2^11 times their value); added once at the end
    f32x4 acc[MT][4], lo[MT][4];
#ifndef MMS_S16_DMA_BEHIND
#define MMS_S16_DMA_BEHIND (MT == 4)
#endif
    constexpr bool kDmaBehindFirstTile = MMS_S16_DMA_BEHIND;

    auto step = [&](int buf, bool more, int kc_next, int buf_next) {
        const uint8_t* base = lds + buf * G::BUF;
        if (!kDmaBehindFirstTile && more) dma_slice(kc_next, buf_next);
        f16x8 wf[4][2];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            wf[nt][0] = *reinterpret_cast<const f16x8*>(base + (wfrag + nt * 16 * kChunk16));
            wf[nt][1] = *reinterpret_cast<const f16x8*>(base + ((wfrag + nt * 16 * kChunk16) ^ 64));
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            f16x8 xf[2];
            xf[0] = *reinterpret_cast<const f16x8*>(base + (xfrag + mt * 16 * kChunk16));
            xf[1] = *reinterpret_cast<const f16x8*>(base + ((xfrag + mt * 16 * kChunk16) ^ 64));
#pragma unroll
            for (int nt = 0; nt < 4; nt++) lo[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt][1], xf[0], lo[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; nt++) lo[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt][0], xf[1], lo[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt][0], xf[0], acc[mt][nt], 0, 0, 0);
            if (kDmaBehindFirstTile && mt == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) dma_slice(kc_next, buf_next);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- the rolling form of a k-step (MMS_S16_ROLL) ----------------------------------------------------------------------------
    // Round 3's step above opens with its fragment reads and waits for them: 10 ds_read_b128 behind the barrier, lgkmcnt(0), only then the
    // first MFMA -- and the two waves of a SIMD do that in lockstep, so the matrix pipe idles for an LDS round trip (all eight waves'
    // reads at once) in every step: per step 1536 cycles of MFMA, ~600 of LDS and ~400 of waits added up to the 2570 measured
    // (profiles/r03_split16_layer_pmc.txt) instead of overlapping.  Here the fragments of slice k sit in registers when step k begins
    // and the reads of slice k + 1 are issued BETWEEN its MFMAs, each into the registers the MFMAs in front of it have just used for
    // the last time: the X fragments of row tile mt behind that tile's twelve MFMAs, the W fragments -- which every row tile needs --
    // pair by pair inside the last row tile, whose MFMAs run column pair by column pair for that.  No register is added (the X
    // fragments of all MT row tiles are live instead of one tile's: + 8 (MT - 1)), every accumulator sees its products in the same
    // order as before (bit-identical results), and by the time a fragment is used its read is a step old.
    // What the barrier at the top of step k then certifies: slice k + 1 has landed for every wave (each waited for its own pieces), and
    // every wave has finished READING slice k (its lgkmcnt(0) in front of the barrier: the last of those reads were issued three MFMAs
    // earlier) -- so slice k's buffer takes the DMA of slice k + 3, and the last slice's buffer is free for the epilogue's scratch
    // without a barrier of its own.  DMA depth in time is unchanged (issued two barriers ahead of the barrier that needs it).
    f16x8 wfr[4][2], xfr[MT][2];
    auto read_w = [&](const uint8_t* base, int nt) {
        wfr[nt][0] = *reinterpret_cast<const f16x8*>(base + (wfrag + nt * 16 * kChunk16));
        wfr[nt][1] = *reinterpret_cast<const f16x8*>(base + ((wfrag + nt * 16 * kChunk16) ^ 64));
    };
    auto read_x = [&](const uint8_t* base, int mt) {
        xfr[mt][0] = *reinterpret_cast<const f16x8*>(base + (xfrag + mt * 16 * kChunk16));
        xfr[mt][1] = *reinterpret_cast<const f16x8*>(base + ((xfrag + mt * 16 * kChunk16) ^ 64));
    };
    // Where a wave issues the DMA of slice k + 3 inside step k: behind row tile `qa` (waves 0-3) / `qb` (waves 4-7) -- -1 = in front of the
    // step's first MFMA, MT - 1 = inside the last row tile, behind its first column pair.  The two waves of a SIMD are waves w and w + 4:
    // with qa != qb their DMA issue phases (6 instructions that hold the wave's in-order stream for 60-185 cycles each,
    // MI355X_MICROARCH.md) do not coincide, and one of them feeds the matrix pipe while the other issues.
#ifndef MMS_S16_DMA_SPREAD
#define MMS_S16_DMA_SPREAD 0
#endif
#ifndef MMS_S16_DMA_QA
#define MMS_S16_DMA_QA (MT == 4 ? 0 : -1)
#endif
#ifndef MMS_S16_DMA_QB
#define MMS_S16_DMA_QB MMS_S16_DMA_QA
#endif
    const int qd = wave >= 4 ? ((MMS_S16_DMA_QB) < MT ? (MMS_S16_DMA_QB) : MT - 1) : ((MMS_S16_DMA_QA) < MT ? (MMS_S16_DMA_QA) : MT - 1);
    auto roll_step = [&](auto more_c, const uint8_t* next, bool dma, int kc_dma, int buf_dma) {
        constexpr bool more = decltype(more_c)::value && !(MMS_S16_EXP & 4);
        if (MMS_S16_EXP & 1) dma = false;
        if (!(MMS_S16_DMA_SPREAD) && dma && qd < 0) dma_slice(kc_dma, buf_dma);
        // MMS_S16_DMA_SPREAD: the pieces one at a time between the groups of four MFMAs of the first MT - 1 row tiles (piece i behind group
        // i NG / NDMA, waves 4-7 one group later) instead of all NDMA together
        constexpr int NG = 3 * (MT - 1);
        auto spread = [&](int g) {                                       // g: compile time
            if (!(MMS_S16_DMA_SPREAD) || !dma) return;
#pragma unroll
            for (int i = 0; i < G::NDMA; i++) {
                const int ga = i * NG / G::NDMA, gb = ga + 1 < NG ? ga + 1 : NG - 1;
                if ((wave >= 4 ? gb : ga) == g) dma_piece(i, kc_dma, buf_dma);
            }
        };
#pragma unroll
        for (int mt = 0; mt < MT - 1; mt++) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++) lo[mt][nt] = MMS_S16_MFMA(wfr[nt][1], xfr[mt][0], lo[mt][nt], 0, 0, 0);
            if (MMS_S16_DMA_SPREAD) { __builtin_amdgcn_sched_barrier(0); spread(3 * mt); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int nt = 0; nt < 4; nt++) lo[mt][nt] = MMS_S16_MFMA(wfr[nt][0], xfr[mt][1], lo[mt][nt], 0, 0, 0);
            if (MMS_S16_DMA_SPREAD) { __builtin_amdgcn_sched_barrier(0); spread(3 * mt + 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int nt = 0; nt < 4; nt++) acc[mt][nt] = MMS_S16_MFMA(wfr[nt][0], xfr[mt][0], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            spread(3 * mt + 2);
            if (!(MMS_S16_DMA_SPREAD) && dma && qd == mt) dma_slice(kc_dma, buf_dma);
            if constexpr (more) read_x(next, mt);
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int ml = MT - 1;
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int q = 0; q < 2; q++) lo[ml][2 * p + q] = MMS_S16_MFMA(wfr[2 * p + q][1], xfr[ml][0], lo[ml][2 * p + q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; q++) lo[ml][2 * p + q] = MMS_S16_MFMA(wfr[2 * p + q][0], xfr[ml][1], lo[ml][2 * p + q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; q++) acc[ml][2 * p + q] = MMS_S16_MFMA(wfr[2 * p + q][0], xfr[ml][0], acc[ml][2 * p + q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(MMS_S16_DMA_SPREAD) && p == 0 && dma && qd == ml) dma_slice(kc_dma, buf_dma);
            if constexpr (more) { read_w(next, 2 * p); read_w(next, 2 * p + 1); }
            if constexpr (more) if (p == 1) read_x(next, ml);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // (the LayerNorm-fold variants keep round 3's loop: their epilogue holds 24 more operand registers across the tile boundary, with the
    //  rolling loop's 64 fragment registers they spill, and the grouped MAPPO pass measured no gain from it: 505 against 496-506 us)
    constexpr bool kRoll = MMS_S16_ROLL != 0 && LN == 0;

    // 16-byte output stores a lane issues per tile BEHIND the next tile's prefetch (the count the next tile's first wait may leave
    // outstanding besides its own second slice); out_mode 2 stores under a predicate: none counted, the wait then covers them too
    constexpr int kStores = OUT == 2 ? 0 : MT * 4;

    int v = blockIdx.x;
    setup_tile(v);
    int cur = 0;                                      // the buffer slice 0 of the current tile lives in
    bool stores_in_flight = false;                    // (= not the block's first tile)
    dma_slice(0, 0);
    if (KC > 1) dma_slice(1, 1);
    if (kRoll && KC > 2) dma_slice(2, 2);             // (later tiles: behind their first barrier -- the third buffer is the epilogue's scratch until then)
    while (true) {
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        int blast;
        if constexpr (kRoll) {
            // The wave's vector-memory queue at this point, oldest first: slice 0, slice 1, then slice 2 (first tile) or the previous
            // tile's kStores output stores (later tiles).  Slice 0 has to have landed.
            if (!stores_in_flight) {
                if (KC > 2) wait_vm<2 * G::NDMA>(); else if (KC > 1) wait_vm<G::NDMA>(); else wait_vm<0>();
            } else {
                if (KC > 1) wait_vm<G::NDMA + kStores>(); else wait_vm<kStores>();
            }
            __builtin_amdgcn_s_barrier();                                // slice 0 is there for everyone; everyone has left the previous epilogue
#if MMS_S16_STAMP
            const uint64_t stamp_r1 = __builtin_amdgcn_s_memrealtime();
#endif
            {
                const int b2 = cur == 0 ? 2 : cur - 1;                   // (cur + 2) % 3
                if (stores_in_flight && KC > 2) dma_slice(2, b2);
                const uint8_t* base = lds + cur * G::BUF;
#pragma unroll
                for (int nt = 0; nt < 4; nt++) read_w(base, nt);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) read_x(base, mt);
            }
            for (int kt = 0; kt + 1 < KC; kt++) {
                const int nxt = cur == 2 ? 0 : cur + 1;                  // the buffer of slice kt + 1
                // slice kt + 1 has to have landed; younger than it: the stores (kt = 0 of a later tile), slice kt + 2
                const bool two = kt + 2 < KC;
                if (kt == 0 && stores_in_flight) {
                    if (two) wait_vm<G::NDMA + kStores>(); else wait_vm<kStores>();
                } else {
                    if (two) wait_vm<G::NDMA>(); else wait_vm<0>();
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's reads of slice kt are complete
                __builtin_amdgcn_s_barrier();
                roll_step(std::true_type{}, lds + nxt * G::BUF, kt + 3 < KC, kt + 3, cur);
                cur = nxt;
            }
            // the last slice (peeled: one MFMA body per path -- as two arms of a branch inside the loop the two bodies' accumulators met in
            // phi nodes the register allocator could not coalesce, 110 dwords of scratch)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            roll_step(std::false_type{}, lds, false, 0, 0);
            cur = cur == 2 ? 0 : cur + 1;
#if MMS_S16_STAMP
            asm volatile("s_nop 0" ::: "memory");
            stamp_r12[0] = stamp_r1 - stamp_r0;
            stamp_r12[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
#endif
            blast = cur == 0 ? 2 : cur - 1;                              // the buffer of the last slice: free (see above)
        } else {
            for (int kt = 0; kt < KC; kt++) {
                const bool ahead = kt + 1 < KC;                             // slice kt + 1 is in flight behind slice kt
                // (slice 1 of a following tile is older than the previous tile's stores as well, so step 1 could leave them outstanding too:
                //  measured in round 4, no difference -- 0.498-0.506 against 0.495-0.501 ms per grouped MARL pass)
                if (kt == 0 && stores_in_flight) {
                    if (ahead) wait_vm<G::NDMA + kStores>(); else wait_vm<kStores>();
                } else {
                    if (ahead) wait_vm<G::NDMA>(); else wait_vm<0>();
                }
                __builtin_amdgcn_s_barrier();
                const int nxt = cur == 0 ? 2 : cur - 1;                      // (cur + 2) % 3: the buffer step kt - 1 read
                step(cur, kt + 2 < KC, kt + 2, nxt);
                cur = cur == 2 ? 0 : cur + 1;
            }
            blast = cur == 0 ? 2 : cur - 1;                              // the buffer the last k-step read
            __builtin_amdgcn_s_barrier();                                // ... which becomes the waves' epilogue scratch
        }

        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));                                // (tile-independent epilogue addresses are re-derived, not kept live across the k-loop)
        const int r16 = lane_e & 15, g4 = lane_e >> 4;
        const int e_gi = gi, e_tn = tn;
        const int mbase = m0 + wm * 16 * MT, nbase = n0 + wn * 64, e_n0 = n0;
        // the epilogue's operands first: an ordinary load's result is waited for with vmcnt(0), which must not find the next tile's
        // prefetch in the queue
        const float* __restrict__ Bv = a.b[e_gi];
        const float* __restrict__ Wi = a.winv[e_gi];
        const float* __restrict__ Xi = a.xinv[e_gi];
        float4 bias[4], wi[4];
        float xi[MT], ys[MT];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            bias[nt] = *reinterpret_cast<const float4*>(Bv + nbase + 16 * nt + 4 * g4);
            wi[nt] = *reinterpret_cast<const float4*>(Wi + nbase + 16 * nt + 4 * g4);
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            xi[mt] = Xi[mbase + 16 * mt + r16];
            ys[mt] = OUT == 1 ? a.yscale[e_gi][mbase + 16 * mt + r16] : 1.f;
        }
        float4 sv[4];
        float2 st[MT];
        if constexpr (LN != 0) {
            const float* __restrict__ Sv = a.s[e_gi];
            const float2* __restrict__ stat = reinterpret_cast<const float2*>(a.stat_in[e_gi]);
#pragma unroll
            for (int nt = 0; nt < 4; nt++) sv[nt] = *reinterpret_cast<const float4*>(Sv + nbase + 16 * nt + 4 * g4);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) st[mt] = stat[mbase + 16 * mt + r16];
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            asm volatile("" : "+v"(bias[nt].x), "+v"(bias[nt].y), "+v"(bias[nt].z), "+v"(bias[nt].w), "+v"(wi[nt].x), "+v"(wi[nt].y), "+v"(wi[nt].z), "+v"(wi[nt].w));
            if constexpr (LN != 0) asm volatile("" : "+v"(sv[nt].x), "+v"(sv[nt].y), "+v"(sv[nt].z), "+v"(sv[nt].w));
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            asm volatile("" : "+v"(xi[mt]), "+v"(ys[mt]));
            if constexpr (LN != 0) asm volatile("" : "+v"(st[mt].x), "+v"(st[mt].y));
        }
        const int vnext = v + a.grid;
        const bool has_next = vnext < total;
        const int nb0 = blast == 2 ? 0 : blast + 1, nb1 = nb0 == 2 ? 0 : nb0 + 1;
        // (the next tile's first wait counts the vmcnt entries issued from here on: exactly these DMAs, then the kStores output stores --
        //  the scheduling barriers keep the compiler from moving a store in front of the prefetch or a DMA behind the stores)
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {                                                 // the next tile's first two slices into the two free buffers
            setup_tile(vnext);
            dma_slice(0, nb0);
            if (KC > 1) dma_slice(1, nb1);
        }
        __builtin_amdgcn_sched_barrier(0);
        uint8_t* scr_base = lds + (G::SCRATCH_IN_BUF ? blast * G::BUF : 3 * G::BUF);
        const int act = a.act;
        // hi hi + 2^-11 cross, the operands' scales undone (powers of two: exact), bias, activation -- in place of the accumulators.
        // Four-wide vector arithmetic (packed fp32 instructions: the accumulators sit in aligned register pairs); with ELU known at
        // compile time the whole block is straight-line code (a per-element switch on `act` puts every element into basic blocks of
        // its own, with scalar branches and a dependent mul -> exp -> add -> select chain each).
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const f32x4 w4 = f32x4{wi[nt].x, wi[nt].y, wi[nt].z, wi[nt].w};
                f32x4 raw = ((acc[mt][nt] + lo[mt][nt] * (1.f / kLoScale)) * w4) * xi[mt];
                if constexpr (LN == 0) {
                    raw += f32x4{bias[nt].x, bias[nt].y, bias[nt].z, bias[nt].w};
                    if constexpr (ELU) raw = elu16_4(raw);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; r++) raw[r] = act16_apply(raw[r], act);
                    }
                }
                acc[mt][nt] = raw;
            }
        if constexpr (LN != 0) {
            float2* __restrict__ part = reinterpret_cast<float2*>(a.part_out[e_gi]) + (size_t)(2 * e_tn + wn) * a.M;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int m = mbase + 16 * mt + r16;
                f32x4 sum4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    const f32x4 s4 = f32x4{sv[nt].x, sv[nt].y, sv[nt].z, sv[nt].w}, b4 = f32x4{bias[nt].x, bias[nt].y, bias[nt].z, bias[nt].w};
                    f32x4 x = (acc[mt][nt] - s4 * st[mt].x) * st[mt].y + b4;        // rstd (W~ h - mean s) + c, four wide (packed fp32)
                    x = elu16_4(x);
                    acc[mt][nt] = x;
                    sum4 += x;
                }
                float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float mean = sum * (1.f / 64.f);
                f32x4 q4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < 4; nt++) { const f32x4 d = acc[mt][nt] - mean; q4 += d * d; }
                float m2 = (q4[0] + q4[1]) + (q4[2] + q4[3]);
                m2 += __shfl_xor(m2, 16, 64);
                m2 += __shfl_xor(m2, 32, 64);
                if (g4 == 0) part[m] = make_float2(sum, m2);
            }
        }
        if constexpr (OUT == 2) {
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
                    if ((j & 3) == g4) {
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
        } else if constexpr (OUT == 1) {
            constexpr int RS = 2 * kChunk16 + 16;                        // scratch row: this wave's two chunks (64 n) of one m, padded
            uint8_t* scr = scr_base + wave * (16 * RS);
            const int ypitch = (N / 32) * kChunk16;
            // this wave's 64 MT rows x 256 bytes of the output through a buffer descriptor of their own: a store's address is then one
            // lane offset (row-in-group x pitch + piece) + a scalar offset per store, no 64-bit vector arithmetic (1.7 VALU per output before)
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<uint8_t*>(a.y[e_gi]) + (size_t)mbase * ypitch + (size_t)(nbase / 32) * kChunk16, (short)0, 16 * MT * ypitch, 0x00020000);
            const int yoff = (lane_e >> 4) * ypitch + (lane_e & 15) * 16;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const float ys2k = ys[mt] * kLoScale;
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    uint32_t h01, h23, l01, l23;
                    planes16_4(acc[mt][nt], ys[mt], ys2k, h01, h23, l01, l23);
                    uint8_t* d = scr + r16 * RS + (nt >> 1) * kChunk16 + (nt & 1) * 32 + g4 * 8;
                    *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(d + 64) = make_uint2(l01, l23);
                }
                // 16 rows x 256 bytes back out as 16-byte pieces: 16 per row, contiguous in HBM
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int idx = lane_e + 64 * j, row = idx >> 4, off = (idx & 15) * 16;
                    const u32x4_16 d = *reinterpret_cast<const u32x4_16*>(scr + row * RS + off);
                    __builtin_amdgcn_raw_buffer_store_b128(d, yr, yoff, (16 * mt + 4 * j) * ypitch, 0);
                }
            }
        } else {
            constexpr int RS = 256 + 16;
            uint8_t* scr = scr_base + wave * (16 * RS);
            const int ypitch = N * 4;
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<uint8_t*>(a.y[e_gi]) + (size_t)mbase * ypitch + (size_t)nbase * 4, (short)0, 16 * MT * ypitch, 0x00020000);
            const int yoff = (lane_e >> 4) * ypitch + (lane_e & 15) * 16;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
                    *reinterpret_cast<float4*>(scr + r16 * RS + (16 * nt + 4 * g4) * 4) = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int idx = lane_e + 64 * j, row = idx >> 4, off = (idx & 15) * 16;
                    const u32x4_16 d = *reinterpret_cast<const u32x4_16*>(scr + row * RS + off);
                    __builtin_amdgcn_raw_buffer_store_b128(d, yr, yoff, (16 * mt + 4 * j) * ypitch, 0);
                }
            }
        }
#if MMS_S16_STAMP
        if (!has_next && blockIdx.x == 0 && t == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint64_t* dbg = reinterpret_cast<uint64_t*>(a.y[0]);
            dbg[0] = __builtin_readcyclecounter() - stamp_c0;
            dbg[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
            dbg[2] = stamp_r12[0];
            dbg[3] = stamp_r12[1];
        }
#endif
        if (!has_next && a.clock_probe && blockIdx.x == 0 && t == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the block's own stores have left
            a.clock_probe[0] = __builtin_readcyclecounter() - probe_c0;
            a.clock_probe[1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
        }
        if (!has_next) break;
        v = vnext;
        cur = nb0;
        stores_in_flight = true;
    }
#endif
}

static hipError_t allow_lds16(const void* kernel, int slot, size_t bytes) {
    static bool done[16][64] = {};
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

// mms_layer_clock_probe: launches from here on report into out[2 (n % slots)], n counting from this call (out = nullptr: off)
static uint64_t* g_probe_out = nullptr;
static int g_probe_slots = 0;
static long g_probe_next = 0;
void set_split16_clock_probe(uint64_t* out, int slots) {
    g_probe_out = (out && slots > 0) ? out : nullptr;
    g_probe_slots = slots;
    g_probe_next = 0;
}

// M a multiple of 128, N of 128 (checked by the caller).  256-row tiles when they still give every CU a block.
hipError_t launch_linear_split16(const Split16LinearArgs& a, int groups, hipStream_t s) {
    if (a.M == 0 || a.N == 0 || groups == 0) return hipSuccess;
    int cus = 256;
    {
        static int cached[64] = {};                                     // (hipGetDeviceProperties per launch is host time an eager rollout step pays)
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
#define MMS_LAUNCH_SPLIT16(MT, OUT, LNF, SLOT)                                                                                 \
    {                                                                                                                          \
        auto kern = (LNF != 0 || a.act == 1) ? linear_split16_kernel<MT, OUT, LNF, true> : linear_split16_kernel<MT, OUT, LNF, false>; \
        if (hipError_t e = allow_lds16(reinterpret_cast<const void*>(kern), SLOT + ((LNF != 0 || a.act == 1) ? 0 : 8), Geom16<MT>::LDS); e != hipSuccess) return e; \
        Split16LinearArgs b = a;                                                                                               \
        b.tiles = (int)((int64_t)groups * (a.M / (64 * MT)) * (a.N / 128));                                                    \
        b.clock_probe = g_probe_out ? g_probe_out + 2 * (g_probe_next++ % g_probe_slots) : nullptr;                           \
        const unsigned grid = (unsigned)(b.tiles < cus ? b.tiles : cus);        /* persistent: at most one block per CU */      \
        b.grid = (int)grid;                                                                                                    \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), Geom16<MT>::LDS, s, b);                                                \
    }
    if (ln) {
        if (big && a.out_mode == 1) MMS_LAUNCH_SPLIT16(4, 1, 3, 4)
        else if (big) MMS_LAUNCH_SPLIT16(4, 2, 3, 5)
        else if (a.out_mode == 1) MMS_LAUNCH_SPLIT16(2, 1, 3, 6)
        else MMS_LAUNCH_SPLIT16(2, 2, 3, 7)
    } else if (big && a.out_mode == 1) MMS_LAUNCH_SPLIT16(4, 1, 0, 0)
    else if (big) MMS_LAUNCH_SPLIT16(4, 0, 0, 1)
    else if (a.out_mode == 1) MMS_LAUNCH_SPLIT16(2, 1, 0, 2)
    else MMS_LAUNCH_SPLIT16(2, 0, 0, 3)
#undef MMS_LAUNCH_SPLIT16
    return hipGetLastError();
}

}  // namespace mms
