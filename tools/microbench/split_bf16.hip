// split_bf16.hip -- the bounded experiment VERDICT r2 item 1(b) asks for, part 1: is an fp32 product formed from operands split
// into three bf16 planes (a = a0 + a1 + a2 exactly; six v_mfma_f32_32x32x16_bf16 products a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0,
// fp32 accumulate) at least as accurate as the exact-fp32 v_mfma_f32_32x32x2_f32 chain the policy layers use now?  And what rate
// does the matrix pipe sustain on it (register-only loop, random data)?
//
// Numerics: Y[256, 256] = X[256, K] W[256, K]^T for K = 388 (padded to 400), 512, 1024; X ~ ELU-like activations, W ~ N(0, 1/sqrt K);
// every variant against the float64 product on the host: max and rms error in units of rms(Y).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }

// lane (r, h) of a 32x32 tile: A row r0 + r, k = kb + 8 h + j.  variant: 0 fp32 MFMA chain; 1 split, one accumulator, small terms
// first; 2 split, one accumulator, large first; 3 split, two accumulators (a0b0 | the rest), summed at the end; 4 as 1 with only
// five products (a2b0 dropped: shows what the sixth is worth)
__global__ void __launch_bounds__(64) gemm_tile(const float* X, const float* W, float* Y, int N, int K, int variant) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc = {}, lo = {};
    if (variant == 0) {
        for (int k = 0; k < K; k += 2) {
            const float a = X[(size_t)(m0 + r) * K + k + h], b = W[(size_t)(n0 + r) * K + k + h];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a0, a1, a2, b0, b1, b2;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float a = X[(size_t)(m0 + r) * K + k + 8 * h + j], b = W[(size_t)(n0 + r) * K + k + 8 * h + j];
                a0[j] = (__bf16)a; a -= bf2f(a0[j]); a1[j] = (__bf16)a; a -= bf2f(a1[j]); a2[j] = (__bf16)a;
                b0[j] = (__bf16)b; b -= bf2f(b0[j]); b1[j] = (__bf16)b; b -= bf2f(b1[j]); b2[j] = (__bf16)b;
            }
            if (variant == 1 || variant == 4) {
                if (variant == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
            } else if (variant == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, lo, 0, 0, 0);
                lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, lo, 0, 0, 0);
            }
        }
        if (variant == 3)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] += lo[i];
    }
#pragma unroll
    for (int i = 0; i < 16; i++) Y[(size_t)(m0 + (i & 3) + 8 * (i >> 2) + 4 * h) * N + n0 + r] = acc[i];
}

// register-only rate loops: one wave per SIMD-slot, 2 x 2 tiles of 32 x 32 per wave, per k16 step 24 bf16 MFMAs on 12 fragments
// (the split product) -- fragments re-seeded from a rotating register set so that the data is not constant
__global__ void __launch_bounds__(256) rate_split(float* out, const float* seed, int iters) {
    f32x16 acc[4] = {};
    bf16x8 fa[6], fb[6];
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            fa[i][j] = (__bf16)seed[(threadIdx.x * 97 + i * 8 + j) & 4095];
            fb[i][j] = (__bf16)seed[(threadIdx.x * 31 + 48 + i * 8 + j) & 4095];
        }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int ta = 0; ta < 2; ta++)
#pragma unroll
            for (int tb = 0; tb < 2; tb++) {
                f32x16& c = acc[ta * 2 + tb];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta * 3 + 2], fb[tb * 3 + 0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta * 3 + 0], fb[tb * 3 + 2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta * 3 + 1], fb[tb * 3 + 1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta * 3 + 1], fb[tb * 3 + 0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta * 3 + 0], fb[tb * 3 + 1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta * 3 + 0], fb[tb * 3 + 0], c, 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same work on the 16x16x32 shape: 4 x 4 tiles of 16 x 16 per wave = the same 64 x 64 wave tile, k32 per instruction
__global__ void __launch_bounds__(256) rate_split16(float* out, const float* seed, int iters) {
    f32x4 acc[16] = {};
    bf16x8 fa[12], fb[12];
#pragma unroll
    for (int i = 0; i < 12; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            fa[i][j] = (__bf16)seed[(threadIdx.x * 97 + i * 8 + j) & 4095];
            fb[i][j] = (__bf16)seed[(threadIdx.x * 31 + 48 + i * 8 + j) & 4095];
        }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int ta = 0; ta < 4; ta++)
#pragma unroll
            for (int tb = 0; tb < 4; tb++) {
                f32x4& c = acc[ta * 4 + tb];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ta * 3 + 2], fb[tb * 3 + 0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ta * 3 + 0], fb[tb * 3 + 2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ta * 3 + 1], fb[tb * 3 + 1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ta * 3 + 1], fb[tb * 3 + 0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ta * 3 + 0], fb[tb * 3 + 1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ta * 3 + 0], fb[tb * 3 + 0], c, 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double gauss() {
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

int main() {
    srand(1234);
    const int M = 256, N = 256;
    for (int K : {400, 512, 1024}) {
        std::vector<float> x((size_t)M * K), w((size_t)N * K), y((size_t)M * N);
        for (auto& v : x) { double g = gauss(); v = (float)(g > 0 ? g : exp(g) - 1.0); }       // ELU of a unit Gaussian
        for (auto& v : w) v = (float)(gauss() / sqrt((double)K));
        std::vector<double> ref((size_t)M * N);
        double rms = 0;
        for (int i = 0; i < M; i++)
            for (int j = 0; j < N; j++) {
                double s = 0;
                for (int k = 0; k < K; k++) s += (double)x[(size_t)i * K + k] * (double)w[(size_t)j * K + k];
                ref[(size_t)i * N + j] = s;
                rms += s * s;
            }
        rms = sqrt(rms / ((double)M * N));
        float *dx, *dw, *dy;
        hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&dy, y.size() * 4);
        hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        const char* names[] = {"fp32 mfma 32x32x2 chain", "split bf16 x3, 6 products, small first", "split bf16 x3, 6 products, large first",
                               "split bf16 x3, 6 products, hi + lo accumulators", "split bf16 x3, 5 products (a2 b0 dropped)"};
        for (int variant = 0; variant < 5; variant++) {
            hipLaunchKernelGGL(gemm_tile, dim3(N / 32, M / 32), dim3(64), 0, 0, dx, dw, dy, N, K, variant);
            hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
            double emax = 0, e2 = 0, bias = 0;
            for (size_t i = 0; i < y.size(); i++) {
                const double e = (double)y[i] - ref[i];
                emax = fmax(emax, fabs(e));
                e2 += e * e;
                bias += e;
            }
            printf("K %4d  %-50s  max err %.3e  rms err %.3e  mean err %+.3e   (units of rms(Y) = %.3f)\n", K, names[variant], emax / rms,
                   sqrt(e2 / y.size()) / rms, bias / y.size() / rms, rms);
        }
        hipFree(dx); hipFree(dw); hipFree(dy);
    }

    float *out, *seed;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipMalloc(&seed, sizeof(float) * 4096);
    {
        std::vector<float> s(4096);
        for (auto& v : s) v = (float)gauss();
        hipMemcpy(seed, s.data(), 4096 * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape = 0; shape < 2; shape++)
        for (int blocks_per_cu = 1; blocks_per_cu <= 2; blocks_per_cu++)
            for (int iters : {64, 512, 8192}) {
                const int grid = 256 * blocks_per_cu;
                float best = 1e30f;
                for (int rep = 0; rep < 4; rep++) {
                    hipEventRecord(e0);
                    if (shape == 0) hipLaunchKernelGGL(rate_split, dim3(grid), dim3(256), 0, 0, out, seed, iters);
                    else hipLaunchKernelGGL(rate_split16, dim3(grid), dim3(256), 0, 0, out, seed, iters);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                // per iteration and wave: shape 0: 24 MFMAs of 32x32x16 (32768 flop each); shape 1: 96 of 16x16x32 (16384 flop each)
                const double bf16_flops = (double)grid * 4 * iters * (shape == 0 ? 24 * 32768.0 : 96 * 16384.0);
                printf("%s  waves/SIMD %d  %5d iters  %8.1f us  %7.1f TFLOP/s bf16 = %6.1f TFLOP/s of fp32-equivalent product (1/6)\n",
                       shape == 0 ? "32x32x16" : "16x16x32", blocks_per_cu, iters, best * 1e3, bf16_flops / (best * 1e-3) / 1e12,
                       bf16_flops / 6 / (best * 1e-3) / 1e12);
            }
    return 0;
}
