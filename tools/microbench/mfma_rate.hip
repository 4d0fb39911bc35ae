// mfma_rate.hip -- sustained rate of v_mfma_f32_32x32x2_f32 (the instruction of the policy-layer kernels) on all 256 CUs:
// register-only loop, 4 independent accumulators per wave, 1 / 2 waves per SIMD, for ~100 us and for ~2 ms (clock behaviour
// under sustained matrix load).  Prints TFLOP/s and the core clock it implies at 64 cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) mfma_kernel(float* out, int iters, float a, float b) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float x = a + threadIdx.x, y = b + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; blocks_per_cu++) {
        for (int iters : {128, 2048, 32768}) {
            const int grid = 256 * blocks_per_cu;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_kernel, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double mfmas_per_simd = (double)iters * 16 * blocks_per_cu;          // one wave of each block per SIMD
            const double flops = (double)grid * 4 * iters * 16 * 4096.0;
            printf("waves/SIMD %d  %6d iters  %8.1f us  %6.1f TFLOP/s  implied clock %.2f GHz\n", blocks_per_cu, iters, best * 1e3,
                   flops / (best * 1e-3) / 1e12, mfmas_per_simd * 64.0 / (best * 1e-3) / 1e9);
        }
    }
    // the same short launch (2 waves per SIMD, 128 iterations) 40 times back to back on one stream: does every launch start its
    // clock ramp again, or only the first one after idle?
    {
        const int iters = 128, grid = 512, n = 40;
        hipLaunchKernelGGL(mfma_kernel, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(mfma_kernel, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)n * grid * 4 * iters * 16 * 4096.0;
        printf("%d launches of 128 iters back to back: %.1f us each, %.1f TFLOP/s, implied clock %.2f GHz\n", n, ms * 1e3 / n,
               flops / (ms * 1e-3) / 1e12, (double)iters * 16 * 2 * 64.0 / (ms * 1e-3 / n) / 1e9);
    }
    return 0;
}
