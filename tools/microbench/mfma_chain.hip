// mfma_chain.hip -- does a chain of DEPENDENT v_mfma_f32_32x32x2_f32 (same accumulator back to back) issue at the pipe's rate?
// Register-only loops with dependency distance 1, 2 and 4 (round robin over that many accumulators), one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D>
__global__ void __launch_bounds__(256) chain_kernel(float* out, int iters, float a, float b) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float x = a + threadIdx.x, y = b + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            // 16 MFMAs per trip; accumulator index cycles with period D inside blocks of 16 / (4 / D) ... kept simple:
            const int k = (D == 1) ? (u >> 2) & 3 : (D == 2) ? ((u >> 3) << 1) | (u & 1) : (u & 3);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[k], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int D>
void run(float* out, int bpc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, grid = 256 * bpc;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain_kernel<D>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)grid * 4 * iters * 16 * 4096.0;
    printf("dependency distance %d  waves/SIMD %d  %8.1f us  %6.1f TFLOP/s\n", D, bpc, best * 1e3, flops / (best * 1e-3) / 1e12);
}
int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    for (int bpc = 1; bpc <= 2; bpc++) { run<1>(out, bpc); run<2>(out, bpc); run<4>(out, bpc); }
    return 0;
}
