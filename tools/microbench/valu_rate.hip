// valu_rate.hip -- calibrates the per-SIMD VALU issue rate on gfx950 for wave64 fp32 code (no packed math):
// N waves per SIMD each running ITER iterations of 16 independent v_fma_f32 (or a dependent chain).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int DEP>
__global__ void __launch_bounds__(64) fma_kernel(float* out, int iters, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = a + (float)(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (DEP) x[(i + 1) & 15] = __builtin_fmaf(x[i], a, b);          // one dependent chain
            else x[i] = __builtin_fmaf(x[i], a, b);                          // 16 independent chains
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 16 * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int dep = 0; dep < 2; dep++)
        for (int wps = 1; wps <= 8; wps *= 2) {
            int grid = 1024 * wps;   // 256 CUs x 4 SIMDs x wps one-wave workgroups
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (dep) hipLaunchKernelGGL(fma_kernel<1>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(fma_kernel<0>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double instr_per_simd = (double)wps * iters * 16;
            printf("%s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", dep ? "dependent  " : "independent", wps, ms,
                   ms * 1e6 / instr_per_simd);
        }
    return 0;
}
