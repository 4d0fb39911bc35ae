// valu_rate2.hip -- second calibration of the gfx950 VALU issue rate for wave64 fp32 code.  valu_rate.hip's
// "independent" case was SLP-packed by the compiler (v_pk_fma_f32), so it measured the PACKED rate; this file pins
// each instruction kind with inline asm-free but unambiguous source forms and is built with -fno-slp-vectorize:
//   scalar  16 independent v_fma_f32 chains
//   packed  8 independent v_pk_fma_f32 chains (float2 ext vectors), counted as 16 fp32 FMAs
//   rcp     16 independent v_rcp_f32
//   cnd     16 independent v_cndmask (select)
//   dpp     16 independent v_add_f32 with a quad_perm DPP operand
// Output: ns per wave-instruction per SIMD at 2, 3, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(64) rate_kernel(float* out, int iters, float a, float b) {
    float x[16];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = a + (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = f2{x[2 * i], x[2 * i + 1]};
    const f2 pa = f2{a, a}, pb = f2{b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if (KIND == 2) x[i] = __builtin_amdgcn_rcpf(x[i]);
            if (KIND == 3) x[i] = (x[i] > b) ? a : x[i];
            if (KIND == 4)
                x[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[i]), 0xB1, 0xF, 0xF, true));
        }
        if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) p[i] = __builtin_elementwise_fma(p[i], pa, pb);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, float* out, int per_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-8s", name);
    for (int wps = 2; wps <= 4; wps++) {
        int grid = 1024 * wps;
        float ms = 0.f;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("  waves/SIMD %d: %.2f ns", wps, ms * 1e6 / ((double)wps * iters * per_iter));
    }
    printf("   (per wave-instruction per SIMD)\n");
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 8 * 64 * sizeof(float));
    run<0>("scalar", out, 16);
    run<1>("packed", out, 8);
    run<2>("rcp", out, 16);
    run<3>("cnd", out, 16);
    run<4>("dpp", out, 16);
    return 0;
}
