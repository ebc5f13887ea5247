// mfma_peak.hip -- sustained v_mfma_f32_32x32x2_f32 rate of this device (development tool).
// Pure register MFMA loop, 1 or 2 waves per SIMD on every CU, random-ish operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; wgs_per_cu++) {
        int grid = 256 * wgs_per_cu;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 0.37f, -0.21f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)grid * 4 * iters * 8 * 2 * 4096.0;
            printf("waves/SIMD %d rep %d: %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, rep, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
