// mfma_shapes.hip -- development: the fp16 matrix shapes under the power limit.  One wave per SIMD on every CU, the B operand
// (rows) in registers, the A operand (components) re-read from LDS per tile by ds_read_b128, pseudo-random operands, K = 112
// per output (the headline D = 100 padded to the k-step): 32x32x16 x 7, against 16x16x32 x 3 + 16x16x16 x 1 and 16x16x32 x 4
// (K padded to 128).  Prints wall time and algorithmic TFLOP/s (2 x outputs x 100) for equal numbers of outputs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// MODE 0: 32x32x16 (per tile of 32 components: 2 row blocks of 32 rows x 7 k-steps = 14 MFMAs, 1024 outputs each)
// MODE 1: 16x16x32 x 3 + 16x16x16 x 1 (per tile: 2 subtiles x 4 row blocks of 16 rows x 4 = 32 MFMAs, 256 outputs each)
// MODE 2: 16x16x32 x 4
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, const _Float16 *img, int tiles, int groups)
{
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // 16 tiles x 32 components x 128 halves
    for (int i = tid; i < tiles * 32 * 128 / 8; i += 256) reinterpret_cast<h8 *>(lds)[i] = reinterpret_cast<const h8 *>(img)[i];
    __syncthreads();
    float m1 = -3e38f, m2 = -3e38f;
    for (int g = 0; g < groups; g++) {
        h8 b[2][7];                                   // 64 rows of 112 halves: 2 x 7 x 16 B per lane either way
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int s = 0; s < 7; s++)
#pragma unroll
                for (int i = 0; i < 8; i++) b[r][s][i] = (_Float16)((float)(((lane * 131 + g * 17 + r * 7 + s * 3 + i) * 2654435761u) >> 20 & 1023) * (1.f / 1024.f) - 0.5f);
        for (int t = 0; t < tiles; t++) {
            const _Float16 *T = lds + t * 32 * 128;
            if (MODE == 0) {
                h8 a[7];
#pragma unroll
                for (int s = 0; s < 7; s++) a[s] = *reinterpret_cast<const h8 *>(T + (s * 64 + lane) * 8);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 16; q++) acc[q] = 0.f;
#pragma unroll
                    for (int s = 0; s < 7; s++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[r][s], acc, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 16; q += 4) {
                        const float x1 = fmaxf(fmaxf(m1, acc[q]), acc[q + 1]);
                        const float t1 = __builtin_amdgcn_fmed3f(m1, acc[q], acc[q + 1]);
                        const float u = __builtin_amdgcn_fmed3f(x1, acc[q + 2], acc[q + 3]);
                        m1 = fmaxf(fmaxf(x1, acc[q + 2]), acc[q + 3]);
                        m2 = fmaxf(fmaxf(m2, t1), u);
                    }
                }
            } else {
#pragma unroll
                for (int st = 0; st < 2; st++) {          // subtiles of 16 components
                    h8 a[4];
#pragma unroll
                    for (int s = 0; s < 4; s++) a[s] = *reinterpret_cast<const h8 *>(T + ((st * 4 + s) * 64 + lane) * 8);
#pragma unroll
                    for (int rb = 0; rb < 4; rb++) {      // row blocks of 16 rows: the lane's B fragments (same registers, other grouping)
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < 3; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], b[rb & 1][(rb >> 1) * 3 + s], acc, 0, 0, 0);
                        if (MODE == 1) {
                            h4 a4, b4;
#pragma unroll
                            for (int i = 0; i < 4; i++) { a4[i] = a[3][i]; b4[i] = b[rb & 1][6][i + 4 * (rb >> 1)]; }
                            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
                        } else {
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[3], b[rb & 1][6], acc, 0, 0, 0);
                        }
                        const float x1 = fmaxf(fmaxf(m1, acc[0]), acc[1]);
                        const float t1 = __builtin_amdgcn_fmed3f(m1, acc[0], acc[1]);
                        const float u = __builtin_amdgcn_fmed3f(x1, acc[2], acc[3]);
                        m1 = fmaxf(fmaxf(x1, acc[2]), acc[3]);
                        m2 = fmaxf(fmaxf(m2, t1), u);
                    }
                }
            }
        }
    }
    out[blockIdx.x * 256 + tid] = m1 + m2;
}

int main()
{
    const int tiles = 16, groups = 60;
    float *out;
    _Float16 *img;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&img, tiles * 32 * 128 * 2);
    {
        _Float16 *h = (_Float16 *)malloc(tiles * 32 * 128 * 2);
        unsigned x = 12345;
        for (int i = 0; i < tiles * 32 * 128; i++) { x = x * 1664525u + 1013904223u; h[i] = (_Float16)(((x >> 9) & 1023) * (1.f / 1024.f) - 0.5f); }
        hipMemcpy(img, h, tiles * 32 * 128 * 2, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = tiles * 32 * 128 * 2;
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; rep++)
        for (int mode = 0; mode < 3; mode++) {
            hipEventRecord(e0);
            for (int l = 0; l < 20; l++) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), lds, 0, out, img, tiles, groups);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), lds, 0, out, img, tiles, groups);
                else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), lds, 0, out, img, tiles, groups);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 20;
            // outputs per launch: 256 workgroups x 4 waves x groups x tiles x 32 components x 64 rows
            const double outs = 256.0 * 4 * groups * tiles * 32 * 64;
            printf("rep %d mode %d (%s): %.3f ms  %.0f TFLOP/s algorithmic (2 x outputs x 100)\n", rep, mode,
                   mode == 0 ? "32x32x16 x 7" : mode == 1 ? "16x16x32 x 3 + 16x16x16" : "16x16x32 x 4", ms, outs * 200 / ms / 1e9);
        }
    return 0;
}
