// Sustained fp32 MFMA rate of the chip: every wave runs a register-only loop of independent
// v_mfma_f32_16x16x4_f32 (no LDS, no memory).  The ceiling the implicit-GEMM kernels are priced
// against in practice (clocks under matrix load are below the 2.4 GHz of the datasheet peak).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void peak_kernel(float *out, int iters, float a0, float b0)
{
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}
int main(int argc, char **argv)
{
    float *out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
        const int grid = 256 * wgs_per_cu;
        hipLaunchKernelGGL(peak_kernel<8>, dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 2.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(peak_kernel<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 8 * 2048.0;
        printf("waves/SIMD %d: %.3f ms %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
    }
    return 0;
}
