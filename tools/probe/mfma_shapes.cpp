// Sustained fp32 MFMA rate by instruction shape: register-only loops of independent
// v_mfma_f32_16x16x4_f32 (8 accumulators of 4 registers) and v_mfma_f32_32x32x2_f32 (4 accumulators
// of 16 registers, and 2), with operands that differ per lane and change every iteration (sign
// flips keep the sums bounded) -- VERDICT r2 item 7: the clock the chip holds may depend on the
// shape.  Prints TFLOP/s and the shader clock (cycles / 100 MHz wall ticks of one wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void peak_kernel(float *out, long long *clk, int iters, const float *seed)
{
    const float a0 = seed[threadIdx.x], b0 = seed[256 + threadIdx.x];
    float a = a0, b = b0;
    const long long c0 = clock64(), w0 = wall_clock64();
    float s = 0;
    if (SHAPE == 16) {
        f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            b = -b; a = a * 1.0000001f;
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            b = -b; a = a * 1.0000001f;
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int SHAPE, int NACC>
void run(const char *name, float *out, long long *clk, const float *seed, int wps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wps;
    const int iters = 20000 * 8 / NACC / (SHAPE == 16 ? 1 : 2);     // equal flops per wave
    hipLaunchKernelGGL((peak_kernel<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, out, clk, 100, seed);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((peak_kernel<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, out, clk, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double flops = (double)grid * 4 * iters * NACC * (SHAPE == 16 ? 2048.0 : 4096.0);
    printf("%-22s waves/SIMD %d: %.3f ms %.1f TFLOP/s  clock %.0f MHz\n", name, wps, best, flops / best / 1e9,
           h[1] ? 100.0 * h[0] / h[1] : 0.0);
}

int main()
{
    float *out, *seed; long long *clk;
    hipMalloc(&out, 4096); hipMalloc(&clk, 64); hipMalloc(&seed, 512 * 4);
    float hs[512];
    srand(1);
    for (int i = 0; i < 512; ++i) hs[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round)
        for (int wps = 1; wps <= 2; ++wps) {
            run<16, 8>("16x16x4 x 8 acc", out, clk, seed, wps);
            run<32, 4>("32x32x2 x 4 acc", out, clk, seed, wps);
            run<32, 2>("32x32x2 x 2 acc", out, clk, seed, wps);
            run<16, 4>("16x16x4 x 4 acc", out, clk, seed, wps);
        }
    return 0;
}
