// synth.hip -- counter-based synthetic data on the device (same hash as weather_model_amd/synth.py).
// The reference reads WRF NetCDF files (multipleRegression.py:167-187); the benchmark shapes of
// BASELINE.json are synthetic, and config 5 (2.56 TB) only exists as a generator.
#include "common.h"

namespace {

constexpr int BLOCK = 256;

__device__ __forceinline__ uint64_t hash64(uint64_t seed, uint64_t idx)
{
    uint64_t z = idx + ((seed + 1) & 0xFFFFFFFFull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float u01(uint64_t z) { return (float)(z >> 40) * 0x1.0p-24f; }

__device__ __forceinline__ float gauss(uint64_t z)
{
    const float u1 = ((float)(z >> 40) + 1.0f) * 0x1.0p-24f;            // (0,1]
    const float u2 = (float)((z >> 8) & 0xFFFFFFull) * 0x1.0p-24f;      // [0,1)
    return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

__global__ __launch_bounds__(BLOCK) void synth_uniform_kernel(float *__restrict__ out,
                                                             long long count, uint64_t seed,
                                                             uint64_t offset, float lo, float span)
{
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < count;
         i += (long long)gridDim.x * BLOCK)
        out[i] = lo + u01(hash64(seed, offset + (uint64_t)i)) * span;
}

// one thread per (row, 4-feature group): writes 16 B of X; the group-0 thread also writes Y.
__global__ __launch_bounds__(BLOCK) void synth_regression_kernel(
    float *__restrict__ X, float *__restrict__ Y, long long n, int F, int P, uint64_t seed,
    uint64_t row_offset, const float *__restrict__ beta, float intercept, float noise, int dist)
{
    for (long long r = (long long)blockIdx.x * BLOCK + threadIdx.x; r < n;
         r += (long long)gridDim.x * BLOCK) {
        const uint64_t gr = row_offset + (uint64_t)r;
        for (int p = 0; p < P; ++p) {
            float s = intercept;
            for (int f = 0; f < F; ++f) {
                const uint64_t z = hash64(seed, gr * (uint64_t)F + f);
                const float xv = dist ? gauss(z) : u01(z);
                if (p == 0) X[r * F + f] = xv;
                s = fmaf(xv, beta[f * P + p], s);
            }
            Y[r * P + p] = s + noise * gauss(hash64(seed + 2000, gr * (uint64_t)P + p));
        }
    }
}

int grid_for(long long count)
{
    long long g = (count + BLOCK - 1) / BLOCK;
    const long long cap = (long long)wm_num_cus() * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int wm_synth_uniform_f32(float *out, int64_t count, uint64_t seed, uint64_t offset,
                                    float lo, float hi, void *stream)
{
    WM_REQUIRE(out && count >= 0, WM_ERR_INVALID, "wm_synth_uniform_f32: bad arguments");
    if (count == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_synth_uniform_f32: no HIP device");
    hipLaunchKernelGGL(synth_uniform_kernel, dim3(grid_for(count)), dim3(BLOCK), 0,
                       (hipStream_t)stream, out, (long long)count, seed, offset, lo, hi - lo);
    WM_LAUNCH_CHECK("synth_uniform_kernel");
    return WM_OK;
}

extern "C" int wm_synth_regression_f32(float *X, float *Y, int64_t n, int F, int P, uint64_t seed,
                                       uint64_t row_offset, const float *beta, float intercept,
                                       float noise, int dist, void *stream)
{
    WM_REQUIRE(X && Y && beta && n >= 0 && F > 0 && P > 0, WM_ERR_INVALID,
               "wm_synth_regression_f32: bad arguments");
    if (n == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_synth_regression_f32: no HIP device");
    hipLaunchKernelGGL(synth_regression_kernel, dim3(grid_for(n)), dim3(BLOCK), 0,
                       (hipStream_t)stream, X, Y, (long long)n, F, P, seed, row_offset, beta,
                       intercept, noise, dist);
    WM_LAUNCH_CHECK("synth_regression_kernel");
    return WM_OK;
}
