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

// dist 2, "paired uniform": TWO 24-bit draws per hash -- elements 2k and 2k+1 take bits 63..40 and
// 39..16 of hash64(seed, k).  Half the 64-bit multiplies per float: the row generator of config 5
// was bound by them (3.9 ms per 5e7 x 64 chunk against 2.3 ms for the Gram pass it feeds).
__device__ __forceinline__ float u01_lo(uint64_t z) { return (float)((z >> 16) & 0xFFFFFFull) * 0x1.0p-24f; }
__device__ __forceinline__ float u01_paired(uint64_t seed, uint64_t e)
{
    const uint64_t z = hash64(seed, e >> 1);
    return (e & 1) ? u01_lo(z) : u01(z);
}

// dist 3, "uniform32": one 32-bit hash per element -- two 32-bit multiplies (the lowbias32 finaliser)
// where a splitmix64 costs eight quarter-rate 32-bit multiply instructions.  The 64-bit element index
// is folded into the 32-bit key with odd multipliers, so keys stay distinct within any window of 2^32
// consecutive elements and the high half selects a different sequence.  The row generator of config
// 5 was thought bound by its integer multiplies; measured 2.6 -> 2.5 ms per 5e7 x 64 chunk against
// dist 2 -- the 13 GB non-temporal write stream (5.2 TB/s) is the bound by now.
__device__ __forceinline__ uint32_t key32(uint64_t seed, uint64_t e)
{
    return (uint32_t)e + 0x9E3779B9u * (uint32_t)(e >> 32) + 0x85EBCA6Bu * (uint32_t)seed + 0x165667B1u;
}
__device__ __forceinline__ float u01_32(uint32_t k)
{
    k ^= k >> 16;
    k *= 0x7FEB352Du;
    k ^= k >> 15;
    k *= 0x846CA68Bu;
    k ^= k >> 16;
    return (float)(k >> 8) * 0x1.0p-24f;
}

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
                const uint64_t e = gr * (uint64_t)F + f;
                const float xv = dist == 3 ? u01_32(key32(seed, e))
                                 : dist == 2 ? u01_paired(seed, e) : (dist ? gauss(hash64(seed, e)) : u01(hash64(seed, e)));
                if (p == 0) X[r * F + f] = xv;
                s = fmaf(xv, beta[f * P + p], s);
            }
            Y[r * P + p] = s + noise * gauss(hash64(seed + 2000, gr * (uint64_t)P + p));
        }
    }
}

// Fast path (F a multiple of 4 with F/4 a power of two <= 64, P = 1): G = F/4 lanes share a row,
// each hashes its four features and stores them as ONE 16-byte piece, so a wave writes a contiguous
// KiB of X per instruction (the kernel above writes 4 bytes per lane at a stride of one row: 0.33
// TB/s on the 1.25e9 x 64 rows of config 5, where generation then took 16x the Gram pass).  The
// row's dot product with beta is summed across its G lanes by a butterfly; X is bit for bit what
// the per-row kernel and weather_model_amd/synth.py produce, Y agrees to float32 rounding.
template <int G>
__global__ __launch_bounds__(BLOCK) void synth_regression_rows_kernel(
    float *__restrict__ X, float *__restrict__ Y, long long n, int F, uint64_t seed,
    uint64_t row_offset, const float *__restrict__ beta, float intercept, float noise, int dist)
{
    constexpr int RPW = 64 / G;                       // rows per wave and iteration
    const int lane = threadIdx.x & 63;
    const int g = lane % G, rl = lane / G;
    const long long wave = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (BLOCK / 64);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + 4 * g);
    // a wave owns blocks of 64 consecutive rows (16 KiB of X at F = 64): G iterations of RPW rows,
    // and lane (g, rl) keeps the row sum of iteration g -- so each lane draws the Box-Muller noise
    // of exactly ONE row per block instead of every lane sitting through lane 0's logf / cospif on
    // every iteration
    for (long long blk = wave; blk * 64 < n; blk += nwaves) {
        float mine = 0.0f;
#pragma unroll(G <= 16 ? G : 1)
        for (int it = 0; it < G; ++it) {
            const long long r_ = blk * 64 + it * RPW + rl;
            const long long r = min(r_, n - 1);       // clamped: every lane takes part in the sum
            const uint64_t e0 = (row_offset + (uint64_t)r) * (uint64_t)F + (uint64_t)(4 * g);
            f32x4 xv;
            if (dist == 3) {                          // e0 is a multiple of 4: e0 .. e0 + 3 share their high half
                const uint32_t k0 = key32(seed, e0);
#pragma unroll
                for (int k = 0; k < 4; ++k) xv[k] = u01_32(k0 + (uint32_t)k);
            } else if (dist == 2) {                   // two hashes, four floats
                const uint64_t za = hash64(seed, e0 >> 1), zb = hash64(seed, (e0 >> 1) + 1);
                xv[0] = u01(za); xv[1] = u01_lo(za); xv[2] = u01(zb); xv[3] = u01_lo(zb);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint64_t z = hash64(seed, e0 + k);
                    xv[k] = dist ? gauss(z) : u01(z);
                }
            }
            float s = fmaf(xv[3], b[3], fmaf(xv[2], b[2], fmaf(xv[1], b[1], xv[0] * b[0])));
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (r_ < n) __builtin_nontemporal_store(xv, reinterpret_cast<f32x4 *>(X + r * F + 4 * g));
            mine = it == g ? s : mine;
        }
        const long long row = blk * 64 + g * RPW + rl;
        if (row < n) Y[row] = (mine + intercept) + noise * gauss(hash64(seed + 2000, row_offset + (uint64_t)row));
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
    const int G = F / 4;
    if (P == 1 && F % 4 == 0 && G <= 64 && (G & (G - 1)) == 0 &&
        (((uintptr_t)X | (uintptr_t)beta) & 15) == 0) {
        const long long waves = (n + 63) / 64;          // one wave per block of 64 rows
        const int grid = grid_for(waves * 64);
#define WM_SYNTH_ROWS(G_)                                                                      \
    hipLaunchKernelGGL(synth_regression_rows_kernel<G_>, dim3(grid), dim3(BLOCK), 0,           \
                       (hipStream_t)stream, X, Y, (long long)n, F, seed, row_offset, beta,     \
                       intercept, noise, dist)
        switch (G) {
        case 1: WM_SYNTH_ROWS(1); break;
        case 2: WM_SYNTH_ROWS(2); break;
        case 4: WM_SYNTH_ROWS(4); break;
        case 8: WM_SYNTH_ROWS(8); break;
        case 16: WM_SYNTH_ROWS(16); break;
        case 32: WM_SYNTH_ROWS(32); break;
        default: WM_SYNTH_ROWS(64); break;
        }
#undef WM_SYNTH_ROWS
        WM_LAUNCH_CHECK("synth_regression_rows_kernel");
        return WM_OK;
    }
    hipLaunchKernelGGL(synth_regression_kernel, dim3(grid_for(n)), dim3(BLOCK), 0,
                       (hipStream_t)stream, X, Y, (long long)n, F, P, seed, row_offset, beta,
                       intercept, noise, dist);
    WM_LAUNCH_CHECK("synth_regression_kernel");
    return WM_OK;
}
