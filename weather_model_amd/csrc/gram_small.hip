// gram_small.hip -- Z'Z for narrow problems (F + P <= 15) as a pure HBM stream on the VALU.
//
// The reference's own multipleRegression fit is F = 2, P = 1 (CUBE_SIZE = 1, one variable per
// iteration: multipleRegression.py:35,43-45; model :273-286; rows x(t), x(t+6) -> x(t+3),
// :217-234) and grows as F = 2p, P = p with the cube; singleRegression's "1-predictor" plumbing
// case is F = 1, P = 1.  A row is 4(F+P) bytes and its Gram update (F+P)(F+P+1)/2 products:
// 12 B and 9 FMAs for F = 2 -- two orders of magnitude below the VALU rate at the HBM rate, and
// far too narrow for an MFMA tile (4 of 16 columns useful), so the products are plain fmaf on
// registers and the kernel is shaped like gram_f16_kernel's memory side:
//   * a lane owns R consecutive rows per step (R = 1, 2 or 4: the smallest with R*F % 4 == 0), so X
//     arrives as 16-byte non-temporal loads of whole rows and Y as the widest vector that fits;
//   * batches of U steps (about 8-12 KiB per wave) are dealt round-robin to the waves, the next
//     batch's loads are issued before the current batch's FMAs (software pipeline, loads
//     unconditional on a clamped batch index);
//   * float32 sums run over at most 1024 rows per lane, then every accumulator is summed over the
//     wave in float64 (xor butterfly: a fixed order) and kept by lane v % 64; per-workgroup float64
//     partials, fixed-order finalize (bitwise reproducible).
#include "gram_internal.h"
#include "small_stream.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int SM_FLUSH_ROWS = 1024;

template <int F, int P>
struct SmallCfg {
    static constexpr int M = F + P;                      // non-constant columns of Z = [X | 1 | Y]
    static constexpr int NP = M * (M + 1) / 2;           // products z_i z_j, i <= j
    static constexpr int NV = NP + M;                    // + column sums
    static constexpr int R0 = (F % 4 == 0) ? 1 : ((F % 2 == 0) ? 2 : 4);
    static constexpr int U0 = (NV > 100 ? 96 : 48) / (R0 * M);
    static constexpr int U = U0 < 1 ? 1 : (U0 > 8 ? 8 : U0);
    using S = wm_small::Stream<F, P, U>;                 // the memory side (small_stream.h)
    static constexpr int R = S::R;
    static constexpr int ROWS_BATCH = S::ROWS_BATCH;
    static constexpr int TOT = (NV + 63) / 64;           // float64 totals per lane
    // waves per SIMD: above ~100 accumulators two register sets of a batch no longer fit in 256
    // VGPRs (F = 10, P = 5: 537 spilled, 2.3 TB/s), so such shapes run one wave per SIMD with 512
    static constexpr int WPS = NV > 100 ? 1 : 2;
    static_assert(M <= 15, "gram_small: F + P <= 15");
};

__host__ __device__ constexpr int small_pair(int M, int i, int j)   // i <= j
{
    return i * M - (i * (i - 1)) / 2 + (j - i);
}

template <int F, int P, int NV_, int M_>
__device__ __forceinline__ void small_row(float (&acc)[NV_], const float (&z)[M_])
{
    using C = SmallCfg<F, P>;
    static_assert(NV_ == C::NV && M_ == C::M, "small_row: array sizes");
#pragma unroll
    for (int i = 0; i < C::M; ++i) {
#pragma unroll
        for (int j = i; j < C::M; ++j)
            acc[small_pair(C::M, i, j)] = fmaf(z[i], z[j], acc[small_pair(C::M, i, j)]);
        acc[C::NP + i] += z[i];
    }
}

// float32 sums -> float64: every accumulator is summed over the wave (xor butterfly, a fixed
// order) and added to the total kept by lane v % 64.
template <int F, int P, int NV_, int TOT_>
__device__ __forceinline__ void small_flush(float (&acc)[NV_], double (&tot)[TOT_], int lane)
{
#pragma unroll
    for (int v = 0; v < NV_; ++v) {
        const double s = wave_sum((double)acc[v]);
        if (lane == (v & 63)) tot[v >> 6] += s;
        acc[v] = 0.f;
        // one value at a time: left alone, hipcc runs all NV butterflies side by side (2 NV more
        // registers than the kernel has); the flush is 1 / 1024 of the rows' work either way
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int F, int P, int NV_>
__device__ __forceinline__ void small_consume(const wm_small::Buf<typename SmallCfg<F, P>::S> &b, float (&acc)[NV_])
{
    using C = SmallCfg<F, P>;
#pragma unroll
    for (int u = 0; u < C::U; ++u) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            float z[C::M];
#pragma unroll
            for (int m = 0; m < F; ++m) z[m] = b.xe(u, r, m);
#pragma unroll
            for (int m = 0; m < P; ++m) z[F + m] = b.ye(u, r, m);
            small_row<F, P>(acc, z);
        }
    }
}

// partial: [gridDim.x][NV] float64.
template <int F, int P>
__global__ __launch_bounds__(BLOCK, (SmallCfg<F, P>::WPS)) void gram_small_kernel(const float *__restrict__ X,
                                                              const float *__restrict__ Y,
                                                              long long n,
                                                              double *__restrict__ partial)
{
    using C = SmallCfg<F, P>;
    using DN = wm_small::Dense<typename C::S>;
    __shared__ double red[WAVES][C::TOT * 64];
    __shared__ __align__(16) char turn[WAVES][DN::BYTES];      // wave-private (small_stream.h)
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long nwaves = (long long)gridDim.x * WAVES;
    const long long wave_id = (long long)blockIdx.x * WAVES + wv;

    float acc[C::NV];
#pragma unroll
    for (int v = 0; v < C::NV; ++v) acc[v] = 0.f;
    double tot[C::TOT];
#pragma unroll
    for (int s = 0; s < C::TOT; ++s) tot[s] = 0.0;
    int since = 0;

    const long long nb = n / C::ROWS_BATCH;
    if (wave_id < nb) {
        // the clamped batch of an out-of-range fetch: loads stay unconditional and in bounds
        const long long spare = wave_id % nb;
        wm_small::Buf<typename C::S> ba, bb;
        wm_small::fetch_dense<typename C::S>(ba, X, Y, wave_id, lane);
        // Two batches per trip, both consumed unconditionally (with the second consume under an
        // `if`, hipcc sinks that batch's loads into the branch and waits for them on the spot);
        // an odd last batch is consumed behind the loop.
        long long b = wave_id;
        for (; b + nwaves < nb; b += 2 * nwaves) {
            wm_small::fetch_dense<typename C::S>(bb, X, Y, b + nwaves, lane);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DN::ANY) wm_small::untangle<typename C::S>(ba, turn[wv], lane);
            small_consume<F, P>(ba, acc);
            __builtin_amdgcn_sched_barrier(0);
            wm_small::fetch_dense<typename C::S>(ba, X, Y, b + 2 * nwaves < nb ? b + 2 * nwaves : spare, lane);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DN::ANY) wm_small::untangle<typename C::S>(bb, turn[wv], lane);
            small_consume<F, P>(bb, acc);
            __builtin_amdgcn_sched_barrier(0);
            since += 2 * C::R * C::U;
            if (since + 2 * C::R * C::U > SM_FLUSH_ROWS) {
                small_flush<F, P>(acc, tot, lane);
                since = 0;
            }
        }
        if (b < nb) {
            if constexpr (DN::ANY) wm_small::untangle<typename C::S>(ba, turn[wv], lane);
            small_consume<F, P>(ba, acc);
        }
    }
    small_flush<F, P>(acc, tot, lane);
    // the rows behind the last full batch (fewer than ROWS_BATCH): one row per lane
    for (long long r0 = nb * C::ROWS_BATCH + wave_id * 64; r0 < n; r0 += nwaves * 64) {
        const long long row = r0 + lane;
        float z[C::M];
#pragma unroll
        for (int m = 0; m < C::M; ++m) z[m] = 0.f;
        if (row < n) {
#pragma unroll
            for (int m = 0; m < F; ++m) z[m] = X[row * F + m];
#pragma unroll
            for (int m = 0; m < P; ++m) z[F + m] = Y[row * P + m];
        }
        small_row<F, P>(acc, z);
    }
    small_flush<F, P>(acc, tot, lane);

#pragma unroll
    for (int s = 0; s < C::TOT; ++s) red[wv][s * 64 + lane] = tot[s];
    __syncthreads();
    for (int v = threadIdx.x; v < C::NV; v += BLOCK) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[w][v];
        partial[(size_t)blockIdx.x * C::NV + v] = s;
    }
}

// acc (D x D) += fixed-order sum of the per-workgroup partials.  One wave per acc entry.
__global__ __launch_bounds__(64) void gram_small_finalize_kernel(const double *__restrict__ partial,
                                                                 int nparts, int F, int P,
                                                                 long long n,
                                                                 double *__restrict__ acc)
{
    const int M = F + P, D = M + 1, NP = M * (M + 1) / 2, NV = NP + M;
    const int e = blockIdx.x;
    const int a = e / D, b = e % D;
    const int ma = a < F ? a : (a == F ? -1 : a - 1);
    const int mb = b < F ? b : (b == F ? -1 : b - 1);
    int idx = -1;
    if (ma >= 0 && mb >= 0) idx = ma <= mb ? small_pair(M, ma, mb) : small_pair(M, mb, ma);
    else if (ma >= 0) idx = NP + ma;
    else if (mb >= 0) idx = NP + mb;
    double s = 0.0;
    if (idx >= 0) {
        for (int p = threadIdx.x; p < nparts; p += 64) s += partial[(size_t)p * NV + idx];
        s = wave_sum(s);
    } else {
        s = (double)n;
    }
    if (threadIdx.x == 0) acc[e] += s;
}

int small_grid()
{
    return wm_num_cus() * 2;     // at most 2 waves per SIMD (launch bounds): every workgroup is resident
}

}  // namespace

bool wm_gram_small_supported(const float *X, const float *Y, int F, int P)
{
#define X_(F_, P_)                                                                             \
    if (F == F_ && P == P_)                                                                    \
        return wm_small::aligned<SmallCfg<F_, P_>::S>(X, Y);
    WM_SMALL_SHAPES(X_)
#undef X_
    return false;
}

size_t wm_gram_small_workspace_bytes(int F, int P)
{
    const int M = F + P;
    if (M > 15 || wm_num_cus() <= 0) return 0;
    return (size_t)small_grid() * (M * (M + 1) / 2 + M) * sizeof(double);
}

int wm_gram_small_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                         double *partial, hipStream_t st)
{
    int grid = small_grid();
    bool launched = false;
#define X_(F_, P_)                                                                             \
    if (F == F_ && P == P_) {                                                                  \
        grid = wm_num_cus() * SmallCfg<F_, P_>::WPS;                                           \
        hipLaunchKernelGGL((gram_small_kernel<F_, P_>), dim3(grid), dim3(BLOCK), 0, st, X, Y, n,  \
                           partial);                                                           \
        launched = true;                                                                       \
    }
    WM_SMALL_SHAPES(X_)
#undef X_
    WM_REQUIRE(launched, WM_ERR_UNSUPPORTED, "gram_small: shape F=%d P=%d not instantiated", F, P);
    WM_LAUNCH_CHECK("gram_small_kernel");
    const int D = F + 1 + P;
    hipLaunchKernelGGL(gram_small_finalize_kernel, dim3(D * D), dim3(64), 0, st, partial, grid, F,
                       P, n, acc);
    WM_LAUNCH_CHECK("gram_small_finalize_kernel");
    return WM_OK;
}
