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
#include "cube_walk.h"

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

// ---- rows straight from the resident volume (wm_gram_walk_f32).  The drop-ins' fit used to gather
// every row [x(t) | x(t + d_end)] ; x(t + d_mid) of the region walk into X and Y (wm_gather_cubes_f32:
// 12 B read + 12 B written per row at one variable) and then read them back here -- at scale the
// gather was 2-3x the Gram pass it fed.  With CUBE_SIZE 1 (the reference's setting,
// multipleRegression.py:35) a row is three values per variable of ONE grid cell, and consecutive rows
// of a batch are consecutive files: lane = row decodes (file, cell) from the walk's closed form
// (cube_walk.h: nine multiply-shift divisions), loads its 3 V values from the volume -- neighbours in
// a wave read neighbouring files, and every value is fetched from HBM once for the three rows that
// use it -- and accumulates exactly as gram_small_kernel does.  V <= 4 variables (F = 2 V, P = V).
// The loop is bound by the latency of its loads (a trip's loads, then its arithmetic): rows per lane
// and trip (UR) and resident waves per SIMD (WPS) are what keep bytes in flight.
template <int V>
struct WalkCfg {
    static constexpr int UR = V == 1 ? 8 : 4;
    static constexpr int WPS = V <= 2 ? 4 : 2;      // V = 3: 128 VGPRs, V = 4: 220 (6 per SIMD at V = 1: slower)
    static constexpr bool BY_ROW = V == 4;          // address per row inside the load loop (measured)
};

template <int V, bool VEC>
__global__ __launch_bounds__(BLOCK, WalkCfg<V>::WPS) void gram_walk_kernel(
    const float *__restrict__ vol, int files, int Z, int Y, int X, int d_mid, int d_end, long long rows,
    wm_walk::CubeWalk w, wm_walk::CubeStrides st, wm_walk::CubeDivs dv, double *__restrict__ partial)
{
    constexpr int F = 2 * V, P = V, UR = WalkCfg<V>::UR, CHUNK = 64 * UR;
    using C = SmallCfg<F, P>;
    __shared__ double red[WAVES][C::TOT * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long nwaves = (long long)gridDim.x * WAVES;
    const long long wave_id = (long long)blockIdx.x * WAVES + wv;
    float acc[C::NV];
#pragma unroll
    for (int v = 0; v < C::NV; ++v) acc[v] = 0.f;
    double tot[C::TOT];
#pragma unroll
    for (int s = 0; s < C::TOT; ++s) tot[s] = 0.0;
    int since = 0;
    const int dmax = d_mid > d_end ? d_mid : d_end;
    const long long nchunks = (rows + CHUNK - 1) / CHUNK;
    for (long long ch = wave_id; ch < nchunks; ch += nwaves) {
        const long long c0 = ch * CHUNK;
        const wm_walk::WalkChunk wc = wm_walk::walk_chunk((unsigned)c0, CHUNK, rows, Z, Y, X, w, st, dv);
        float z[UR][C::M];
        long long off[UR];
        if constexpr (!WalkCfg<V>::BY_ROW)
            wm_walk::walk_chunk_rows<UR>(wc, c0, lane, rows, files, Z, Y, X, dmax, w, st, dv, off);
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            float a[V], c[V], m[V];
            bool ok = c0 + u * 64 + lane < rows;
            if constexpr (WalkCfg<V>::BY_ROW) {
                const float *src = wm_walk::walk_chunk_row(wc, vol, c0 + u * 64 + lane, (unsigned)(u * 64 + lane),
                                                           rows, files, Z, Y, X, dmax, w, st, dv, ok);
                wm_walk::walk_load<V, VEC>(src, st, d_mid, d_end, a, c, m);
            } else {
                wm_walk::walk_load<V, VEC>(vol + off[u], st, d_mid, d_end, a, c, m);
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                z[u][v] = ok ? a[v] : 0.f;
                z[u][V + v] = ok ? c[v] : 0.f;
                z[u][2 * V + v] = ok ? m[v] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) small_row<F, P>(acc, z[u]);
        since += UR;
        if (since + UR > SM_FLUSH_ROWS) {
            small_flush<F, P>(acc, tot, lane);
            since = 0;
        }
    }
    small_flush<F, P>(acc, tot, lane);
#pragma unroll
    for (int s = 0; s < C::TOT; ++s) red[wv][s * 64 + lane] = tot[s];
    __syncthreads();
    for (int v = threadIdx.x; v < C::NV; v += BLOCK) {
        double s = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < WAVES; ++w2) s += red[w2][v];
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

// ---- the walk's host side, shared with the metrics kernel (linmetrics.hip)
int wm_walk_setup(const char *who, const float *vol, int files, int nvars, int Z, int Y, int X,
                  const int64_t *strides, int cz, int cy, int cx, int d_mid, int d_end, int64_t rows,
                  int batch, int64_t length, int64_t index0, int64_t r0, int64_t r1, int64_t r2,
                  int dim_z, int dim_y, int dim_x, wm_walk::CubeWalk *w, wm_walk::CubeStrides *cs,
                  wm_walk::CubeDivs *dv)
{
    WM_REQUIRE(vol && strides, WM_ERR_INVALID, "%s: null pointer", who);
    for (int a = 0; a < 5; ++a)
        WM_REQUIRE(strides[a] >= 0, WM_ERR_INVALID, "%s: negative stride %lld", who, (long long)strides[a]);
    WM_REQUIRE(files > 0 && nvars >= 1 && nvars <= 4 && Z > 0 && Y > 0 && X > 0, WM_ERR_UNSUPPORTED,
               "%s: 1..4 variables are implemented (nvars=%d): gather the rows with "
               "wm_gather_cubes_f32 and use wm_gram_f32 / wm_linear_metrics_f32", who, nvars);
    WM_REQUIRE(cz == 1 && cy == 1 && cx == 1, WM_ERR_UNSUPPORTED,
               "%s: CUBE_SIZE 1 is implemented (cube %d,%d,%d)", who, cz, cy, cx);
    WM_REQUIRE(d_mid >= 0 && d_end >= 0 && d_mid < files && d_end < files, WM_ERR_INVALID,
               "%s: file offsets (%d, %d) outside %d files", who, d_mid, d_end, files);
    WM_REQUIRE(rows >= 0 && batch > 0 && length > 0 && index0 >= 0 && r0 >= 0 && r1 >= 0 && r2 >= 0,
               WM_ERR_INVALID, "%s: bad walk state", who);
    WM_REQUIRE(dim_z > 0 && dim_y > 0 && dim_x > 0 && dim_z <= Z && dim_y <= Y && dim_x <= X,
               WM_ERR_INVALID, "%s: walk region (%d,%d,%d) leaves the grid", who, dim_z, dim_y, dim_x);
    const int dmax = d_mid > d_end ? d_mid : d_end;
    WM_REQUIRE((length - 1) * batch + batch - 1 + dmax < files, WM_ERR_INVALID,
               "%s: walk of %lld batches of %d files + offset %d needs more than the %d resident files",
               who, (long long)length, batch, dmax, files);
    const long long lim = 1ll << 31;
    const long long last_b = index0 + rows / batch + 2;
    WM_REQUIRE(rows < lim && last_b < lim && length < lim && r0 < lim && r1 < lim && r2 < lim &&
                   (length + 1) * batch < lim && batch <= (1 << 24),
               WM_ERR_UNSUPPORTED, "%s: walk counts beyond 2^31 (gather in pieces instead)", who);
    *w = {length, index0, r0, r1, r2, batch, dim_z, dim_y, dim_x};
    *cs = {strides[0], strides[1], strides[2], strides[3], strides[4]};
    *dv = {wm_walk::fast_div(1u), wm_walk::fast_div((unsigned)batch), wm_walk::fast_div((unsigned)length),
           wm_walk::fast_div((unsigned)dim_z), wm_walk::fast_div((unsigned)dim_y),
           wm_walk::fast_div((unsigned)dim_x), wm_walk::fast_div(1u), wm_walk::fast_div(1u),
           wm_walk::fast_div(1u)};
    return WM_OK;
}

extern "C" size_t wm_gram_walk_workspace_bytes(int nvars)
{
    if (nvars < 1 || nvars > 4 || wm_num_cus() <= 0) return 0;
    const int M = 3 * nvars;
    return (size_t)wm_num_cus() * 8 * (M * (M + 1) / 2 + M) * sizeof(double);     // WalkCfg::WPS <= 8
}

extern "C" int wm_gram_walk_f32(const float *vol, int files, int nvars, int Z, int Y, int X,
                                const int64_t *strides, int d_mid, int d_end, int64_t rows, int batch,
                                int64_t length, int64_t index0, int64_t r0, int64_t r1, int64_t r2,
                                int dim_z, int dim_y, int dim_x, double *acc, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    wm_walk::CubeWalk w;
    wm_walk::CubeStrides cs;
    wm_walk::CubeDivs dv;
    const int rc = wm_walk_setup("wm_gram_walk_f32", vol, files, nvars, Z, Y, X, strides, 1, 1, 1, d_mid,
                                 d_end, rows, batch, length, index0, r0, r1, r2, dim_z, dim_y, dim_x, &w,
                                 &cs, &dv);
    if (rc) return rc;
    WM_REQUIRE(acc, WM_ERR_INVALID, "wm_gram_walk_f32: null accumulator");
    if (rows == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_gram_walk_f32: no HIP device");
    WM_REQUIRE(workspace && workspace_bytes >= wm_gram_walk_workspace_bytes(nvars), WM_ERR_WORKSPACE,
               "wm_gram_walk_f32: workspace of %zu bytes needed, %zu given",
               wm_gram_walk_workspace_bytes(nvars), workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    double *partial = (double *)workspace;
    int grid = 0;                            // every workgroup resident (launch bounds)
    const bool vec = wm_walk::walk_vec_ok(nvars, cs);
#define WM_WALK(V_)                                                                              \
    grid = wm_num_cus() * WalkCfg<V_>::WPS;                                                      \
    if (vec)                                                                                     \
        hipLaunchKernelGGL((gram_walk_kernel<V_, true>), dim3(grid), dim3(BLOCK), 0, st, vol, files, Z, Y, \
                           X, d_mid, d_end, (long long)rows, w, cs, dv, partial);                \
    else                                                                                         \
        hipLaunchKernelGGL((gram_walk_kernel<V_, false>), dim3(grid), dim3(BLOCK), 0, st, vol, files, Z,  \
                           Y, X, d_mid, d_end, (long long)rows, w, cs, dv, partial)
    switch (nvars) {
    case 1: WM_WALK(1); break;
    case 2: WM_WALK(2); break;
    case 3: WM_WALK(3); break;
    default: WM_WALK(4); break;
    }
#undef WM_WALK
    WM_LAUNCH_CHECK("gram_walk_kernel");
    const int F = 2 * nvars, P = nvars, D = F + 1 + P;
    hipLaunchKernelGGL(gram_small_finalize_kernel, dim3(D * D), dim3(64), 0, st, (const double *)partial,
                       grid, F, P, (long long)rows, acc);
    WM_LAUNCH_CHECK("gram_small_finalize_kernel");
    return WM_OK;
}
