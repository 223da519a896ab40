// gram_quad.hip -- the Gram of Z = [X | 1 | Y] for the drop-ins' multi-variable fits whose rows are
// four equal pieces: F = 4 RX, P = 4 RY (eight variables at CUBE_SIZE 1: F = 16, P = 8;
// multipleRegression.py:273-286 with F = 2p, P = p).  gram_f16_kernel's scheme with Y as further
// operand registers: a wave reads a tile of 16 rows as ONE contiguous run of X (lane 4s+a gets
// columns RX a .. RX a + RX - 1 of row s) and one of Y, and every pair (r1 <= r2) of the R = RX + RY
// registers is the A and B operand of v_mfma_f32_4x4x1_16b_f32 -- sixteen independent 4x4 outer
// products, one per row, every MFMA flop a useful Gram entry: R (R + 1) / 2 MFMAs of 8 cycles per 16
// rows (21 at R = 6: 10.5 cycles per row) where gram_mid_kernel's 16x16x4 tiles spend 24 cycles per
// row on a Z padded to 32 columns.  At F = 16, P = 8 that moves the kernel from the MFMA pipe
// (0.67-0.69 of HBM) to the memory system.  Column sums (the ones column) are VALU adds.
// float32 sums for at most 1024 rows per accumulator; at a flush the 16 row blocks of a wave are
// added in float64 across lanes and kept in LDS (4 lane residues per value), so no float64 register
// file: 2 waves per SIMD with two batches of 8 tiles in flight.  Per-workgroup float64 partials,
// fixed-order one-wave-per-entry finalize (bitwise reproducible), as the other Gram kernels.
#include "gram_internal.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int Q_UNROLL = 8;            // tiles per batch
constexpr int Q_FLUSH_TILES = 1024;    // 1024 products per float32 accumulator

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N> struct QVec;
template <> struct QVec<1> { typedef float type; };
template <> struct QVec<2> { typedef f32x2 type; };
template <> struct QVec<4> { typedef f32x4 type; };

__host__ __device__ constexpr int q_pair(int R, int r1, int r2)      // r1 <= r2
{
    return r1 * R - (r1 * (r1 - 1)) / 2 + (r2 - r1);
}

template <int N> __device__ __forceinline__ float q_get(const typename QVec<N>::type &v, int r)
{
    if constexpr (N == 1) return v;
    else return v[r];
}

template <int R>
__device__ __forceinline__ void q_step_z(f32x4 (&g)[R * (R + 1) / 2], float (&sz)[R], const float (&z)[R])
{
#pragma unroll
    for (int r1 = 0; r1 < R; ++r1)
#pragma unroll
        for (int r2 = r1; r2 < R; ++r2)
            g[q_pair(R, r1, r2)] =
                __builtin_amdgcn_mfma_f32_4x4x1f32(z[r1], z[r2], g[q_pair(R, r1, r2)], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < R; ++r) sz[r] += z[r];
}

template <int RX, int RY>
__device__ __forceinline__ void q_step(f32x4 (&g)[(RX + RY) * (RX + RY + 1) / 2], float (&sz)[RX + RY],
                                       const typename QVec<RX>::type &x,
                                       const typename QVec<RY>::type &y)
{
    constexpr int R = RX + RY;
    float z[R];
#pragma unroll
    for (int r = 0; r < RX; ++r) z[r] = q_get<RX>(x, r);
#pragma unroll
    for (int r = 0; r < RY; ++r) z[RX + r] = q_get<RY>(y, r);
    q_step_z<R>(g, sz, z);
}

// add the wave's float32 sums into its LDS float64 sums: the 16 row blocks (lanes with equal
// lane & 3) first, across lanes, in float64
template <int NP, int R>
__device__ __forceinline__ void q_flush(f32x4 (&g)[NP], float (&sz)[R], double (*red)[4], int lane)
{
#pragma unroll
    for (int v = 0; v < NP * 4 + R; ++v) {
        double s = v < NP * 4 ? (double)g[v >> 2][v & 3] : (double)sz[v - NP * 4];
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane < 4) red[v][lane] += s;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) g[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) sz[r] = 0.f;
}

// partial: [gridDim.x][NVAL][4] float64 (value index, lane residue lane & 3)
template <int RX, int RY>
__global__ __launch_bounds__(BLOCK, 2) void gram_quad_kernel(const float *__restrict__ X,
                                                             const float *__restrict__ Y, long long n,
                                                             double *__restrict__ partial)
{
    constexpr int R = RX + RY, NP = R * (R + 1) / 2, NVAL = NP * 4 + R, PART = NVAL * 4;
    typedef typename QVec<RX>::type VX;
    typedef typename QVec<RY>::type VY;
    __shared__ double red[WAVES][NVAL][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long nwaves = (long long)gridDim.x * WAVES;
    const long long wave_id = (long long)blockIdx.x * WAVES + wv;
    const long long tiles = (n + 15) >> 4, full_tiles = n >> 4;
    for (int i = lane; i < PART; i += 64) (&red[wv][0][0])[i] = 0.0;     // wave-private: LDS runs a
                                                                         // wave's accesses in order
    f32x4 g[NP];
    float sz[R];
#pragma unroll
    for (int p = 0; p < NP; ++p) g[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) sz[r] = 0.f;

    const VX *__restrict__ Xv = reinterpret_cast<const VX *>(X);
    const VY *__restrict__ Yv = reinterpret_cast<const VY *>(Y);
    int since_flush = 0;
    // batches of Q_UNROLL tiles dealt to the waves round-robin, the next batch's loads issued before
    // this batch's MFMAs, loads unconditional on a clamped batch index: gram_f16_kernel's loop
    const long long nb = full_tiles / Q_UNROLL;
#define Q_FETCH(xv, yv, bq)                                                                      \
    {                                                                                            \
        const long long bq_ = (bq);                                                              \
        const long long t_ = (bq_ < nb ? bq_ : wave_id % (nb > 0 ? nb : 1)) * Q_UNROLL;          \
        _Pragma("unroll") for (int u = 0; u < Q_UNROLL; ++u) {                                   \
            xv[u] = __builtin_nontemporal_load(Xv + (t_ + u) * 64 + lane);                       \
            yv[u] = __builtin_nontemporal_load(Yv + (t_ + u) * 64 + lane);                       \
        }                                                                                        \
    }
#define Q_CONSUME(xv, yv)                                                                        \
    {                                                                                            \
        _Pragma("unroll") for (int u = 0; u < Q_UNROLL; ++u) q_step<RX, RY>(g, sz, xv[u], yv[u]); \
        since_flush += Q_UNROLL;                                                                 \
        if (since_flush >= Q_FLUSH_TILES) {                                                      \
            q_flush<NP, R>(g, sz, red[wv], lane);                                                \
            since_flush = 0;                                                                     \
        }                                                                                        \
    }
    if (wave_id < nb) {
        VX xa[Q_UNROLL], xb[Q_UNROLL];
        VY ya[Q_UNROLL], yb[Q_UNROLL];
        Q_FETCH(xa, ya, wave_id)
        for (long long b = wave_id; b < nb; b += 2 * nwaves) {
            Q_FETCH(xb, yb, b + nwaves)
            __builtin_amdgcn_sched_barrier(0);
            Q_CONSUME(xa, ya)
            __builtin_amdgcn_sched_barrier(0);
            Q_FETCH(xa, ya, b + 2 * nwaves)
            __builtin_amdgcn_sched_barrier(0);
            if (b + nwaves < nb) Q_CONSUME(xb, yb)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef Q_FETCH
#undef Q_CONSUME
    // the at most Q_UNROLL tiles behind the last full batch (the last may be partial): one per wave
    for (long long t = nb * Q_UNROLL + wave_id; t < tiles; t += nwaves) {
        const long long row = t * 16 + (lane >> 2);
        VX xv = {};
        VY yv = {};
        if (row < n) {
            xv = Xv[t * 64 + lane];
            yv = Yv[t * 64 + lane];
        }
        q_step<RX, RY>(g, sz, xv, yv);
    }
    q_flush<NP, R>(g, sz, red[wv], lane);
    __syncthreads();
    for (int i = threadIdx.x; i < PART; i += BLOCK) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += (&red[w][0][0])[i];
        partial[(size_t)blockIdx.x * PART + i] = s;
    }
}

// ---- any F, P with R = ceil(F / 4) + ceil(P / 4) <= 9 and any (4-byte) alignment: the same scheme on
// rows that are NOT four equal pieces (six variables: F = 12, P = 6; seven: 14, 7; nine, ten, eleven).
// Lane 4s + a still takes RX = ceil(F / 4) consecutive floats of X row s from column RX a, and RY of
// Y, as dword-aligned vector loads (the hardware takes dword-aligned dwordx2/x3/x4); where 4 RX > F the
// last lanes' pieces run into the next row -- in bounds for every row but the last, which the tail
// loop reads element by element -- and those registers are zeroed by a per-lane mask computed once.
// A wave's loads of a tile still cover one contiguous run of X (16 F floats) with a few floats read
// twice, so the memory side is gram_quad_kernel's.  gram_mid_kernel, which these shapes took before,
// pads Z to 32 columns of 16x16x4 tiles: 24 MFMA cycles per row against R (R + 1) / 2 / 2 here
// (7.5 at F=12,P=6, 18 at F=20,P=10).
template <int N>
__device__ __forceinline__ void qu_load(const float *__restrict__ p, float *__restrict__ dst)
{
    if constexpr (N >= 4) {
        const f4u v = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(p));
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = v[k];
        qu_load<N - 4>(p + 4, dst + 4);
    } else if constexpr (N == 3) {
        const f2u v = __builtin_nontemporal_load(reinterpret_cast<const f2u *>(p));
        dst[0] = v[0];
        dst[1] = v[1];
        dst[2] = __builtin_nontemporal_load(p + 2);
    } else if constexpr (N == 2) {
        const f2u v = __builtin_nontemporal_load(reinterpret_cast<const f2u *>(p));
        dst[0] = v[0];
        dst[1] = v[1];
    } else if constexpr (N == 1) {
        dst[0] = __builtin_nontemporal_load(p);
    }
}

template <int RX_, int RY_>
struct QuCfg {
    static constexpr int RX = RX_, RY = RY_, R = RX + RY;
    static constexpr int NP = R * (R + 1) / 2, NVAL = NP * 4 + R, PART = NVAL * 4;
    // tiles per batch: two batches of U (R floats each) beside 4 NP accumulator registers
    static constexpr int U = R <= 6 ? 8 : (R <= 8 ? 4 : 2);
    // beyond R = 9 the 4 NP accumulator registers alone pass 256; F=32,P=16 (R = 12) with one wave per
    // SIMD and 512 registers was tried: 360 VGPRs, no spill reported, 71 ms against gram_mid's 1.9
    static_assert(R <= 9, "gram_quadu: ceil(F/4) + ceil(P/4) <= 9");
};

// F in (4 RX - 4, 4 RX], P in (4 RY - 4, 4 RY]: one instantiation per (RX, RY), F and P at run time
template <int RX, int RY>
__global__ __launch_bounds__(BLOCK, 2) void gram_quadu_kernel(const float *__restrict__ X,
                                                              const float *__restrict__ Y, long long n,
                                                              int F, int P, double *__restrict__ partial)
{
    using C = QuCfg<RX, RY>;
    constexpr int R = C::R, NP = C::NP, U = C::U;
    __shared__ double red[WAVES][C::NVAL][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int a = lane & 3, srow = lane >> 2;
    const long long nwaves = (long long)gridDim.x * WAVES;
    const long long wave_id = (long long)blockIdx.x * WAVES + wv;
    const long long tiles = (n + 15) >> 4;
    // tiles whose vector loads stay inside the arrays: all full tiles but the one holding the last row
    const bool padded = 4 * RX != F || 4 * RY != P;
    const long long safe_tiles = padded ? (n - 1 > 0 ? (n - 1) >> 4 : 0) : (n >> 4);
    for (int i = lane; i < C::PART; i += 64) (&red[wv][0][0])[i] = 0.0;
    f32x4 g[NP];
    float sz[R];
#pragma unroll
    for (int p = 0; p < NP; ++p) g[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) sz[r] = 0.f;
    bool live[R];                                  // this lane's register holds a column of Z
#pragma unroll
    for (int r = 0; r < RX; ++r) live[r] = a * RX + r < F;
#pragma unroll
    for (int r = 0; r < RY; ++r) live[RX + r] = a * RY + r < P;
    const float *__restrict__ xl = X + (long long)srow * F + a * RX;
    const float *__restrict__ yl = Y + (long long)srow * P + a * RY;
    const long long xstep = 16ll * F, ystep = 16ll * P;           // floats per tile
    int since_flush = 0;
    const long long nb = safe_tiles / U;
    auto fetch = [&](float(&z)[U][R], long long bq) __attribute__((always_inline)) {
        const long long t_ = (bq < nb ? bq : wave_id % (nb > 0 ? nb : 1)) * U;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            qu_load<RX>(xl + (t_ + u) * xstep, &z[u][0]);
            qu_load<RY>(yl + (t_ + u) * ystep, &z[u][RX]);
        }
    };
    auto consume = [&](float(&z)[U][R]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) z[u][r] = live[r] ? z[u][r] : 0.f;
            q_step_z<R>(g, sz, z[u]);
        }
        since_flush += U;
        if (since_flush >= Q_FLUSH_TILES) {
            q_flush<NP, R>(g, sz, red[wv], lane);
            since_flush = 0;
        }
    };
    if (wave_id < nb) {
        float za[U][R], zb[U][R];
        fetch(za, wave_id);
        for (long long b = wave_id; b < nb; b += 2 * nwaves) {
            fetch(zb, b + nwaves);
            __builtin_amdgcn_sched_barrier(0);
            consume(za);
            __builtin_amdgcn_sched_barrier(0);
            fetch(za, b + 2 * nwaves);
            __builtin_amdgcn_sched_barrier(0);
            if (b + nwaves < nb) consume(zb);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the tiles behind the last full batch (and the one with the last row): element by element
    for (long long t = nb * U + wave_id; t < tiles; t += nwaves) {
        const long long row = t * 16 + srow;
        float z[R];
#pragma unroll
        for (int r = 0; r < RX; ++r) z[r] = (row < n && a * RX + r < F) ? X[row * F + a * RX + r] : 0.f;
#pragma unroll
        for (int r = 0; r < RY; ++r) z[RX + r] = (row < n && a * RY + r < P) ? Y[row * P + a * RY + r] : 0.f;
        q_step_z<R>(g, sz, z);
    }
    q_flush<NP, R>(g, sz, red[wv], lane);
    __syncthreads();
    for (int i = threadIdx.x; i < C::PART; i += BLOCK) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += (&red[w][0][0])[i];
        partial[(size_t)blockIdx.x * C::PART + i] = s;
    }
}

// acc (D x D, D = F + 1 + P) += fixed-order sum of the per-workgroup partials; one wave per entry.
// Column c of Z: c < F -> lane residue c / RX, register c % RX; c = F -> the ones; c > F -> residue
// (c - F - 1) / RY, register RX + (c - F - 1) % RY.  mfma(A = reg r1, B = reg r2): accumulator
// register a', lane residue b' holds sum_rows Z[row][col(a', r1)] * Z[row][col(b', r2)].
__global__ __launch_bounds__(64) void gram_quad_finalize_kernel(const double *__restrict__ partial,
                                                                int nparts, long long n, int RX, int RY,
                                                                int F, int P, double *__restrict__ acc)
{
    const int R = RX + RY, NP = R * (R + 1) / 2, PART = (NP * 4 + R) * 4;
    const int D = F + 1 + P;                     // F <= 4 RX, P <= 4 RY (gram_quadu: padded pieces)
    const int e = blockIdx.x;
    const int ci = e / D, cj = e % D;
    int ai = 0, ri = -1, aj = 0, rj = -1;            // r = -1: the ones column
    if (ci < F) { ai = ci / RX; ri = ci % RX; }
    else if (ci > F) { ai = (ci - F - 1) / RY; ri = RX + (ci - F - 1) % RY; }
    if (cj < F) { aj = cj / RX; rj = cj % RX; }
    else if (cj > F) { aj = (cj - F - 1) / RY; rj = RX + (cj - F - 1) % RY; }
    int idx = -1;
    if (ri >= 0 && rj >= 0) {
        if (ri <= rj) idx = (q_pair(R, ri, rj) * 4 + ai) * 4 + aj;
        else          idx = (q_pair(R, rj, ri) * 4 + aj) * 4 + ai;
    } else if (ri >= 0) idx = (NP * 4 + ri) * 4 + ai;
    else if (rj >= 0) idx = (NP * 4 + rj) * 4 + aj;
    double s;
    if (idx >= 0) {
        s = 0.0;
        for (int p = threadIdx.x; p < nparts; p += 64) s += partial[(size_t)p * PART + idx];
        s = wave_sum(s);
    } else {
        s = (double)n;
    }
    if (threadIdx.x == 0) acc[e] += s;
}

int quad_grid() { return wm_num_cus() * 2; }

constexpr int q_part(int RX, int RY)
{
    return (((RX + RY) * (RX + RY + 1) / 2) * 4 + RX + RY) * 4;
}

bool quad_shape(int F, int P, int *RX, int *RY)
{
    if (F % 4 || P % 4) return false;
    const int rx = F / 4, ry = P / 4;
    if (!((rx == 4 && ry == 2) || (rx == 4 && ry == 1) || (rx == 2 && ry == 2) || (rx == 2 && ry == 1)))
        return false;
    *RX = rx;
    *RY = ry;
    return true;
}

}  // namespace

bool wm_gram_quad_supported(const float *X, const float *Y, int F, int P)
{
    int rx, ry;
    if (!quad_shape(F, P, &rx, &ry)) return false;
    static const bool off = [] { const char *e = getenv("WM_GRAM_QUAD"); return e && e[0] == '0'; }();
    if (off) return false;                     // probe switch: A/B against gram_mid / gram_small
    return (((uintptr_t)X) & (size_t)(4 * rx - 1)) == 0 && (((uintptr_t)Y) & (size_t)(4 * ry - 1)) == 0;
}

size_t wm_gram_quad_workspace_bytes(int F, int P)
{
    int rx, ry;
    if (!quad_shape(F, P, &rx, &ry) || wm_num_cus() <= 0) return 0;
    return (size_t)quad_grid() * q_part(rx, ry) * sizeof(double);
}

int wm_gram_quad_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                        double *partial, hipStream_t st)
{
    int rx = 0, ry = 0;
    WM_REQUIRE(quad_shape(F, P, &rx, &ry), WM_ERR_UNSUPPORTED, "gram_quad: F=%d P=%d not instantiated",
               F, P);
    const int grid = quad_grid();
#define Q_CASE(RX_, RY_)                                                                         \
    if (rx == RX_ && ry == RY_)                                                                  \
        hipLaunchKernelGGL((gram_quad_kernel<RX_, RY_>), dim3(grid), dim3(BLOCK), 0, st, X, Y, n, partial);
    Q_CASE(4, 2)
    Q_CASE(4, 1)
    Q_CASE(2, 2)
    Q_CASE(2, 1)
#undef Q_CASE
    WM_LAUNCH_CHECK("gram_quad_kernel");
    const int D = F + 1 + P;
    hipLaunchKernelGGL(gram_quad_finalize_kernel, dim3(D * D), dim3(64), 0, st,
                       (const double *)partial, grid, n, rx, ry, F, P, acc);
    WM_LAUNCH_CHECK("gram_quad_finalize_kernel");
    return WM_OK;
}

// the padded / unaligned variant: every F, P with ceil(F / 4) + ceil(P / 4) <= 9 and P <= 12 that the
// narrow-row kernel and the aligned kernel above leave: one instantiation per (RX, RY)
#define WM_QUADU_RXRY(X_)                                                                        \
    X_(1, 1) X_(2, 1) X_(3, 1) X_(4, 1) X_(5, 1) X_(6, 1) X_(7, 1) X_(8, 1)                      \
    X_(1, 2) X_(2, 2) X_(3, 2) X_(4, 2) X_(5, 2) X_(6, 2) X_(7, 2)                               \
    X_(1, 3) X_(2, 3) X_(3, 3) X_(4, 3) X_(5, 3) X_(6, 3)

bool wm_gram_quadu_supported(int F, int P)
{
    static const bool off = [] { const char *e = getenv("WM_GRAM_QUADU"); return e && e[0] == '0'; }();
    if (off || F < 1 || P < 1) return false;   // probe switch: A/B against gram_mid
    const int rx = (F + 3) / 4, ry = (P + 3) / 4;
    return ry <= 3 && rx + ry <= 9;
}

size_t wm_gram_quadu_workspace_bytes(int F, int P)
{
    if (!wm_gram_quadu_supported(F, P) || wm_num_cus() <= 0) return 0;
    return (size_t)quad_grid() * q_part((F + 3) / 4, (P + 3) / 4) * sizeof(double);
}

int wm_gram_quadu_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                         double *partial, hipStream_t st)
{
    const int grid = quad_grid();
    const int rx = (F + 3) / 4, ry = (P + 3) / 4;
    bool launched = false;
#define X_(RX_, RY_)                                                                             \
    if (rx == RX_ && ry == RY_) {                                                                \
        hipLaunchKernelGGL((gram_quadu_kernel<RX_, RY_>), dim3(grid), dim3(BLOCK), 0, st, X, Y, n, F, P, \
                           partial);                                                             \
        launched = true;                                                                         \
    }
    WM_QUADU_RXRY(X_)
#undef X_
    WM_REQUIRE(launched, WM_ERR_UNSUPPORTED, "gram_quadu: F=%d P=%d not instantiated", F, P);
    WM_LAUNCH_CHECK("gram_quadu_kernel");
    const int D = F + 1 + P;
    hipLaunchKernelGGL(gram_quad_finalize_kernel, dim3(D * D), dim3(64), 0, st,
                       (const double *)partial, grid, n, rx, ry, F, P, acc);
    WM_LAUNCH_CHECK("gram_quad_finalize_kernel");
    return WM_OK;
}
