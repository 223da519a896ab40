// gram.hip -- X'X / X'Y Gram build on fp32 MFMA, float64 combination, and the normal-equation solve.
//
// Replaces the iterative Adam/MSE fit of the reference's linear Dense stacks
// (multipleRegression.py:273-298 + train_step :98-106; singleRegression.py:74-96) by the closed
// form they converge to: ordinary least squares with intercept.  With Z = [X | 1 | Y] the whole
// problem is the D x D Gram Z'Z (D = F+1+P); see include/wm.h.
//
// Kernels
//   gram_f16_kernel     F = 16, P = 1 (BASELINE config 2).  HBM-bound: 68 B per row.  A wave reads
//                       16 rows per 1 KiB dwordx4 load; lane 4s+a then holds features 4a..4a+3 of
//                       row s, which is exactly the operand shape of v_mfma_f32_4x4x1_16b_f32
//                       (16 independent 4x4 outer products, one per row): A = register r1,
//                       B = register r2 gives Gram entries (4a+r1, 4b+r2).  The 10 register pairs
//                       r1 <= r2 cover the symmetric 16x16 block with every MFMA flop useful and no
//                       LDS round trip.  X'y, column sums, y'y are VALU FMAs on the same registers.
//   gram_x64_kernel     F = 64, P = 1 (BASELINE config 5): the same idea on v_mfma_f32_16x16x4_f32.
//   gram_small.hip      F + P <= 15 (the reference's multipleRegression shape F = 2, P = 1): VALU stream.
//   gram_syrk.hip       every other shape (singleRegression's F = 432, P = 216): LDS-staged blocked
//                       SYRK on MFMA.
//   all keep float32 sums for at most 1024 rows per accumulator, then add them into float64
//   registers; per-block float64 partials go to the workspace and a finalize kernel adds them to
//   acc in a fixed order.
//   normal_solve_*      float64 Cholesky of the centred system: one wave (F <= 24), one workgroup
//                       in LDS, or blocked over many workgroups (solve_blocked.hip).
#include "gram_internal.h"
#include "solve_internal.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;

// ---------------------------------------------------------------------------- F = 16, P = 1
constexpr int F16_NPAIR = 10;                         // (r1,r2), r1 <= r2
constexpr int F16_NVAL = F16_NPAIR * 4 + 4 + 4 + 2;   // per-lane values: 40 mfma + xy[4] + sx[4] + sy + syy
constexpr int F16_PART = F16_NVAL * 4;                // per-block doubles (4 lane residues each)
constexpr int F16_FLUSH_TILES = 1024;                 // 1024 tiles = 1024 products per accumulator
constexpr int F16_UNROLL = 8;

__device__ __forceinline__ constexpr int f16_pair(int r1, int r2)
{
    // index of (r1,r2), r1 <= r2, in the order (0,0)(0,1)(0,2)(0,3)(1,1)(1,2)(1,3)(2,2)(2,3)(3,3)
    return r1 * 4 - (r1 * (r1 - 1)) / 2 + (r2 - r1);
}

struct F16Acc {
    f32x4 g[F16_NPAIR];
    float xy[4], sx[4], sy, syy;
};

__device__ __forceinline__ void f16_zero(F16Acc &a)
{
#pragma unroll
    for (int p = 0; p < F16_NPAIR; ++p) a.g[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) { a.xy[r] = 0.f; a.sx[r] = 0.f; }
    a.sy = 0.f;
    a.syy = 0.f;
}

__device__ __forceinline__ void f16_step(F16Acc &a, const f32x4 x, const float y)
{
    // 16 outer products per instruction: block s (lanes 4s..4s+3) is row s of the tile.
    a.g[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, x.x, a.g[0], 0, 0, 0);
    a.g[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, x.y, a.g[1], 0, 0, 0);
    a.g[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, x.z, a.g[2], 0, 0, 0);
    a.g[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, x.w, a.g[3], 0, 0, 0);
    a.g[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, x.y, a.g[4], 0, 0, 0);
    a.g[5] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, x.z, a.g[5], 0, 0, 0);
    a.g[6] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, x.w, a.g[6], 0, 0, 0);
    a.g[7] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, x.z, a.g[7], 0, 0, 0);
    a.g[8] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, x.w, a.g[8], 0, 0, 0);
    a.g[9] = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, x.w, a.g[9], 0, 0, 0);
    a.xy[0] = fmaf(x.x, y, a.xy[0]);
    a.xy[1] = fmaf(x.y, y, a.xy[1]);
    a.xy[2] = fmaf(x.z, y, a.xy[2]);
    a.xy[3] = fmaf(x.w, y, a.xy[3]);
    a.sx[0] += x.x;
    a.sx[1] += x.y;
    a.sx[2] += x.z;
    a.sx[3] += x.w;
    a.sy += y;
    a.syy = fmaf(y, y, a.syy);
}

__device__ __forceinline__ void f16_flush(F16Acc &a, double (&d)[F16_NVAL])
{
#pragma unroll
    for (int p = 0; p < F16_NPAIR; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[p * 4 + r] += (double)a.g[p][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d[40 + r] += (double)a.xy[r];
        d[44 + r] += (double)a.sx[r];
    }
    d[48] += (double)a.sy;
    d[49] += (double)a.syy;
    f16_zero(a);
}

// partial: [gridDim.x][F16_NVAL][4] float64 (value index, lane residue lane&3).
__global__ __launch_bounds__(BLOCK, 2) void gram_f16_kernel(const float *__restrict__ X,
                                                            const float *__restrict__ Y,
                                                            long long n,
                                                            double *__restrict__ partial)
{
    __shared__ double red[WAVES][F16_NVAL][4];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const long long nwaves = (long long)gridDim.x * WAVES;
    const long long wave_id = (long long)blockIdx.x * WAVES + wv;
    const long long tiles = (n + 15) >> 4;          // 16-row tiles, the last may be partial
    const long long full_tiles = n >> 4;

    F16Acc a;
    f16_zero(a);
    double d[F16_NVAL];
#pragma unroll
    for (int v = 0; v < F16_NVAL; ++v) d[v] = 0.0;

    const f32x4 *__restrict__ Xv = reinterpret_cast<const f32x4 *>(X);
    const int rsub = lane >> 2;
    int since_flush = 0;
    // Batches of F16_UNROLL tiles (8 KiB of X) are dealt to the waves round-robin, so the waves that
    // are active together read ONE compact window that moves through X.  Giving every wave a
    // contiguous range instead (ranges 3.3 MB apart) was 4-5 % slower on the same box (1.131 ->
    // 1.078 ms at n = 1e8, tools/probe/gram16_time.py), and runs of 4 or 16 batches per wave far
    // slower (1.43 / 1.35 ms); batch sizes of 4 to 12 tiles and 3 or 4 waves per SIMD (float64
    // sums moved to LDS) all ran within 1 % of this.
    const long long nb = full_tiles / F16_UNROLL;
    // Software pipeline: the loads of the wave's next batch are issued before the MFMAs of the
    // current one, so a wave always has 8.5 KB in flight (1.097 -> 1.05 ms on the box above; with
    // the MFMAs removed the same loads take 0.97 ms, without the Y loads 0.99).  Loads are
    // unconditional on a clamped batch index (a branch around a load drains vmcnt(0) at the join);
    // 256 VGPRs, no spills, 2 waves per SIMD.
    auto fetch = [&](f32x4 (&xv)[F16_UNROLL], float (&yv)[F16_UNROLL], long long bq) {
        const long long t = (bq < nb ? bq : wave_id % (nb > 0 ? nb : 1)) * F16_UNROLL;
#pragma unroll
        for (int u = 0; u < F16_UNROLL; ++u) {
            // non-temporal: a stream read once; with the nt bit a read runs at 7.0 TB/s on this
            // chip against 6.3 without (tools/probe/read_probe.hip)
            xv[u] = __builtin_nontemporal_load(Xv + (t + u) * 64 + lane);
            yv[u] = __builtin_nontemporal_load(Y + (t + u) * 16 + rsub);
        }
    };
    auto consume = [&](const f32x4 (&xv)[F16_UNROLL], const float (&yv)[F16_UNROLL]) {
#pragma unroll
        for (int u = 0; u < F16_UNROLL; ++u) f16_step(a, xv[u], yv[u]);
        since_flush += F16_UNROLL;
        if (since_flush >= F16_FLUSH_TILES) {
            f16_flush(a, d);
            since_flush = 0;
        }
    };
    if (wave_id < nb) {
        f32x4 xa[F16_UNROLL], xb[F16_UNROLL];
        float ya[F16_UNROLL], yb[F16_UNROLL];
        fetch(xa, ya, wave_id);
        for (long long b = wave_id; b < nb; b += 2 * nwaves) {
            fetch(xb, yb, b + nwaves);
            __builtin_amdgcn_sched_barrier(0);
            consume(xa, ya);
            __builtin_amdgcn_sched_barrier(0);
            fetch(xa, ya, b + 2 * nwaves);
            __builtin_amdgcn_sched_barrier(0);
            if (b + nwaves < nb) consume(xb, yb);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the at most F16_UNROLL tiles behind the last full batch (the last may be partial): one per wave
    for (long long t = nb * F16_UNROLL + wave_id; t < tiles; t += nwaves) {
        const long long row = t * 16 + rsub;
        f32x4 xv = (f32x4){0.f, 0.f, 0.f, 0.f};
        float yv = 0.f;
        if (row < n) {
            xv = Xv[t * 64 + lane];
            yv = Y[row];
        }
        f16_step(a, xv, yv);
    }
    f16_flush(a, d);

    // sum the 16 lane blocks (rows) of the wave: lanes with equal lane&3 hold the same entries
#pragma unroll
    for (int v = 0; v < F16_NVAL; ++v) {
        double s = d[v];
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane < 4) red[wv][v][lane] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F16_PART; i += BLOCK) {
        const int v = i >> 2, b = i & 3;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[w][v][b];
        partial[(size_t)blockIdx.x * F16_PART + i] = s;
    }
}

// acc (18x18) += fixed-order sum of the per-block partials.  One wave per acc entry.
__global__ __launch_bounds__(64) void gram_f16_finalize_kernel(const double *__restrict__ partial,
                                                               int nparts, long long n,
                                                               double *__restrict__ acc)
{
    constexpr int D = 18;
    const int e = blockIdx.x;
    const int i = e / D, j = e % D;
    int idx = -1;   // index into the F16_PART values of one block
    if (i < 16 && j < 16) {
        const int a = i >> 2, r1 = i & 3, b = j >> 2, r2 = j & 3;
        // mfma(A = reg r1, B = reg r2): register a', lane residue b' holds
        //   sum_rows X[row][4a'+r1] * X[row][4b'+r2]
        if (r1 <= r2) idx = (f16_pair(r1, r2) * 4 + a) * 4 + b;
        else          idx = (f16_pair(r2, r1) * 4 + b) * 4 + a;
    } else if (i < 16 && j == 16) idx = (44 + (i & 3)) * 4 + (i >> 2);
    else if (j < 16 && i == 16) idx = (44 + (j & 3)) * 4 + (j >> 2);
    else if (i < 16 && j == 17) idx = (40 + (i & 3)) * 4 + (i >> 2);
    else if (j < 16 && i == 17) idx = (40 + (j & 3)) * 4 + (j >> 2);
    else if ((i == 16 && j == 17) || (i == 17 && j == 16)) idx = 48 * 4;   // sum y (residue 0)
    else if (i == 17 && j == 17) idx = 49 * 4;                             // sum y^2
    double s = 0.0;
    if (idx >= 0) {
        for (int p = threadIdx.x; p < nparts; p += 64) s += partial[(size_t)p * F16_PART + idx];
        s = wave_sum(s);
    } else {
        s = (double)n;   // (16,16): the row count
    }
    if (threadIdx.x == 0) acc[e] += s;
}

// ---------------------------------------------------------------------------- F = 64, P = 1
// BASELINE config 5.  A row is 256 B, so one 1 KiB dwordx4 wave-load holds 4 rows: lane (i, g) =
// (lane & 15, lane >> 4) gets features 4i..4i+3 of row g.  Register q of all 64 lanes is then both
// the A and the B operand of v_mfma_f32_16x16x4_f32 for the feature set S_q = {4i+q}: A = reg q1,
// B = reg q2 accumulates the 16x16 block of the Gram between S_q1 and S_q2 (row a, column b =
// features 4a+q1, 4b+q2), k = the 4 rows.  The 10 pairs q1 <= q2 cover the symmetric 64x64 block
// (10/16 of the full product), with no LDS, no transposition and one coalesced load per 4 rows.
constexpr int X64_NPAIR = 10;
constexpr int X64_NVAL = X64_NPAIR * 4 + 4 + 4 + 2;     // per-lane float64 values
constexpr int X64_PART = X64_NPAIR * 256 + 64 + 64 + 2;  // per-wave doubles written out
constexpr int X64_FLUSH_STEPS = 256;                     // 256 steps x 4 rows = 1024 products
constexpr int X64_UNROLL = 8;

struct X64Acc {
    f32x4 g[X64_NPAIR];
    float xy[4], sx[4], sy, syy;
};

__device__ __forceinline__ void x64_zero(X64Acc &a)
{
#pragma unroll
    for (int p = 0; p < X64_NPAIR; ++p) a.g[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) { a.xy[r] = 0.f; a.sx[r] = 0.f; }
    a.sy = 0.f;
    a.syy = 0.f;
}

__device__ __forceinline__ void x64_step(X64Acc &a, const f32x4 x, const float y)
{
    a.g[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, x.x, a.g[0], 0, 0, 0);
    a.g[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, x.y, a.g[1], 0, 0, 0);
    a.g[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, x.z, a.g[2], 0, 0, 0);
    a.g[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, x.w, a.g[3], 0, 0, 0);
    a.g[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, x.y, a.g[4], 0, 0, 0);
    a.g[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, x.z, a.g[5], 0, 0, 0);
    a.g[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, x.w, a.g[6], 0, 0, 0);
    a.g[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, x.z, a.g[7], 0, 0, 0);
    a.g[8] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, x.w, a.g[8], 0, 0, 0);
    a.g[9] = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, x.w, a.g[9], 0, 0, 0);
    a.xy[0] = fmaf(x.x, y, a.xy[0]);
    a.xy[1] = fmaf(x.y, y, a.xy[1]);
    a.xy[2] = fmaf(x.z, y, a.xy[2]);
    a.xy[3] = fmaf(x.w, y, a.xy[3]);
    a.sx[0] += x.x;
    a.sx[1] += x.y;
    a.sx[2] += x.z;
    a.sx[3] += x.w;
    a.sy += y;
    a.syy = fmaf(y, y, a.syy);
}

__device__ __forceinline__ void x64_flush(X64Acc &a, double (&d)[X64_NVAL])
{
#pragma unroll
    for (int p = 0; p < X64_NPAIR; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[p * 4 + r] += (double)a.g[p][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d[40 + r] += (double)a.xy[r];
        d[44 + r] += (double)a.sx[r];
    }
    d[48] += (double)a.sy;
    d[49] += (double)a.syy;
    x64_zero(a);
}

// partial: [gridDim.x * WAVES][X64_PART] float64, one slab per wave.
__global__ __launch_bounds__(BLOCK, 2) void gram_x64_kernel(const float *__restrict__ X,
                                                            const float *__restrict__ Y,
                                                            long long n,
                                                            double *__restrict__ partial)
{
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int g = lane >> 4, i = lane & 15;
    const long long nwaves = (long long)gridDim.x * WAVES;
    const long long wave_id = (long long)blockIdx.x * WAVES + wv;
    const long long steps = (n + 3) >> 2;           // 4-row steps, the last may be partial
    const long long full_steps = n >> 2;
    const long long s0 = steps * wave_id / nwaves;
    const long long s1 = steps * (wave_id + 1) / nwaves;

    X64Acc a;
    x64_zero(a);
    double d[X64_NVAL];
#pragma unroll
    for (int v = 0; v < X64_NVAL; ++v) d[v] = 0.0;

    const f32x4 *__restrict__ Xv = reinterpret_cast<const f32x4 *>(X);
    int since_flush = 0;
    long long s = s0;
    const long long s1_full = (s1 < full_steps) ? s1 : full_steps;
    for (; s + X64_UNROLL <= s1_full; s += X64_UNROLL) {
        f32x4 xv[X64_UNROLL];
        float yv[X64_UNROLL];
#pragma unroll
        for (int u = 0; u < X64_UNROLL; ++u) {
            xv[u] = __builtin_nontemporal_load(Xv + (s + u) * 64 + lane);     // streamed once
            yv[u] = __builtin_nontemporal_load(Y + (s + u) * 4 + g);
        }
#pragma unroll
        for (int u = 0; u < X64_UNROLL; ++u) x64_step(a, xv[u], yv[u]);
        since_flush += X64_UNROLL;
        if (since_flush >= X64_FLUSH_STEPS) {
            x64_flush(a, d);
            since_flush = 0;
        }
    }
    for (; s < s1; ++s) {
        const long long row = s * 4 + g;
        f32x4 xv = (f32x4){0.f, 0.f, 0.f, 0.f};
        float yv = 0.f;
        if (row < n) {
            xv = Xv[s * 64 + lane];
            yv = Y[row];
        }
        x64_step(a, xv, yv);
    }
    x64_flush(a, d);

    double *dst = partial + (size_t)wave_id * X64_PART;
#pragma unroll
    for (int p = 0; p < X64_NPAIR; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[p * 256 + r * 64 + lane] = d[p * 4 + r];
    // X'y and column sums: lane (i, g) holds rows = g (mod 4); add the four row groups
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        double v = d[40 + q];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) dst[2560 + q * 16 + i] = v;      // q<4: xy[q][i], q>=4: sx[q-4][i]
    }
    {
        // y sums: the 16 lanes of a group hold the same y; take one lane per group
        double sy = d[48], syy = d[49];
        sy += __shfl_xor(sy, 16, 64);   sy += __shfl_xor(sy, 32, 64);
        syy += __shfl_xor(syy, 16, 64); syy += __shfl_xor(syy, 32, 64);
        if (lane == 0) { dst[2688] = sy; dst[2689] = syy; }
    }
}

// acc (66x66) += fixed-order sum over the per-wave slabs.  One wave per acc entry.
__global__ __launch_bounds__(64) void gram_x64_finalize_kernel(const double *__restrict__ partial,
                                                               int nparts, long long n,
                                                               double *__restrict__ acc)
{
    constexpr int D = 66;
    const int e = blockIdx.x;
    const int I = e / D, J = e % D;
    int idx = -1;
    if (I < 64 && J < 64) {
        const int a = I >> 2, q1 = I & 3, b = J >> 2, q2 = J & 3;
        // tile (q1<=q2): row a, col b  ->  register a&3, lane (a>>2)*16 + b
        if (q1 <= q2) idx = f16_pair(q1, q2) * 256 + (a & 3) * 64 + (a >> 2) * 16 + b;
        else          idx = f16_pair(q2, q1) * 256 + (b & 3) * 64 + (b >> 2) * 16 + a;
    } else if (I < 64 && J == 64) idx = 2560 + (4 + (I & 3)) * 16 + (I >> 2);
    else if (J < 64 && I == 64) idx = 2560 + (4 + (J & 3)) * 16 + (J >> 2);
    else if (I < 64 && J == 65) idx = 2560 + (I & 3) * 16 + (I >> 2);
    else if (J < 64 && I == 65) idx = 2560 + (J & 3) * 16 + (J >> 2);
    else if ((I == 64 && J == 65) || (I == 65 && J == 64)) idx = 2688;
    else if (I == 65 && J == 65) idx = 2689;
    double s = 0.0;
    if (idx >= 0) {
        for (int p = threadIdx.x; p < nparts; p += 64) s += partial[(size_t)p * X64_PART + idx];
        s = wave_sum(s);
    } else {
        s = (double)n;
    }
    if (threadIdx.x == 0) acc[e] += s;
}

// ---------------------------------------------------------------------------- predict
__global__ __launch_bounds__(BLOCK) void linear_predict_kernel(const float *__restrict__ X,
                                                              const double *__restrict__ beta,
                                                              float *__restrict__ Yhat,
                                                              long long n, int F, int P)
{
    const long long total = n * P;
    for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < total;
         t += (long long)gridDim.x * BLOCK) {
        const long long r = t / P;
        const int p = (int)(t % P);
        double s = beta[(size_t)F * P + p];
        for (int f = 0; f < F; ++f) s += (double)X[r * F + f] * beta[(size_t)f * P + p];
        Yhat[t] = (float)s;
    }
}

// ---------------------------------------------------------------------------- planning
bool use_f16(const float *X, int F, int P)
{
    return F == 16 && P == 1 && (((uintptr_t)X) & 15) == 0;
}

bool use_x64(const float *X, int F, int P)
{
    return F == 64 && P == 1 && (((uintptr_t)X) & 15) == 0;
}

int x64_grid(long long n)
{
    // 2 blocks of 4 waves per CU, fewer for small inputs (>= 64 steps per wave)
    long long g = (long long)wm_num_cus() * 2;
    const long long cap = (n / 4 + 64 * WAVES - 1) / (64 * WAVES);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int f16_grid()
{
    // 2 waves per SIMD (launch bounds) = 2 blocks of 4 waves per CU; every block is resident.
    return wm_num_cus() * 2;
}

}  // namespace

// where gram_mid_kernel (register-fed, any alignment) beats the LDS-staged SYRK: see DESIGN.md
static bool mid_preferred(int F, int P)
{
#ifdef WM_PROBE_ENV
    if (const char *e = getenv("WM_GRAM_MID")) return atoi(e) != 0;
#endif
    (void)F; (void)P;
    return true;
}

extern "C" size_t wm_gram_workspace_bytes(int64_t n, int F, int P)
{
    if (n <= 0 || F <= 0 || P <= 0 || wm_num_cus() <= 0) return 0;
    // the alignment of X and Y is not known here: the generic kernel must always fit
    size_t b = wm_gram_syrk_workspace_bytes(n, F, P);
    if (F == 16 && P == 1) {
        const size_t b16 = (size_t)f16_grid() * F16_PART * sizeof(double);
        if (b16 > b) b = b16;
    }
    if (F == 64 && P == 1) {
        const size_t b64 = (size_t)x64_grid(n) * WAVES * X64_PART * sizeof(double);
        if (b64 > b) b = b64;
    }
    const size_t bs = wm_gram_small_workspace_bytes(F, P);
    if (bs > b) b = bs;
    const size_t bm = wm_gram_mid_workspace_bytes(F, P);
    if (bm > b) b = bm;
    const size_t bq = wm_gram_quad_workspace_bytes(F, P);
    if (bq > b) b = bq;
    const size_t bu = wm_gram_quadu_workspace_bytes(F, P);
    if (bu > b) b = bu;
    return wm_align_up(b, 256);
}

extern "C" int wm_gram_f32(const float *X, const float *Y, int64_t n, int F, int P, double *acc,
                           void *workspace, size_t workspace_bytes, void *stream)
{
    WM_REQUIRE(X && Y && acc, WM_ERR_INVALID, "wm_gram_f32: null pointer");
    WM_REQUIRE(F > 0 && P > 0 && n >= 0, WM_ERR_INVALID, "wm_gram_f32: bad sizes n=%lld F=%d P=%d",
               (long long)n, F, P);
    if (n == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_gram_f32: no HIP device");
    const size_t need = wm_gram_workspace_bytes(n, F, P);
    WM_REQUIRE(workspace && workspace_bytes >= need, WM_ERR_WORKSPACE,
               "wm_gram_f32: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    double *partial = (double *)workspace;
    if (use_f16(X, F, P)) {
        const int grid = f16_grid();
        hipLaunchKernelGGL(gram_f16_kernel, dim3(grid), dim3(BLOCK), 0, st, X, Y, (long long)n,
                           partial);
        WM_LAUNCH_CHECK("gram_f16_kernel");
        hipLaunchKernelGGL(gram_f16_finalize_kernel, dim3(18 * 18), dim3(64), 0, st, partial, grid,
                           (long long)n, acc);
        WM_LAUNCH_CHECK("gram_f16_finalize_kernel");
        return WM_OK;
    }
    if (use_x64(X, F, P)) {
        const int grid = x64_grid(n);
        hipLaunchKernelGGL(gram_x64_kernel, dim3(grid), dim3(BLOCK), 0, st, X, Y, (long long)n,
                           partial);
        WM_LAUNCH_CHECK("gram_x64_kernel");
        hipLaunchKernelGGL(gram_x64_finalize_kernel, dim3(66 * 66), dim3(64), 0, st, partial,
                           grid * WAVES, (long long)n, acc);
        WM_LAUNCH_CHECK("gram_x64_finalize_kernel");
        return WM_OK;
    }
    if (wm_gram_small_supported(X, Y, F, P))        // (8,4): 6.2 TB/s here, 5.95 on the quad kernel
        return wm_gram_small_launch(X, Y, (long long)n, F, P, acc, partial, st);
    if (wm_gram_quad_supported(X, Y, F, P))
        return wm_gram_quad_launch(X, Y, (long long)n, F, P, acc, partial, st);
    if (wm_gram_quadu_supported(F, P))              // padded / unaligned rows of the same family
        return wm_gram_quadu_launch(X, Y, (long long)n, F, P, acc, partial, st);
    if (wm_gram_mid_supported(F, P) && mid_preferred(F, P))
        return wm_gram_mid_launch(X, Y, (long long)n, F, P, acc, partial, st);
    return wm_gram_syrk_launch(X, Y, (long long)n, F, P, acc, workspace, st);
}

namespace {

// One wave, matrix in registers: lane i holds row i of the centred system (F <= 24, P <= 4).
// The LDS kernel above spends its 20-27 us in ~3F barrier phases of float64 LDS round trips; here
// an elimination step is two v_readlane per broadcast double (compile-time lane and register
// indices: the loops are fully unrolled to WS_F with uniform guards) and one fused update per
// element, with no barrier anywhere.  Same algorithm (right-looking Cholesky, rsqrt, forward and
// back substitution); L goes through LDS once so that the back substitution can read a column.
constexpr int WS_F = 24, WS_P = 4;

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void normal_solve_wave_kernel(const double *__restrict__ acc, int F,
                                                               int P, int fit_intercept,
                                                               double ridge,
                                                               double *__restrict__ beta,
                                                               int *__restrict__ status)
{
    __shared__ double Ls[WS_F][WS_F + 1];
    const int lane = threadIdx.x, D = F + 1 + P;
    const int i = lane < F ? lane : F - 1;               // clamped row for the loads
    const double n = acc[(size_t)F * D + F];
    const double inv_n = (fit_intercept && n > 0.0) ? 1.0 / n : 0.0;
    bool bad = fit_intercept && !(n > 0.0);
    const double si = acc[(size_t)i * D + F];
    double a[WS_F], b[WS_P], T[WS_P], d0 = 0.0;
#pragma unroll
    for (int p = 0; p < WS_P; ++p) {
        T[p] = p < P ? acc[(size_t)F * D + F + 1 + p] : 0.0;
        b[p] = p < P ? acc[(size_t)i * D + F + 1 + p] - si * T[p] * inv_n : 0.0;
    }
#pragma unroll
    for (int j = 0; j < WS_F; ++j) {
        double v = 0.0;
        if (j < F) {
            v = acc[(size_t)i * D + j] - si * acc[(size_t)j * D + F] * inv_n;
            if (j == lane) v += ridge;
        }
        a[j] = v;
        if (j == lane) d0 = v;                            // the column's own centred diagonal
    }
    double r[WS_F];                                       // 1 / L[k][k], wave-uniform
#pragma unroll
    for (int k = 0; k < WS_F; ++k) {
        r[k] = 0.0;
        if (k < F && !bad) {
            const double dk = readlane_f64(a[k], k);
            if (!(dk > WM_PIVOT_RTOL * readlane_f64(d0, k))) {    // collinear: solve_internal.h
                bad = true;
            } else {
                const double rs = rsqrt(dk), inv = rs * rs;
                r[k] = rs;
                const double aik = a[k];                  // A[i][k]
#pragma unroll
                for (int j = k + 1; j < WS_F; ++j)
                    if (j < F) a[j] -= aik * (readlane_f64(aik, j) * inv);   // rows i >= j matter
                a[k] = aik * rs;                          // L[i][k]
            }
        }
    }
    double z[WS_P], bt[WS_P];
#pragma unroll
    for (int p = 0; p < WS_P; ++p) { z[p] = 0.0; bt[p] = 0.0; }
    if (!bad) {
        if (lane < F) {
#pragma unroll
            for (int j = 0; j < WS_F; ++j)
                if (j < F) Ls[lane][j] = a[j];
        }
        // forward substitution L z = b: z_k lives in lane k
#pragma unroll
        for (int k = 0; k < WS_F; ++k)
            if (k < F) {
#pragma unroll
                for (int p = 0; p < WS_P; ++p) {
                    const double zk = readlane_f64(b[p], k) * r[k];
                    if (lane == k) z[p] = zk;
                    if (lane > k) b[p] -= a[k] * zk;
                }
            }
        // back substitution L' beta = z: beta_k lives in lane k; lane i < k needs L[k][i]
#pragma unroll
        for (int k = WS_F - 1; k >= 0; --k)
            if (k < F) {
                const double lki = lane < k ? Ls[k][lane] : 0.0;
#pragma unroll
                for (int p = 0; p < WS_P; ++p) {
                    const double bk = readlane_f64(z[p], k) * r[k];
                    if (lane == k) bt[p] = bk;
                    z[p] -= lki * bk;
                }
            }
    }
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
    for (int p = 0; p < WS_P; ++p)
        if (p < P) {
            if (lane < F) beta[(size_t)lane * P + p] = bad ? qnan : bt[p];
            double icpt = 0.0;
            if (fit_intercept && !bad) {
                const double part = lane < F ? si * inv_n * bt[p] : 0.0;
                icpt = T[p] * inv_n - wave_sum(part);
            }
            if (lane == 0) beta[(size_t)F * P + p] = bad ? qnan : icpt;
        }
    if (lane == 0 && status) *status = bad ? 1 : 0;
}

}  // namespace

extern "C" size_t wm_normal_solve_workspace_bytes(int F, int P)
{
    if (F <= 0 || P <= 0) return 0;
    if (F <= WS_F && P <= WS_P) return 0;          // one wave, matrix in registers
    return wm_solve_blocked_workspace_bytes(F, P);
}

extern "C" int wm_normal_solve_f64(const double *acc, int F, int P, int fit_intercept, double ridge,
                                   double *beta, int *status, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    WM_REQUIRE(acc && beta, WM_ERR_INVALID, "wm_normal_solve_f64: null pointer");
    WM_REQUIRE(F > 0 && P > 0, WM_ERR_INVALID, "wm_normal_solve_f64: bad sizes F=%d P=%d", F, P);
    WM_REQUIRE(ridge >= 0.0, WM_ERR_INVALID, "wm_normal_solve_f64: negative ridge");
    hipStream_t st = (hipStream_t)stream;
    if (F <= WS_F && P <= WS_P) {
        hipLaunchKernelGGL(normal_solve_wave_kernel, dim3(1), dim3(64), 0, st, acc, F, P,
                           fit_intercept, ridge, beta, status);
        WM_LAUNCH_CHECK("normal_solve_wave_kernel");
        return WM_OK;
    }
    const size_t need = wm_normal_solve_workspace_bytes(F, P);
    WM_REQUIRE(workspace && workspace_bytes >= need, WM_ERR_WORKSPACE,
               "wm_normal_solve_f64: workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
    WM_REQUIRE((((uintptr_t)workspace) & 15) == 0, WM_ERR_INVALID,
               "wm_normal_solve_f64: workspace must be 16-byte aligned");
    return wm_solve_blocked_launch(acc, F, P, fit_intercept, ridge, beta, status, workspace, st);
}

extern "C" int wm_linear_predict_f32(const float *X, const double *beta, float *Yhat, int64_t n,
                                     int F, int P, void *stream)
{
    WM_REQUIRE(X && beta && Yhat, WM_ERR_INVALID, "wm_linear_predict_f32: null pointer");
    WM_REQUIRE(F > 0 && P > 0 && n >= 0, WM_ERR_INVALID, "wm_linear_predict_f32: bad sizes");
    if (n == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_linear_predict_f32: no HIP device");
    long long g = ((long long)n * P + BLOCK - 1) / BLOCK;
    const long long cap = (long long)wm_num_cus() * 16;
    if (g > cap) g = cap;
    hipLaunchKernelGGL(linear_predict_kernel, dim3((int)g), dim3(BLOCK), 0, (hipStream_t)stream, X,
                       beta, Yhat, (long long)n, F, P);
    WM_LAUNCH_CHECK("linear_predict_kernel");
    return WM_OK;
}
