// solve_blocked.hip -- the normal-equation solve for wide systems, blocked over many workgroups.
//
// The reference's singleRegression fit is 432 predictors -> 216 targets (singleRegression.py:21,33,
// 74-85): a 432 x 432 float64 Cholesky with 216 right-hand sides, 1.5 MB + 0.75 MB -- far beyond one
// workgroup's LDS.  The one-workgroup kernel walked it in global memory with ~3F barrier phases
// (55 ms).  Here the centred system [A | B] lives in the workspace and is factorised by 64-column
// block steps, each step two launches:
//   chol_panel_kernel   every workgroup factorises the 64 x 64 diagonal block in LDS (redundantly:
//                       no cross-workgroup wait); workgroup b > 0 then forms its block of the
//                       panel, L_ik = A_ik L_kk^-T, and the right-hand-side workgroups
//                       Z_k = L_kk^-1 B_k, by substitution: one lane per row (column) of the block
//                       with the 64 unknowns in registers and L read from LDS at wave-uniform
//                       addresses -- 2016 independent-enough FMAs per lane, no barrier;
//   chol_update_kernel  A_ij -= L_ik L_jk' (trailing blocks, lower triangle), B_i -= L_ik Z_k as
//                       64^3 register-tiled products from LDS.
// Back substitution is one launch per block step: workgroup (i, pc) solves
// beta_k = L_kk^-T Z_k for its 64 target columns (again one lane per column) and applies
// Z_i -= L_ki' beta_k; the workgroups with i = k write beta_k.  A pivot that is not above
// WM_PIVOT_RTOL times the column's own centred diagonal (a column that the ones before it explain
// to rounding error: collinear) sets *status; every later launch sees it and the last one writes
// NaN.  About 3 nb + 2 launches (23 for F = 432); fixed summation order, float64 throughout.
// Every tile comes in through 16 unconditional (clamped) loads per thread issued together: with
// the loads inside a bounds test each one waited for the last, and a 64^3 product took 19 us.
#include "common.h"
#include "solve_internal.h"

namespace {

// Ablation switches for tools/probe/solve_ablate.sh (timing-only builds; results are wrong, and
// SB_ABLATE keeps the garbage from tripping the pivot test, which would end the later launches):
// -DSB_NO_MM skips the 64^3 products, -DSB_NO_SUBST the substitutions, -DSB_NO_FACTOR the diagonal
// block's factorisation.
constexpr int NB = 64;
constexpr int BLOCK = 256;
constexpr int TILE = NB * NB;                       // doubles per LDS tile (32 KB)
constexpr int VLD = NB + 1;                         // pitch of a tile read one ROW per lane
constexpr int VTILE = NB * VLD;
constexpr int PER = TILE / BLOCK;                   // elements per thread and tile

// ---- tile movers.  A 64 x 64 float64 tile of a row-major matrix (leading dimension ld), zero
// outside (rows, cols), lands at s[pos(r, c)].  rows, cols >= 1.
struct PosNatural { __device__ static int at(int r, int c) { return r * NB + c; } };
// transposed, the 4-double groups of a row XOR-swizzled with the row index (consecutive lanes write
// rows 512 B apart, i.e. one bank: 8-way instead of 64-way conflicts); tile_mm<true,..> reads it
struct PosTransSwz {
    __device__ static int at(int r, int c) { return c * NB + ((((r >> 2) ^ (c & 15)) << 2) | (r & 3)); }
};
struct PosRowVec { __device__ static int at(int r, int c) { return r * VLD + c; } };   // vector = row
struct PosColVec { __device__ static int at(int r, int c) { return c * VLD + r; } };   // vector = column

__device__ __forceinline__ void tile_fetch(double (&v)[PER], const double *__restrict__ g, int ld,
                                           int rows, int cols)
{
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = threadIdx.x + k * BLOCK;
        const int r = e >> 6, c = e & 63;
        const int rr = r < rows ? r : rows - 1, cc = c < cols ? c : cols - 1;
        const double x = g[(size_t)rr * ld + cc];
        v[k] = (r < rows && c < cols) ? x : 0.0;
    }
}
template <class Pos>
__device__ __forceinline__ void tile_put(double *s, const double (&v)[PER])
{
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = threadIdx.x + k * BLOCK;
        s[Pos::at(e >> 6, e & 63)] = v[k];
    }
}

// acc[a][b] += sum_m As[m][4ty + a] * Bs[m][4tx + b], thread (ty, tx) = (tid >> 4, tid & 15).
template <bool SWA, bool SWB>
__device__ __forceinline__ void tile_mm(const double *As, const double *Bs, double (&acc)[4][4])
{
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#ifdef SB_NO_MM
    acc[0][0] += As[ty] * Bs[tx];
    return;
#endif
#pragma unroll 4
    for (int m = 0; m < NB; ++m) {
        double a[4], b[4];
        const int ao = SWA ? ((ty ^ (m & 15)) << 2) : 4 * ty;
        const int bo = SWB ? ((tx ^ (m & 15)) << 2) : 4 * tx;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = As[m * NB + ao + q];
            b[q] = Bs[m * NB + bo + q];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[p][q] = fma(a[p], b[q], acc[p][q]);
    }
}

// Broadcast of lane (lane & ~3) | R to its quad, in the VALU (DPP quad_perm), for a double.
template <int R>
__device__ __forceinline__ double quad_bcast(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, R * 0x55, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, R * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Substitution with FOUR lanes per vector: lane t of the quad holds x[4q + t], q = 0..15, in
// registers; step j scales x_j on its owner lane, broadcasts it through the quad and every lane
// updates its share.  All 256 threads work (64 vectors), a step is <= 16 FMAs per lane, and the
// fully unrolled code is a quarter of the one-lane-per-vector form -- which ran at the speed of
// instruction fetch (28 KB of straight-line code: 15 us).
//   forward:  x <- L^-1 x,  sLT[j][m] = L[m][j] (zero for m < j), sR[j] = 1 / L[j][j]
//   backward: x <- L^-T x,  sL[j][m]  = L[j][m] (zero for m > j)
template <int J>
__device__ __forceinline__ void subst_forward_step(double (&x)[16], const double *sLT,
                                                   const double *sR, int t)
{
    constexpr int Q = J >> 2, R = J & 3;
    const double xj = quad_bcast<R>(x[Q] * sR[J]);
#pragma unroll
    for (int q = Q; q < 16; ++q) x[q] = fma(-xj, sLT[J * NB + 4 * q + t], x[q]);
    if (t == R) x[Q] = xj;            // the owner's own update above used L[j][j]: overwrite
    // keep the steps apart: nothing orders the L reads of later steps, and hipcc hoists all 544
    // of them to the top (512 VGPRs, 142 spilled)
    __builtin_amdgcn_sched_barrier(0);
}
template <int J>
__device__ __forceinline__ void subst_backward_step(double (&x)[16], const double *sL,
                                                    const double *sR, int t)
{
    constexpr int Q = J >> 2, R = J & 3;
    const double xj = quad_bcast<R>(x[Q] * sR[J]);
#pragma unroll
    for (int q = 0; q <= Q; ++q) x[q] = fma(-xj, sL[J * NB + 4 * q + t], x[q]);
    if (t == R) x[Q] = xj;
    __builtin_amdgcn_sched_barrier(0);
}
template <int J>
__device__ __forceinline__ void subst_forward_from(double (&x)[16], const double *sLT,
                                                   const double *sR, int t)
{
    if constexpr (J < NB) {
        subst_forward_step<J>(x, sLT, sR, t);
        subst_forward_from<J + 1>(x, sLT, sR, t);
    }
}
template <int J>
__device__ __forceinline__ void subst_backward_from(double (&x)[16], const double *sL,
                                                    const double *sR, int t)
{
    if constexpr (J >= 0) {
        subst_backward_step<J>(x, sL, sR, t);
        subst_backward_from<J - 1>(x, sL, sR, t);
    }
}
// x of vector v = tid >> 2 lives at sV[v * VLD + m]; solved in place.
__device__ __forceinline__ void subst_forward(double *sV, const double *sLT, const double *sR)
{
    const int v = threadIdx.x >> 2, t = threadIdx.x & 3;
    double x[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = sV[v * VLD + 4 * q + t];
#ifndef SB_NO_SUBST
    subst_forward_from<0>(x, sLT, sR, t);
#endif
#pragma unroll
    for (int q = 0; q < 16; ++q) sV[v * VLD + 4 * q + t] = x[q];
}
// result to out[m * NB + v] (the natural [row][column] tile of the solved block)
__device__ __forceinline__ void subst_backward(const double *sV, double *out, const double *sL,
                                               const double *sR)
{
    const int v = threadIdx.x >> 2, t = threadIdx.x & 3;
    double x[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = sV[v * VLD + 4 * q + t];
#ifndef SB_NO_SUBST
    subst_backward_from<NB - 1>(x, sL, sR, t);
#endif
#pragma unroll
    for (int q = 0; q < 16; ++q) out[(4 * q + t) * NB + v] = x[q];
}

// One step of the right-looking Cholesky of the 64 x 64 block with the matrix in REGISTERS: thread
// (i, t3) = (tid >> 2, tid & 3) holds A[i][4q + t3], q = 0..15.  The pivot column is published
// through a double-buffered 64-double LDS line (one barrier per step); every read of a step is
// issued before its first use.  (With the block in LDS, the read-modify-write chains of a step
// waited on one another: 45 us for the block.)
template <int J>
__device__ __forceinline__ void chol_step(double (&own)[16], double *colb, double *sL, double *sLT,
                                          double *sR, const double *sD0, int i, int t3, bool &bad)
{
    constexpr int Q = J >> 2, R = J & 3;
    double *col = colb + (J & 1) * NB;
    if (t3 == R) col[i] = own[Q];
    __syncthreads();
    const double dk = col[J], aij = col[i];
    double cv[16];
#pragma unroll
    for (int q = Q; q < 16; ++q) cv[q] = col[4 * q + t3];
#ifndef SB_ABLATE
    bad = bad || !(dk > sD0[J]);                     // uniform; garbage below is never published
#endif
    const double rs = rsqrt(dk), inv = rs * rs;
    if (threadIdx.x == 0) sR[J] = rs;
    if (t3 == R && i >= J) {
        sL[i * NB + J] = aij * rs;
        sLT[J * NB + i] = aij * rs;
    }
    const double lij = aij * inv;
#pragma unroll
    for (int q = Q; q < 16; ++q) {
        const int c = 4 * q + t3;
        own[q] = fma(-((c > J && c <= i) ? lij : 0.0), cv[q], own[q]);
        // pin the update to this step: left free, hipcc sinks the 4q updates of own[q] to the step
        // that first reads it and keeps every (lij, column) pair alive until then (512 VGPRs + spills)
        asm volatile("" : "+v"(own[q]));
    }
}
template <int J>
__device__ __forceinline__ void chol_from(double (&own)[16], double *colb, double *sL, double *sLT,
                                          double *sR, const double *sD0, int i, int t3, bool &bad)
{
    if constexpr (J < NB) {
        chol_step<J>(own, colb, sL, sLT, sR, sD0, i, t3, bad);
        chol_from<J + 1>(own, colb, sL, sLT, sR, sD0, i, t3, bad);
    }
}

// ---- centring: A = X'X - s s'/n (+ ridge), B = X'Y - s t'/n from the Gram
__global__ __launch_bounds__(BLOCK) void solve_prep_kernel(const double *__restrict__ acc, int F,
                                                           int P, int fit_intercept, double ridge,
                                                           double *__restrict__ A,
                                                           double *__restrict__ B,
                                                           double *__restrict__ diag0,
                                                           int *__restrict__ status)
{
    const int D = F + 1 + P;
    const double n = acc[(size_t)F * D + F];
    const double inv_n = (fit_intercept && n > 0.0) ? 1.0 / n : 0.0;
    const long long total = (long long)F * F + (long long)F * P;
    for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total;
         e += (long long)gridDim.x * BLOCK) {
        if (e < (long long)F * F) {
            const int i = (int)(e / F), j = (int)(e % F);
            double v = acc[(size_t)i * D + j] - acc[(size_t)i * D + F] * acc[(size_t)j * D + F] * inv_n;
            if (i == j) {
                v += ridge;
                diag0[i] = v;
            }
            A[e] = v;
        } else {
            const long long f = e - (long long)F * F;
            const int i = (int)(f / P), p = (int)(f % P);
            B[f] = acc[(size_t)i * D + F + 1 + p] -
                   acc[(size_t)i * D + F] * acc[(size_t)F * D + F + 1 + p] * inv_n;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *status = (fit_intercept && !(n > 0.0)) ? 1 : 0;
}

// LDS of the panel kernel (doubles): sA | sL | sLT | sV (pitch 65) | sR | sD0
constexpr size_t PANEL_LDS = (3 * TILE + VTILE + 2 * NB) * sizeof(double);
constexpr size_t UPDATE_LDS = 2 * TILE * sizeof(double);
// back kernel: sL | sLki | sBeta | sV | sR
constexpr size_t BACK_LDS = (3 * TILE + VTILE + NB) * sizeof(double);

// ---- block step kb, part 1.  grid = (nb - kb) + npb.
__global__ __launch_bounds__(BLOCK) void chol_panel_kernel(double *__restrict__ A,
                                                           double *__restrict__ B,
                                                           double *__restrict__ Ld,
                                                           const double *__restrict__ diag0, int F,
                                                           int P, int kb, int nb,
                                                           int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sA = sm;                 // the diagonal block while it is factorised
    double *sL = sm + TILE;          // L_kk
    double *sLT = sm + 2 * TILE;     // L_kk'
    double *sV = sm + 3 * TILE;      // this workgroup's block, one vector per lane (pitch 65)
    double *sR = sV + VTILE;         // 1 / L[j][j]
    double *sD0 = sR + NB;           // pivot floors of this block's columns
    const int tid = threadIdx.x;
    const int k0 = kb * NB;
    const int bs = F - k0 < NB ? F - k0 : NB;
    const int nrb = nb - kb;
    const int b = blockIdx.x;
    // every global read of the workgroup is issued before the first wait: the status word, the
    // diagonal block, the pivot floors and the block this workgroup will solve for
    const int st0 = *status;
    double vd[PER], vb[PER];
    tile_fetch(vd, A + (size_t)k0 * F + k0, F, bs, bs);
    const double d0 = diag0[k0 + (tid < bs ? tid : bs - 1)];
    int i0 = 0, rows = 0, c0 = 0, cols = 0;
    if (b > 0 && b < nrb) {
        i0 = (kb + b) * NB;
        rows = F - i0 < NB ? F - i0 : NB;
        tile_fetch(vb, A + (size_t)i0 * F + k0, F, rows, bs);
    } else if (b >= nrb) {
        c0 = (b - nrb) * NB;
        cols = P - c0 < NB ? P - c0 : NB;
        tile_fetch(vb, B + (size_t)k0 * P + c0, P, bs, cols);
    }
    if (st0 != 0) return;            // written by an earlier launch: uniform
    tile_put<PosNatural>(sA, vd);
    if (b > 0 && b < nrb) tile_put<PosRowVec>(sV, vb);
    else if (b >= nrb) tile_put<PosColVec>(sV, vb);
    for (int e = tid; e < TILE; e += BLOCK) {
        sL[e] = 0.0;
        sLT[e] = 0.0;
    }
    if (tid < NB) sD0[tid] = tid < bs ? WM_PIVOT_RTOL * d0 : 0.0;
    __syncthreads();
    if (tid >= bs && tid < NB) sA[tid * NB + tid] = 1.0;      // pad with the identity
    __syncthreads();
    // right-looking Cholesky of the diagonal block, rows in registers (chol_step)
    bool bad = false;
    {
        const int i = tid >> 2, t3 = tid & 3;
        double own[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) own[q] = sA[i * NB + 4 * q + t3];
        __syncthreads();             // sA's first two rows now serve as the pivot-column line
#ifdef SB_NO_FACTOR
        chol_from<NB - 1>(own, sA, sL, sLT, sR, sD0, i, t3, bad);
#else
        chol_from<0>(own, sA, sL, sLT, sR, sD0, i, t3, bad);
#endif
        __syncthreads();
    }
    if (bad) {
        if (b == 0 && tid == 0) *status = 1;
        return;
    }
    if (b == 0) {
        // the writer: L_kk (zero above the diagonal and past the system's end) goes to its own
        // tile, NOT back into A -- the other workgroups of this launch are still reading A_kk
        for (int e = tid; e < TILE; e += BLOCK) Ld[(size_t)kb * TILE + e] = sL[e];
        return;
    }
    // row r of L_ik solves L_kk x = (row r of A_ik)'; column c of Z_k solves L_kk z = column c of
    // B_k: the same substitution, one lane per vector
    subst_forward(sV, sLT, sR);
    __syncthreads();
    if (b < nrb) {
        for (int e = tid; e < TILE; e += BLOCK) {
            const int r = e >> 6, j = e & 63;
            if (r < rows && j < bs) A[(size_t)(i0 + r) * F + k0 + j] = sV[r * VLD + j];
        }
    } else {
        for (int e = tid; e < TILE; e += BLOCK) {
            const int j = e >> 6, c = e & 63;
            if (j < bs && c < cols) B[(size_t)(k0 + j) * P + c0 + c] = sV[c * VLD + j];
        }
    }
}

// ---- block step kb, part 2.  grid = T + (nb-kb-1) * npb, T = trailing lower-triangle blocks.
__global__ __launch_bounds__(BLOCK) void chol_update_kernel(double *__restrict__ A,
                                                            double *__restrict__ B, int F, int P,
                                                            int kb, int nb, int npb,
                                                            const int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sAs = sm, *sBs = sm + TILE;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int k0 = kb * NB;
    const int bs = F - k0 < NB ? F - k0 : NB;
    const int nt = nb - kb - 1;                    // trailing block rows
    const int T = nt * (nt + 1) / 2;
    int b = blockIdx.x;
    const int st0 = *status;
    double va[PER], vb[PER];
    double *C;                                     // the block this workgroup updates
    int ldc, rows, cols;
    const bool tri = b < T;
    if (tri) {
        // (i, j) with j <= i: A_ij[r][c] -= sum_m L_ik[r][m] L_jk[c][m]
        int i = 0;
        while (b > i) { b -= i + 1; ++i; }
        const int j = b;
        const int i0 = (kb + 1 + i) * NB, j0 = (kb + 1 + j) * NB;
        rows = F - i0 < NB ? F - i0 : NB;
        cols = F - j0 < NB ? F - j0 : NB;
        tile_fetch(va, A + (size_t)i0 * F + k0, F, rows, bs);
        tile_fetch(vb, A + (size_t)j0 * F + k0, F, cols, bs);
        C = A + (size_t)i0 * F + j0;
        ldc = F;
    } else {
        // B_i[r][c] -= sum_m L_ik[r][m] Z_k[m][c]
        b -= T;
        const int i = b / npb, pc = b % npb;
        const int i0 = (kb + 1 + i) * NB, c0 = pc * NB;
        rows = F - i0 < NB ? F - i0 : NB;
        cols = P - c0 < NB ? P - c0 : NB;
        tile_fetch(va, A + (size_t)i0 * F + k0, F, rows, bs);
        tile_fetch(vb, B + (size_t)k0 * P + c0, P, bs, cols);
        C = B + (size_t)i0 * P + c0;
        ldc = P;
    }
    // the block itself, clamped like the tiles, so that its latency hides behind the product
    double cv[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * ty + p, c = 4 * tx + q;
            cv[p][q] = C[(size_t)(r < rows ? r : rows - 1) * ldc + (c < cols ? c : cols - 1)];
        }
    if (st0 != 0) return;
    tile_put<PosTransSwz>(sAs, va);
    if (tri) tile_put<PosTransSwz>(sBs, vb);
    else tile_put<PosNatural>(sBs, vb);
    __syncthreads();
    double acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = 0.0;
    if (tri) tile_mm<true, true>(sAs, sBs, acc);
    else tile_mm<true, false>(sAs, sBs, acc);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * ty + p, c = 4 * tx + q;
            if (r < rows && c < cols) C[(size_t)r * ldc + c] = cv[p][q] - acc[p][q];
        }
}

// ---- back substitution, block step k (from nb-1 down).  grid = (k + 1) * npb.
__global__ __launch_bounds__(BLOCK) void chol_back_kernel(const double *__restrict__ A,
                                                          double *__restrict__ B,
                                                          const double *__restrict__ Ld,
                                                          double *__restrict__ beta, int F, int P,
                                                          int k, int npb,
                                                          const int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *sL = sm, *sLki = sm + TILE, *sBeta = sm + 2 * TILE, *sV = sm + 3 * TILE;
    double *sR = sV + VTILE;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int i = blockIdx.x / npb, pc = blockIdx.x % npb;
    const int k0 = k * NB, c0 = pc * NB, i0 = i * NB;      // block i is full when i < k
    const int bs = F - k0 < NB ? F - k0 : NB;
    const int cols = P - c0 < NB ? P - c0 : NB;
    const int st0 = *status;
    double vl[PER], vz[PER], vk[PER];
    tile_fetch(vl, Ld + (size_t)k * TILE, NB, NB, NB);
    tile_fetch(vz, B + (size_t)k0 * P + c0, P, bs, cols);
    double cv[4][4];
    if (i < k) {
        tile_fetch(vk, A + (size_t)k0 * F + i0, F, bs, NB);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 4 * tx + q;
                cv[p][q] = B[(size_t)(i0 + 4 * ty + p) * P + c0 + (c < cols ? c : cols - 1)];
            }
    }
    if (st0 != 0) return;
    tile_put<PosNatural>(sL, vl);
    tile_put<PosColVec>(sV, vz);
    if (i < k) tile_put<PosNatural>(sLki, vk);
    __syncthreads();
    // rows past the system's end act as the identity
    if (tid < NB) sR[tid] = tid < bs ? 1.0 / sL[tid * NB + tid] : 1.0;
    __syncthreads();
    // beta_k = L_kk^-T Z_k, one lane per target column
    subst_backward(sV, sBeta, sL, sR);
    __syncthreads();
    if (i == k) {
        for (int e = tid; e < TILE; e += BLOCK) {
            const int r = e >> 6, c = e & 63;
            if (r < bs && c < cols) beta[(size_t)(k0 + r) * P + c0 + c] = sBeta[e];
        }
        return;
    }
    // Z_i[r][c] -= sum_m L_ki[m][r] beta_k[m][c]
    double acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = 0.0;
    tile_mm<false, false>(sLki, sBeta, acc);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * ty + p, c = 4 * tx + q;
            if (c < cols) B[(size_t)(i0 + r) * P + c0 + c] = cv[p][q] - acc[p][q];
        }
}

// ---- intercept row, or NaN everywhere when the system was not positive definite.  grid = P.
__global__ __launch_bounds__(64) void solve_finish_kernel(const double *__restrict__ acc, int F,
                                                          int P, int fit_intercept,
                                                          double *__restrict__ beta,
                                                          const int *__restrict__ status)
{
    const int D = F + 1 + P, p = blockIdx.x, lane = threadIdx.x;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    if (*status != 0) {
        for (int i = lane; i <= F; i += 64) beta[(size_t)i * P + p] = qnan;
        return;
    }
    double icpt = 0.0;
    if (fit_intercept) {
        const double inv_n = 1.0 / acc[(size_t)F * D + F];
        double s = 0.0;
        for (int i = lane; i < F; i += 64) s += acc[(size_t)i * D + F] * inv_n * beta[(size_t)i * P + p];
        icpt = acc[(size_t)F * D + F + 1 + p] * inv_n - wave_sum(s);
    }
    if (lane == 0) beta[(size_t)F * P + p] = icpt;
}

int set_lds_once(const void *fn, size_t bytes, int slot)
{
    static bool done[64][3] = {};
    int dev = 0;
    WM_HIP(hipGetDevice(&dev));
    bool &d = done[dev & 63][slot];
    if (!d) {
        WM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        d = true;
    }
    return WM_OK;
}

}  // namespace

size_t wm_solve_blocked_workspace_bytes(int F, int P)
{
    // A (F x F) | B (F x P) | L_kk tiles (nb) | diag0 (F) | own status word
    const size_t nb = (F + NB - 1) / NB;
    return wm_align_up(((size_t)F * F + (size_t)F * P + nb * TILE + F) * sizeof(double) + 256, 256);
}

int wm_solve_blocked_launch(const double *acc, int F, int P, int fit_intercept, double ridge,
                            double *beta, int *status, void *workspace, hipStream_t st)
{
    const int nb = (F + NB - 1) / NB, npb = (P + NB - 1) / NB;
    double *A = (double *)workspace;
    double *B = A + (size_t)F * F;
    double *Ld = B + (size_t)F * P;
    double *diag0 = Ld + (size_t)nb * TILE;
    int *st_own = (int *)(diag0 + F);
    int *stat = status ? status : st_own;
    int rc;
    if ((rc = set_lds_once((const void *)chol_panel_kernel, PANEL_LDS, 0)) != WM_OK) return rc;
    if ((rc = set_lds_once((const void *)chol_update_kernel, UPDATE_LDS, 1)) != WM_OK) return rc;
    if ((rc = set_lds_once((const void *)chol_back_kernel, BACK_LDS, 2)) != WM_OK) return rc;
    const long long total = (long long)F * F + (long long)F * P;
    int g = (int)((total + BLOCK - 1) / BLOCK);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(solve_prep_kernel, dim3(g), dim3(BLOCK), 0, st, acc, F, P, fit_intercept,
                       ridge, A, B, diag0, stat);
    WM_LAUNCH_CHECK("solve_prep_kernel");
    for (int kb = 0; kb < nb; ++kb) {
        hipLaunchKernelGGL(chol_panel_kernel, dim3(nb - kb + npb), dim3(BLOCK), PANEL_LDS, st, A, B,
                           Ld, (const double *)diag0, F, P, kb, nb, stat);
        WM_LAUNCH_CHECK("chol_panel_kernel");
        const int nt = nb - kb - 1;
        if (nt > 0) {
            hipLaunchKernelGGL(chol_update_kernel, dim3(nt * (nt + 1) / 2 + nt * npb), dim3(BLOCK),
                               UPDATE_LDS, st, A, B, F, P, kb, nb, npb, (const int *)stat);
            WM_LAUNCH_CHECK("chol_update_kernel");
        }
    }
    for (int k = nb - 1; k >= 0; --k) {
        hipLaunchKernelGGL(chol_back_kernel, dim3((k + 1) * npb), dim3(BLOCK), BACK_LDS, st,
                           (const double *)A, B, (const double *)Ld, beta, F, P, k, npb,
                           (const int *)stat);
        WM_LAUNCH_CHECK("chol_back_kernel");
    }
    hipLaunchKernelGGL(solve_finish_kernel, dim3(P), dim3(64), 0, st, acc, F, P, fit_intercept, beta,
                       (const int *)stat);
    WM_LAUNCH_CHECK("solve_finish_kernel");
    return WM_OK;
}
