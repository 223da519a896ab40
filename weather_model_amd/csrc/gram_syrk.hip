// gram_syrk.hip -- Z'Z for every shape without a dedicated kernel: LDS-staged blocked SYRK on
// v_mfma_f32_16x16x4_f32, float64 combination.
//
// The shape it is built for is the reference's singleRegression fit, Dense(432) -> Dense(216) on
// 6^3 cubes of one variable (singleRegression.py:21,33,74-85): F = 432, P = 216, 2592 B and
// 2F(F+P) = 560 k flop per row -- 216 flop/B, MFMA-bound by a factor of eleven.  The generic
// kernel it replaces gave every 16 x 16 tile of the 649 x 649 Gram its own workgroup, each
// re-reading its columns of X from global memory one scalar per lane per MFMA operand (861 tile
// pairs, X streamed ~41 times).
//
// Layout.  Z = [X | Y | 1 | 0...] is cut into 64-column blocks (the virtual column order puts the
// ones column behind Y, where the padding of the last block is; the finalize kernel maps it back
// to acc's [X | 1 | Y] order).  A workgroup owns a PAIR of 128-column superblocks (I <= J) and a
// slice of the rows; its four waves own the 64 x 64 tiles (2I+wi, 2J+wj) on or above the
// diagonal.  Per stage of 64 rows the two 64 x 128 panels are copied into LDS by 16-byte
// LDS-direct loads (one KiB = two panel rows per wave instruction, the image filled linearly);
// an operand read is ONE ds_read_b128 per side and four rows: lane (i, g) = (lane & 15,
// lane >> 4) takes columns 4i..4i+3 of row g, and register q of all lanes is the 16x16x4 operand
// of the column set {4i + q} -- the trick of gram_x64_kernel, fed from LDS instead of from global
// memory -- so sixteen MFMAs (ten on a diagonal tile, q1 <= q2) follow two LDS reads.
// Columns past F + P come from a constant line in the workspace ([1,0,0,0] / zeros, row stride 0),
// so the fill has no branch and the column sums ride on the MFMA as the ones column.
// Workgroups that share a row slice are placed on one XCD (blockIdx % 8) and walk their stages
// in step, so of the npairs reads of a panel all but the first hit that XCD's L2.
// float32 accumulators take 1024 rows (16 stages), then fold into float64 registers; per-wave
// float64 slabs, fixed-order finalize.  A pair with fewer than three tiles splits the k-steps of
// a stage over its idle waves (two tiles: 2 x 2, one tile: 1 x 4), which is also what a narrow
// problem (D <= 64) gets.  Shapes whose rows are not 16-byte aligned (F or P not a multiple of 4)
// and the last partial stage take a scalar fill.
#include "gram_internal.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int SY_ROWS = 64;                       // rows per stage
constexpr int SY_PC = 128;                        // panel columns
constexpr int SY_PANEL = SY_ROWS * SY_PC;         // floats per panel (32 KB)
constexpr int SY_KSTEPS = SY_ROWS / 4;
constexpr int SY_FLUSH_STAGES = 1024 / SY_ROWS;   // float32 sums of at most 1024 rows
constexpr int SY_SLAB = 4096;                     // doubles per wave: [q1][q2][reg][lane]
constexpr size_t SY_CONST_BYTES = 256;            // [1,0,0,0, 0,0,0,0] in front of the slabs

#define WM_GLDS16(src, dst)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),    \
                                     (__attribute__((address_space(3))) void *)(dst), 16, 0, 0)

struct SyTile {
    int bi, bj;      // 64-column blocks of the A and B side
    int wi, wj;      // which half of panel I / panel J
    int ks, kpart;   // the k-steps of a stage are dealt kpart, kpart + ks, ...
    bool active;
};

// Tile and k-share of wave w in the workgroup of superblock pair (I, J); NB64 = 64-column blocks.
__host__ __device__ inline SyTile sy_wave_tile(int I, int J, int NB64, int w)
{
    int tb[4][2];
    int T = 0;
    for (int wi = 0; wi < 2; ++wi)
        for (int wj = 0; wj < 2; ++wj) {
            const int bi = 2 * I + wi, bj = 2 * J + wj;
            if (bi <= bj && bj < NB64) {
                tb[T][0] = wi;
                tb[T][1] = wj;
                ++T;
            }
        }
    SyTile t;
    t.ks = T == 1 ? 4 : (T == 2 ? 2 : 1);
    const int ti = w / t.ks;
    t.kpart = w % t.ks;
    t.active = ti < T;
    t.wi = t.active ? tb[ti][0] : 0;
    t.wj = t.active ? tb[ti][1] : 0;
    t.bi = 2 * I + t.wi;
    t.bj = 2 * J + t.wj;
    return t;
}

__host__ __device__ inline void sy_pair_decode(int pair, int NSB, int *I, int *J)
{
    int i = 0;
    while (pair >= NSB - i) {
        pair -= NSB - i;
        ++i;
    }
    *I = i;
    *J = i + pair;
}

__device__ __forceinline__ float sy_zval(const float *__restrict__ X, const float *__restrict__ Y,
                                         long long row, int v, long long n, int F, int P)
{
    const int M = F + P;
    if (row >= n || v > M) return 0.0f;
    if (v < F) return X[row * F + v];
    if (v < M) return Y[row * P + (v - F)];
    return 1.0f;
}

template <bool DIAG>
__device__ __forceinline__ void sy_mfma(const f32x4 a, const f32x4 b, f32x4 (&c)[16])
{
#pragma unroll
    for (int q1 = 0; q1 < 4; ++q1)
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2)
            if (!DIAG || q1 <= q2)
                c[q1 * 4 + q2] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(a[q1], b[q2], c[q1 * 4 + q2], 0, 0, 0);
}

// The k-steps kpart, kpart + ks, ... of one stage (ks = 1, 2 or 4).  Operands of the next step are
// read while the MFMAs of this one run (two register sets); the loop is NOT unrolled: fully
// unrolled, hipcc hoists the reads of all sixteen steps (92 VGPRs spilled).
template <bool DIAG>
__device__ __forceinline__ void sy_stage(const float *pa, const float *pb, int kpart, int ks,
                                         f32x4 (&c)[16])
{
    const int stride = ks * 4 * SY_PC;          // floats between two of this wave's steps
    pa += kpart * 4 * SY_PC;
    pb += kpart * 4 * SY_PC;
    f32x4 a0 = wm_load4(pa), b0 = DIAG ? a0 : wm_load4(pb);
#pragma unroll 1
    for (int s = SY_KSTEPS / 2; s > 0; s -= ks) {       // two steps per trip
        const f32x4 a1 = wm_load4(pa + stride), b1 = DIAG ? a1 : wm_load4(pb + stride);
        sy_mfma<DIAG>(a0, b0, c);
        pa += 2 * stride;
        pb += 2 * stride;
        if (s > ks) {
            a0 = wm_load4(pa);
            b0 = DIAG ? a0 : wm_load4(pb);
        }
        sy_mfma<DIAG>(a1, b1, c);
    }
}

// grid = npairs * S workgroups.  slabs: [S][npairs][WAVES][SY_SLAB] float64.
// FAST: rows are 16-byte aligned and n is a multiple of SY_ROWS (LDS-direct fill); otherwise a
// scalar fill that takes any shape and zero-fills the rows past n.
template <bool FAST>
__global__ __launch_bounds__(BLOCK, 2) void gram_syrk_kernel(const float *__restrict__ X,
                                                             const float *__restrict__ Y,
                                                             long long n, int F, int P, int NB64,
                                                             int NSB, int npairs, int S,
                                                             const float *__restrict__ consts,
                                                             double *__restrict__ slabs)
{
    __shared__ __attribute__((aligned(16))) float smem[2 * SY_PANEL];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int M = F + P;
    // workgroups of one row slice share an XCD (blockIdx % 8) when the slices divide over the eight
    int pair, slice;
    if (S % 8 == 0) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        pair = local % npairs;
        slice = (local / npairs) * 8 + xcd;
    } else {
        pair = blockIdx.x % npairs;
        slice = blockIdx.x / npairs;
    }
    int I, J;
    sy_pair_decode(pair, NSB, &I, &J);
    const bool two_panels = I != J;
    const SyTile tile = sy_wave_tile(I, J, NB64, wv);
    const bool diag = tile.bi == tile.bj;

    const long long stages = (n + SY_ROWS - 1) / SY_ROWS;
    const long long t0 = stages * slice / S, t1 = stages * (slice + 1) / S;

    // fill: lane (lr, lc) of a wave instruction takes 4 columns at lc of panel row 2k + lr
    const int lr = lane >> 5, lc = 4 * (lane & 31);
    const float *src[2];
    long long sstride[2];
    if (FAST) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int v = (p == 0 ? I : J) * SY_PC + lc;
            if (v < F) { src[p] = X + v; sstride[p] = F; }
            else if (v < M) { src[p] = Y + (v - F); sstride[p] = P; }
            else if (v == M) { src[p] = consts; sstride[p] = 0; }
            else { src[p] = consts + 4; sstride[p] = 0; }
            src[p] += (t0 * SY_ROWS + 2 * wv + lr) * sstride[p];
            sstride[p] *= 2 * WAVES;                       // rows between two of a wave's loads
        }
    }
    // operands: lane (i, g) reads columns 4i..4i+3 of row 4s + g of its tile's 64-column half
    const int oi = lane & 15, og = lane >> 4;
    const float *pa = smem + og * SY_PC + tile.wi * 64 + 4 * oi;
    const float *pb = smem + (two_panels ? SY_PANEL : 0) + og * SY_PC + tile.wj * 64 + 4 * oi;

    f32x4 c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    double d[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) d[k] = 0.0;
    int since = 0;

    for (long long t = t0; t < t1; ++t) {
        __syncthreads();                                  // every wave is past its reads
        if (FAST) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (p == 1 && !two_panels) break;
#pragma unroll
                for (int m = 0; m < SY_ROWS / 2 / WAVES; ++m) {
                    WM_GLDS16(src[p], smem + p * SY_PANEL + 2 * (wv + WAVES * m) * SY_PC);
                    src[p] += sstride[p];
                }
            }
        } else {
            const long long row0 = t * SY_ROWS;
            for (int p = 0; p < (two_panels ? 2 : 1); ++p) {
                const int vb = (p == 0 ? I : J) * SY_PC;
                for (int e = tid; e < SY_PANEL; e += BLOCK)
                    smem[p * SY_PANEL + e] =
                        sy_zval(X, Y, row0 + e / SY_PC, vb + e % SY_PC, n, F, P);
            }
        }
        __syncthreads();                                  // vmcnt(0) + barrier: the panels landed
        if (tile.active) {
            if (diag) sy_stage<true>(pa, pa, tile.kpart, tile.ks, c);
            else sy_stage<false>(pa, pb, tile.kpart, tile.ks, c);
        }
        if (++since == SY_FLUSH_STAGES) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d[k * 4 + r] += (double)c[k][r];
                    c[k][r] = 0.f;
                }
            since = 0;
        }
    }
    if (!tile.active) return;
    double *dst = slabs + (((size_t)slice * npairs + pair) * WAVES + wv) * SY_SLAB;
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(k * 4 + r) * 64 + lane] = d[k * 4 + r] + (double)c[k][r];
}

__global__ void gram_syrk_consts_kernel(float *consts)
{
    if (threadIdx.x < 8) consts[threadIdx.x] = threadIdx.x == 0 ? 1.0f : 0.0f;
}

// grid = (64 x 64 tiles (bi <= bj), 16), one slab element per thread.  Sums the slabs of every
// slice and of every wave that shared the tile, in a fixed order, and adds to acc (both triangles).
// (One workgroup per tile with 16 elements per thread took 1.5 ms at F = 432: 66 workgroups, each
// thread a chain of 16 x S dependent-latency loads.)
__global__ __launch_bounds__(BLOCK) void gram_syrk_finalize_kernel(const double *__restrict__ slabs,
                                                                   int F, int P, int NB64, int NSB,
                                                                   int npairs, int S,
                                                                   double *__restrict__ acc)
{
    const int M = F + P, D = M + 1;
    int bi = 0, rest = blockIdx.x;
    while (rest >= NB64 - bi) {
        rest -= NB64 - bi;
        ++bi;
    }
    const int bj = bi + rest;
    const int I = bi >> 1, J = bj >> 1;
    int pair = 0;
    for (int i = 0; i < I; ++i) pair += NSB - i;
    pair += J - I;
    int wlist[WAVES], nw = 0;
    for (int w = 0; w < WAVES; ++w) {
        const SyTile t = sy_wave_tile(I, J, NB64, w);
        if (t.active && t.bi == bi && t.bj == bj) wlist[nw++] = w;
    }
    const bool diag = bi == bj;
    {
        const int e = blockIdx.y * BLOCK + threadIdx.x;
        const int lane = e & 63, r = (e >> 6) & 3, q2 = (e >> 8) & 3, q1 = e >> 10;
        if (diag && q1 > q2) return;
        const int va = bi * 64 + 4 * ((lane >> 4) * 4 + r) + q1;
        const int vb = bj * 64 + 4 * (lane & 15) + q2;
        if (va > M || vb > M) return;
        double s = 0.0;
        for (int k = 0; k < nw; ++k) {
            const double *src = slabs + ((size_t)pair * WAVES + wlist[k]) * SY_SLAB + e;
            const size_t step = (size_t)npairs * WAVES * SY_SLAB;
#pragma unroll 8
            for (int sl = 0; sl < S; ++sl) s += src[sl * step];
        }
        // virtual order [X | Y | 1] -> acc order [X | 1 | Y]
        const int a = va < F ? va : (va < M ? va + 1 : F);
        const int b = vb < F ? vb : (vb < M ? vb + 1 : F);
        acc[(size_t)a * D + b] += s;
        if (a != b && !(diag && q1 == q2)) acc[(size_t)b * D + a] += s;
    }
}

struct SyPlan {
    int NB64, NSB, npairs, S;
};

SyPlan sy_plan(long long n, int F, int P)
{
    SyPlan p;
    const int D = F + P + 1;
    p.NB64 = (D + 63) / 64;
    p.NSB = (p.NB64 + 1) / 2;
    p.npairs = p.NSB * (p.NSB + 1) / 2;
    const long long stages = (n + SY_ROWS - 1) / SY_ROWS;
    long long S = ((long long)wm_num_cus() * 2) / p.npairs;      // two workgroups per CU
    if (S >= 8) S -= S % 8;                                      // whole XCD rounds
    if (S > stages) S = stages;
    if (S < 1) S = 1;
    p.S = (int)S;
    return p;
}

}  // namespace

size_t wm_gram_syrk_workspace_bytes(long long n, int F, int P)
{
    if (n <= 0 || F <= 0 || P <= 0 || wm_num_cus() <= 0) return 0;
    const SyPlan p = sy_plan(n, F, P);
    return SY_CONST_BYTES + (size_t)p.S * p.npairs * WAVES * SY_SLAB * sizeof(double);
}

int wm_gram_syrk_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                        void *workspace, hipStream_t st)
{
    float *consts = (float *)workspace;
    double *slabs = (double *)((char *)workspace + SY_CONST_BYTES);
    const bool fast = F % 4 == 0 && P % 4 == 0 && (((uintptr_t)X) & 15) == 0 &&
                      (((uintptr_t)Y) & 15) == 0;
    // aligned shapes: whole stages through the LDS-direct kernel, the last n % 64 rows through the
    // scalar-fill kernel (a second pass over the same slabs, stream-ordered)
    const long long n_fast = fast ? n - n % SY_ROWS : 0;
    for (int pass = 0; pass < 2; ++pass) {
        const long long rows = pass == 0 ? n_fast : n - n_fast;
        if (rows == 0) continue;
        const float *Xp = pass == 0 ? X : X + n_fast * F;
        const float *Yp = pass == 0 ? Y : Y + n_fast * P;
        const SyPlan p = sy_plan(rows, F, P);
        if (pass == 0) {
            hipLaunchKernelGGL(gram_syrk_consts_kernel, dim3(1), dim3(64), 0, st, consts);
            WM_LAUNCH_CHECK("gram_syrk_consts_kernel");
            hipLaunchKernelGGL(gram_syrk_kernel<true>, dim3(p.npairs * p.S), dim3(BLOCK), 0, st, Xp,
                               Yp, rows, F, P, p.NB64, p.NSB, p.npairs, p.S,
                               (const float *)consts, slabs);
        } else {
            hipLaunchKernelGGL(gram_syrk_kernel<false>, dim3(p.npairs * p.S), dim3(BLOCK), 0, st, Xp,
                               Yp, rows, F, P, p.NB64, p.NSB, p.npairs, p.S,
                               (const float *)consts, slabs);
        }
        WM_LAUNCH_CHECK("gram_syrk_kernel");
        const int ntiles = p.NB64 * (p.NB64 + 1) / 2;
        hipLaunchKernelGGL(gram_syrk_finalize_kernel, dim3(ntiles, SY_SLAB / BLOCK), dim3(BLOCK), 0, st,
                           (const double *)slabs, F, P, p.NB64, p.NSB, p.npairs, p.S, acc);
        WM_LAUNCH_CHECK("gram_syrk_finalize_kernel");
    }
    return WM_OK;
}
