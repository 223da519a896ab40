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
#include <stdlib.h>

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
    int ti;
    if (T == 3) {
        // a diagonal superblock: two diagonal tiles (10 MFMAs per k-step) and one full tile (16);
        // the full one is split over two waves, so the waves carry 10 / 8 / 8 / 10
        ti = w == 0 ? 0 : (w == 3 ? 2 : 1);
        t.ks = ti == 1 ? 2 : 1;
        t.kpart = ti == 1 ? w - 1 : 0;
    } else {
        t.ks = T == 1 ? 4 : (T == 2 ? 2 : 1);
        ti = w / t.ks;
        t.kpart = w % t.ks;
    }
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

// ---- who does what.  Two schedules:
//  mode 0  grid = npairs x S: workgroup (pair, slice) walks one contiguous slice of the stages.
//          Simple, but a full pair costs 16 MFMAs per k-step and wave, a diagonal one 10, an edge
//          one 8: with one workgroup per (pair, slice) the CUs that drew two full pairs set the
//          time and a quarter of the chip idles (F = 432, P = 216: 97 TFLOP/s; four times as many
//          slices, dealt dynamically: 117, but 528 MB of slabs).
//  mode 1  balanced and persistent: ONE workgroup of eight waves per CU -- two halves of four waves,
//          each the workgroup of mode 0 on its own stage and LDS buffer, so what shares a SIMD is
//          two waves of the same cost.  The chip is cut into G groups of XCDs (blockIdx % 8), every
//          group takes a contiguous share of the stages, and inside a group pair p gets w[p]
//          workgroups IN PROPORTION TO ITS COST, whose 2 w[p] halves deal the group's stages
//          round-robin (half h: stages h, h + 2 w[p], ...).  All pairs then advance through the
//          rows at the same rate, so the workgroups of a group stream one compact window (cache
//          hits for all but the first reader) and finish together; one slab per wave.
//          (Two independent four-wave workgroups per CU with the same dealing gained 4 %: the
//          dispatcher co-locates workgroups of different cost, and a SIMD shared by a 16- and a
//          10-MFMA wave runs both at the pace of 26.)
constexpr int SY_MAXP = 64;     // pairs a mode-1 plan can hold
constexpr int SY_G = 2;         // XCD groups (four XCDs each)

struct SyPlan {
    int NB64, NSB, npairs, mode;
    int S;                      // mode 0: slices
    int slots, used;            // mode 1: workgroup slots per group, slots in use
    int yslow;                  // FAST kernels: Y rows are not 16-byte pieces -- its columns are filled by hand
    short w[SY_MAXP], off[SY_MAXP + 1];
};

struct SyWork {
    int pair;
    long long t_first, t_end, t_step;
    size_t slab_wg;             // slab index of the workgroup (x WAVES + wave)
    bool valid;
};

__device__ __forceinline__ SyWork sy_assign(const SyPlan &pl, int bid, int half, long long stages)
{
    SyWork k;
    k.valid = true;
    if (pl.mode == 0) {
        int slice;
        if (pl.S % 8 == 0) {     // workgroups of one slice share an XCD
            const int xcd = bid & 7, local = bid >> 3;
            k.pair = local % pl.npairs;
            slice = (local / pl.npairs) * 8 + xcd;
        } else {
            k.pair = bid % pl.npairs;
            slice = bid / pl.npairs;
        }
        k.t_first = stages * slice / pl.S;
        k.t_end = stages * (slice + 1) / pl.S;
        k.t_step = 1;
        k.slab_wg = (size_t)slice * pl.npairs + k.pair;
    } else {
        constexpr int PER = 8 / SY_G;                    // XCDs per group
        const int xcd = bid & 7, g = xcd / PER;
        const int l = (bid >> 3) * PER + xcd % PER;      // slot inside the group
        k.valid = l < pl.used;
        int p = 0;
        while (p + 1 < pl.npairs && pl.off[p + 1] <= l) ++p;
        k.pair = p;
        k.t_first = stages * g / SY_G + 2 * (l - pl.off[p]) + half;
        k.t_end = stages * (g + 1) / SY_G;
        k.t_step = 2 * pl.w[p];
        k.slab_wg = (size_t)bid * 2 + half;
    }
    return k;
}

// slabs: [workgroup (x 2 halves)][WAVES][SY_SLAB] float64.
// FAST: rows are 16-byte aligned and n is a multiple of SY_ROWS (LDS-direct fill); otherwise a
// scalar fill that takes any shape and zero-fills the rows past n.
// HALVES: 1 = mode 0 (four waves), 2 = mode 1 (eight waves, two stages in flight).
template <bool FAST, int HALVES, bool YSLOW = false>
__global__ __launch_bounds__(BLOCK * HALVES, 2) void gram_syrk_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, long long n, int F, int P,
    const SyPlan pl, const float *__restrict__ consts, double *__restrict__ slabs)
{
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int tid = threadIdx.x & (BLOCK - 1), lane = tid & 63, wv = tid >> 6;
    const int half = threadIdx.x / BLOCK;                 // wave-uniform
    float *smem = smem_all + half * 2 * SY_PANEL;
    const int M = F + P;
    const long long stages = (n + SY_ROWS - 1) / SY_ROWS;
    const SyWork wk = sy_assign(pl, blockIdx.x, half, stages);
    if (!wk.valid) return;                                // a spare slot: uniform, before any barrier
    int I, J;
    sy_pair_decode(wk.pair, pl.NSB, &I, &J);
    const bool two_panels = I != J;
    const SyTile tile = sy_wave_tile(I, J, pl.NB64, wv);
    const bool diag = tile.bi == tile.bj;

    // fill: lane (lr, lc) of a wave instruction takes 4 columns at lc of panel row 2k + lr
    const int lr = lane >> 5, lc = 4 * (lane & 31);
    const float *src[2];
    long long sstride[2], sskip[2];
    bool dma[2] = {true, true};
    if (FAST) {
        // Y with P % 4 != 0 (P = 1 above all) has no 16-byte pieces to hand to the LDS-direct loads:
        // then only the X columns come by DMA (lanes on other columns sit the instruction out), the
        // padding columns are zeroed once here, and every fill writes the P + 1 columns of Y and of
        // the ones by hand.  (Before round 3 such shapes took the scalar-fill kernel for EVERYTHING:
        // F = 32, P = 1 at 0.3 TB/s.)
        if (YSLOW) {
            for (int e = tid * 4; e < 2 * SY_PANEL; e += BLOCK * 4) wm_store4(smem + e, (f32x4){0.f, 0.f, 0.f, 0.f});
            __syncthreads();
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int v = (p == 0 ? I : J) * SY_PC + lc;
            dma[p] = !(YSLOW && v >= F);
            if (v < F) { src[p] = X + v; sstride[p] = F; }
            else if (v < M) { src[p] = Y + (v - F); sstride[p] = P; }
            else if (v == M) { src[p] = consts; sstride[p] = 0; }
            else { src[p] = consts + 4; sstride[p] = 0; }
            src[p] += (wk.t_first * SY_ROWS + 2 * wv + lr) * sstride[p];
            sskip[p] = (wk.t_step - 1) * SY_ROWS * sstride[p];    // the stages of the other workgroups
            sstride[p] *= 2 * WAVES;                              // rows between two of a wave's loads
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

    // both halves run the same number of trips (the barriers are the workgroup's): half 0 never
    // has fewer stages than half 1, and a half past its last stage idles through the barriers
    const long long first0 = wk.t_first - half;
    const long long trips = first0 < wk.t_end ? (wk.t_end - first0 + wk.t_step - 1) / wk.t_step : 0;
    long long t_load = wk.t_first;                        // the stage the next fill brings in

    auto fill = [&]() {
        if (t_load >= wk.t_end) return;
        if (FAST) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (p == 1 && !two_panels) break;
#pragma unroll
                for (int m = 0; m < SY_ROWS / 2 / WAVES; ++m) {
                    if (!YSLOW || dma[p]) WM_GLDS16(src[p], smem + p * SY_PANEL + 2 * (wv + WAVES * m) * SY_PC);
                    src[p] += sstride[p];
                }
                src[p] += sskip[p];
            }
            if (YSLOW) {
                const long long row0 = t_load * SY_ROWS;
                for (int p = 0; p < (two_panels ? 2 : 1); ++p) {
                    const int vb = (p == 0 ? I : J) * SY_PC;
                    const int c0 = F > vb ? F : vb, c1 = M < vb + SY_PC - 1 ? M : vb + SY_PC - 1;
                    const int nc = c1 - c0 + 1;
                    for (int e = tid; e < SY_ROWS * nc; e += BLOCK) {
                        const int r = e / nc, c = c0 + e - r * nc;
                        smem[p * SY_PANEL + r * SY_PC + (c - vb)] =
                            c < M ? Y[(row0 + r) * P + (c - F)] : 1.0f;
                    }
                }
            }
        } else {
            const long long row0 = t_load * SY_ROWS;
            for (int p = 0; p < (two_panels ? 2 : 1); ++p) {
                const int vb = (p == 0 ? I : J) * SY_PC;
                for (int e = tid; e < SY_PANEL; e += BLOCK)
                    smem[p * SY_PANEL + e] =
                        sy_zval(X, Y, row0 + e / SY_PC, vb + e % SY_PC, n, F, P);
            }
        }
        t_load += wk.t_step;
    };
    auto multiply = [&](bool live) {
        if (tile.active && live) {
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
    };
    if (HALVES == 1) {
        for (long long it = 0; it < trips; ++it) {
            __syncthreads();                              // every wave is past its reads
            fill();
            __syncthreads();                              // vmcnt(0) + barrier: the panels landed
            multiply(true);
        }
    } else {
        // The halves alternate: while one fills its buffer the other multiplies, so the LDS-direct
        // loads of a CU always run beside MFMAs.  (In step -- both filling, then both multiplying
        // -- the pass took fill time PLUS multiply time: 5.6 ms where the MFMAs alone take 3.85.)
        long long t_mul = wk.t_first;                     // the stage the next multiply works on
        if (half == 1) fill();
        __syncthreads();
        for (long long it = 0; it < trips; ++it) {
            if (half == 0) fill();
            else { multiply(t_mul < wk.t_end); t_mul += wk.t_step; }
            __syncthreads();
            if (half == 0) { multiply(t_mul < wk.t_end); t_mul += wk.t_step; }
            else fill();
            __syncthreads();
        }
    }
    if (!tile.active) return;
    double *dst = slabs + (wk.slab_wg * WAVES + wv) * SY_SLAB;
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
// workgroup that worked on the tile's pair and of every wave that shared the tile, in a fixed
// order, and adds to acc (both triangles).  (One workgroup per tile with 16 elements per thread
// took 1.5 ms at F = 432: 66 workgroups, each thread a chain of 16 x S dependent-latency loads.)
__global__ __launch_bounds__(BLOCK) void gram_syrk_finalize_kernel(const double *__restrict__ slabs,
                                                                   int F, int P, const SyPlan pl,
                                                                   double *__restrict__ acc)
{
    const int M = F + P, D = M + 1;
    const int NB64 = pl.NB64, NSB = pl.NSB;
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
    const int e = blockIdx.y * BLOCK + threadIdx.x;
    const int lane = e & 63, r = (e >> 6) & 3, q2 = (e >> 8) & 3, q1 = e >> 10;
    if (diag && q1 > q2) return;
    const int va = bi * 64 + 4 * ((lane >> 4) * 4 + r) + q1;
    const int vb = bj * 64 + 4 * (lane & 15) + q2;
    if (va > M || vb > M) return;
    double s = 0.0;
    if (pl.mode == 0) {
        for (int k = 0; k < nw; ++k) {
            const double *src = slabs + ((size_t)pair * WAVES + wlist[k]) * SY_SLAB + e;
            const size_t step = (size_t)pl.npairs * WAVES * SY_SLAB;
#pragma unroll 8
            for (int sl = 0; sl < pl.S; ++sl) s += src[sl * step];
        }
    } else {
        constexpr int PER = 8 / SY_G;
        const int wp = pl.w[pair], o = pl.off[pair];
        for (int k = 0; k < nw; ++k)
            for (int g = 0; g < SY_G; ++g) {
#pragma unroll 4
                for (int j = 0; j < 2 * wp; ++j) {
                    const int l = o + (j >> 1);
                    const size_t bid = (size_t)(l / PER) * 8 + g * PER + l % PER;
                    s += slabs[((bid * 2 + (j & 1)) * WAVES + wlist[k]) * SY_SLAB + e];
                }
            }
    }
    // virtual order [X | Y | 1] -> acc order [X | 1 | Y]
    const int a = va < F ? va : (va < M ? va + 1 : F);
    const int b = vb < F ? vb : (vb < M ? vb + 1 : F);
    acc[(size_t)a * D + b] += s;
    if (a != b && !(diag && q1 == q2)) acc[(size_t)b * D + a] += s;
}

// MFMAs per k-step on the busiest wave of the workgroup of pair (I, J): what its stages cost
float sy_pair_cost(int I, int J, int NB64)
{
    int T = 0;
    for (int wi = 0; wi < 2; ++wi)
        for (int wj = 0; wj < 2; ++wj)
            if (2 * I + wi <= 2 * J + wj && 2 * J + wj < NB64) ++T;
    return T == 4 ? 16.f : (T == 3 ? 10.f : (T == 2 ? 8.f : 2.5f));
}

SyPlan sy_plan(long long n, int F, int P, bool fast)
{
    SyPlan p = {};
    const int D = F + P + 1;
    p.NB64 = (D + 63) / 64;
    p.NSB = (p.NB64 + 1) / 2;
    p.npairs = p.NSB * (p.NSB + 1) / 2;
    const long long stages = (n + SY_ROWS - 1) / SY_ROWS;
    const int wgs = wm_num_cus() * 2;                            // two four-wave workgroups per CU
    int slots = wm_num_cus() / SY_G;                             // mode 1: one eight-wave workgroup
    slots -= slots % (8 / SY_G);
    bool balanced = fast && wm_num_cus() % 8 == 0 && p.npairs <= SY_MAXP && p.npairs <= slots &&
                    stages >= 4LL * wgs;
#ifdef WM_PROBE_ENV
    if (const char *e = getenv("WM_SY_MODE")) balanced = balanced && atoi(e) != 0;
#endif
    if (balanced) {
        p.mode = 1;
        p.slots = slots;
        float cost[SY_MAXP], total = 0.f;
        for (int q = 0; q < p.npairs; ++q) {
            int I, J;
            sy_pair_decode(q, p.NSB, &I, &J);
            cost[q] = sy_pair_cost(I, J, p.NB64);
            total += cost[q];
        }
        // proportional shares, at least one workgroup each, the rest by largest remainder
        float frac[SY_MAXP];
        int given = 0;
        for (int q = 0; q < p.npairs; ++q) {
            const float share = slots * cost[q] / total;
            int w = (int)share;
            if (w < 1) w = 1;
            p.w[q] = (short)w;
            frac[q] = share - w;
            given += w;
        }
        while (given > slots) {               // the "at least one" floor overshot: trim the largest
            int big = 0;
            for (int q = 1; q < p.npairs; ++q)
                if (p.w[q] > p.w[big]) big = q;
            --p.w[big];
            --given;
        }
        while (given < slots) {
            int best = 0;
            for (int q = 1; q < p.npairs; ++q)
                if (frac[q] > frac[best]) best = q;
            if (frac[best] <= 0.f) break;
            ++p.w[best];
            frac[best] = -1.f;
            ++given;
        }
        p.off[0] = 0;
        for (int q = 0; q < p.npairs; ++q) p.off[q + 1] = (short)(p.off[q] + p.w[q]);
        p.used = given;
        return p;
    }
    p.mode = 0;
    long long S = (long long)wgs / p.npairs;
    if (S >= 8) S -= S % 8;                                      // whole XCD rounds
#ifdef WM_PROBE_ENV
    if (const char *e = getenv("WM_SY_SLICES")) S = atoll(e);
#endif
    if (S > stages) S = stages;
    if (S < 1) S = 1;
    p.S = (int)S;
    return p;
}

int sy_lds_once(const void *fn, size_t bytes, int slot)
{
    static bool done[64][5] = {};
    int dev = 0;
    WM_HIP(hipGetDevice(&dev));
    bool &d = done[dev & 63][slot];
    if (!d) {
        WM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        d = true;
    }
    return WM_OK;
}

size_t sy_plan_wgs(const SyPlan &p)      // workgroups launched
{
    return p.mode == 0 ? (size_t)p.npairs * p.S : (size_t)p.slots * SY_G;
}
size_t sy_plan_slabs(const SyPlan &p)    // four-wave slab groups written
{
    return p.mode == 0 ? (size_t)p.npairs * p.S : (size_t)p.slots * SY_G * 2;
}

}  // namespace

size_t wm_gram_syrk_workspace_bytes(long long n, int F, int P)
{
    if (n <= 0 || F <= 0 || P <= 0 || wm_num_cus() <= 0) return 0;
    // the alignment of X and Y is not known here: room for either schedule
    const size_t a = sy_plan_slabs(sy_plan(n, F, P, true)), b = sy_plan_slabs(sy_plan(n, F, P, false));
    return SY_CONST_BYTES + (a > b ? a : b) * WAVES * SY_SLAB * sizeof(double);
}

int wm_gram_syrk_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                        void *workspace, hipStream_t st)
{
    float *consts = (float *)workspace;
    double *slabs = (double *)((char *)workspace + SY_CONST_BYTES);
    const bool fast = F % 4 == 0 && (((uintptr_t)X) & 15) == 0;
    const bool yslow = P % 4 != 0 || (((uintptr_t)Y) & 15) != 0;
    // aligned shapes: whole stages through the LDS-direct kernel, the last n % 64 rows through the
    // scalar-fill kernel (a second pass over the same slabs, stream-ordered)
    const long long n_fast = fast ? n - n % SY_ROWS : 0;
    for (int pass = 0; pass < 2; ++pass) {
        const long long rows = pass == 0 ? n_fast : n - n_fast;
        if (rows == 0) continue;
        const float *Xp = pass == 0 ? X : X + n_fast * F;
        const float *Yp = pass == 0 ? Y : Y + n_fast * P;
        SyPlan p = sy_plan(rows, F, P, pass == 0);
        p.yslow = pass == 0 && yslow;
        const int grid = (int)sy_plan_wgs(p);
        const size_t lds = 2 * SY_PANEL * sizeof(float);
        if (pass == 0) {
            hipLaunchKernelGGL(gram_syrk_consts_kernel, dim3(1), dim3(64), 0, st, consts);
            WM_LAUNCH_CHECK("gram_syrk_consts_kernel");
#define SY_LAUNCH_FAST(HALVES_, YSLOW_, SLOT_)                                                    \
    {                                                                                             \
        int rc = sy_lds_once((const void *)gram_syrk_kernel<true, HALVES_, YSLOW_>, HALVES_ * lds, \
                             SLOT_);                                                              \
        if (rc != WM_OK) return rc;                                                               \
        hipLaunchKernelGGL((gram_syrk_kernel<true, HALVES_, YSLOW_>), dim3(grid),                 \
                           dim3(HALVES_ * BLOCK), HALVES_ * lds, st, Xp, Yp, rows, F, P, p,       \
                           (const float *)consts, slabs);                                         \
    }
            if (p.mode == 1) {
                if (yslow) SY_LAUNCH_FAST(2, true, 3) else SY_LAUNCH_FAST(2, false, 0)
            } else {
                if (yslow) SY_LAUNCH_FAST(1, true, 4) else SY_LAUNCH_FAST(1, false, 1)
            }
#undef SY_LAUNCH_FAST
        } else {
            int rc = sy_lds_once((const void *)gram_syrk_kernel<false, 1>, lds, 2);
            if (rc != WM_OK) return rc;
            hipLaunchKernelGGL((gram_syrk_kernel<false, 1>), dim3(grid), dim3(BLOCK), lds, st, Xp, Yp,
                               rows, F, P, p, (const float *)consts, slabs);
        }
        WM_LAUNCH_CHECK("gram_syrk_kernel");
        const int ntiles = p.NB64 * (p.NB64 + 1) / 2;
        hipLaunchKernelGGL(gram_syrk_finalize_kernel, dim3(ntiles, SY_SLAB / BLOCK), dim3(BLOCK), 0, st,
                           (const double *)slabs, F, P, p, acc);
        WM_LAUNCH_CHECK("gram_syrk_finalize_kernel");
    }
    return WM_OK;
}
