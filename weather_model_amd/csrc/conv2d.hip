// conv2d.hip -- valid / stride-1 cross-correlation forward, backward-data and backward-weight+bias.
//
// Replaces torch.nn.functional.conv2d and its autograd triple as the reference reaches them from
// nn.Conv2d (newConvolution.py:129-133 layers, :149-158 forward calls, :184 backward).
//
// Two kernel families:
//   * stencil (Cin == Cout == 1, small kH x kW): the memory-bound single-channel grid stencil of
//     BASELINE config 3.  Every thread owns a 4-column strip and walks a contiguous range of the
//     N*OH output rows with a kH-row register window, so each input row is read once per thread
//     (plus kH-1 halo rows per range) with 16-byte row-major loads; the next row is prefetched
//     while the current one is computed.  The row ranges are cut so that the launch fills the
//     chip exactly once (no tail wave).  Algorithmic traffic: 8 B per grid point for forward,
//     8 B for backward-weight+bias.
//   * generic direct kernels for every other shape (correctness for the whole operator class).
//
// The weight-gradient reduction is wavefront-shuffle -> LDS -> float64 per-block partials in the
// caller's workspace -> a fixed-order float64 finalize kernel (bitwise reproducible).
#include "common.h"

#include <map>
#include <mutex>
#include <tuple>

namespace {

constexpr int BLOCK = 256;

// ------------------------------------------------------------------------------------------
// Stencil kernels (Cin = Cout = 1).
//
// The batch is ONE stream of input rows (images are contiguous, so the row after the last row of
// image n is row 0 of image n+1).  A thread owns a 4-column strip and a contiguous range of the
// N*OH output rows; it loads every input row that range needs exactly once, in order, and keeps
// the last KH rows in a register window.  Output row (n, h-KH+1) is produced at the step that
// loads input row (n, h), for h >= hmin.  Rows are taken RB at a time: the loads of all RB rows
// are issued back to back (a sched_barrier keeps hipcc from sinking them to their uses), so
// RB x 24-40 B per thread are in flight while the previous rows are consumed.
//
// Every load is unconditional and in bounds, so hipcc never drains vmcnt at a control-flow join
// (a bounds branch around a load costs a full memory latency per row): the last strip is shifted
// left to start at column OW-4 instead of being masked, so a strip always reads columns
// j0 .. j0+3+KW-1 <= W-1 and writes columns j0 .. j0+3 <= OW-1; the columns it shares with its
// neighbour are written twice with identical values (forward) or masked out of the sums by a
// select (backward).  All per-thread index arithmetic is 32-bit; only row pointers are 64-bit.
#ifndef WM_STENCIL_RB
#define WM_STENCIL_RB 4
#endif
#ifndef WM_STENCIL_BPC
#define WM_STENCIL_BPC 6
#endif
constexpr int RB = WM_STENCIL_RB;
// forward flat kernel: load batches per range, consecutive ranges per XCD, minimum dynamic LDS per
// workgroup (an occupancy cap: 160 KB per CU / this)
#ifndef WM_FWD_RB
#define WM_FWD_RB 2
#endif
constexpr int FRB = WM_FWD_RB;      // rows per load batch of the flat forward kernel
// load batches of FRB rows per range: with non-temporal main loads the forward is fastest at
// three (89-91 us against 92-93 at two), backward-data at two (90 against 94)
#ifndef WM_FWD_BATCHES
#define WM_FWD_BATCHES 3
#endif
#ifndef WM_BWDD_BATCHES
#define WM_BWDD_BATCHES 2
#endif
#ifndef WM_FWD_XCD_RUN
#define WM_FWD_XCD_RUN 16
#endif
// Non-temporal loads of the single-use rows in the flat stencil kernels, measured per kernel on
// (64,1,721,1440) (tools/probe/stencil_time.py, same box): weight gradient 98.2 -> 91.4 us with its
// finalize launch, forward 93.3 -> 92.0 (89.4 with three batches), backward-data 90.7 -> 102.6 (its
// halo pair re-reads the neighbour lane's line).  The forward nevertheless loads TEMPORALLY: alone
// it is 4 us slower (93.5), but the step it belongs to is faster -- forward + weight gradient of the
// same x back to back 185.8-188.8 -> 182.8 us, with backward-data 285-289 -> 280.8 (in bench.py's
// loop 189.9/187.2 -> 186.8/185.8 = 0.71 of HBM peak): the rows of x it leaves in the 256 MiB
// Infinity Cache are rows the weight gradient does not fetch from HBM again.
#ifndef WM_NT_FWD
#define WM_NT_FWD false
#endif
#ifndef WM_NT_FWD_TAIL
#define WM_NT_FWD_TAIL false
#endif
#ifndef WM_NT_BWDW
#define WM_NT_BWDW true
#endif
// gy in the weight gradient is loaded temporally: the kernel itself does not care (89.8 us either
// way), but a backward-data pass that follows it re-reads gy, and the three passes of a step take
// 278 -> 264 us when part of gy is still in the Infinity Cache (walking backward-data from the
// last range to the first instead: 3 us slower)
#ifndef WM_NT_BWDW_G
#define WM_NT_BWDW_G false
#endif
#ifndef WM_NT_BWDD
#define WM_NT_BWDD false
#endif
// relu_mask_kernel (three streams: g, mask, out): non-temporal on all three and 2 workgroups per CU
// -- 572 -> 519 us on (8,32,721,1440); 4 per CU 555, 1 per CU 687 (tools/probe/mask_time.py): a
// read+write stream on this chip wants few bytes in flight
#ifndef WM_RM_NT
#define WM_RM_NT 1
#endif
#ifndef WM_RM_CAP
#define WM_RM_CAP 2
#endif
#ifndef WM_BWD_OCC_CAP
#define WM_BWD_OCC_CAP 6
#endif
#ifndef WM_FWD_LDS_MIN
#define WM_FWD_LDS_MIN 0
#endif
// probe builds (tools/probe/conv_variants.py, -DWM_PROBE_ENV) read the knobs from the environment
// at every call; the product build compiles them in
#ifdef WM_PROBE_ENV
#include <stdlib.h>
static int wm_knob(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}
#define WM_KNOB(name) wm_knob(#name, name)
#else
#define WM_KNOB(name) (name)
#endif

struct RowWalk {
    int steps;           // input rows this thread streams
    int h;               // row-in-image of the first input row
    int j0;              // first column of the strip
    int qmin;            // strip columns below qmin belong to the left neighbour
    size_t in_off;       // element offset of the first input row, at column j0
    size_t out_off;      // element offset of the first output row, at column j0
};

// thread -> (strip, contiguous range of output rows).  rows_base/rows_rem: the N*OH rows split
// into `segs` ranges of rows_base rows, the first rows_rem of them one row longer.
__device__ __forceinline__ bool make_walk(RowWalk &k, int H, int W, int OH, int OW, int KH,
                                          int strips, int segs, unsigned rows_base,
                                          unsigned rows_rem)
{
    const unsigned tid = blockIdx.x * BLOCK + threadIdx.x;
    if (tid >= (unsigned)strips * (unsigned)segs) return false;
    const unsigned seg = tid / (unsigned)strips;
    const unsigned strip = tid - seg * (unsigned)strips;
    const unsigned r0 = seg * rows_base + min(seg, rows_rem);
    const unsigned cnt = rows_base + (seg < rows_rem ? 1u : 0u);
    if (cnt == 0) return false;
    const unsigned n0 = r0 / (unsigned)OH, i0 = r0 - n0 * (unsigned)OH;
    const unsigned rl = r0 + cnt - 1;
    const unsigned n1 = rl / (unsigned)OH, i1 = rl - n1 * (unsigned)OH;
    k.steps = (int)((n1 - n0) * (unsigned)H + i1 + KH - i0);
    k.h = (int)i0;
    k.j0 = min((int)strip * 4, OW - 4);
    k.qmin = (int)strip * 4 - k.j0;
    k.in_off = ((size_t)n0 * H + i0) * W + k.j0;
    k.out_off = (size_t)r0 * OW + k.j0;
    return true;
}

// EXT = 4 + KW - 1 floats starting at p (always in bounds, 4-byte aligned).
template <int KW>
__device__ __forceinline__ void load_row(float (&dst)[4 + KW - 1], const float *__restrict__ p)
{
    constexpr int EXT = 4 + KW - 1;
    const f4u v = *reinterpret_cast<const f4u *>(p);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    if constexpr (KW - 1 == 2) {
        const f2u u = *reinterpret_cast<const f2u *>(p + 4);
        dst[4] = u.x; dst[5] = u.y;
    } else if constexpr (KW - 1 == 4) {
        const f4u u = *reinterpret_cast<const f4u *>(p + 4);
        dst[4] = u.x; dst[5] = u.y; dst[6] = u.z; dst[7] = u.w;
    } else {
#pragma unroll
        for (int e = 4; e < EXT; ++e) dst[e] = p[e];
    }
}

template <int KH, int KW, int NB>
__device__ __forceinline__ void fwd_batch(const float *__restrict__ &xq, float *__restrict__ &yq,
                                          float (&win)[KH][4 + KW - 1], const float (&wr)[KH][KW],
                                          float bias, int relu, int &h, int &hmin, int H, int W,
                                          int OW)
{
    constexpr int EXT = 4 + KW - 1;
    float rows[NB][EXT];
#pragma unroll
    for (int u = 0; u < NB; ++u) load_row<KW>(rows[u], xq + (size_t)u * W);
    xq += (size_t)NB * W;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
#pragma unroll
        for (int kh = 0; kh < KH - 1; ++kh)
#pragma unroll
            for (int e = 0; e < EXT; ++e) win[kh][e] = win[kh + 1][e];
#pragma unroll
        for (int e = 0; e < EXT; ++e) win[KH - 1][e] = rows[u][e];
        if (h >= hmin) {
            float out[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float acc = bias;
#pragma unroll
                for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw) acc = fmaf(wr[kh][kw], win[kh][q + kw], acc);
                out[q] = relu ? fmaxf(acc, 0.0f) : acc;
            }
            f4u v;
            v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
            // non-temporal: y is written once and not re-read by this kernel; a cached store leaves
            // ~256 MB of dirty lines for the NEXT kernel to pay for (measured: fwd+bwd 243 -> 218 us)
            __builtin_nontemporal_store(v, reinterpret_cast<f4u *>(yq));
            yq += OW;
        }
        const bool wrap = (h + 1 == H);
        h = wrap ? 0 : h + 1;
        hmin = wrap ? KH - 1 : hmin;
    }
}

// stencil forward.
template <int KH, int KW>
__global__ __launch_bounds__(BLOCK) void conv_fwd_c1_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
    float *__restrict__ y, int H, int W, int OH, int OW, int strips, int segs,
    unsigned rows_base, unsigned rows_rem, int relu)
{
    constexpr int EXT = 4 + KW - 1;
    RowWalk wk;
    if (!make_walk(wk, H, W, OH, OW, KH, strips, segs, rows_base, rows_rem)) return;

    float wr[KH][KW];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) wr[kh][kw] = w[kh * KW + kw];
    const float bias = b ? b[0] : 0.0f;

    const float *__restrict__ xq = x + wk.in_off;
    float *__restrict__ yq = y + wk.out_off;
    int h = wk.h, hmin = wk.h + KH - 1;

    float win[KH][EXT];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int e = 0; e < EXT; ++e) win[kh][e] = 0.0f;

    int s = 0;
    for (; s + RB <= wk.steps; s += RB)
        fwd_batch<KH, KW, RB>(xq, yq, win, wr, bias, relu, h, hmin, H, W, OW);
    for (; s < wk.steps; ++s)
        fwd_batch<KH, KW, 1>(xq, yq, win, wr, bias, relu, h, hmin, H, W, OW);
}

// ------------------------------------------------------------------------------------------
// Forward, flat-aligned variant (W % 4 == 0, rows up to ~4000 wide): what BASELINE config 3 runs.
//
// Measured on the 64 x 721 x 1440 grid: the strip kernel above runs (3,1) and (1,1) -- where
// OW == W and every wave stores one 128-byte-aligned KiB -- in 106-109 us, but (3,3), (1,3),
// (5,5), (2,2) in 125-133 us, whatever the load scheme, the output pitch or the tail handling;
// with a padded output pitch AND 256-column wave spacing (3,3) drops to 112 us.  The cost is the
// wave-level store that straddles 128-byte lines: rows of OW = W-KW+1 floats start at every
// 8-byte phase.  So here the output leaves through LDS: one workgroup owns one contiguous range
// of output rows = one contiguous range of the FLAT output; lanes drop their results into an LDS
// ring at their flat position, and after every batch of rows the workgroup flushes the finished
// part of the ring with 16-byte stores whose wave-level KiB chunks are aligned in flat space.
// Loads: a lane loads only its own four columns (one aligned dwordx4 per row) and takes the KW-1
// halo columns from lane+1 by shuffle; the last lane of a wave is a halo-only provider.
template <int KW>
struct WaveCols {
    static constexpr int HALO = (KW - 1 + 3) / 4;          // halo lanes per wave (0 or 1)
    static constexpr int CW = 4 * (64 - HALO);             // output columns per wave
};

struct FlatOut {
    float *yb;           // y + (flat start of the range rounded down to a multiple of 4)
    int prow;            // flat offset (from yb) of the next output row
    int head, tail;      // offsets below head / from tail on are stored directly (ragged ends)
    int flushed;         // everything in [head, flushed) has left the ring
    int phase;           // (yb - y) mod 256: makes the flush chunks KiB-aligned in flat space
    int mask;            // ring size - 1 (power of two)
};

// NT: non-temporal loads, for rows nobody reads again (a contiguous stream reads 12 % faster with
// the nt bit on this chip, tools/probe/read_probe.hip).  The KH - 1 rows a range shares with its
// successor are loaded normally by the range that meets them first (its tail), so that they stay
// in L2 for the second reader.
template <int KH, int KW, int NB, bool NT = false>
__device__ __forceinline__ void load_rows_f(f32x4 (&rows)[NB], const float *__restrict__ p, int W)
{
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const f32x4 *q = reinterpret_cast<const f32x4 *>(p + (size_t)u * W);
        rows[u] = NT ? __builtin_nontemporal_load(q) : *q;
    }
}

// consume NB loaded rows: window shift, halo shuffle, outputs into the ring, then flush
template <int KH, int KW, int NB>
__device__ __forceinline__ void fwd_batch_f(const f32x4 (&rows)[NB], float *ring, FlatOut &fo,
                                            float (&win)[KH][4 + KW - 1],
                                            const float (&wr)[KH][KW], float bias, int relu,
                                            int j0, int nst, int &h, int &hmin, int H, int OW)
{
    constexpr int EXT = 4 + KW - 1;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
#pragma unroll
        for (int kh = 0; kh < KH - 1; ++kh)
#pragma unroll
            for (int e = 0; e < EXT; ++e) win[kh][e] = win[kh + 1][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) win[KH - 1][e] = rows[u][e];
#pragma unroll
        for (int e = 4; e < EXT; ++e) win[KH - 1][e] = __shfl_down(rows[u][e - 4], 1, 64);
        if (h >= hmin) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float acc = bias;
#pragma unroll
                for (int kh = 0; kh < KH; ++kh)
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw) acc = fmaf(wr[kh][kw], win[kh][q + kw], acc);
                acc = relu ? fmaxf(acc, 0.0f) : acc;
                if (q < nst) {
                    const int f = fo.prow + j0 + q;
                    if (f < fo.head || f >= fo.tail) fo.yb[f] = acc;
                    else ring[f & fo.mask] = acc;
                }
            }
            fo.prow += OW;
        }
        const bool wrap = (h + 1 == H);
        h = wrap ? 0 : h + 1;
        hmin = wrap ? KH - 1 : hmin;
    }
    __syncthreads();
    const int lim = min(fo.prow & ~3, fo.tail);
    if (lim > fo.flushed) {
        const int start = ((fo.flushed + fo.phase) & ~255) - fo.phase;
        for (int f = start + 4 * (int)threadIdx.x; f < lim; f += 4 * (int)blockDim.x)
            if (f >= fo.flushed) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(ring + (f & fo.mask));
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(fo.yb + f));
            }
        fo.flushed = lim;
    }
    // No second barrier: the ring holds TWO batches.  A wave that runs ahead writes batch k+1 into
    // the half that flush k does not read, and cannot reach batch k+2 before the barrier of batch
    // k+1, which every wave passes only after it has finished flush k.
}

// one workgroup (wq_per_row waves) = one contiguous range of output rows
template <int KH, int KW>
__global__ __launch_bounds__(1024) void conv_fwd_c1f_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
    float *__restrict__ y, int H, int W, int OH, int OW, unsigned rows_base, unsigned rows_rem,
    int relu, int ring_mask, int xcd_run)
{
    extern __shared__ __attribute__((aligned(16))) float ring[];
    constexpr int EXT = 4 + KW - 1;
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one L2).  Inside every
    // run of 8*XS workgroups, XCD x takes the XS CONSECUTIVE ranges x*XS .. x*XS+XS-1, so the
    // KH-1 input rows two neighbouring ranges share are read twice through the SAME L2 (the
    // mapping is speed only: any placement gives the same result).
    unsigned seg = blockIdx.x;
    if (xcd_run > 0) {
        const unsigned run = 8u * (unsigned)xcd_run;
        const unsigned base = seg / run * run;
        if (base + run <= gridDim.x) {
            const unsigned within = seg - base;
            seg = base + (within & 7u) * (unsigned)xcd_run + (within >> 3);
        }
    }
    const unsigned r0 = seg * rows_base + min(seg, rows_rem);
    const unsigned cnt = rows_base + (seg < rows_rem ? 1u : 0u);
    const unsigned n0 = r0 / (unsigned)OH, i0 = r0 - n0 * (unsigned)OH;
    const unsigned rl = r0 + cnt - 1;
    const unsigned n1 = rl / (unsigned)OH, i1 = rl - n1 * (unsigned)OH;
    const int steps = (int)((n1 - n0) * (unsigned)H + i1 + KH - i0);
    const int j0 = wq * WaveCols<KW>::CW + 4 * lane;
    const int nst = lane < 64 - WaveCols<KW>::HALO ? max(0, min(4, OW - j0)) : 0;
    const float *__restrict__ xq = x + ((size_t)n0 * H + i0) * W + min(j0, W - 4);

    const long long F0 = (long long)r0 * OW, Fb = F0 & ~3LL;
    FlatOut fo;
    fo.yb = y + Fb;
    fo.prow = (int)(F0 - Fb);
    fo.head = (fo.prow + 3) & ~3;
    fo.tail = (int)((F0 + (long long)cnt * OW - Fb) & ~3LL);
    if (fo.tail < fo.head) fo.tail = fo.head;
    fo.flushed = fo.head;
    fo.phase = (int)(Fb & 255);
    fo.mask = ring_mask;

    float wr[KH][KW];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) wr[kh][kw] = w[kh * KW + kw];
    const float bias = b ? b[0] : 0.0f;
    int h = (int)i0, hmin = (int)i0 + KH - 1;
    float win[KH][EXT];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int e = 0; e < EXT; ++e) win[kh][e] = 0.0f;

    // The last KH-1 input rows of this range are the first rows of the next range, which its
    // workgroup (same XCD, dispatched a moment later) loads at the START of its life.  They are
    // loaded here first as well and kept in registers until the end: both reads then fall within
    // microseconds of each other and one of them is an L2 hit instead of a second trip through the
    // fabric 10 us later (PMC: 1.28x the algorithmic read traffic before).
    constexpr int TR = KH - 1;
    f32x4 tail[TR > 0 ? TR : 1];
    if constexpr (TR > 0) {
        load_rows_f<KH, KW, TR, WM_NT_FWD_TAIL>(tail, xq + (size_t)(steps - TR) * W, W);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int main_steps = steps - TR;
    const int nfull = main_steps / FRB;
    // One batch in flight per wave, no software prefetch: the other workgroups of the CU cover the
    // latency, and FEWER bytes in flight measured faster (a plain grid-stride copy of the same
    // buffers runs 88.6 us with 3 MB in flight and 100-109 us with 8 MB; this kernel 96 -> 90 us
    // on a cool chip when the double-buffered batch went away).
    for (int bi = 0; bi < nfull; ++bi) {
        f32x4 cur[FRB];
        load_rows_f<KH, KW, FRB, WM_NT_FWD>(cur, xq, W);
        xq += (size_t)FRB * W;
        __builtin_amdgcn_sched_barrier(0);
        fwd_batch_f<KH, KW, FRB>(cur, ring, fo, win, wr, bias, relu, j0, nst, h, hmin, H, OW);
    }
    for (int s = nfull * FRB; s < main_steps; ++s) {
        f32x4 one[1];
        load_rows_f<KH, KW, 1, WM_NT_FWD>(one, xq, W);
        xq += W;
        fwd_batch_f<KH, KW, 1>(one, ring, fo, win, wr, bias, relu, j0, nst, h, hmin, H, OW);
    }
    if constexpr (TR > 0)
        fwd_batch_f<KH, KW, TR>(tail, ring, fo, win, wr, bias, relu, j0, nst, h, hmin, H, OW);
}

template <int KH, int KW, int NB>
__device__ __forceinline__ void bwd_batch(const float *__restrict__ &xq,
                                          const float *__restrict__ &gq,
                                          float (&win)[KH][4 + KW - 1], float (&acc)[KH][KW],
                                          float &accb, int qmin, int &h, int &hmin, int H, int W,
                                          int OW)
{
    constexpr int EXT = 4 + KW - 1;
    float rows[NB][EXT];
    f4u gv[NB];
    bool prod[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        load_row<KW>(rows[u], xq + (size_t)u * W);
        // the gy row of a non-producing step (window refill at an image start) is loaded and
        // ignored: gq then still points at the next row this thread consumes, always in bounds
        gv[u] = *reinterpret_cast<const f4u *>(gq);
        prod[u] = h >= hmin;
        gq += prod[u] ? OW : 0;
        const bool wrap = (h + 1 == H);
        h = wrap ? 0 : h + 1;
        hmin = wrap ? KH - 1 : hmin;
    }
    xq += (size_t)NB * W;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
#pragma unroll
        for (int kh = 0; kh < KH - 1; ++kh)
#pragma unroll
            for (int e = 0; e < EXT; ++e) win[kh][e] = win[kh + 1][e];
#pragma unroll
        for (int e = 0; e < EXT; ++e) win[KH - 1][e] = rows[u][e];
        float g[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = (prod[u] && q >= qmin) ? g[q] : 0.0f;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int kw = 0; kw < KW; ++kw)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[kh][kw] = fmaf(g[q], win[kh][q + kw], acc[kh][kw]);
        accb += (g[0] + g[1]) + (g[2] + g[3]);
    }
}

// ------------------------------------------------------------------------------------------
// stencil backward weight+bias.  partial: [gridDim.x][KH*KW+1] float64.
template <int KH, int KW>
__global__ __launch_bounds__(BLOCK) void conv_bwd_wb_c1_kernel(
    const float *__restrict__ x, const float *__restrict__ gy, double *__restrict__ partial,
    int H, int W, int OH, int OW, int strips, int segs, unsigned rows_base, unsigned rows_rem)
{
    constexpr int EXT = 4 + KW - 1;
    constexpr int NV = KH * KW + 1;
    __shared__ float red[BLOCK / 64][NV];

    float acc[KH][KW];
    float accb = 0.0f;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) acc[kh][kw] = 0.0f;

    RowWalk wk;
    if (make_walk(wk, H, W, OH, OW, KH, strips, segs, rows_base, rows_rem)) {
        const float *__restrict__ xq = x + wk.in_off;
        const float *__restrict__ gq = gy + wk.out_off;
        int h = wk.h, hmin = wk.h + KH - 1;
        float win[KH][EXT];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int e = 0; e < EXT; ++e) win[kh][e] = 0.0f;
        int s = 0;
        for (; s + RB <= wk.steps; s += RB)
            bwd_batch<KH, KW, RB>(xq, gq, win, acc, accb, wk.qmin, h, hmin, H, W, OW);
        for (; s < wk.steps; ++s)
            bwd_batch<KH, KW, 1>(xq, gq, win, acc, accb, wk.qmin, h, hmin, H, W, OW);
    }
    // wavefront-shuffle reduction, then LDS across the block's 4 waves, float64 partial out.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            const float s = wave_sum(acc[kh][kw]);
            if (lane == 0) red[wv][kh * KW + kw] = s;
        }
    {
        const float s = wave_sum(accb);
        if (lane == 0) red[wv][NV - 1] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int v = 0; v < BLOCK / 64; ++v) s += (double)red[v][threadIdx.x];
        partial[(size_t)blockIdx.x * NV + threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------
// Backward weight+bias, flat variant: the same decomposition as conv_fwd_c1f_kernel (a workgroup
// spans the row, lanes load only their own four x columns and shuffle the halo in, short row
// ranges taken in order), but persistent: a workgroup keeps its KH*KW+1 sums in registers over all
// the ranges it takes (range b, b + grid, ...), so there is one float64 partial per workgroup.
template <int KH, int KW, int NB>
__device__ __forceinline__ void bwd_batch_f(const float *__restrict__ &xq,
                                            const float *__restrict__ &gq,
                                            float (&win)[KH][4 + KW - 1], float (&acc)[KH][KW],
                                            float &accb, int nst, int gshift, int &h, int &hmin,
                                            int H, int W, int OW)
{
    constexpr int EXT = 4 + KW - 1;
    f32x4 rows[NB];
    f4u gv[NB];
    bool prod[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const f32x4 *xp_ = reinterpret_cast<const f32x4 *>(xq + (size_t)u * W);
        const f4u *gp_ = reinterpret_cast<const f4u *>(gq);
        rows[u] = WM_NT_BWDW ? __builtin_nontemporal_load(xp_) : *xp_;
        gv[u] = WM_NT_BWDW_G ? __builtin_nontemporal_load(gp_) : *gp_;   // ignored on a non-producing step
        prod[u] = h >= hmin;
        gq += prod[u] ? OW : 0;
        const bool wrap = (h + 1 == H);
        h = wrap ? 0 : h + 1;
        hmin = wrap ? KH - 1 : hmin;
    }
    xq += (size_t)NB * W;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NB; ++u) {
#pragma unroll
        for (int kh = 0; kh < KH - 1; ++kh)
#pragma unroll
            for (int e = 0; e < EXT; ++e) win[kh][e] = win[kh + 1][e];
#pragma unroll
        for (int e = 0; e < 4; ++e) win[KH - 1][e] = rows[u][e];
#pragma unroll
        for (int e = 4; e < EXT; ++e) win[KH - 1][e] = __shfl_down(rows[u][e - 4], 1, 64);
        // the last strip of a row loads its gy four columns early (in bounds) and shifts
        const float l0 = gv[u].x, l1 = gv[u].y, l2 = gv[u].z, l3 = gv[u].w;
        float g[4];
        g[0] = gshift == 0 ? l0 : (gshift == 1 ? l1 : (gshift == 2 ? l2 : l3));
        g[1] = gshift == 0 ? l1 : (gshift == 1 ? l2 : l3);
        g[2] = gshift == 0 ? l2 : l3;
        g[3] = l3;
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = (prod[u] && q < nst) ? g[q] : 0.0f;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int kw = 0; kw < KW; ++kw)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[kh][kw] = fmaf(g[q], win[kh][q + kw], acc[kh][kw]);
        accb += (g[0] + g[1]) + (g[2] + g[3]);
    }
}

template <int KH, int KW>
__global__ __launch_bounds__(1024) void conv_bwd_wb_c1f_kernel(
    const float *__restrict__ x, const float *__restrict__ gy, double *__restrict__ partial,
    int H, int W, int OH, int OW, unsigned rows_base, unsigned rows_rem, unsigned segs)
{
    constexpr int EXT = 4 + KW - 1;
    constexpr int NV = KH * KW + 1;
    __shared__ float red[16][NV];
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
    const int j0 = wq * WaveCols<KW>::CW + 4 * lane;
    const int nst = lane < 64 - WaveCols<KW>::HALO ? max(0, min(4, OW - j0)) : 0;
    const int jx = min(j0, W - 4), jg = min(j0, OW - 4);
    const int gshift = min(j0 - jg, 3);

    float acc[KH][KW];
    float accb = 0.0f;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) acc[kh][kw] = 0.0f;

    for (unsigned seg = blockIdx.x; seg < segs; seg += gridDim.x) {
        const unsigned r0 = seg * rows_base + min(seg, rows_rem);
        const unsigned cnt = rows_base + (seg < rows_rem ? 1u : 0u);
        const unsigned n0 = r0 / (unsigned)OH, i0 = r0 - n0 * (unsigned)OH;
        const unsigned rl = r0 + cnt - 1;
        const unsigned n1 = rl / (unsigned)OH, i1 = rl - n1 * (unsigned)OH;
        const int steps = (int)((n1 - n0) * (unsigned)H + i1 + KH - i0);
        const float *__restrict__ xq = x + ((size_t)n0 * H + i0) * W + jx;
        const float *__restrict__ gq = gy + (size_t)r0 * OW + jg;
        int h = (int)i0, hmin = (int)i0 + KH - 1;
        float win[KH][EXT];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int e = 0; e < EXT; ++e) win[kh][e] = 0.0f;
        int s = 0;
        for (; s + RB <= steps; s += RB)
            bwd_batch_f<KH, KW, RB>(xq, gq, win, acc, accb, nst, gshift, h, hmin, H, W, OW);
        for (; s < steps; ++s)
            bwd_batch_f<KH, KW, 1>(xq, gq, win, acc, accb, nst, gshift, h, hmin, H, W, OW);
    }
    const int nwaves = blockDim.x >> 6;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            const float s = wave_sum(acc[kh][kw]);
            if (lane == 0) red[wq][kh * KW + kw] = s;
        }
    {
        const float s = wave_sum(accb);
        if (lane == 0) red[wq][NV - 1] = s;
    }
    __syncthreads();
    // entry-major partials [NV][gridDim.x]: the finalize kernel then reads them coalesced
    if (threadIdx.x < NV) {
        double s = 0.0;
        for (int v = 0; v < nwaves; ++v) s += (double)red[v][threadIdx.x];
        partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------
// Backward-data, flat stencil variant (Cin = Cout = 1, W % 4 == 0): gx = full correlation of gy with
// the flipped kernel,  gx[n,p,q] = sum_{kh,kw} w[kh,kw] gy[n,p-kh,q-kw]  (gy = 0 outside).
// Same decomposition as conv_fwd_c1f_kernel -- a workgroup spans the row and owns one short range
// of the N*H output rows, ranges handed out in order with XCD-local runs, the KH-1 gy rows a range
// shares with its successor loaded first and kept in registers -- but the roles of the two tensors
// are swapped: the OUTPUT rows (W floats) are 16-byte aligned, so a lane stores its four columns
// directly (a wave stores one aligned KiB, no LDS ring, no barrier), and the INPUT rows (OW floats)
// start at any dword, which loads do not mind.  A lane loads its own four gy columns (16 B) plus
// the two columns to their left (8 B) per row.
// Rows are walked in a virtual padded space: every image has H rows, of which rows >= OH are the
// zero padding below it -- which is also the zero padding above the next image, so the register
// window needs no reset at an image boundary.
struct GyRow {
    f4u own;
    f2u halo;
    bool valid;
};

// virtual row v -> row in image p (negative above image 0) and flat gy row gr of the next valid row
__device__ __forceinline__ void virtual_row(long long v, int H, int OH, int &p, unsigned &gr)
{
    if (v >= 0) {
        const unsigned n = (unsigned)v / (unsigned)H;
        p = (int)((unsigned)v - n * (unsigned)H);
        gr = n * (unsigned)OH + (unsigned)min(p, OH);
    } else {
        p = (int)v;
        gr = 0;
    }
}

template <int NB, bool NT = false>
__device__ __forceinline__ void load_gy_rows(GyRow (&rows)[NB], const float *__restrict__ gy,
                                             int OW, int OH, int H, unsigned last_row, int jg,
                                             int jh, int &p, unsigned &gr)
{
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const bool valid = p >= 0 && p < OH;
        const size_t row = (size_t)min(gr, last_row) * OW;     // always in bounds; zeroed if !valid
        const f4u *own_ = reinterpret_cast<const f4u *>(gy + row + jg);
        rows[u].own = NT ? __builtin_nontemporal_load(own_) : *own_;      // see load_rows_f
        rows[u].halo = *reinterpret_cast<const f2u *>(gy + row + jh);
        rows[u].valid = valid;
        gr += valid ? 1u : 0u;
        p = (p + 1 == H) ? 0 : p + 1;
    }
}

template <int KH, int KW, int NB>
__device__ __forceinline__ void bwd_data_batch(const GyRow (&rows)[NB], float *__restrict__ &gq,
                                               float (&win)[KH][4 + KW - 1],
                                               const float (&wr)[KH][KW], int shift, bool has_halo,
                                               bool active, int &skip, int W)
{
    constexpr int EXT = 4 + KW - 1;
    constexpr int HO = KW - 1 <= 2 ? 2 - (KW - 1) : 0;     // first halo column used of the two loaded
#pragma unroll
    for (int u = 0; u < NB; ++u) {
#pragma unroll
        for (int kh = 0; kh < KH - 1; ++kh)
#pragma unroll
            for (int e = 0; e < EXT; ++e) win[kh][e] = win[kh + 1][e];
        const float l0 = rows[u].own.x, l1 = rows[u].own.y, l2 = rows[u].own.z, l3 = rows[u].own.w;
        float g[4];
        g[0] = shift == 0 ? l0 : (shift == 1 ? l1 : (shift == 2 ? l2 : l3));
        g[1] = shift == 0 ? l1 : (shift == 1 ? l2 : (shift == 2 ? l3 : 0.0f));
        g[2] = shift == 0 ? l2 : (shift == 1 ? l3 : 0.0f);
        g[3] = shift == 0 ? l3 : 0.0f;
        const bool v = rows[u].valid;
        const float hl[2] = {rows[u].halo.x, rows[u].halo.y};
#pragma unroll
        for (int e = 0; e < KW - 1; ++e)
            win[KH - 1][e] = (v && has_halo) ? hl[(HO + e) & 1] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) win[KH - 1][KW - 1 + e] = v ? g[e] : 0.0f;
        if (skip > 0) {
            --skip;                     // window refill: the first KH-1 rows of a range
        } else {
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < KH; ++i)
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw)
                        acc = fmaf(wr[KH - 1 - i][kw], win[i][e + (KW - 1) - kw], acc);
                out[e] = acc;
            }
            if (active) __builtin_nontemporal_store(out, reinterpret_cast<f32x4 *>(gq));
            gq += W;
        }
    }
}

template <int KH, int KW>
__global__ __launch_bounds__(1024) void conv_bwd_data_c1f_kernel(
    const float *__restrict__ gy, const float *__restrict__ w, float *__restrict__ gx, int H,
    int W, int OH, int OW, unsigned last_row, unsigned rows_base, unsigned rows_rem, int xcd_run)
{
    constexpr int EXT = 4 + KW - 1;
    constexpr int TR = KH - 1;
    unsigned seg = blockIdx.x;
    if (xcd_run > 0) {                              // XCD-local runs: see conv_fwd_c1f_kernel
        const unsigned run = 8u * (unsigned)xcd_run;
        const unsigned base = seg / run * run;
        if (base + run <= gridDim.x) {
            const unsigned within = seg - base;
            seg = base + (within & 7u) * (unsigned)xcd_run + (within >> 3);
        }
    }
    const unsigned r0 = seg * rows_base + min(seg, rows_rem);      // first of the N*H output rows
    const unsigned cnt = rows_base + (seg < rows_rem ? 1u : 0u);
    const int j0 = 4 * (int)threadIdx.x;
    const bool active = j0 < W;
    const int jc = min(j0, W - 4);
    const int jg = min(jc, OW - 4), shift = jc - jg;
    const bool has_halo = jc >= 2;
    const int jh = max(jc - 2, 0);

    float wr[KH][KW];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) wr[kh][kw] = w[kh * KW + kw];

    // virtual row (H rows per image) of the first gy row this range needs; negative above image 0
    const long long v0 = (long long)r0 - TR;
    int p;
    unsigned gr;
    virtual_row(v0, H, OH, p, gr);
    const int steps = (int)cnt + TR;
    float *__restrict__ gq = gx + (size_t)r0 * W + jc;
    float win[KH][EXT];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int e = 0; e < EXT; ++e) win[kh][e] = 0.0f;
    int skip = TR;

    // the last KH-1 gy rows first (they are the first rows of the next range: same XCD, same time)
    GyRow tail[TR > 0 ? TR : 1];
    if constexpr (TR > 0) {
        int pt;
        unsigned grt;
        virtual_row(v0 + steps - TR, H, OH, pt, grt);
        load_gy_rows<TR>(tail, gy, OW, OH, H, last_row, jg, jh, pt, grt);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int main_steps = steps - TR;
    const int nfull = main_steps / FRB;
    for (int bi = 0; bi < nfull; ++bi) {
        GyRow cur[FRB];
        load_gy_rows<FRB, WM_NT_BWDD>(cur, gy, OW, OH, H, last_row, jg, jh, p, gr);
        __builtin_amdgcn_sched_barrier(0);
        bwd_data_batch<KH, KW, FRB>(cur, gq, win, wr, shift, has_halo, active, skip, W);
    }
    for (int s = nfull * FRB; s < main_steps; ++s) {
        GyRow one[1];
        load_gy_rows<1, WM_NT_BWDD>(one, gy, OW, OH, H, last_row, jg, jh, p, gr);
        bwd_data_batch<KH, KW, 1>(one, gq, win, wr, shift, has_halo, active, skip, W);
    }
    if constexpr (TR > 0)
        bwd_data_batch<KH, KW, TR>(tail, gq, win, wr, shift, has_halo, active, skip, W);
}

// ------------------------------------------------------------------------------------------
// fixed-order float64 reduction of per-block partials: one wave per output entry.
// partial is [nparts][nent]; out_w gets entries [0, nw), out_b gets [nw, nent).
__device__ __forceinline__ double sum_part_major(const double *__restrict__ partial, int nparts,
                                                 int nent, int e)
{
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) s += partial[(size_t)p * nent + e];
    return wave_sum(s);
}

__global__ __launch_bounds__(64) void reduce_partials_kernel(const double *__restrict__ partial,
                                                            int nparts, int nent, int nw,
                                                            float *__restrict__ out_w,
                                                            float *__restrict__ out_b)
{
    const int e = blockIdx.x;
    const double s = sum_part_major(partial, nparts, nent, e);
    if (threadIdx.x == 0) {
        if (e < nw) out_w[e] = (float)s;
        else if (out_b) out_b[e - nw] = (float)s;
    }
}

// The same for entry-major partials [nent][nparts] (the flat stencil kernel): one 256-thread
// workgroup per entry, coalesced 8-byte loads, fixed order (thread t adds partials t, t+256, ...;
// butterfly across the wave; the four wave sums in wave order).  The total is valid in thread 0.
__device__ __forceinline__ double sum_entry_major(const double *__restrict__ partial, int nparts,
                                                  int e, double (&wsum)[4])
{
    const double *__restrict__ src = partial + (size_t)e * nparts;
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 256) s += src[p];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    return ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(
    const double *__restrict__ partial, int nparts, int nw, float *__restrict__ out_w,
    float *__restrict__ out_b)
{
    __shared__ double wsum[4];
    const int e = blockIdx.x;
    const double t = sum_entry_major(partial, nparts, e, wsum);
    if (threadIdx.x == 0) {
        if (e < nw) out_w[e] = (float)t;
        else if (out_b) out_b[e - nw] = (float)t;
    }
}

// Deferred finalize (wm_adam_step_partials_f32): the consumer of the gradient does the fixed-order
// sum itself -- one workgroup per parameter sums that entry's partials exactly as the two kernels
// above do, rounds to float32 once, and applies the Adam update.  One launch instead of
// finalize + Adam, bit-identical parameters.
template <bool ENTRY_MAJOR>
__global__ __launch_bounds__(ENTRY_MAJOR ? 256 : 64) void adam_partials_kernel(
    float *__restrict__ prm, const double *__restrict__ partial, int nparts, int nent,
    float *__restrict__ m, float *__restrict__ v, float *__restrict__ grads_out, AdamScalars a)
{
    __shared__ double wsum[4];
    const int e = blockIdx.x;
    const double t = ENTRY_MAJOR ? sum_entry_major(partial, nparts, e, wsum)
                                 : sum_part_major(partial, nparts, nent, e);
    if (threadIdx.x == 0) {
        const float g = (float)t;
        if (grads_out) grads_out[e] = g;
        float pv = prm[e], mv = m[e], vv = v[e];
        adam_update(pv, mv, vv, g, a);
        m[e] = mv;
        v[e] = vv;
        prm[e] = pv;
    }
}

// ------------------------------------------------------------------------------------------
// generic direct forward: one thread = 4 consecutive outputs of one (n, o, i) row.
__global__ __launch_bounds__(BLOCK) void conv_fwd_generic_kernel(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
    float *__restrict__ y, int N, int Cin, int H, int W, int Cout, int kH, int kW, int OH, int OW,
    int strips, int relu)
{
    const long long total = (long long)N * Cout * OH * strips;
    for (long long tid = (long long)blockIdx.x * BLOCK + threadIdx.x; tid < total;
         tid += (long long)gridDim.x * BLOCK) {
        const int strip = (int)(tid % strips);
        long long t = tid / strips;
        const int i = (int)(t % OH); t /= OH;
        const int o = (int)(t % Cout);
        const int n = (int)(t / Cout);
        const int j0 = strip * 4;
        const float bias = b ? b[o] : 0.0f;
        float s0 = bias, s1 = bias, s2 = bias, s3 = bias;
        for (int c = 0; c < Cin; ++c)
            for (int kh = 0; kh < kH; ++kh) {
                const float *xr = x + (((size_t)n * Cin + c) * H + i + kh) * W + j0;
                const float *wr = w + (((size_t)o * Cin + c) * kH + kh) * kW;
                float x0 = xr[0];
                float x1 = (j0 + 1 < W) ? xr[1] : 0.0f;
                float x2 = (j0 + 2 < W) ? xr[2] : 0.0f;
                float x3 = (j0 + 3 < W) ? xr[3] : 0.0f;
                for (int kw = 0; kw < kW; ++kw) {
                    const float wv = wr[kw];
                    s0 = fmaf(wv, x0, s0);
                    s1 = fmaf(wv, x1, s1);
                    s2 = fmaf(wv, x2, s2);
                    s3 = fmaf(wv, x3, s3);
                    x0 = x1; x1 = x2; x2 = x3;
                    x3 = (j0 + 4 + kw < W) ? xr[4 + kw] : 0.0f;
                }
            }
        float *yr = y + (((size_t)n * Cout + o) * OH + i) * OW + j0;
        if (relu) { s0 = fmaxf(s0, 0.f); s1 = fmaxf(s1, 0.f); s2 = fmaxf(s2, 0.f); s3 = fmaxf(s3, 0.f); }
        yr[0] = s0;
        if (j0 + 1 < OW) yr[1] = s1;
        if (j0 + 2 < OW) yr[2] = s2;
        if (j0 + 3 < OW) yr[3] = s3;
    }
}

// generic direct backward-data: one thread = one gx element.
__global__ __launch_bounds__(BLOCK) void conv_bwd_data_generic_kernel(
    const float *__restrict__ gy, const float *__restrict__ w, float *__restrict__ gx, int N,
    int Cin, int H, int W, int Cout, int kH, int kW, int OH, int OW)
{
    const long long total = (long long)N * Cin * H * W;
    for (long long tid = (long long)blockIdx.x * BLOCK + threadIdx.x; tid < total;
         tid += (long long)gridDim.x * BLOCK) {
        const int q = (int)(tid % W);
        long long t = tid / W;
        const int p = (int)(t % H); t /= H;
        const int c = (int)(t % Cin);
        const int n = (int)(t / Cin);
        const int kh_lo = max(0, p - OH + 1), kh_hi = min(kH - 1, p);
        const int kw_lo = max(0, q - OW + 1), kw_hi = min(kW - 1, q);
        float s = 0.0f;
        for (int o = 0; o < Cout; ++o)
            for (int kh = kh_lo; kh <= kh_hi; ++kh) {
                const float *gr = gy + (((size_t)n * Cout + o) * OH + (p - kh)) * OW + q;
                const float *wr = w + (((size_t)o * Cin + c) * kH + kh) * kW;
                for (int kw = kw_lo; kw <= kw_hi; ++kw) s = fmaf(wr[kw], gr[-kw], s);
            }
        gx[tid] = s;
    }
}

// generic backward weight+bias: grid = (entries, splits); entry e < nw is a weight, else a bias.
// Each block reduces its slice of the N*OH*OW positions; partial is [splits][nent] float64.
__global__ __launch_bounds__(BLOCK) void conv_bwd_wb_generic_kernel(
    const float *__restrict__ x, const float *__restrict__ gy, double *__restrict__ partial, int N,
    int Cin, int H, int W, int Cout, int kH, int kW, int OH, int OW, int nw, int nent)
{
    __shared__ double red[BLOCK / 64];
    const int e = blockIdx.x;
    const int split = blockIdx.y, splits = gridDim.y;
    const long long L = (long long)N * OH * OW;
    const long long l0 = L * split / splits, l1 = L * (split + 1) / splits;
    float s = 0.0f;
    double sd = 0.0;
    int cnt = 0;
    if (e < nw) {
        const int kw = e % kW;
        const int kh = (e / kW) % kH;
        const int c = (e / (kW * kH)) % Cin;
        const int o = e / (kW * kH * Cin);
        for (long long l = l0 + threadIdx.x; l < l1; l += BLOCK) {
            const int j = (int)(l % OW);
            const long long t = l / OW;
            const int i = (int)(t % OH);
            const int n = (int)(t / OH);
            s = fmaf(gy[(((size_t)n * Cout + o) * OH + i) * OW + j],
                     x[(((size_t)n * Cin + c) * H + i + kh) * W + j + kw], s);
            if (++cnt == 256) { sd += (double)s; s = 0.0f; cnt = 0; }
        }
    } else {
        const int o = e - nw;
        for (long long l = l0 + threadIdx.x; l < l1; l += BLOCK) {
            const int j = (int)(l % OW);
            const long long t = l / OW;
            const int i = (int)(t % OH);
            const int n = (int)(t / OH);
            s += gy[(((size_t)n * Cout + o) * OH + i) * OW + j];
            if (++cnt == 256) { sd += (double)s; s = 0.0f; cnt = 0; }
        }
    }
    sd += (double)s;
    sd = wave_sum(sd);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sd;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tsum = 0.0;
#pragma unroll
        for (int v = 0; v < BLOCK / 64; ++v) tsum += red[v];
        partial[(size_t)split * nent + e] = tsum;
    }
}

// hipOccupancyMaxActiveBlocksPerMultiprocessor is a host-side query of a few microseconds and is
// not legal under stream capture: ask once per (device, kernel, block size) and keep the answer.
// (a map under a mutex: a fixed table that fills up would redo the query on every call)
template <typename K>
int cached_occupancy(K kernel, int block, int *out)
{
    static std::mutex mu;
    static std::map<std::tuple<int, const void *, int>, int> cache;
    int dev = 0;
    WM_HIP(hipGetDevice(&dev));
    const void *fn = reinterpret_cast<const void *>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(dev, fn, block);
    const auto hit = cache.find(key);
    if (hit != cache.end()) { *out = hit->second; return WM_OK; }
    int occ = 0;
    WM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block, 0));
    cache[key] = occ;
    *out = occ;
    return WM_OK;
}

// ------------------------------------------------------------------------------------------
struct StencilPlan {
    int strips, segs, blocks;
    unsigned rows_base, rows_rem;
};

// Cut the N*OH output rows of every 4-column strip into `segs` contiguous ranges so that
// strips*segs threads fill the chip once: blocks_per_cu resident blocks on every CU.
template <typename K>
int stencil_plan(K kernel, int N, int OH, int OW, StencilPlan *p)
{
    const int cus = wm_num_cus();
    WM_REQUIRE(cus > 0, WM_ERR_NO_DEVICE, "no HIP device");
    int occ = 0;
    if (int rc = cached_occupancy(kernel, BLOCK, &occ)) return rc;
    // The occupancy query over-reports only at the 8-blocks-per-CU end (SGPR-heavy kernels,
    // MI355X_MICROARCH residency note); a launch that is not fully resident would double the
    // kernel time, so never plan for more than 6 blocks per CU.
    int bpc = occ;
    if (bpc > WM_STENCIL_BPC) bpc = WM_STENCIL_BPC;
    if (bpc < 1) bpc = 1;
    p->strips = (OW + 3) / 4;
    const long long total_rows = (long long)N * OH;
    long long segs = (long long)cus * bpc * BLOCK / p->strips;
    const long long max_segs = (total_rows + 7) / 8;   // at least ~8 rows per range
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    p->segs = (int)segs;
    p->rows_base = (unsigned)(total_rows / segs);
    p->rows_rem = (unsigned)(total_rows % segs);
    p->blocks = (int)(((long long)p->strips * segs + BLOCK - 1) / BLOCK);
    return WM_OK;
}

// Launch shape of the flat-aligned forward: one workgroup of `waves` waves per row range, `segs`
// ranges filling the chip once, an LDS ring of `ring` floats.  false: shape outside its limits.
struct FlatPlan {
    int waves, segs, ring;
    unsigned rows_base, rows_rem;
};

// Row ranges are SHORT (3*RB - (kH-1) output rows, i.e. three full load batches) and one workgroup
// takes one range: the dispatcher hands ranges out in order, so at any moment the resident
// workgroups work on one compact window of x and of y that slides through memory.  Measured on
// 64 x 721 x 1440, (3,3): ranges that fill the chip once (36 rows each, every workgroup streaming
// its own far-apart 200 KB) 114 us; 6-row ranges in order 100 us; pure copy (1,1) 110 -> 90 us.
// Swept later: two / three / four / six batches per range 4.83 / 5.01 / 4.89 / 4.43 TB/s forward
// (the halo rows are 33 / 20 / 14 / 9 % of the reads; longer ranges lose the compact window).
// The halo rows two neighbouring ranges both read come out of L2 / Infinity Cache.
template <typename K>
bool flat_plan(K kernel, int N, int OH, int OW, int W, int kH, const void *x, const void *y,
               int cols_per_wave, FlatPlan *p, long long target_segs = 0)
{
    if (W % 4 != 0 || (((uintptr_t)x | (uintptr_t)y) & 15) != 0) return false;
    p->waves = (OW + cols_per_wave - 1) / cols_per_wave;
    if (p->waves > 16) return false;
    int ring = 1024;
    const int ring_rows = FRB > kH - 1 ? FRB : kH - 1;     // the largest batch: FRB rows or the tail
    while (ring < 2 * ring_rows * OW + 8) ring *= 2;      // two batches: see fwd_batch_f
    if (ring > 16384) return false;                                  // 64 KB of LDS
    p->ring = ring;
    (void)kernel;
    const long long total_rows = (long long)N * OH;
    // forward: WM_FWD_BATCHES full load batches per range; the KH-1 rows the range shares with
    // its successor are loaded up front on top of them (kept in registers)
    int rows = WM_KNOB(WM_FWD_BATCHES) * FRB;
    if (target_segs > 0) rows = 3 * RB - (kH - 1);
    long long segs = (total_rows + rows - 1) / rows;
    // target_segs > 0 (backward, reads only: long ranges cost nothing and halve the halo): one
    // range per workgroup, as even as possible
    if (target_segs > 0 && target_segs < segs) segs = target_segs;
    if (segs < 1) segs = 1;
    if (segs > 0x7fffffffLL) return false;
    p->segs = (int)segs;
    p->rows_base = (unsigned)(total_rows / segs);
    p->rows_rem = (unsigned)(total_rows % segs);
    // 32-bit flat offsets inside a range
    return ((long long)p->rows_base + 2) * OW < (1ll << 30);
}

bool stencil_shape(int Cin, int Cout, int kH, int kW, int OW, long long total_rows)
{
    // 32-bit row arithmetic inside the kernels: N*H rows and strips*segs threads below 2^31
    if (Cin != 1 || Cout != 1 || OW < 16 || total_rows >= (1ll << 30)) return false;
    return (kH == 3 && kW == 3) || (kH == 5 && kW == 5) || (kH == 3 && kW == 1) ||
           (kH == 1 && kW == 3) || (kH == 2 && kW == 1) || (kH == 1 && kW == 2) ||
           (kH == 2 && kW == 2) || (kH == 1 && kW == 1);
}

#define WM_STENCIL_DISPATCH(MACRO)                  \
    if (kH == 3 && kW == 3) { MACRO(3, 3) }         \
    else if (kH == 5 && kW == 5) { MACRO(5, 5) }    \
    else if (kH == 3 && kW == 1) { MACRO(3, 1) }    \
    else if (kH == 1 && kW == 3) { MACRO(1, 3) }    \
    else if (kH == 2 && kW == 1) { MACRO(2, 1) }    \
    else if (kH == 1 && kW == 2) { MACRO(1, 2) }    \
    else if (kH == 2 && kW == 2) { MACRO(2, 2) }    \
    else { MACRO(1, 1) }

int check_dims(int N, int Cin, int H, int W, int Cout, int kH, int kW)
{
    WM_REQUIRE(N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && kH > 0 && kW > 0, WM_ERR_INVALID,
               "conv2d: non-positive dimension (N=%d Cin=%d H=%d W=%d Cout=%d kH=%d kW=%d)", N,
               Cin, H, W, Cout, kH, kW);
    WM_REQUIRE(kH <= H && kW <= W, WM_ERR_INVALID,
               "conv2d: kernel (%d,%d) larger than input (%d,%d) -- valid padding has no output",
               kH, kW, H, W);
    return WM_OK;
}

// Multi-channel shapes are dense contractions: they go to the implicit-GEMM MFMA kernels of
// conv3d.hip with a depth of one (measured on (2,32,721,1440) x (32,32,3,3): forward 18 TFLOP/s
// against 3.9 for the direct kernel below, backward-data 29 against 3.2, backward-weights 10
// against 0.9).  Few-channel shapes stay on the direct kernels (an MFMA tile would be mostly
// padding).
bool mfma_shape(int N, int Cin, int H, int W, int Cout, int kH, int kW)
{
    if (Cin > 64 || Cout > 64 || kH >= 256 || kW >= 256) return false;
    if (Cout < 8 || Cin * kH * kW < 16) return false;
    const long long xin = (long long)N * Cin * H * W;
    const long long yout = (long long)N * Cout * (H - kH + 1) * (W - kW + 1);
    return xin < (1ll << 31) && yout < (1ll << 31);
}

int generic_grid(long long total)
{
    long long g = (total + BLOCK - 1) / BLOCK;
    const long long cap = (long long)wm_num_cus() * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int bwd_wb_generic_splits(int N, int OH, int OW, int nent)
{
    const long long L = (long long)N * OH * OW;
    long long s = (L + 16383) / 16384;
    long long cap = (long long)wm_num_cus() * 8 / nent;
    if (cap < 1) cap = 1;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    if (s > 65535) s = 65535;
    return (int)s;
}

// ------------------------------------------------------------------------------------------
// Scratch of the allocation-free entry point (wm_conv2d_bwd_wb_noalloc_f32): the float64
// per-workgroup partial sums that wm_conv2d_bwd_wb_f32 keeps in the caller's workspace.  One copy
// per GPU, held until the process ends.  The array's NAME is only a host-side handle; the device
// address comes from hipGetSymbolAddress, once per device, and only that pointer is ever handed
// to a kernel.
constexpr size_t SCRATCH_BYTES = WM_CONV2D_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

int scratch_ptr(void **out)
{
    static std::mutex mu;
    static void *ptrs[64] = {};
    int dev = 0;
    WM_HIP(hipGetDevice(&dev));
    WM_REQUIRE(dev >= 0 && dev < 64, WM_ERR_NO_DEVICE, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        void *p = nullptr;
        WM_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        WM_REQUIRE(p != nullptr, WM_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
        ptrs[dev] = p;
    }
    *out = ptrs[dev];
    return WM_OK;
}

}  // namespace

extern "C" int wm_conv2d_fwd_f32(const float *x, const float *w, const float *b, float *y, int N,
                                 int Cin, int H, int W, int Cout, int kH, int kW, int relu,
                                 void *stream)
{
    WM_REQUIRE(x && w && y, WM_ERR_INVALID, "wm_conv2d_fwd_f32: null pointer");
    if (int rc = check_dims(N, Cin, H, W, Cout, kH, kW)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int OH = H - kH + 1, OW = W - kW + 1;
    if (stencil_shape(Cin, Cout, kH, kW, OW, (long long)N * H)) {
#define FWDF(KH_, KW_)                                                                          \
    FlatPlan fp;                                                                                \
    if (KH_ * KW_ <= 9 &&                                                                       \
        flat_plan(conv_fwd_c1f_kernel<KH_, KW_>, N, OH, OW, W, kH, x, y, WaveCols<KW_>::CW,     \
                  &fp)) {                                                                       \
        size_t lds_ = (size_t)fp.ring * sizeof(float);                                         \
        if (lds_ < (size_t)WM_KNOB(WM_FWD_LDS_MIN)) lds_ = (size_t)WM_KNOB(WM_FWD_LDS_MIN);                     \
        hipLaunchKernelGGL((conv_fwd_c1f_kernel<KH_, KW_>), dim3(fp.segs), dim3(fp.waves * 64), \
                           lds_, st, x, w, b, y, H, W, OH, OW,                                  \
                           fp.rows_base, fp.rows_rem, relu, fp.ring - 1, WM_KNOB(WM_FWD_XCD_RUN)); \
        WM_LAUNCH_CHECK("conv_fwd_c1f_kernel");                                                 \
        return WM_OK;                                                                           \
    }
        WM_STENCIL_DISPATCH(FWDF)
#undef FWDF
    }
    if (stencil_shape(Cin, Cout, kH, kW, OW, (long long)N * H)) {
#define FWD(KH_, KW_)                                                                       \
    StencilPlan p;                                                                          \
    if (int rc = stencil_plan(conv_fwd_c1_kernel<KH_, KW_>, N, OH, OW, &p)) return rc;      \
    hipLaunchKernelGGL((conv_fwd_c1_kernel<KH_, KW_>), dim3(p.blocks), dim3(BLOCK), 0, st, x, \
                       w, b, y, H, W, OH, OW, p.strips, p.segs, p.rows_base, p.rows_rem, relu);
        WM_STENCIL_DISPATCH(FWD)
#undef FWD
        WM_LAUNCH_CHECK("conv_fwd_c1_kernel");
        return WM_OK;
    }
    if (mfma_shape(N, Cin, H, W, Cout, kH, kW))
        return wm_conv3d_fwd_f32(x, w, b, y, N, Cin, 1, H, W, Cout, 1, kH, kW, relu, stream);
    const int strips = (OW + 3) / 4;
    const long long total = (long long)N * Cout * OH * strips;
    hipLaunchKernelGGL(conv_fwd_generic_kernel, dim3(generic_grid(total)), dim3(BLOCK), 0, st, x,
                       w, b, y, N, Cin, H, W, Cout, kH, kW, OH, OW, strips, relu);
    WM_LAUNCH_CHECK("conv_fwd_generic_kernel");
    return WM_OK;
}

extern "C" int wm_conv2d_bwd_data_f32(const float *gy, const float *w, float *gx, int N, int Cin,
                                      int H, int W, int Cout, int kH, int kW, void *stream)
{
    WM_REQUIRE(gy && w && gx, WM_ERR_INVALID, "wm_conv2d_bwd_data_f32: null pointer");
    if (int rc = check_dims(N, Cin, H, W, Cout, kH, kW)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (mfma_shape(N, Cin, H, W, Cout, kH, kW))
        return wm_conv3d_bwd_data_f32(gy, w, gx, N, Cin, 1, H, W, Cout, 1, kH, kW, stream);
    const int OH = H - kH + 1, OW = W - kW + 1;
    // single-channel stencil with 16-byte aligned output rows: the flat kernel
    if (stencil_shape(Cin, Cout, kH, kW, OW, (long long)N * H) && kH * kW <= 9 && kW <= 3 &&
        W % 4 == 0 && OW >= 4 && W <= 4096 && ((uintptr_t)gx & 15) == 0) {
        const long long total_rows = (long long)N * H;
        const int rows = WM_KNOB(WM_BWDD_BATCHES) * FRB;
        const long long segs = (total_rows + rows - 1) / rows;
        if (segs <= 0x7fffffffLL) {
            const unsigned rows_base = (unsigned)(total_rows / segs);
            const unsigned rows_rem = (unsigned)(total_rows % segs);
            const unsigned last_row = (unsigned)((long long)N * OH - 1);
            const int threads = (W / 4 + 63) / 64 * 64;
#define BWDD(KH_, KW_)                                                                          \
    hipLaunchKernelGGL((conv_bwd_data_c1f_kernel<KH_, KW_>), dim3((unsigned)segs),              \
                       dim3(threads), 0, st, gy, w, gx, H, W, OH, OW, last_row, rows_base,      \
                       rows_rem, WM_KNOB(WM_FWD_XCD_RUN));
            WM_STENCIL_DISPATCH(BWDD)
#undef BWDD
            WM_LAUNCH_CHECK("conv_bwd_data_c1f_kernel");
            return WM_OK;
        }
    }
    const long long total = (long long)N * Cin * H * W;
    hipLaunchKernelGGL(conv_bwd_data_generic_kernel, dim3(generic_grid(total)), dim3(BLOCK), 0, st,
                       gy, w, gx, N, Cin, H, W, Cout, kH, kW, OH, OW);
    WM_LAUNCH_CHECK("conv_bwd_data_generic_kernel");
    return WM_OK;
}

// the same with the ReLU backward of the layer below fused in: gx = ymask > 0 ? gx : 0, ymask = that
// layer's (post-ReLU) output, shaped like gx.  Multi-channel shapes fuse it into the row kernel's
// store; the others apply it in a second element-wise launch.
extern "C" int wm_relu_mask_f32(float *gx, const float *ymask, int64_t n, void *stream);

extern "C" int wm_conv2d_bwd_data_relu_f32(const float *gy, const float *w, const float *ymask,
                                           float *gx, int N, int Cin, int H, int W, int Cout, int kH,
                                           int kW, void *stream)
{
    WM_REQUIRE(gy && w && gx && ymask, WM_ERR_INVALID, "wm_conv2d_bwd_data_relu_f32: null pointer");
    if (int rc = check_dims(N, Cin, H, W, Cout, kH, kW)) return rc;
    if (mfma_shape(N, Cin, H, W, Cout, kH, kW))
        return wm_conv3d_bwd_data_relu_f32(gy, w, ymask, gx, N, Cin, 1, H, W, Cout, 1, kH, kW,
                                           stream);
    if (int rc = wm_conv2d_bwd_data_f32(gy, w, gx, N, Cin, H, W, Cout, kH, kW, stream)) return rc;
    return wm_relu_mask_f32(gx, ymask, (int64_t)N * Cin * H * W, stream);
}

// out = ymask > 0 ? g : 0 (nn.ReLU's backward, newConvolution.py:150-159 under :184); out may be g.
// 16 bytes per lane when all three pointers allow it.
__global__ __launch_bounds__(BLOCK) void relu_mask_kernel(const float *g, const float *__restrict__ m,
                                                          float *out, long long n, int vec)
{
    const long long stride = (long long)gridDim.x * BLOCK;
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (vec) {
        const long long n4 = n / 4;
        for (; i < n4; i += stride) {
#if WM_RM_NT
            const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(g) + i);
#else
            const f32x4 gv = reinterpret_cast<const f32x4 *>(g)[i];
#endif
            const f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(m) + i);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = mv[e] > 0.0f ? gv[e] : 0.0f;
#if WM_RM_NT
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(out) + i);
#else
            reinterpret_cast<f32x4 *>(out)[i] = o;
#endif
        }
        i = n4 * 4 + (long long)blockIdx.x * BLOCK + threadIdx.x;
        if (blockIdx.x != 0) return;
    }
    for (; i < n; i += stride) out[i] = m[i] > 0.0f ? g[i] : 0.0f;
}

static int relu_mask_launch(const float *g, const float *ymask, float *out, int64_t n, void *stream)
{
    if (n == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_relu_mask_f32: no HIP device");
    const int vec = ((((uintptr_t)g | (uintptr_t)ymask | (uintptr_t)out) & 15) == 0) ? 1 : 0;
    long long grid = ((vec ? n / 4 : n) + BLOCK - 1) / BLOCK;
    const long long cap = (long long)wm_num_cus() * WM_RM_CAP;
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)grid), dim3(BLOCK), 0, (hipStream_t)stream,
                       g, ymask, out, (long long)n, vec);
    WM_LAUNCH_CHECK("relu_mask_kernel");
    return WM_OK;
}

extern "C" int wm_relu_mask_f32(float *gx, const float *ymask, int64_t n, void *stream)
{
    WM_REQUIRE(gx && ymask && n >= 0, WM_ERR_INVALID, "wm_relu_mask_f32: bad arguments");
    return relu_mask_launch(gx, ymask, gx, n, stream);
}

extern "C" int wm_relu_mask_to_f32(const float *g, const float *ymask, float *out, int64_t n,
                                   void *stream)
{
    WM_REQUIRE(g && ymask && out && n >= 0, WM_ERR_INVALID, "wm_relu_mask_to_f32: bad arguments");
    return relu_mask_launch(g, ymask, out, n, stream);
}

extern "C" size_t wm_conv2d_bwd_wb_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kH,
                                                   int kW)
{
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || kH <= 0 || kW <= 0 || kH > H ||
        kW > W)
        return 0;
    const int OH = H - kH + 1, OW = W - kW + 1;
    const int nw = Cout * Cin * kH * kW, nent = nw + Cout;
    if (stencil_shape(Cin, Cout, kH, kW, OW, (long long)N * H)) {
        // upper bound on the stencil grid: 6 blocks per CU (stencil_plan) + 1
        const size_t blocks = (size_t)wm_num_cus() * 6 + 1;
        return wm_align_up(blocks * (size_t)nent * sizeof(double), 256);
    }
    if (mfma_shape(N, Cin, H, W, Cout, kH, kW))
        return wm_conv3d_bwd_wb_workspace_bytes(N, Cin, 1, H, W, Cout, 1, kH, kW);
    const int splits = bwd_wb_generic_splits(N, OH, OW, nent);
    return wm_align_up((size_t)splits * nent * sizeof(double), 256);
}

// The partial-sum pass of the weight/bias gradient for the shapes that do not go to the MFMA
// kernels: float64 partials in `partial`, nparts of them per entry, entry-major ([nent][nparts],
// the flat stencil kernel) or part-major ([nparts][nent]).
static int bwd_wb_partials_launch(const float *x, const float *gy, double *partial,
                                  size_t workspace_bytes, int N, int Cin, int H, int W, int Cout,
                                  int kH, int kW, hipStream_t st, int *nparts_out,
                                  int *entry_major_out)
{
    const int OH = H - kH + 1, OW = W - kW + 1;
    const int nw = Cout * Cin * kH * kW, nent = nw + Cout;
    int nparts = 0;
    *entry_major_out = 0;
    if (stencil_shape(Cin, Cout, kH, kW, OW, (long long)N * H)) {
#define BWDF(KH_, KW_)                                                                          \
    FlatPlan fp;                                                                                \
    const int waves_ = (OW + WaveCols<KW_>::CW - 1) / WaveCols<KW_>::CW;                        \
    int occ = 0;                                                                                \
    if (KH_ * KW_ <= 9 && waves_ <= 16) {                                                       \
        if (int rc = cached_occupancy(conv_bwd_wb_c1f_kernel<KH_, KW_>, waves_ * 64, &occ))     \
            return rc;                                                                          \
        if (occ * waves_ > 30) occ = 30 / waves_;                                               \
        if (occ > WM_KNOB(WM_BWD_OCC_CAP)) occ = WM_KNOB(WM_BWD_OCC_CAP);                       \
        if (occ < 1) occ = 1;                                                                   \
    }                                                                                           \
    if (occ > 0 &&                                                                              \
        flat_plan(conv_bwd_wb_c1f_kernel<KH_, KW_>, N, OH, OW, W, kH, x, gy,                    \
                  WaveCols<KW_>::CW, &fp, (long long)wm_num_cus() * occ)) {                     \
        hipLaunchKernelGGL((conv_bwd_wb_c1f_kernel<KH_, KW_>), dim3((unsigned)fp.segs),         \
                           dim3(fp.waves * 64), 0, st, x, gy, partial, H, W, OH, OW,            \
                           fp.rows_base, fp.rows_rem, (unsigned)fp.segs);                       \
        WM_LAUNCH_CHECK("conv_bwd_wb_c1f_kernel");                                              \
        *nparts_out = (int)fp.segs;                                                             \
        *entry_major_out = 1;                                                                   \
        return WM_OK;                                                                           \
    }
        WM_STENCIL_DISPATCH(BWDF)
#undef BWDF
    }
    if (stencil_shape(Cin, Cout, kH, kW, OW, (long long)N * H)) {
#define BWD(KH_, KW_)                                                                         \
    StencilPlan p;                                                                            \
    if (int rc = stencil_plan(conv_bwd_wb_c1_kernel<KH_, KW_>, N, OH, OW, &p)) return rc;     \
    WM_REQUIRE((size_t)p.blocks * nent * sizeof(double) <= workspace_bytes, WM_ERR_WORKSPACE, \
               "wm_conv2d_bwd_wb_f32: workspace too small for %d blocks", p.blocks);          \
    hipLaunchKernelGGL((conv_bwd_wb_c1_kernel<KH_, KW_>), dim3(p.blocks), dim3(BLOCK), 0, st, \
                       x, gy, partial, H, W, OH, OW, p.strips, p.segs, p.rows_base,      \
                       p.rows_rem);         \
    nparts = p.blocks;
        WM_STENCIL_DISPATCH(BWD)
#undef BWD
        WM_LAUNCH_CHECK("conv_bwd_wb_c1_kernel");
    } else {
        const int splits = bwd_wb_generic_splits(N, OH, OW, nent);
        hipLaunchKernelGGL(conv_bwd_wb_generic_kernel, dim3(nent, splits), dim3(BLOCK), 0, st, x,
                           gy, partial, N, Cin, H, W, Cout, kH, kW, OH, OW, nw, nent);
        WM_LAUNCH_CHECK("conv_bwd_wb_generic_kernel");
        nparts = splits;
    }
    *nparts_out = nparts;
    return WM_OK;
}

static int partials_finish_launch(const double *partial, int nparts, int entry_major, int nw,
                                  int nb, float *gw, float *gb, hipStream_t st)
{
    const int nent = nw + nb;
    if (entry_major) {
        hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3(nent), dim3(256), 0, st, partial,
                           nparts, nw, gw, gb);
        WM_LAUNCH_CHECK("reduce_partials_wide_kernel");
    } else {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(nent), dim3(64), 0, st, partial, nparts,
                           nent, nw, gw, gb);
        WM_LAUNCH_CHECK("reduce_partials_kernel");
    }
    return WM_OK;
}

extern "C" int wm_conv2d_bwd_wb_f32(const float *x, const float *gy, float *gw, float *gb,
                                    void *workspace, size_t workspace_bytes, int N, int Cin, int H,
                                    int W, int Cout, int kH, int kW, void *stream)
{
    WM_REQUIRE(x && gy && gw, WM_ERR_INVALID, "wm_conv2d_bwd_wb_f32: null pointer");
    if (int rc = check_dims(N, Cin, H, W, Cout, kH, kW)) return rc;
    const size_t need = wm_conv2d_bwd_wb_workspace_bytes(N, Cin, H, W, Cout, kH, kW);
    WM_REQUIRE(workspace && workspace_bytes >= need, WM_ERR_WORKSPACE,
               "wm_conv2d_bwd_wb_f32: workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    if (mfma_shape(N, Cin, H, W, Cout, kH, kW))
        return wm_conv3d_bwd_wb_f32(x, gy, gw, gb, workspace, workspace_bytes, N, Cin, 1, H, W,
                                    Cout, 1, kH, kW, stream);
    int nparts = 0, entry_major = 0;
    if (int rc = bwd_wb_partials_launch(x, gy, (double *)workspace, workspace_bytes, N, Cin, H, W,
                                        Cout, kH, kW, st, &nparts, &entry_major))
        return rc;
    return partials_finish_launch((const double *)workspace, nparts, entry_major,
                                  Cout * Cin * kH * kW, Cout, gw, gb, st);
}

// Deferred finalize: the partial-sum pass alone.  layout[0] = partials per entry, layout[1] = 1 if
// the workspace holds them entry-major.  wm_partials_finish_f32 or wm_adam_step_partials_f32
// consume them; the sums are the ones wm_conv2d_bwd_wb_f32 forms, in the same order.
extern "C" int wm_conv2d_bwd_wb_partials_f32(const float *x, const float *gy, void *workspace,
                                             size_t workspace_bytes, int N, int Cin, int H, int W,
                                             int Cout, int kH, int kW, int *layout, void *stream)
{
    WM_REQUIRE(x && gy && layout, WM_ERR_INVALID, "wm_conv2d_bwd_wb_partials_f32: null pointer");
    if (int rc = check_dims(N, Cin, H, W, Cout, kH, kW)) return rc;
    WM_REQUIRE(!mfma_shape(N, Cin, H, W, Cout, kH, kW), WM_ERR_UNSUPPORTED,
               "wm_conv2d_bwd_wb_partials_f32: this shape takes the MFMA weight-gradient kernels, "
               "whose slab reduction is not deferrable -- use wm_conv2d_bwd_wb_f32");
    const size_t need = wm_conv2d_bwd_wb_workspace_bytes(N, Cin, H, W, Cout, kH, kW);
    WM_REQUIRE(workspace && workspace_bytes >= need, WM_ERR_WORKSPACE,
               "wm_conv2d_bwd_wb_partials_f32: workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
    return bwd_wb_partials_launch(x, gy, (double *)workspace, workspace_bytes, N, Cin, H, W, Cout,
                                  kH, kW, (hipStream_t)stream, &layout[0], &layout[1]);
}

extern "C" int wm_partials_finish_f32(const void *partials, int nparts, int entry_major, int nw,
                                      int nb, float *gw, float *gb, void *stream)
{
    WM_REQUIRE(partials && gw && nparts > 0 && nw > 0 && nb >= 0, WM_ERR_INVALID,
               "wm_partials_finish_f32: bad arguments");
    return partials_finish_launch((const double *)partials, nparts, entry_major, nw, nb, gw, gb,
                                  (hipStream_t)stream);
}

extern "C" int wm_adam_step_partials_f32(float *params, const void *partials, int nparts,
                                         int entry_major, float *m, float *v, int count,
                                         float *grads_out, int step, double lr, double beta1,
                                         double beta2, double eps, float grad_scale, void *stream)
{
    WM_REQUIRE(params && partials && m && v, WM_ERR_INVALID,
               "wm_adam_step_partials_f32: null pointer");
    WM_REQUIRE(count > 0 && nparts > 0 && step >= 1, WM_ERR_INVALID,
               "wm_adam_step_partials_f32: bad count/nparts/step");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const AdamScalars a{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                        (float)(lr / bc1), (float)sqrt(bc2), grad_scale};
    if (entry_major)
        hipLaunchKernelGGL(adam_partials_kernel<true>, dim3(count), dim3(256), 0,
                           (hipStream_t)stream, params, (const double *)partials, nparts, count, m,
                           v, grads_out, a);
    else
        hipLaunchKernelGGL(adam_partials_kernel<false>, dim3(count), dim3(64), 0,
                           (hipStream_t)stream, params, (const double *)partials, nparts, count, m,
                           v, grads_out, a);
    WM_LAUNCH_CHECK("adam_partials_kernel");
    return WM_OK;
}

// No counterpart in the reference: the same weight/bias gradient for callers that cannot allocate
// (HIP graph capture, tight serving loops).  The partial sums live in this library's static
// device array instead of a caller workspace; every kernel of the path overwrites the part of it
// that the finalize kernel reads, so nothing has to be cleared between calls.
extern "C" size_t wm_conv2d_bwd_wb_noalloc_max_bytes(void) { return SCRATCH_BYTES; }

extern "C" int wm_conv2d_bwd_wb_noalloc_f32(const float *x, const float *gy, float *gw, float *gb,
                                            int N, int Cin, int H, int W, int Cout, int kH, int kW,
                                            void *stream)
{
    WM_REQUIRE(x && gy && gw, WM_ERR_INVALID, "wm_conv2d_bwd_wb_noalloc_f32: null pointer");
    if (int rc = check_dims(N, Cin, H, W, Cout, kH, kW)) return rc;
    const size_t need = wm_conv2d_bwd_wb_workspace_bytes(N, Cin, H, W, Cout, kH, kW);
    WM_REQUIRE(need > 0 && need <= SCRATCH_BYTES, WM_ERR_UNSUPPORTED,
               "wm_conv2d_bwd_wb_noalloc_f32: this shape needs %zu bytes of partial sums, the "
               "static scratch holds %zu -- use wm_conv2d_bwd_wb_f32 with a caller workspace",
               need, SCRATCH_BYTES);
    // every GPU holds its own copy of the scratch array, and scratch_ptr() hands out the CURRENT
    // device's: a stream that belongs to another device would get this device's array
    if (stream) {
        int cur = 0, sdev = 0;
        WM_HIP(hipGetDevice(&cur));
        WM_HIP(hipStreamGetDevice((hipStream_t)stream, &sdev));
        WM_REQUIRE(cur == sdev, WM_ERR_INVALID,
                   "wm_conv2d_bwd_wb_noalloc_f32: the stream belongs to device %d but device %d is "
                   "current -- make the stream's device current (hipSetDevice) before the call",
                   sdev, cur);
    }
    void *ptr = nullptr;
    if (int rc = scratch_ptr(&ptr)) return rc;
    return wm_conv2d_bwd_wb_f32(x, gy, gw, gb, ptr, SCRATCH_BYTES, N, Cin, H, W, Cout, kH, kW,
                                stream);
}
