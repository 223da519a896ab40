// conv3d.hip -- 3-D cross-correlation (valid, stride 1, channels_first) forward / backward and the
// 2x2x2 max pool of the reference's Conv3D model:
//   keras.layers.Conv3D(32, 7, strides=1, data_format="channels_first")   convolution.py:164,170
//   keras.layers.Conv3D(64, 5, strides=1, data_format="channels_first")   convolution.py:166,172
//   keras.layers.MaxPool3D(pool_size=2, strides=2)                        convolution.py:165,167
//
// All three conv passes are implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32 products and
// fp32 accumulation, MI355X_MICROARCH.md), one 16x16 output tile per wave, operands gathered
// straight from global memory / L2 (the tensors of this model are a few MB):
//   forward   Y[o, n]  = sum_k  W[o, k]   Xcol[k, n]     k = (c,kd,kh,kw), n = (b,od,oh,ow)
//   bwd data  GX[c, m] = sum_kk W[kk, c]  GYcol[kk, m]   kk = (o,kd,kh,kw), m = (b,d,h,w)
//   bwd w,b   GW[o, k] = sum_n  GY[o, n]  Xcol[k, n]     (+ a column of ones for the bias)
// The k -> tensor-offset decode is done once per workgroup into an LDS table (chunks of 4096
// entries); the weight/bias gradient is split over n-chunks of 1024 outputs, written as float64
// partials and combined in a fixed order (bitwise reproducible, no atomics).
#include "common.h"
#include "conv_internal.h"
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int KCHUNK = 4096;      // table entries resident in LDS at a time
constexpr int NCHUNK = 1024;      // outputs per weight-gradient split
constexpr int UNR = 4;            // MFMA steps whose operand loads are issued together

struct Dims3 {
    int B, Cin, D, H, W, Cout, kD, kH, kW, OD, OH, OW;
};

__device__ __forceinline__ f32x4 mma(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// offset of tap k = (c,kd,kh,kw) relative to the window origin inside one batch item of x
__device__ __forceinline__ int tap_offset(const Dims3 &d, int k)
{
    const int kw = k % d.kW;
    int t = k / d.kW;
    const int kh = t % d.kH;
    t /= d.kH;
    const int kd = t % d.kD;
    const int c = t / d.kD;
    return ((c * d.D + kd) * d.H + kh) * d.W + kw;
}

// Combine the WAVES per-wave partial accumulators of one column tile (fixed order) and hand every
// thread its share of the OT*256 outputs: element idx = t*256 + lane*4 + r.
template <int OT>
__device__ __forceinline__ void reduce_tiles(float *part, const f32x4 (&acc)[OT], int wave, int lane)
{
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[((wave * OT + t) * 64 + lane) * 4 + r] = acc[t][r];
    __syncthreads();
}

// ------------------------------------------------------------------------------ forward
// One workgroup = one 16-column tile of outputs (OT channel tiles; grid.y splits the channel tiles
// when there are few column tiles); its NW waves split the contraction (16-tap super-steps, round
// robin) and the partial tiles are summed through LDS in a fixed order.  With one wave per tile
// the second layer of the reference ((16,16,7,7,14) -> (16,64,3,3,10): only 90 column tiles,
// K = 2000) ran 23 workgroups of 500 dependent steps.  Inside a super-step MFMA step u takes taps
// 4g + u (g = lane group), so lane (o, g) needs four CONSECUTIVE weights: one dwordx4 load for
// four steps instead of four 16-line gathers -- the kernel is bound by its gathers (66 -> 42 us
// with the channel tiles in grid.y, -> 20 us with the 16-byte weight loads).  Taps past K read
// into the next row of w and meet x = 0; the chunk that holds the end of the rows is loaded
// element-wise with a clamp.
template <int OT, int NW>
__global__ __launch_bounds__(NW * 64) void conv3d_fwd_tap_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ w,
                                                               const float *__restrict__ bias,
                                                               float *__restrict__ y, Dims3 d, int K,
                                                               int Ntot, int relu)
{
    __shared__ int off[KCHUNK];
    __shared__ float part[NW * OT * 256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int S = d.OD * d.OH * d.OW;
    const int tile = blockIdx.x, ot0 = blockIdx.y * OT;   // grid.y splits the channel tiles
    int n = tile * 16 + j;
    if (n >= Ntot) n = Ntot - 1;                     // keep the gathers in bounds; never stored
    const int b_ = n / S, sp = n % S;
    const int ow = sp % d.OW, oh = (sp / d.OW) % d.OH, od = sp / (d.OW * d.OH);
    const size_t nbase = (((size_t)b_ * d.Cin * d.D + od) * d.H + oh) * d.W + ow;
    const long long wtotal = (long long)d.Cout * K;

    f32x4 acc[OT];
#pragma unroll
    for (int t = 0; t < OT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < K; kc += KCHUNK) {
        const int kn = min(KCHUNK, K - kc);
        const int nsup = (kn + 15) >> 4;                  // 16-tap super-steps of this chunk
        __syncthreads();
        for (int k = tid; k < nsup * 16; k += (NW * 64)) off[k] = k < kn ? tap_offset(d, kc + k) : -1;
        __syncthreads();
        for (int s0 = wave; s0 < nsup; s0 += 2 * NW) {
            float xv[2][4];
            f32x4 wq[2][OT];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ss = s0 + h * NW;
                const bool live = ss < nsup;              // wave-uniform
                const int k0 = (live ? ss : 0) * 16 + 4 * g;
#pragma unroll
                for (int t = 0; t < OT; ++t) {
                    const int o = min((ot0 + t) * 16 + j, d.Cout - 1);
                    const long long wi = (long long)o * K + kc + k0;
                    if (kc + (live ? ss : 0) * 16 + 16 > K) {   // the chunk that holds the end of the rows
#pragma unroll
                        for (int u = 0; u < 4; ++u) wq[h][t][u] = w[wi + u < wtotal ? wi + u : wtotal - 1];
                    } else {
                        wq[h][t] = *(const f4u *)(w + wi);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int o_ = off[k0 + u];
                    const bool in = live && o_ >= 0;
                    const float xl = x[nbase + (in ? o_ : 0)];
                    xv[h][u] = in ? xl : 0.0f;
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < OT; ++t) acc[t] = mma(wq[h][t][u], xv[h][u], acc[t]);
        }
    }
    reduce_tiles<OT>(part, acc, wave, lane);
    for (int idx = tid; idx < OT * 256; idx += (NW * 64)) {
        const int t = idx >> 8, l = (idx >> 2) & 63, r = idx & 3;
        const int o = (ot0 + t) * 16 + 4 * (l >> 4) + r;
        const int nn = tile * 16 + (l & 15);
        if (o >= d.Cout || nn >= Ntot) continue;
        float v = bias ? bias[o] : 0.0f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) v += part[((wv * OT + t) * 64 + l) * 4 + r];
        if (relu) v = fmaxf(v, 0.0f);
        y[((size_t)(nn / S) * d.Cout + o) * S + nn % S] = v;
    }
}

// ------------------------------------------------------------------------------ forward, small volumes
// The first layer of the reference's Conv3D model ((16,1,20^3) x 7^3 -> 32 channels): the whole
// packed weight matrix (44 KB) and the input slab of one output plane (b, od) fit in LDS together,
// so a workgroup per plane reads both MFMA operands from LDS (tap offset + position offset, as in
// conv3d_bwd_wb_vol_kernel) instead of gathering them per step (50 us for 1 GFLOP).  A wave takes
// position tiles wave, wave+4, .. two at a time (2 OT independent chains).
template <int OT>
__global__ __launch_bounds__(BLOCK) void conv3d_fwd_vol_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ w,
                                                               const float *__restrict__ bias,
                                                               float *__restrict__ y, Dims3 d, int K,
                                                               int relu)
{
    extern __shared__ __align__(16) float smem[];
    const int HW = d.H * d.W, slabn = d.Cin * d.kD * HW, P = d.OH * d.OW, PT = (P + 15) >> 4;
    const int steps = (((K + 3) >> 2) + 1) & ~1;         // even, padded with zero weights
    // [slabn] x slab | [steps][OT][64] packed weights | [steps*4] tap offsets | [PT*16] positions
    const int woff = slabn, toff_ = woff + steps * OT * 64, poff = toff_ + steps * 4;
    float *wl = smem + woff;
    int *tof = (int *)(smem + toff_), *pos = (int *)(smem + poff);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int plane = blockIdx.x, b_ = plane / d.OD, od = plane - b_ * d.OD;
    {
        const int run = d.kD * HW;
        const float *src = x + ((size_t)b_ * d.Cin * d.D + od) * HW;
        for (int e = tid; e < slabn; e += BLOCK) {
            const int c = e / run, r = e - c * run;
            smem[e] = src[(size_t)c * d.D * HW + r];
        }
        for (int e = tid; e < steps * OT * 64; e += BLOCK) wl[e] = 0.0f;
        for (int k = tid; k < steps * 4; k += BLOCK) {
            int o_ = 0;
            if (k < K) {
                const int kw = k % d.kW, t1 = k / d.kW, kh = t1 % d.kH, t2 = t1 / d.kH;
                o_ = t2 * HW + kh * d.W + kw;
            }
            tof[k] = o_;
        }
        for (int n = tid; n < PT * 16; n += BLOCK) {
            const int nc = n < P ? n : P - 1;             // clamped: computed, never stored
            pos[n] = (nc / d.OW) * d.W + nc % d.OW;
        }
    }
    __syncthreads();
    for (int e = tid; e < d.Cout * K; e += BLOCK) {
        const int o = e / K, k = e - o * K;
        wl[((k >> 2) * OT + (o >> 4)) * 64 + (k & 3) * 16 + (o & 15)] = w[e];
    }
    __syncthreads();
    float bv[OT][4];
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = t * 16 + 4 * g + r;
            bv[t][r] = (bias && o < d.Cout) ? bias[o] : 0.0f;
        }
    const size_t ybase = ((size_t)b_ * d.Cout * d.OD + od) * P;
    for (int pt = wave; pt < PT; pt += 2 * WAVES) {
        const int pt1 = pt + WAVES < PT ? pt + WAVES : pt;   // second tile (or a repeat)
        const int p0 = pos[pt * 16 + j], p1 = pos[pt1 * 16 + j];
        f32x4 acc0[OT], acc1[OT];
#pragma unroll
        for (int t = 0; t < OT; ++t) { acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[t] = acc0[t]; }
        for (int s_ = 0; s_ < steps; s_ += 2) {
            const int o0 = tof[4 * s_ + g], o1 = tof[4 * s_ + 4 + g];
            const float b00 = smem[o0 + p0], b01 = smem[o0 + p1];
            const float b10 = smem[o1 + p0], b11 = smem[o1 + p1];
#pragma unroll
            for (int t = 0; t < OT; ++t) {
                const float a0 = wl[(s_ * OT + t) * 64 + lane];
                acc0[t] = mma(a0, b00, acc0[t]);
                acc1[t] = mma(a0, b01, acc1[t]);
            }
#pragma unroll
            for (int t = 0; t < OT; ++t) {
                const float a1 = wl[((s_ + 1) * OT + t) * 64 + lane];
                acc0[t] = mma(a1, b10, acc0[t]);
                acc1[t] = mma(a1, b11, acc1[t]);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tile = half ? pt1 : pt, n = tile * 16 + j;
            if ((half && pt1 == pt) || n >= P) continue;
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = t * 16 + 4 * g + r;
                    if (o < d.Cout) {
                        float v = (half ? acc1[t][r] : acc0[t][r]) + bv[t][r];
                        if (relu) v = fmaxf(v, 0.0f);
                        y[ybase + (size_t)o * d.OD * P + n] = v;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------ forward, wide
// For large outputs (the 721 x 1440 grids of BASELINE config 4b): a workgroup owns WIDE_TILES
// consecutive 16-column tiles, each wave CPW of them for every channel tile, and the weights of a
// K-chunk are packed once into LDS in MFMA operand order and reused by all sixteen column tiles
// (in the split-K kernel above every wave gathers its weights from L1/L2 per step: 16 scattered
// lines per operand against 8 cycles per MFMA).
constexpr int CPW = 4;                    // column tiles per wave
constexpr int WIDE_TILES = CPW * WAVES;   // column tiles per workgroup
constexpr int WIDE_STEPS = 32;            // MFMA k-steps (of 4 taps) per LDS chunk
#ifndef WB_WIDE_BPC
#define WB_WIDE_BPC 3                     // persistent weight-gradient workgroups per CU
#endif

template <int OT>
__global__ __launch_bounds__(BLOCK) void conv3d_fwd_wide_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ w,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ y, Dims3 d,
                                                                int K, int Ntot, int relu)
{
    __shared__ int off[WIDE_STEPS * 4];
    __shared__ float wl[WIDE_STEPS * OT * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int S = d.OD * d.OH * d.OW;
    size_t nbase[CPW];
    int nn[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        int n = ((int)blockIdx.x * WIDE_TILES + wave + c * WAVES) * 16 + j;
        nn[c] = n;
        if (n >= Ntot) n = Ntot - 1;
        const int b_ = n / S, sp = n % S;
        const int ow = sp % d.OW, oh = (sp / d.OW) % d.OH, od = sp / (d.OW * d.OH);
        nbase[c] = (((size_t)b_ * d.Cin * d.D + od) * d.H + oh) * d.W + ow;
    }
    f32x4 acc[CPW][OT];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int t = 0; t < OT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < K; kc += WIDE_STEPS * 4) {
        const int kn = min(WIDE_STEPS * 4, K - kc);
        const int steps = (kn + 3) >> 2;
        __syncthreads();
        if (tid < WIDE_STEPS * 4) off[tid] = tid < kn ? tap_offset(d, kc + tid) : 0;
        // operand (step, t) for lane l: W[o = 16 t + (l & 15)][k = kc + 4 step + (l >> 4)]
        for (int e = tid; e < steps * OT * 64; e += BLOCK) {
            const int l = e & 63, q = e >> 6, t = q % OT, st = q / OT;
            const int o = t * 16 + (l & 15), k = 4 * st + (l >> 4);
            wl[e] = (k < kn && o < d.Cout) ? w[(size_t)o * K + kc + k] : 0.0f;
        }
        __syncthreads();
        for (int sb = 0; sb < steps; sb += UNR) {
            float xv[UNR][CPW];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int st = min(sb + u, steps - 1);        // repeats are skipped below
                const int o_ = off[4 * st + g];
#pragma unroll
                for (int c = 0; c < CPW; ++c) xv[u][c] = x[nbase[c] + o_];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (sb + u < steps) {
#pragma unroll
                    for (int t = 0; t < OT; ++t) {
                        const float a = wl[((sb + u) * OT + t) * 64 + lane];
#pragma unroll
                        for (int c = 0; c < CPW; ++c) acc[c][t] = mma(a, xv[u][c], acc[c][t]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        if (nn[c] >= Ntot) continue;
        const int b_ = nn[c] / S, sp = nn[c] % S;
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = t * 16 + 4 * g + r;
                if (o < d.Cout) {
                    float v = acc[c][t][r] + (bias ? bias[o] : 0.0f);
                    if (relu) v = fmaxf(v, 0.0f);
                    y[((size_t)b_ * d.Cout + o) * S + sp] = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------ forward, rows
// The kernel for the large grids (BASELINE config 4b: 32 -> 32 channels, 3x3, 721 x 1440).
// Persistent workgroups; the whole packed weight matrix stays in LDS (36 KB for that shape) and
// there is no barrier after the pack.  A wave's unit of work is 64 consecutive outputs of ONE
// output row (the last segment of a row is shifted left to end at the row's end and recomputes a
// few columns), so the position decode is two scalar divisions per unit instead of five vector
// divisions per column tile, the four column tiles of the wave are immediate offsets of one
// address, and every store instruction covers whole 64-byte runs of four channels' rows.
// Operand gathers are double-buffered by hand (set A in flight while set B feeds the MFMAs;
// hipcc does not pipeline across the MFMA builtin and otherwise sinks each load to its use).
// Workgroups of one XCD (blockIdx & 7) walk one contiguous eighth of the units, so the kH-fold
// re-read of the input rows is served by that XCD's L2.
constexpr int RESIDENT_LDS = 64 * 1024;
constexpr int ROWS_SEG = 16 * CPW;            // outputs per wave unit

__host__ __device__ inline int rows_steps(int K2) { return ((K2 + 3) >> 2) + 2 * UNR - 1 & ~(2 * UNR - 1); }
__host__ __device__ inline size_t rows_lds_bytes(int K2, int tiles)
{
    const size_t steps = (size_t)rows_steps(K2);
    return steps * tiles * 64 * sizeof(float) + steps * 4 * sizeof(int);
}

// (NC_: the unit's 16-column chunks, CPW or -- the short last segment of a row -- CPW / 2)
#define ROWS_LOAD(dst, sbase)                                                          \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                  \
        const unsigned a_ = lane0 + off[4 * ((sbase) + u) + g];                        \
        _Pragma("unroll") for (int c = 0; c < NC_; ++c)                                \
            dst[u][c] = *(const float *)(xb + a_ + 64u * c);                           \
    }
#define ROWS_MMA(src, sbase)                                                           \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u)                                    \
        _Pragma("unroll") for (int t = 0; t < OT; ++t) {                               \
            const float a = wl[(((sbase) + u) * OT + t) * 64 + lane];                  \
            _Pragma("unroll") for (int c = 0; c < NC_; ++c)                            \
                acc[c][t] = mma(a, src[u][c], acc[c][t]);                              \
        }

// A row's last segment holds OW - 64 (segs - 1) outputs; when that is at most half a unit it is
// run as a unit of CPW / 2 chunks ending at the row's end instead of a full one shifted left
// (1438 outputs: 22.5 units of work per row instead of 23).
__host__ __device__ inline bool rows_short_tail(int width)
{
    const int rem = width - (width - 1) / ROWS_SEG * ROWS_SEG;
    return width >= ROWS_SEG && rem <= ROWS_SEG / 2;
}

template <int OT>
__global__ __launch_bounds__(BLOCK) void conv3d_fwd_rows_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ w,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ y, Dims3 d,
                                                                int K, int relu)
{
    extern __shared__ __align__(16) float smem[];
    const int steps = rows_steps(K);                          // padded with zero weights
    float *wl = smem;
    unsigned *off = (unsigned *)(smem + (size_t)steps * OT * 64);   // byte offsets of the taps
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned S = (unsigned)(d.OD * d.OH * d.OW);

    for (int k = tid; k < steps * 4; k += BLOCK) off[k] = k < K ? 4u * (unsigned)tap_offset(d, k) : 0u;
    for (int e = tid; e < steps * OT * 64; e += BLOCK) wl[e] = 0.0f;
    __syncthreads();
    // operand (step, t) for lane l: W[o = 16 t + (l & 15)][k = 4 step + (l >> 4)]; read in row order
    for (int e = tid; e < d.Cout * K; e += BLOCK) {
        const int o = e / K, k = e - o * K;
        wl[((k >> 2) * OT + (o >> 4)) * 64 + (k & 3) * 16 + (o & 15)] = w[e];
    }
    __syncthreads();

    float bv[OT][4];                                      // this lane's output channels' biases
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = t * 16 + 4 * g + r;
            bv[t][r] = (bias && o < d.Cout) ? bias[o] : 0.0f;
        }
    const int segs = (d.OW + ROWS_SEG - 1) / ROWS_SEG;
    const bool short_tail = rows_short_tail(d.OW);
    const int units = d.B * d.OD * d.OH * segs;
    const int groups = (units + WAVES - 1) / WAVES;
    const int gpx = (groups + 7) >> 3;
    const int xcd = blockIdx.x & 7, nslot = gridDim.x >> 3;
    const int gend = min(groups, (xcd + 1) * gpx);
    const size_t xsample = (size_t)d.Cin * d.D * d.H * d.W, ysample = (size_t)d.Cout * S;
    for (int grp = xcd * gpx + ((int)blockIdx.x >> 3); grp < gend; grp += nslot) {
        const int unit = grp * WAVES + wave;
        if (unit >= units) continue;                       // wave-uniform; no barrier below
        const int row = unit / segs, seg = unit - row * segs;
        const bool half = short_tail && seg == segs - 1;
        const int ow0 = half ? d.OW - ROWS_SEG / 2 : min(seg * ROWS_SEG, d.OW - ROWS_SEG);
        const int oh = row % d.OH, q = row / d.OH, od = q % d.OD, b_ = q / d.OD;
        const char *xb = (const char *)(x + (size_t)b_ * xsample);
        const unsigned lane0 = 4u * (unsigned)((od * d.H + oh) * d.W + ow0 + j);
        char *yb = (char *)(y + (size_t)b_ * ysample);
        const unsigned sp0 = 4u * ((unsigned)((od * d.OH + oh) * d.OW + ow0 + j) + (unsigned)(4 * g) * S);
        auto run = [&](auto nc_tag) {
            constexpr int NC_ = decltype(nc_tag)::value;
            f32x4 acc[CPW][OT];
#pragma unroll
            for (int c = 0; c < NC_; ++c)
#pragma unroll
                for (int t = 0; t < OT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float xa[UNR][CPW], xc[UNR][CPW];
            ROWS_LOAD(xa, 0)
            for (int sb = 0; sb < steps; sb += 2 * UNR) {
                ROWS_LOAD(xc, sb + UNR)
                __builtin_amdgcn_sched_barrier(0);
                ROWS_MMA(xa, sb)
                __builtin_amdgcn_sched_barrier(0);
                const int nx_ = sb + 2 * UNR < steps ? sb + 2 * UNR : sb;     // last round: unused re-read
                ROWS_LOAD(xa, nx_)
                __builtin_amdgcn_sched_barrier(0);
                ROWS_MMA(xc, sb + UNR)
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = t * 16 + 4 * g + r;
                    if (o < d.Cout) {
                        const unsigned yo = sp0 + 4u * (unsigned)(t * 16 + r) * S;
#pragma unroll
                        for (int c = 0; c < NC_; ++c) {
                            float v = acc[c][t][r] + bv[t][r];
                            if (relu) v = fmaxf(v, 0.0f);
                            *(float *)(yb + yo + 64u * c) = v;
                        }
                    }
                }
        };
        if (half) run(std::integral_constant<int, CPW / 2>{});
        else run(std::integral_constant<int, CPW>{});
    }
}

// ------------------------------------------------------------------------------ backward data
// table entry for kk = (o,kd,kh,kw): o << 24 | kd << 16 | kh << 8 | kw.  Same decomposition as the
// forward kernel: one workgroup per 16-position tile, the four waves split the contraction.
template <int CT, int NW>   // CT 16-row tiles of input channels
__global__ __launch_bounds__(NW * 64) void conv3d_bwd_data_kernel(const float *__restrict__ gy,
                                                                const float *__restrict__ w,
                                                                float *__restrict__ gx, Dims3 d,
                                                                int K2, int Mtot)
{
    __shared__ int tab[KCHUNK];
    __shared__ float part[NW * CT * 256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int k3 = d.kD * d.kH * d.kW;
    const int S = d.OD * d.OH * d.OW, V = d.D * d.H * d.W;
    const int tile = blockIdx.x;
    int m = tile * 16 + j;
    if (m >= Mtot) m = Mtot - 1;
    const int b_ = m / V, pos = m % V;
    const int pw = pos % d.W, ph = (pos / d.W) % d.H, pd = pos / (d.W * d.H);
    const size_t gybase = (size_t)b_ * d.Cout * S;

    f32x4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < K2; kc += KCHUNK) {
        const int kn = min(KCHUNK, K2 - kc);
        const int kn4 = (kn + 3) & ~3;
        __syncthreads();
        for (int k = tid; k < kn4; k += (NW * 64)) {
            int e = -1;                               // padding: never in range
            if (k < kn) {
                const int kk = kc + k;
                const int o = kk / k3, r = kk % k3;
                const int kw = r % d.kW, kh = (r / d.kW) % d.kH, kd = r / (d.kW * d.kH);
                e = (o << 24) | (kd << 16) | (kh << 8) | kw;
            }
            tab[k] = e;
        }
        __syncthreads();
        for (int kb = 4 * wave; kb < kn4; kb += 4 * NW * UNR) {
            float gv[UNR], wv[UNR][CT];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int k = kb + u * 4 * NW + g;
                const int e = k < kn4 ? tab[k] : -1;
                const int o = (e >> 24) & 0x7f, kd = (e >> 16) & 0xff, kh = (e >> 8) & 0xff,
                          kw = e & 0xff;
                const int od = pd - kd, oh = ph - kh, ow = pw - kw;
                const bool in = e >= 0 && od >= 0 && od < d.OD && oh >= 0 && oh < d.OH &&
                                ow >= 0 && ow < d.OW;
                // out-of-range taps read element 0 of this batch item's gy (in bounds) times zero
                const size_t gi = in ? (size_t)o * S + (od * d.OH + oh) * d.OW + ow : 0;
                gv[u] = in ? gy[gybase + gi] : 0.0f;
                const int wsp = (kd * d.kH + kh) * d.kW + kw;
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    const int c = t * 16 + j;
                    wv[u][t] = (e >= 0 && c < d.Cin) ? w[((size_t)o * d.Cin + c) * k3 + wsp] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = mma(wv[u][t], gv[u], acc[t]);
        }
    }
    reduce_tiles<CT>(part, acc, wave, lane);
    for (int idx = tid; idx < CT * 256; idx += (NW * 64)) {
        const int t = idx >> 8, l = (idx >> 2) & 63, r = idx & 3;
        const int c = t * 16 + 4 * (l >> 4) + r;
        const int mm = tile * 16 + (l & 15);
        if (c >= d.Cin || mm >= Mtot) continue;
        float v = 0.0f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) v += part[((wv * CT + t) * 64 + l) * 4 + r];
        gx[((size_t)(mm / V) * d.Cin + c) * V + mm % V] = v;
    }
}

// ------------------------------------------------------------------------------ backward data, tap order
// The split-K kernel above with the contraction re-ordered so that the weights come in 16-byte
// pieces: within a 16-tap chunk of one output channel, MFMA step u takes taps 4g + u (g = lane
// group), so lane (c, g) needs taps 4g..4g+3 of row (o, c) -- four CONSECUTIVE floats, one
// dwordx4 load for four steps (16 lines per wave instruction instead of 64).  The kernel is bound
// by its gathers (rocprof: 137 us for the 2.8 GFLOP of the reference's second layer).  Taps past
// the end of a row read into the next row and meet gy = 0; only the last chunk of the very last
// rows is loaded element-wise with a clamp.  NW waves take super-steps (o, chunk) round-robin.
template <int CT, int NW>
__global__ __launch_bounds__(NW * 64) void conv3d_bwd_data_tap_kernel(const float *__restrict__ gy,
                                                                    const float *__restrict__ w,
                                                                    float *__restrict__ gx, Dims3 d,
                                                                    int Mtot)
{
    __shared__ int tab[KCHUNK + 16];
    __shared__ float part[NW * CT * 256];
    __shared__ int tpd[16], tph[16], tpw[16], clive[KCHUNK / 16 + 1], clist[KCHUNK / 16 + 1], ncl;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int k3 = d.kD * d.kH * d.kW, nch = (k3 + 15) >> 4;
    const int S = d.OD * d.OH * d.OW, V = d.D * d.H * d.W;
    const int tile = blockIdx.x;
    int m = tile * 16 + j;
    if (m >= Mtot) m = Mtot - 1;
    const int b_ = m / V, pos = m % V;
    const int pw = pos % d.W, ph = (pos / d.W) % d.H, pd = pos / (d.W * d.H);
    const size_t gybase = (size_t)b_ * d.Cout * S;
    const int pb = (pd * d.OH + ph) * d.OW + pw;
    for (int k = tid; k < nch * 16; k += NW * 64) {
        int e = -1;                                      // padding taps: never in range
        if (k < k3) {
            const int kw = k % d.kW, kh = (k / d.kW) % d.kH, kd = k / (d.kW * d.kH);
            e = (kd << 16) | (kh << 8) | kw;
        }
        tab[k] = e;
    }
    // Only the 16-tap chunks with a tap in range of one of the tile's 16 positions are walked: the
    // input gradient of a small volume meets mostly out-of-range taps (the reference's second
    // layer: 13 % of the (position, tap) pairs are in range, about half of the chunks of a tile).
    if (tid < 16) { tpd[tid] = pd; tph[tid] = ph; tpw[tid] = pw; }
    for (int c_ = tid; c_ < nch; c_ += NW * 64) clive[c_] = 0;
    __syncthreads();
    for (int e_ = tid; e_ < nch * 16; e_ += NW * 64) {
        const int c_ = e_ >> 4, jj = e_ & 15;
        bool any = false;
        for (int u = 0; u < 16; ++u) {
            const int e = tab[c_ * 16 + u];
            const int kd = (e >> 16) & 0xff, kh = (e >> 8) & 0xff, kw = e & 0xff;
            any |= e >= 0 && (unsigned)(tpd[jj] - kd) < (unsigned)d.OD &&
                   (unsigned)(tph[jj] - kh) < (unsigned)d.OH && (unsigned)(tpw[jj] - kw) < (unsigned)d.OW;
        }
        if (any) clive[c_] = 1;                          // same value from every writer
    }
    __syncthreads();
    if (tid == 0) {
        int n_ = 0;
        for (int c_ = 0; c_ < nch; ++c_)
            if (clive[c_]) clist[n_++] = c_;
        ncl = n_;
    }
    __syncthreads();
    f32x4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nlive = ncl, nsup = d.Cout * nlive;
    const long long wtotal = (long long)d.Cout * d.Cin * k3;
    for (int sp = wave; sp < nsup; sp += 2 * NW) {
        float gv[2][4];
        f32x4 wq[2][CT];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ss = sp + h * NW;
            const bool live = ss < nsup;                  // wave-uniform
            const int sc = live ? ss : 0;
            const int o = sc / nlive, ch = clist[sc - o * nlive], tap0 = ch * 16 + 4 * g;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int c = min(t * 16 + j, d.Cin - 1);
                const long long wi = ((long long)o * d.Cin + c) * k3 + tap0;
                if (o == d.Cout - 1 && ch == nch - 1) {  // the last rows' tail: stay inside w
#pragma unroll
                    for (int u = 0; u < 4; ++u) wq[h][t][u] = w[wi + u < wtotal ? wi + u : wtotal - 1];
                } else {
                    wq[h][t] = *(const f4u *)(w + wi);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tab[tap0 + u];
                const int kd = (e >> 16) & 0xff, kh = (e >> 8) & 0xff, kw = e & 0xff;
                const bool in = live && e >= 0 && (unsigned)(pd - kd) < (unsigned)d.OD &&
                                (unsigned)(ph - kh) < (unsigned)d.OH &&
                                (unsigned)(pw - kw) < (unsigned)d.OW;
                // out-of-range taps read element 0 of this batch item's gy (in bounds) times zero
                const int gi = in ? o * S + pb - ((kd * d.OH + kh) * d.OW + kw) : 0;
                gv[h][u] = in ? gy[gybase + gi] : 0.0f;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = mma(wq[h][t][u], gv[h][u], acc[t]);
    }
    reduce_tiles<CT>(part, acc, wave, lane);
    for (int idx = tid; idx < CT * 256; idx += (NW * 64)) {
        const int t = idx >> 8, l = (idx >> 2) & 63, r = idx & 3;
        const int c = t * 16 + 4 * (l >> 4) + r;
        const int mm = tile * 16 + (l & 15);
        if (c >= d.Cin || mm >= Mtot) continue;
        float v = 0.0f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) v += part[((wv * CT + t) * 64 + l) * 4 + r];
        gx[((size_t)(mm / V) * d.Cin + c) * V + mm % V] = v;
    }
}

// ------------------------------------------------------------------------------ backward data, row tiles
// Small volumes with large kernels (the reference's second layer: (16,16,7,7,14) <- (16,64,3,3,10),
// 5^3): an input position meets an output position for 13 % of the taps only, and the tap-order
// kernel above, which drops 16-tap chunks without a live tap, still walks half of them.  Here a
// workgroup owns ONE ROW (b, z, y) of gx (W <= 16 positions): the taps (kd, kh) whose output plane
// and row exist are the same for the whole row, so exactly those are walked -- (OD/D)(OH/H) of
// the kD kH pairs, 18 % for that layer -- and only the kw range is masked per lane.  One MFMA
// step contracts four output channels at one tap (lane group g <-> channel o0 + g), so a lane's
// weights for the kW taps of (o, c, kd, kh) are consecutive floats (one 16-byte load for the
// first four) and its gy values are one row shifted by kw.  The NW waves take (pair, channel
// group) super-steps round-robin; the partial tiles are summed through LDS in a fixed order.
template <int CT, int NW, int NB>
__global__ __launch_bounds__(NW * 64) void conv3d_bwd_data_rowtap_kernel(const float *__restrict__ gy,
                                                                       const float *__restrict__ w,
                                                                       float *__restrict__ gx, Dims3 d)
{
    // NB batch items share a workgroup: the weights of a super-step are gathered once (20 useful
    // bytes of every 128-byte line they touch: the gathers, served by L2, are what the kernel
    // waits for) and multiplied with NB rows of gy
    __shared__ float part[NW * CT * 256];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k3 = d.kD * d.kH * d.kW;
    const int S = d.OD * d.OH * d.OW, V = d.D * d.H * d.W;
    const int row = blockIdx.x, y_ = row % d.H, q_ = row / d.H, z = q_ % d.D, b0 = (q_ / d.D) * NB;
    const int kd0 = max(0, z - d.OD + 1), kd1 = min(d.kD - 1, z);
    const int kh0 = max(0, y_ - d.OH + 1), kh1 = min(d.kH - 1, y_);
    const int nkh = kh1 - kh0 + 1, npair = (kd1 - kd0 + 1) * nkh;
    const int ogrp = (d.Cout + 3) >> 2, nsup = npair * ogrp;
    const size_t gstep = (size_t)d.Cout * S;               // one batch item of gy
    f32x4 acc[NB][CT];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[n][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool bv[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) bv[n] = b0 + n < d.B;
    for (int sp = wave; sp < nsup; sp += NW) {
        const int pair = sp / ogrp, o = (sp - pair * ogrp) * 4 + g;
        const int kd = kd0 + pair / nkh, kh = kh0 + pair % nkh;
        const bool ov = o < d.Cout;
        const int oc = ov ? o : d.Cout - 1;
        const float *gr = gy + (size_t)b0 * gstep + (size_t)oc * S +
                          (size_t)((z - kd) * d.OH + (y_ - kh)) * d.OW;
        size_t wb[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t)
            wb[t] = ((size_t)oc * d.Cin + min(t * 16 + j, d.Cin - 1)) * k3 + (size_t)((kd * d.kH + kh) * d.kW);
        int kw = 0;
        if (d.kW >= 4) {
            f32x4 w4[CT];
            float gv[NB][4];
#pragma unroll
            for (int t = 0; t < CT; ++t) w4[t] = *(const f4u *)(w + wb[t]);
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int xo = j - u;
                    const bool in = ov && bv[n] && (unsigned)xo < (unsigned)d.OW;
                    const float v_ = gr[in ? (size_t)n * gstep + xo : 0];
                    gv[n][u] = in ? v_ : 0.0f;
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int t = 0; t < CT; ++t) acc[n][t] = mma(w4[t][u], gv[n][u], acc[n][t]);
            kw = 4;
        }
        for (; kw < d.kW; ++kw) {
            const int xo = j - kw;
            float wv_[CT], gv[NB];
#pragma unroll
            for (int t = 0; t < CT; ++t) wv_[t] = w[wb[t] + kw];
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const bool in = ov && bv[n] && (unsigned)xo < (unsigned)d.OW;
                const float v_ = gr[in ? (size_t)n * gstep + xo : 0];
                gv[n] = in ? v_ : 0.0f;
            }
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[n][t] = mma(wv_[t], gv[n], acc[n][t]);
        }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (n) __syncthreads();                            // part is reused
        reduce_tiles<CT>(part, acc[n], wave, lane);
        if (!bv[n]) continue;                              // uniform
        for (int idx = tid; idx < CT * 256; idx += NW * 64) {
            const int t = idx >> 8, l = (idx >> 2) & 63, r = idx & 3;
            const int c = t * 16 + 4 * (l >> 4) + r, x_ = l & 15;
            if (c >= d.Cin || x_ >= d.W) continue;
            float v = 0.0f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) v += part[((wv * CT + t) * 64 + l) * 4 + r];
            gx[((size_t)(b0 + n) * d.Cin + c) * V + (size_t)(z * d.H + y_) * d.W + x_] = v;
        }
    }
}

// ------------------------------------------------------------------------------ backward data, wide
// Same decomposition as conv3d_fwd_wide_kernel for the input gradient of large tensors.
template <int CT>
__global__ __launch_bounds__(BLOCK) void conv3d_bwd_data_wide_kernel(const float *__restrict__ gy,
                                                                     const float *__restrict__ w,
                                                                     float *__restrict__ gx, Dims3 d,
                                                                     int K2, int Mtot)
{
    __shared__ int tab[WIDE_STEPS * 4];
    __shared__ float wl[WIDE_STEPS * CT * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int k3 = d.kD * d.kH * d.kW;
    const int S = d.OD * d.OH * d.OW, V = d.D * d.H * d.W;
    int mm[CPW], pd[CPW], ph[CPW], pw[CPW];
    size_t gybase[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        int m = ((int)blockIdx.x * WIDE_TILES + wave + c * WAVES) * 16 + j;
        mm[c] = m;
        if (m >= Mtot) m = Mtot - 1;
        const int b_ = m / V, pos = m % V;
        pw[c] = pos % d.W; ph[c] = (pos / d.W) % d.H; pd[c] = pos / (d.W * d.H);
        gybase[c] = (size_t)b_ * d.Cout * S;
    }
    f32x4 acc[CPW][CT];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < K2; kc += WIDE_STEPS * 4) {
        const int kn = min(WIDE_STEPS * 4, K2 - kc);
        const int steps = (kn + 3) >> 2;
        __syncthreads();
        if (tid < WIDE_STEPS * 4) {
            int e = -1;
            if (tid < kn) {
                const int kk = kc + tid;
                const int o = kk / k3, r = kk % k3;
                const int kw = r % d.kW, kh = (r / d.kW) % d.kH, kd = r / (d.kW * d.kH);
                e = (o << 24) | (kd << 16) | (kh << 8) | kw;
            }
            tab[tid] = e;
        }
        // operand (step, t) for lane l: W[o][c = 16 t + (l & 15)][tap], (o, tap) = kc + 4 step + (l >> 4)
        for (int e = tid; e < steps * CT * 64; e += BLOCK) {
            const int l = e & 63, q = e >> 6, t = q % CT, st = q / CT;
            const int c = t * 16 + (l & 15), k = 4 * st + (l >> 4), kk = kc + k;
            wl[e] = (k < kn && c < d.Cin) ? w[((size_t)(kk / k3) * d.Cin + c) * k3 + kk % k3] : 0.0f;
        }
        __syncthreads();
        for (int sb = 0; sb < steps; sb += UNR) {
            float gv[UNR][CPW];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int st = min(sb + u, steps - 1);
                const int e = tab[4 * st + g];
                const int o = (e >> 24) & 0x7f, kd = (e >> 16) & 0xff, kh = (e >> 8) & 0xff,
                          kw = e & 0xff;
#pragma unroll
                for (int c = 0; c < CPW; ++c) {
                    const int od = pd[c] - kd, oh = ph[c] - kh, ow = pw[c] - kw;
                    const bool in = e >= 0 && od >= 0 && od < d.OD && oh >= 0 && oh < d.OH &&
                                    ow >= 0 && ow < d.OW;
                    const size_t gi = in ? (size_t)o * S + (od * d.OH + oh) * d.OW + ow : 0;
                    const float v = gy[gybase[c] + gi];
                    gv[u][c] = in ? v : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (sb + u < steps) {
#pragma unroll
                    for (int t = 0; t < CT; ++t) {
                        const float a = wl[((sb + u) * CT + t) * 64 + lane];
#pragma unroll
                        for (int c = 0; c < CPW; ++c) acc[c][t] = mma(a, gv[u][c], acc[c][t]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        if (mm[c] >= Mtot) continue;
        const int b_ = mm[c] / V, pos = mm[c] % V;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = t * 16 + 4 * g + r;
                if (ch < d.Cin) gx[((size_t)b_ * d.Cin + ch) * V + pos] = acc[c][t][r];
            }
    }
}

// ------------------------------------------------------------------------------ backward data, rows
// Input gradient with the tiling of conv3d_fwd_rows_kernel: a wave's unit is 64 consecutive
// positions of one row of gx.  Units whose every tap is in range for every lane (all but the
// first/last kH-1 rows and the first and last segment of a row) take the unmasked path; border
// units evaluate the range per tap and lane (MASKED 2), or only the columns of the first and last
// chunk when the rows are in range (MASKED 1).  Tap offsets into gy are negative-going and kept
// as wrapped 32-bit byte offsets.
#define ROWSB_LOAD(dst, sbase)                                                         \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                  \
        const int k4_ = 4 * ((sbase) + u) + g;                                         \
        const unsigned a_ = lane0 + off[k4_];                                          \
        if (MASKED == 0) {                                                             \
            _Pragma("unroll") for (int c = 0; c < NC_; ++c)                            \
                dst[u][c] = *(const float *)(gb + a_ + 64u * c);                       \
        } else if (MASKED == 1) {       /* rows in range; columns: first and last chunk only */ \
            const int wq_ = w0 + j - (int)(tap[k4_] & 255u);                           \
            _Pragma("unroll") for (int c = 0; c < NC_; ++c) {                          \
                if (c == 0 || c == NC_ - 1) {                                          \
                    const bool ok_ = (unsigned)(wq_ + 16 * c) < (unsigned)d.OW;        \
                    const float v_ = *(const float *)(gb + (ok_ ? a_ + 64u * c : 0u)); \
                    dst[u][c] = ok_ ? v_ : 0.0f;                                       \
                } else {            /* a_ may be a wrapped negative: add in 32 bits */  \
                    dst[u][c] = *(const float *)(gb + (a_ + 64u * c));                 \
                }                                                                      \
            }                                                                          \
        } else {                                                                       \
            const unsigned tp_ = tap[k4_];                                             \
            const bool rv_ = (unsigned)(dz - (int)(tp_ >> 16)) < (unsigned)d.OD &&     \
                             (unsigned)(h - (int)((tp_ >> 8) & 255u)) < (unsigned)d.OH; \
            const int wq_ = w0 + j - (int)(tp_ & 255u);                                \
            _Pragma("unroll") for (int c = 0; c < NC_; ++c) {                          \
                const bool ok_ = rv_ && (unsigned)(wq_ + 16 * c) < (unsigned)d.OW;     \
                const float v_ = *(const float *)(gb + (ok_ ? a_ + 64u * c : 0u));     \
                dst[u][c] = ok_ ? v_ : 0.0f;                                           \
            }                                                                          \
        }                                                                              \
    }

template <int CT, int MASKED, int NC_>
__device__ __forceinline__ void bwd_rows_unit(const char *gb, const float *wl, const unsigned *off,
                                              const unsigned *tap, const Dims3 &d, int steps,
                                              unsigned lane0, int dz, int h, int w0, int lane,
                                              f32x4 (&acc)[CPW][CT])
{
    constexpr int OT = CT;                                // ROWS_MMA's tile count
    const int j = lane & 15, g = lane >> 4;
    float xa[UNR][CPW], xc[UNR][CPW];
    ROWSB_LOAD(xa, 0)
    for (int sb = 0; sb < steps; sb += 2 * UNR) {
        ROWSB_LOAD(xc, sb + UNR)
        __builtin_amdgcn_sched_barrier(0);
        ROWS_MMA(xa, sb)
        __builtin_amdgcn_sched_barrier(0);
        const int nx_ = sb + 2 * UNR < steps ? sb + 2 * UNR : sb;
        ROWSB_LOAD(xa, nx_)
        __builtin_amdgcn_sched_barrier(0);
        ROWS_MMA(xc, sb + UNR)
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int CT>
__global__ __launch_bounds__(BLOCK) void conv3d_bwd_data_rows_kernel(const float *__restrict__ gy,
                                                                     const float *__restrict__ w,
                                                                     float *__restrict__ gx, Dims3 d,
                                                                     int K2,
                                                                     const float *__restrict__ mask)
{
    extern __shared__ __align__(16) float smem[];
    const int steps = rows_steps(K2);
    float *wl = smem;
    unsigned *off = (unsigned *)(smem + (size_t)steps * CT * 64);
    unsigned *tap = off + steps * 4;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k3 = d.kD * d.kH * d.kW;
    const unsigned S = (unsigned)(d.OD * d.OH * d.OW), V = (unsigned)(d.D * d.H * d.W);

    for (int k = tid; k < steps * 4; k += BLOCK) {
        unsigned o_ = 0u, t_ = 0u;                        // padding: weight 0, in-range address
        if (k < K2) {
            const int o = k / k3, r = k % k3;
            const int kw = r % d.kW, kh = (r / d.kW) % d.kH, kd = r / (d.kW * d.kH);
            o_ = 4u * (unsigned)(((o * d.OD - kd) * d.OH - kh) * d.OW - kw);
            t_ = (unsigned)(kd << 16 | kh << 8 | kw);
        }
        off[k] = o_;
        tap[k] = t_;
    }
    for (int e = tid; e < steps * CT * 64; e += BLOCK) wl[e] = 0.0f;
    __syncthreads();
    // operand (step, t) for lane l: W[o][c = 16 t + (l & 15)][tap], (o, tap) = 4 step + (l >> 4)
    for (int e = tid; e < d.Cout * d.Cin * k3; e += BLOCK) {
        const int oc = e / k3, tv = e - oc * k3, o = oc / d.Cin, c = oc - o * d.Cin;
        const int kk = o * k3 + tv;
        wl[((kk >> 2) * CT + (c >> 4)) * 64 + (kk & 3) * 16 + (c & 15)] = w[e];
    }
    __syncthreads();

    const int segs = (d.W + ROWS_SEG - 1) / ROWS_SEG;
    const bool short_tail = rows_short_tail(d.W);
    const int units = d.B * d.D * d.H * segs;
    const int groups = (units + WAVES - 1) / WAVES;
    const int gpx = (groups + 7) >> 3;
    const int xcd = blockIdx.x & 7, nslot = gridDim.x >> 3;
    const int gend = min(groups, (xcd + 1) * gpx);
    const size_t gysample = (size_t)d.Cout * S, gxsample = (size_t)d.Cin * V;
    for (int grp = xcd * gpx + ((int)blockIdx.x >> 3); grp < gend; grp += nslot) {
        const int unit = grp * WAVES + wave;
        if (unit >= units) continue;                       // wave-uniform; no barrier below
        const int row = unit / segs, seg = unit - row * segs;
        const bool half = short_tail && seg == segs - 1;
        const int w0 = half ? d.W - ROWS_SEG / 2 : min(seg * ROWS_SEG, d.W - ROWS_SEG);
        const int h = row % d.H, q = row / d.H, dz = q % d.D, b_ = q / d.D;
        const char *gb = (const char *)(gy + (size_t)b_ * gysample);
        const unsigned lane0 = 4u * (unsigned)((dz * d.OH + h) * d.OW + w0 + j);   // wraps with off[]
        char *xb = (char *)(gx + (size_t)b_ * gxsample);
        const unsigned sp0 = 4u * ((unsigned)((dz * d.H + h) * d.W + w0 + j) + (unsigned)(4 * g) * V);
        // the first and last segment of a row (2 of 23 units at 1440 columns) leave the row only in
        // their first and last 16 columns as long as kW <= 17: one compare and two selects per tap
        // for those chunks instead of the full range test for all four
        const bool rows_in = dz >= d.kD - 1 && dz < d.OD && h >= d.kH - 1 && h < d.OH;
        auto run = [&](auto nc_tag) {
            constexpr int NC_ = decltype(nc_tag)::value;
            f32x4 acc[CPW][CT];
#pragma unroll
            for (int c = 0; c < NC_; ++c)
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bool cols_in = w0 >= d.kW - 1 && w0 + 16 * NC_ <= d.OW;
            if (rows_in && cols_in)
                bwd_rows_unit<CT, 0, NC_>(gb, wl, off, tap, d, steps, lane0, dz, h, w0, lane, acc);
            else if (rows_in && d.kW <= 17 && NC_ >= 2)
                bwd_rows_unit<CT, 1, NC_>(gb, wl, off, tap, d, steps, lane0, dz, h, w0, lane, acc);
            else
                bwd_rows_unit<CT, 2, NC_>(gb, wl, off, tap, d, steps, lane0, dz, h, w0, lane, acc);
            if (mask) {
                // ReLU backward of the layer BELOW fused into this store: mask is that layer's
                // output (same shape as gx), and its input gradient is gx where that was positive
                const char *mb = (const char *)(mask + (size_t)b_ * gxsample);
#pragma unroll
                for (int t = 0; t < CT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ch = t * 16 + 4 * g + r;
                        if (ch < d.Cin) {
                            const unsigned xo = sp0 + 4u * (unsigned)(t * 16 + r) * V;
                            float mv[CPW];
#pragma unroll
                            for (int c = 0; c < NC_; ++c) mv[c] = *(const float *)(mb + xo + 64u * c);
#pragma unroll
                            for (int c = 0; c < NC_; ++c)
                                *(float *)(xb + xo + 64u * c) = mv[c] > 0.0f ? acc[c][t][r] : 0.0f;
                        }
                    }
                return;
            }
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = t * 16 + 4 * g + r;
                    if (ch < d.Cin) {
                        const unsigned xo = sp0 + 4u * (unsigned)(t * 16 + r) * V;
#pragma unroll
                        for (int c = 0; c < NC_; ++c) *(float *)(xb + xo + 64u * c) = acc[c][t][r];
                    }
                }
        };
        if (half) run(std::integral_constant<int, CPW / 2>{});
        else run(std::integral_constant<int, CPW>{});
    }
}

// gx = mask > 0 ? gx : 0 -- the ReLU backward for the kernels that do not fuse it
__global__ __launch_bounds__(BLOCK) void relu_mask_kernel(float *__restrict__ gx,
                                                          const float *__restrict__ mask,
                                                          long long n)
{
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n;
         i += (long long)gridDim.x * BLOCK)
        gx[i] = mask[i] > 0.0f ? gx[i] : 0.0f;
}

// ------------------------------------------------------------------------------ backward w, b
// grid = (ceil(tiles / WAVES), splits); tile = (o-tile, k-tile) with k in [0, K] (k == K is the
// bias column of ones).  partial is [splits][nent] float64, nent = Cout*K + Cout.
__global__ __launch_bounds__(BLOCK) void conv3d_bwd_wb_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ gy,
                                                              double *__restrict__ partial,
                                                              Dims3 d, int K, int Ntot, int ktiles,
                                                              int tiles)
{
    __shared__ int xbase[NCHUNK];     // window origin in x of output n
    __shared__ int gbase[NCHUNK];     // index of gy[b, 0, sp]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int S = d.OD * d.OH * d.OW;
    const int n0 = blockIdx.y * NCHUNK;
    const int nn = min(NCHUNK, Ntot - n0);
    const int nn4 = (nn + 3) & ~3;
    for (int t = tid; t < nn4; t += BLOCK) {
        int xb = 0, gb = 0;
        if (t < nn) {
            const int n = n0 + t;
            const int b_ = n / S, sp = n % S;
            const int ow = sp % d.OW, oh = (sp / d.OW) % d.OH, od = sp / (d.OW * d.OH);
            xb = ((b_ * d.Cin * d.D + od) * d.H + oh) * d.W + ow;
            gb = b_ * d.Cout * S + sp;
        }
        xbase[t] = xb;
        gbase[t] = gb;
    }
    __syncthreads();
    const int tile = blockIdx.x * WAVES + wave;
    if (tile >= tiles) return;
    const int ot = tile / ktiles, kt = tile % ktiles;
    const int o_a = ot * 16 + j;                       // A operand row (output channel)
    const int k_b = kt * 16 + j;                       // B operand column (tap)
    const int koff = k_b < K ? tap_offset(d, k_b) : 0;
    const bool a_ok = o_a < d.Cout;
    const size_t a_off = (size_t)o_a * S;

    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int tb = 0; tb < nn4; tb += 4 * UNR) {
        float gv[UNR], xv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int t = tb + 4 * u + g;
            const bool tv = t < nn;
            const int tt = tv ? t : 0;
            gv[u] = (tv && a_ok) ? gy[gbase[tt] + a_off] : 0.0f;
            const float xl = x[xbase[tt] + koff];           // koff = 0 for the bias column
            xv[u] = !tv ? 0.0f : (k_b < K ? xl : (k_b == K ? 1.0f : 0.0f));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc = mma(gv[u], xv[u], acc);
    }
    const int nw = d.Cout * K, nent = nw + d.Cout;
    double *dst = partial + (size_t)blockIdx.y * nent;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = ot * 16 + 4 * g + r;
        if (o < d.Cout) {
            if (k_b < K) dst[(size_t)o * K + k_b] = (double)acc[r];
            else if (k_b == K) dst[nw + o] = (double)acc[r];
        }
    }
}

// ------------------------------------------------------------------------------ backward w, b, wide
// For large outputs: persistent workgroups walk chunks of CH outputs.  The gy chunk is staged in
// LDS once (coalesced) and serves every (channel tile, tap tile) pair; each wave owns the tap
// tiles wave, wave+4, ... for ALL channel tiles and keeps their accumulators in registers across
// chunks, so one x gather feeds OT MFMAs (the kernel above gathers both operands per MFMA and
// re-gathers them for every tile pair).  fp32 accumulators are added into the workgroup's float64
// slab every FLUSH chunks (<= 1024 products per accumulator, as elsewhere) -- only the owning lane
// touches an entry, so there are no atomics; the slabs are summed by conv3d_reduce_kernel.
template <int OT, int KTW>
__global__ __launch_bounds__(BLOCK) void conv3d_bwd_wb_wide_kernel(const float *__restrict__ x,
                                                                   const float *__restrict__ gy,
                                                                   double *__restrict__ slabs,
                                                                   Dims3 d, int K, int Ntot,
                                                                   int ktiles)
{
    constexpr int CH = OT == 4 ? 128 : 256;
    constexpr int GP = CH + 2;                    // row pitch: conflict-free (o, n) operand reads
    constexpr int FLUSH = 1024 / CH;
    constexpr int UNW = 8;                        // x gathers in flight per wave
    __shared__ int xbase[CH];
    __shared__ int gbase[CH];
    __shared__ float gl[OT * 16 * GP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int S = d.OD * d.OH * d.OW;
    const int nw = d.Cout * K, nent = nw + d.Cout;
    double *slab = slabs + (size_t)blockIdx.x * nent;

    int koff[KTW];
#pragma unroll
    for (int q = 0; q < KTW; ++q) {
        const int k_b = (wave + WAVES * q) * 16 + j;
        koff[q] = k_b < K ? tap_offset(d, k_b) : 0;
    }
    f32x4 acc[KTW][OT];
#pragma unroll
    for (int q = 0; q < KTW; ++q)
#pragma unroll
        for (int t = 0; t < OT; ++t) acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = (Ntot + CH - 1) / CH;
    int pending = 0;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int n0 = chunk * CH;
        const int nn = min(CH, Ntot - n0);
        __syncthreads();
        if (tid < CH) {
            int xb = 0, gb = 0;
            if (tid < nn) {
                const int n = n0 + tid;
                const int b_ = n / S, sp = n % S;
                const int ow = sp % d.OW, oh = (sp / d.OW) % d.OH, od = sp / (d.OW * d.OH);
                xb = ((b_ * d.Cin * d.D + od) * d.H + oh) * d.W + ow;
                gb = b_ * d.Cout * S + sp;
            }
            xbase[tid] = xb;
            gbase[tid] = gb;
        }
        __syncthreads();
        for (int e = tid; e < OT * 16 * CH; e += BLOCK) {
            const int o = e / CH, t = e % CH;
            gl[o * GP + t] = (t < nn && o < d.Cout) ? gy[gbase[t] + (size_t)o * S] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KTW; ++q) {
            const int kt = wave + WAVES * q;
            if (kt >= ktiles) continue;                       // uniform per wave
            const int k_b = kt * 16 + j;
            for (int tb = 0; tb < CH; tb += 4 * UNW) {
                float xv[UNW];
#pragma unroll
                for (int u = 0; u < UNW; ++u) {
                    const float xl = x[xbase[tb + 4 * u + g] + koff[q]];
                    xv[u] = k_b < K ? xl : (k_b == K ? 1.0f : 0.0f);   // gy is zero past the end
                }
#pragma unroll
                for (int u = 0; u < UNW; ++u)
#pragma unroll
                    for (int t = 0; t < OT; ++t)
                        acc[q][t] = mma(gl[(t * 16 + j) * GP + tb + 4 * u + g], xv[u], acc[q][t]);
            }
        }
        if (++pending == FLUSH || chunk + (int)gridDim.x >= nchunks) {
            pending = 0;
#pragma unroll
            for (int q = 0; q < KTW; ++q) {
                const int k_b = (wave + WAVES * q) * 16 + j;
#pragma unroll
                for (int t = 0; t < OT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = t * 16 + 4 * g + r;
                        if (o < d.Cout) {
                            if (k_b < K) slab[(size_t)o * K + k_b] += (double)acc[q][t][r];
                            else if (k_b == K) slab[nw + o] += (double)acc[q][t][r];
                        }
                        acc[q][t][r] = 0.0f;
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------ backward w, b, rows
// Weight/bias gradient of the large grids with both operands staged in LDS.  In the wide kernel
// above every B operand is a 16-tap x 4-position gather from global memory (sixteen scattered
// 16-byte pieces per wave instruction, each x element fetched kH*kW times per workgroup); here a
// workgroup's unit is 64 consecutive outputs of one output row, whose input window
// (Cin*kD*kH rows of 68 floats: 64 + halo) and gy rows (Cout rows of 64 floats) are copied
// into LDS by 16-byte LDS-direct loads (global_load_lds_dwordx4: the LDS image is filled
// linearly, one KiB per wave instruction, and each lane's SOURCE address picks the piece of the
// row that belongs there, pitch padding included), and the MFMA operands are LDS reads at
// immediate offsets of per-lane bases.  (Row-by-row dword copies of the same tile were
// instruction-bound: 224 wave instructions per unit, 1.26 ms of a 2.5 ms pass.)
// A workgroup walks DOWN a 64-column stripe, one output row per unit, and the window is a ring
// over kH: input row ih lives in slot ih % kH, so a unit fetches only the Cin*kD rows of its
// bottom input row (the others are still there from the units before); only the first unit of a
// workgroup's run or of a stripe fills all kH slots.  That takes the fetches of this pass from
// kH times x to once x; the time moved little (-4 %: per-workgroup clocks show the three resident
// workgroups of a CU finishing at 1.46 / 1.55 / 1.62 ms whether a unit fetches 35 KB or 17 KB,
// and four resident workgroups at 104 registers took the same total).
// Pitches (72 and 68 words; slot stride = 16 mod 64) keep the 16 taps x 4 positions of a B read
// and the 16 channels x 4 positions of an A read on distinct banks.  Accumulators stay in
// registers across units and go to the workgroup's float64 slab every WBR_FLUSH units (1024
// outputs), as in the wide kernel.
constexpr int WBR_SEG = 64;        // outputs per unit
constexpr int WBR_XP = 72;         // x row pitch in LDS (words): 18 pieces of 16 bytes, = 8 mod 64
constexpr int WBR_XW = 68;         // x columns fetched per row (17 pieces): kW - 1 <= 4
constexpr int WBR_GP = 68;         // gy row pitch (words): 17 pieces, = 4 mod 64
constexpr int WBR_FLUSH = 16;
constexpr int WBR_MAXS = 5;        // x pieces / 256 per thread and slot  (Cin*kD <= 71 rows)
constexpr int WBR_MAXG = 5;        // gy pieces / 256            (Cout <= 64)

// float64 sums in registers (80 of them at most) instead of the slab
__host__ __device__ constexpr bool wbr_regsum(int ot, int ktw) { return ot * ktw <= 10; }

// words between two slots of the ring: Cin*kD rows, padded to 16 mod 64 banks
__host__ __device__ inline int wbr_slot_stride(int cd)
{
    const int w = cd * WBR_XP;
    return w + ((16 - w % 64) + 64) % 64;
}

__host__ __device__ inline size_t wbr_lds_bytes(int cd, int kH, int ot)
{
    return ((size_t)kH * wbr_slot_stride(cd) + (size_t)ot * 16 * WBR_GP + 2 * WBR_XP) * sizeof(float);
}

#define WM_GLDS16(src, dst)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),    \
                                     (__attribute__((address_space(3))) void *)(dst), 16, 0, 0)

template <int OT, int KTW>
__global__ __launch_bounds__(BLOCK) void conv3d_bwd_wb_rows_kernel(const float *__restrict__ x,
                                                                   const float *__restrict__ gy,
                                                                   double *__restrict__ partial,
                                                                   Dims3 d, int K, int ktiles)
{
    extern __shared__ __align__(16) float smem[];
    const int CD = d.Cin * d.kD, SS = wbr_slot_stride(CD);
    // [kH slots][CD][WBR_XP] x | [OT*16][WBR_GP] gy | [WBR_XP] ones | [WBR_XP] zeros
    const int goff = d.kH * SS, ooff = goff + OT * 16 * WBR_GP, zoff = ooff + WBR_XP;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = d.OD * d.OH * d.OW;
    // the slab is [Cout][K + 1] float64: column K is the bias gradient
    char *slab = (char *)(partial + (size_t)blockIdx.x * d.Cout * (K + 1));

    for (int e = tid; e < OT * 16 * WBR_GP; e += BLOCK) smem[goff + e] = 0.0f;   // rows >= Cout stay zero
    for (int e = tid; e < WBR_XP; e += BLOCK) { smem[ooff + e] = 1.0f; smem[zoff + e] = 0.0f; }

    // this lane's B operand of tap tile q: tap k = (wave + WAVES q) 16 + j = ((c kD + kd) kH + kh) kW + kw
    // is at word  slot(oh + kh) SS + (c kD + kd) WBR_XP + kw  of the ring; kh = -1: a fixed row
    int kbase[KTW], khq[KTW];
#pragma unroll
    for (int q = 0; q < KTW; ++q) {
        const int k = (wave + WAVES * q) * 16 + j;
        int o_ = zoff, kh_ = -1;
        if (k < K) {
            const int kw = k % d.kW, r_ = k / d.kW;
            kh_ = r_ % d.kH;
            o_ = (r_ / d.kH) * WBR_XP + kw;
        } else if (k == K) {
            o_ = ooff;
        }
        kbase[q] = o_ + g;
        khq[q] = kh_;
    }
    const int abase = goff + j * WBR_GP + g;

    // 16-byte pieces of the LDS image this thread fills, fixed for the kernel: piece P = tid +
    // 256 i of a slot is row P / 18, columns 4 (P % 18).. (the 18th piece of a row is pitch
    // padding and repeats the 17th); of the gy part row P / 17, columns 4 (P % 17).. (the 17th is
    // padding).  Element offsets from the unit's first column; -1 = nothing to fill.
    const int nxs = (CD * 18 + BLOCK - 1) / BLOCK, ngs = (d.Cout * 17 + BLOCK - 1) / BLOCK;
    int xsrc_off[WBR_MAXS], gsrc_off[WBR_MAXG];
#pragma unroll
    for (int i = 0; i < WBR_MAXS; ++i) {
        const int P = tid + BLOCK * i, r_ = P / 18, pc = P - 18 * r_;
        int o_ = -1;
        if (r_ < CD) {
            const int kd = r_ % d.kD, c = r_ / d.kD;
            o_ = (c * d.D + kd) * d.H * d.W + 4 * min(pc, 16);
        }
        xsrc_off[i] = o_;
    }
#pragma unroll
    for (int i = 0; i < WBR_MAXG; ++i) {
        const int P = tid + BLOCK * i, o = P / 17, pc = P - 17 * o;
        gsrc_off[i] = o < d.Cout ? o * S + 4 * min(pc, 15) : -1;
    }

    // byte offset in the slab of this lane's entry (o = 4 g, tap k_b) of tile q
    unsigned voff[KTW];
#pragma unroll
    for (int q = 0; q < KTW; ++q)
        voff[q] = 8u * (unsigned)((wave + WAVES * q) * 16 + j + 4 * g * (K + 1));

    // units in (b, od, segment, oh) order: consecutive units are consecutive output rows of one
    // 64-column stripe.  An XCD takes a contiguous eighth, each of its workgroups a contiguous run.
    const int segs = (d.OW + WBR_SEG - 1) / WBR_SEG;
    const int units = d.B * d.OD * d.OH * segs;
    const int upx = (units + 7) >> 3;
    const int xcd = blockIdx.x & 7, nslot = gridDim.x >> 3;
    const int xfirst = xcd * upx, xn = max(0, min(units, xfirst + upx) - xfirst);
    const int slot_ = (int)blockIdx.x >> 3;
    const int ufirst = xfirst + (int)((long long)xn * slot_ / nslot);
    const int uend = xfirst + (int)((long long)xn * (slot_ + 1) / nslot);
    const size_t xsample = (size_t)d.Cin * d.D * d.H * d.W, gsample = (size_t)d.Cout * S;
    int stripe = ufirst / d.OH, oh = ufirst - stripe * d.OH, ohm = oh % d.kH, seg = 0, od = 0, b_ = 0;
    // float64 sums of the workgroup: in registers when they fit beside the accumulators (two
    // workgroups per CU then, but no read-modify-write traffic: 32 -> 32 channels 3x3 1.87 ->
    // 1.77 ms), else in the slab
    constexpr bool REGSUM = wbr_regsum(OT, KTW);
    double dsum[REGSUM ? KTW : 1][REGSUM ? OT : 1][4];
    if constexpr (REGSUM) {
#pragma unroll
        for (int q = 0; q < KTW; ++q)
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dsum[q][t][r] = 0.0;
    }
    for (int ubase = ufirst; ubase < uend; ubase += WBR_FLUSH) {
    f32x4 acc[KTW][OT];
#pragma unroll
    for (int q = 0; q < KTW; ++q)
#pragma unroll
        for (int t = 0; t < OT; ++t) acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ustop = min(uend, ubase + WBR_FLUSH);
    for (int unit = ubase; unit < ustop; ++unit) {
        if (unit != ufirst) {                              // next row of the stripe, or next stripe
            ++oh;
            if (++ohm == d.kH) ohm = 0;
            if (oh == d.OH) { oh = 0; ohm = 0; ++stripe; }
        }
        // the shifted last segment of a row (an extra barrier per unit) comes FIRST among the
        // stripes of an (image, plane): the workgroups that were dispatched first also get the
        // larger share of the MFMA pipe (measured finish times of the three resident rounds:
        // 1.46 / 1.55 / 1.62 ms), so the dearer units go to them
        if (oh == 0 || unit == ufirst) {
            const int sq = stripe % segs, q_ = stripe / segs;
            seg = sq == 0 ? segs - 1 : sq - 1;
            od = q_ % d.OD;
            b_ = q_ / d.OD;
        }
        // the last segment of a row is moved left to end at the row's end (the outputs it then
        // shares with the segment before are zeroed in the gy tile once it has landed), and its
        // 68-column input window a little further so that it stays inside the row: B reads then
        // start delta words in
        const int ow0 = min(seg * WBR_SEG, d.OW - WBR_SEG);
        const int xs0 = min(ow0, d.W - WBR_XW), delta = ow0 - xs0;
        const float *xsrc = x + (size_t)b_ * xsample + (size_t)((od * d.H + oh) * d.W + xs0);
        const float *gsrc = gy + (size_t)b_ * gsample + (size_t)((od * d.OH + oh) * d.OW + ow0);
        __syncthreads();                                   // every wave is past its reads of the tile
        // input rows oh .. oh+kH-1: all of them at the top of a run or stripe, else the last
        for (int kh = (unit == ufirst || oh == 0) ? 0 : d.kH - 1; kh < d.kH; ++kh) {
            int sl = ohm + kh;
            if (sl >= d.kH) sl -= d.kH;
            const float *xr = xsrc + (size_t)kh * d.W;
            float *dst = smem + sl * SS + 64 * wave * 4;
#pragma unroll
            for (int i = 0; i < WBR_MAXS; ++i)
                if (i < nxs) {
                    if (xsrc_off[i] >= 0) WM_GLDS16(xr + xsrc_off[i], dst + BLOCK * i * 4);
                }
        }
#pragma unroll
        for (int i = 0; i < WBR_MAXG; ++i)
            if (i < ngs) {
                if (gsrc_off[i] >= 0)
                    WM_GLDS16(gsrc + gsrc_off[i], smem + goff + (BLOCK * i + 64 * wave) * 4);
            }
        __syncthreads();                                   // vmcnt(0) + barrier: the tile has landed
        if (ow0 != seg * WBR_SEG) {                        // one stripe in `segs`
            const int m = seg * WBR_SEG - ow0;
            for (int o = tid >> 2; o < d.Cout; o += BLOCK / 4)
                for (int n = tid & 3; n < m; n += 4) smem[goff + o * WBR_GP + n] = 0.0f;
            __syncthreads();
        }
        // operands of step s+1 are read while the MFMAs of step s run; the barriers keep hipcc
        // from hoisting all 16 steps' LDS reads to the top (it did: 244 VGPRs)
#define WBR_OPS(a_, b_, s_)                                                                  \
    _Pragma("unroll") for (int t = 0; t < OT; ++t) a_[t] = smem[abase + t * 16 * WBR_GP + 4 * (s_)]; \
    _Pragma("unroll") for (int q = 0; q < KTW; ++q) b_[q] = smem[kofd[q] + 4 * (s_)];
#define WBR_MMA(a_, b_)                                                                      \
    _Pragma("unroll") for (int q = 0; q < KTW; ++q)                                          \
        _Pragma("unroll") for (int t = 0; t < OT; ++t) acc[q][t] = mma(a_[t], b_[q], acc[q][t]);
        float a0[OT], b0[KTW], a1[OT], b1[KTW];
        int kofd[KTW];
#pragma unroll
        for (int q = 0; q < KTW; ++q) {
            int sl = khq[q] + ohm;
            if (sl >= d.kH) sl -= d.kH;
            kofd[q] = kbase[q] + delta + (khq[q] >= 0 ? sl * SS : 0);
        }
        WBR_OPS(a0, b0, 0)
#pragma unroll
        for (int s_ = 0; s_ < WBR_SEG / 4; s_ += 2) {
            WBR_OPS(a1, b1, s_ + 1)
            __builtin_amdgcn_sched_barrier(0);
            WBR_MMA(a0, b0)
            __builtin_amdgcn_sched_barrier(0);
            if (s_ + 2 < WBR_SEG / 4) { WBR_OPS(a0, b0, s_ + 2) }
            __builtin_amdgcn_sched_barrier(0);
            WBR_MMA(a1, b1)
            __builtin_amdgcn_sched_barrier(0);
        }
#undef WBR_OPS
#undef WBR_MMA
    }
    if constexpr (!REGSUM) {
    // WBR_FLUSH units (1024 outputs) of fp32 accumulation go to the float64 slab
#pragma unroll
    for (int q = 0; q < KTW; ++q) {
        const bool kin = (wave + WAVES * q) * 16 + j <= K;
        // opaque copy: otherwise hipcc hoists all KTW*OT*4 64-bit entry addresses out of the
        // unit loops and keeps them in VGPRs (80 registers for 32 -> 32 channels 3x3)
        unsigned vo = voff[q];
        asm volatile("" : "+v"(vo));
#pragma unroll
        for (int t = 0; t < OT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                char *sb = slab + (size_t)(t * 16 + r) * (K + 1) * sizeof(double);   // wave-uniform
                if (kin && t * 16 + 4 * g + r < d.Cout) *(double *)(sb + vo) += (double)acc[q][t][r];
            }
            __builtin_amdgcn_sched_barrier(0);             // four read-modify-writes in flight, not all
        }
    }
    } else {
#pragma unroll
        for (int q = 0; q < KTW; ++q)
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dsum[q][t][r] += (double)acc[q][t][r];
    }
    }
    if constexpr (REGSUM) {
#pragma unroll
        for (int q = 0; q < KTW; ++q) {
            const bool kin = (wave + WAVES * q) * 16 + j <= K;
#pragma unroll
            for (int t = 0; t < OT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    char *sb = slab + (size_t)(t * 16 + r) * (K + 1) * sizeof(double);
                    if (kin && t * 16 + 4 * g + r < d.Cout) *(double *)(sb + voff[q]) = dsum[q][t][r];
                }
        }
    }
}

// ------------------------------------------------------------------------------ backward w, b, rows, staged
// The row kernel above, re-cut after cycle stamps of its 32 -> 32 channels 3x3 run (tools/probe/
// wb_stamps.py: shader clock 2.07 GHz under this load, 12.4 k clk per unit and workgroup, of which
// 3.6 k between the two barriers, two workgroups per CU; 160 MFMAs x 32 clk = 5.1 k clk of issue
// per unit and wave, a ninth of them on the bias tile and on an empty tile).  The time between
// the barriers was not memory latency: while the other resident workgroup's wave streams MFMAs
// on the same SIMD, every VALU instruction of this wave waits for an issue slot behind a 32-clk
// MFMA, so the ~150 address and select instructions per unit cost as much as a third of the MFMAs.
//  * the next unit's rows (one input row per (channel, kd) and the gy tile) are fetched into
//    REGISTERS while this unit's MFMAs run (scalar base + 32-bit lane offset: no address
//    arithmetic on the VALU) and written to LDS between the barriers (global_load ... lds cannot
//    run ahead here: hipcc orders every later LDS read behind it with vmcnt(0));
//  * the LDS address of every B operand for each of the kH phases of the row ring is computed
//    once into an LDS table; a unit reads its KTW addresses back (LDS issue, not VALU);
//  * taps >= K and channels >= Cout read whatever is there: a column or row of an MFMA product
//    depends on nothing but its own operand, and those are never stored;
//  * the bias gradient leaves the MFMA: its column of ones cost a whole tap tile (K + 1 = 289 ->
//    19 tiles); the gy pieces are summed on their way from the stage registers into LDS;
//  * the 4 waves can be split WOS ways over the channel tiles, the rest over the tap tiles
//    (32 -> 32, 3x3: 2 x 2, nine tap tiles and one channel tile per wave: 9 MFMAs per step and
//    no empty tile, where 1 x 4 ran 10 with two of 20 tiles idle).
// Sums are float64 in registers (OTW * KTW <= 10), written once to the workgroup's slab.
#ifdef WM_WB_STAMPS
// debug build only (tools/probe/wb_stamps.py): per workgroup, the shader cycles spent in
// top-of-stripe fills, and the wall / shader clocks at both ends
__device__ unsigned long long wm_wb_stamps[1024 * 8 + 1024];
#define WB_CLK() __builtin_readcyclecounter()
#endif

typedef float f32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned long long WBS_CLOCK_MAGIC = 0x574d434c4f434b31ull;   // "WMCLOCK1"
constexpr int WBS_NX = 3;          // staged 16-byte pieces per thread: an x row set, Cin*kD*17 <= 768
constexpr int WBS_NG = 2;          // the gy tile, 32 channels x 16 pieces

__host__ __device__ inline size_t wbs_lds_bytes(int cd, int kH, int ot, int ktw)
{
    return ((size_t)(kH + 1) * wbr_slot_stride(cd) + 2 * (size_t)ot * 16 * WBR_GP) * sizeof(float) +
           (size_t)(kH + 1) * ktw * BLOCK * sizeof(unsigned short);
}

template <int OTW, int KTW, int WOS>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3d_bwd_wb_staged_kernel(const float *__restrict__ x,
                                                                     const float *__restrict__ gy,
                                                                     double *__restrict__ partial,
                                                                     Dims3 d, int K)
{
    constexpr int OT = OTW * WOS;
    constexpr int GSZ = OT * 16 * WBR_GP;                  // words of one gy buffer
    extern __shared__ __align__(16) float smem[];
    const int CD = d.Cin * d.kD, SS = wbr_slot_stride(CD), NS = d.kH + 1;
    // [kH + 1 slots][CD][WBR_XP] x | 2 x [OT*16][WBR_GP] gy | [kH + 1][KTW][BLOCK] B addresses (u16)
    const int goff = NS * SS, toff = goff + 2 * GSZ;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = wave % WOS, tg = wave / WOS;            // channel-tile group, tap-tile group
    const int S = d.OD * d.OH * d.OW;
    double *slab = partial + (size_t)blockIdx.x * d.Cout * (K + 1);   // [Cout][K + 1], column K = bias

    for (int e = tid; e < toff; e += BLOCK) smem[e] = 0.0f;
    // B operand of tap tile q in phase p: tap k = (tg KTW + q) 16 + j = ((c kD + kd) kH + kh) kW + kw
    // is at word ((p + kh) mod (kH + 1)) SS + (c kD + kd) WBR_XP + kw + g
    unsigned short *tab = (unsigned short *)(smem + toff) + tid;
#pragma unroll
    for (int q = 0; q < KTW; ++q) {
        int k = (tg * KTW + q) * 16 + j;
        if (k >= K) k = 0;
        const int kw = k % d.kW, r_ = k / d.kW, kh = r_ % d.kH;
        const int w_ = (r_ / d.kH) * WBR_XP + kw + g;
        for (int p_ = 0; p_ < NS; ++p_)
            tab[(p_ * KTW + q) * BLOCK] = (unsigned short)(((kh + p_) % NS) * SS + w_);
    }
    const int abase = goff + (og * OTW * 16 + j) * WBR_GP + g;

    // staged pieces of this thread: x piece P = tid + 256 i is row P / 17, columns 4 (P % 17)..;
    // gy piece is channel P / 16, columns 4 (tid % 16)..  A piece past the end goes to the pad
    // columns of a row (never read) and loads from offset 0.
    unsigned xsrc[WBS_NX], gsrc[WBS_NG];
    int xdst[WBS_NX], gdst[WBS_NG];
#pragma unroll
    for (int i = 0; i < WBS_NX; ++i) {
        const int P = tid + BLOCK * i, r_ = P / 17, pc = P - 17 * r_;
        xsrc[i] = 0;
        xdst[i] = (tid % CD) * WBR_XP + WBR_XW;
        if (r_ < CD) {
            xsrc[i] = 4u * (unsigned)(((r_ / d.kD) * d.D + r_ % d.kD) * d.H * d.W + 4 * pc);
            xdst[i] = r_ * WBR_XP + 4 * pc;
        }
    }
    bool gval[WBS_NG];
#pragma unroll
    for (int i = 0; i < WBS_NG; ++i) {
        const int o = (tid + BLOCK * i) >> 4;
        gval[i] = o < d.Cout;
        gsrc[i] = gval[i] ? 4u * (unsigned)(o * S + 4 * (tid & 15)) : 0u;
        gdst[i] = goff + (gval[i] ? o * WBR_GP + 4 * (tid & 15) : (tid % d.Cout) * WBR_GP + WBR_SEG);
    }
    const int gcol = 4 * (tid & 15);

    // units in (b, od, segment, oh) order, an XCD a contiguous eighth, a workgroup a contiguous run
    const int segs = (d.OW + WBR_SEG - 1) / WBR_SEG;
    const int units = d.B * d.OD * d.OH * segs;
    const int upx = (units + 7) >> 3;
    const int xcd = blockIdx.x & 7, nslot = gridDim.x >> 3;
    const int xfirst = xcd * upx, xn = max(0, min(units, xfirst + upx) - xfirst);
    const int slot_ = (int)blockIdx.x >> 3;
    const int ufirst = xfirst + (int)((long long)xn * slot_ / nslot);
    const int uend = xfirst + (int)((long long)xn * (slot_ + 1) / nslot);
    const size_t xsample = (size_t)d.Cin * d.D * d.H * d.W, gsample = (size_t)d.Cout * S;

    // where a unit's rows are.  sq counts the stripes of an (image, plane); the shifted last
    // segment of a row comes first among them, as in the row kernel.  ow0: first output column
    // (the last segment is moved left to end at the row's end; the outputs it then shares with the
    // segment before are zeroed on their way into the gy tile: mz of them), xs0: first input column.
    // All of it is wave-uniform and branch-free: it is scheduled between the MFMAs of a step.
    struct Cur { int oh, sq, od, b, ow0, xs0, mz; };
    auto derive = [&](Cur &c) {
        const int seg = c.sq == 0 ? segs - 1 : c.sq - 1;
        c.ow0 = min(seg * WBR_SEG, d.OW - WBR_SEG);
        c.xs0 = min(c.ow0, d.W - WBR_XW);
        c.mz = seg * WBR_SEG - c.ow0;
    };
    auto locate = [&](Cur &c, int unit) {
        const int stripe = unit / d.OH, q_ = stripe / segs;
        c.oh = unit - stripe * d.OH;
        c.sq = stripe - q_ * segs;
        c.od = q_ % d.OD;
        c.b = q_ / d.OD;
        derive(c);
    };
    auto advance = [&](Cur &c, int by) {                   // by = 0: stay
        c.oh += by;
        const int w0 = c.oh == d.OH;
        c.oh = w0 ? 0 : c.oh;
        c.sq += w0;
        const int w1 = c.sq == segs;
        c.sq = w1 ? 0 : c.sq;
        c.od += w1;
        const int w2 = c.od == d.OD;
        c.od = w2 ? 0 : c.od;
        c.b += w2;
        derive(c);
    };
    f32x4 sx[WBS_NX], sg[WBS_NG];
    auto fetch = [&](const Cur &c) {                       // the unit's LAST input row and its gy tile
        const char *xr = (const char *)(x + (size_t)c.b * xsample + (size_t)((c.od * d.H + c.oh + d.kH - 1) * d.W + c.xs0));
        const char *gr = (const char *)(gy + (size_t)c.b * gsample + (size_t)((c.od * d.OH + c.oh) * d.OW + c.ow0));
#pragma unroll
        for (int i = 0; i < WBS_NX; ++i) sx[i] = *(const f32x4a4 *)(xr + xsrc[i]);
#pragma unroll
        for (int i = 0; i < WBS_NG; ++i) sg[i] = *(const f32x4a4 *)(gr + gsrc[i]);
    };
    f32x2 bacc[WBS_NG];
#pragma unroll
    for (int i = 0; i < WBS_NG; ++i) bacc[i] = (f32x2){0.f, 0.f};
    // stage registers -> ring slot and gy buffer; the tile counts `live` (0 or 1) times in the bias
    auto put = [&](int slot, int par, int mz, float live) {
#pragma unroll
        for (int i = 0; i < WBS_NX; ++i) wm_store4(smem + slot * SS + xdst[i], sx[i]);
#pragma unroll
        for (int i = 0; i < WBS_NG; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sg[i][e] = gcol + e < mz ? 0.0f : sg[i][e];
            wm_store4(smem + par * GSZ + gdst[i], sg[i]);
            bacc[i] += live * (f32x2){sg[i][0], sg[i][1]};
            bacc[i] += live * (f32x2){sg[i][2], sg[i][3]};
        }
    };

    // workgroup 0 leaves its shader-cycle and 100 MHz wall-clock counts behind the slabs: their
    // ratio is the clock the chip ran this kernel at (bench.py prints it beside the roofline)
    const unsigned long long clk0 = __builtin_readcyclecounter(), wall0 = wall_clock64();
    double dsum[KTW][OTW][4];
#pragma unroll
    for (int q = 0; q < KTW; ++q)
#pragma unroll
        for (int t = 0; t < OTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dsum[q][t][r] = 0.0;
    double bsum[WBS_NG];
#pragma unroll
    for (int i = 0; i < WBS_NG; ++i) bsum[i] = 0.0;
    f32x4 acc[KTW][OTW];
#pragma unroll
    for (int q = 0; q < KTW; ++q)
#pragma unroll
        for (int t = 0; t < OTW; ++t) acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef WM_WB_STAMPS
    unsigned long long t_top = 0;
    const unsigned long long c_begin = WB_CLK(), w_begin = wall_clock64();
#endif

    if (ufirst < uend) {
        // tile(ufirst): its last row and gy tile; then tile(ufirst + 1) into the stage registers.
        // cf: the unit whose rows were fetched last; mz1, dl1, top1: of the unit after this one
        Cur cf;
        locate(cf, ufirst);
        int p = 0, par = 0;
        fetch(cf);
        put(d.kH - 1, 0, cf.mz, 1.0f);
        int kofd[KTW];
#pragma unroll
        for (int q = 0; q < KTW; ++q) kofd[q] = (int)tab[q * BLOCK] + (cf.ow0 - cf.xs0);
        advance(cf, ufirst + 1 < uend);
        fetch(cf);
        int mz1 = cf.mz, dl1 = cf.ow0 - cf.xs0;
        bool top1 = cf.oh == 0;
        int ab = abase;
        bool top = true;
        int nfl = 0;
        // B operands are read one step ahead, the channel operand (which every MFMA of a step needs)
        // two steps ahead into a ring of four, so no MFMA waits for a read issued just before it
        float aq[4][OTW], b0[KTW], b1[KTW];
#define WBS_OPSB(b_, s_, KO)                                                                 \
    _Pragma("unroll") for (int q = 0; q < KTW; ++q) b_[q] = smem[KO[q] + 4 * (s_)];
#define WBS_OPSA(r_, s_, AB)                                                                 \
    _Pragma("unroll") for (int t = 0; t < OTW; ++t) aq[r_][t] = smem[AB + t * 16 * WBR_GP + 4 * (s_)];
#define WBS_MMA(r_, b_)                                                                      \
    _Pragma("unroll") for (int q = 0; q < KTW; ++q)                                          \
        _Pragma("unroll") for (int t = 0; t < OTW; ++t) acc[q][t] = mma(aq[r_][t], b_[q], acc[q][t]);
        // behind every MFMA of a step: R LDS reads, V VALU, S SALU, M global loads, W LDS writes of
        // whatever else the step carries, so that a wave alone on its SIMD keeps the pipe fed
#define WBS_MIX(R, V, S, M, W)                                                               \
    _Pragma("unroll") for (int m_ = 0; m_ < OTW * KTW; ++m_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
        if (R) __builtin_amdgcn_sched_group_barrier(0x100, R, 0);                            \
        if (V) __builtin_amdgcn_sched_group_barrier(0x002, V, 0);                            \
        if (S) __builtin_amdgcn_sched_group_barrier(0x004, S, 0);                            \
        if (M) __builtin_amdgcn_sched_group_barrier(0x020, M, 0);                            \
        if (W) __builtin_amdgcn_sched_group_barrier(0x200, W, 0);                            \
    }                                                                                        \
    __builtin_amdgcn_sched_barrier(0);
        for (int unit = ufirst; unit < uend; ++unit) {
            const bool has1 = unit + 1 < uend, has2 = unit + 2 < uend;
            if (top) {
                // top of a run or stripe: the kH - 1 rows above the staged one, once every wave
                // is through with the unit before; no operands were read ahead
#ifdef WM_WB_STAMPS
                const unsigned long long ct = WB_CLK();
#endif
                __syncthreads();
                Cur c0;
                locate(c0, unit);
                const char *xr = (const char *)(x + (size_t)c0.b * xsample + (size_t)((c0.od * d.H + c0.oh) * d.W + c0.xs0));
                for (int kh = 0; kh + 1 < d.kH; ++kh) {
                    int sl = p + kh;
                    if (sl >= NS) sl -= NS;
                    f32x4 row[WBS_NX];
#pragma unroll
                    for (int i = 0; i < WBS_NX; ++i)
                        row[i] = *(const f32x4a4 *)(xr + (size_t)kh * d.W * sizeof(float) + xsrc[i]);
#pragma unroll
                    for (int i = 0; i < WBS_NX; ++i) wm_store4(smem + sl * SS + xdst[i], row[i]);
                }
                __syncthreads();
                WBS_OPSB(b0, 0, kofd)
                WBS_OPSA(0, 0, ab)
                WBS_OPSA(1, 1, ab)
                __builtin_amdgcn_sched_barrier(0);
#ifdef WM_WB_STAMPS
                t_top += WB_CLK() - ct;
#endif
            }
            int p1 = p + 1;
            p1 = p1 == NS ? 0 : p1;
            int mz2 = 0, dl2 = 0;
            bool top2 = false;
#pragma unroll
            for (int s_ = 0; s_ < WBR_SEG / 4; s_ += 2) {
                // ---- step s_
                WBS_OPSB(b1, s_ + 1, kofd)
                if (s_ + 2 < WBR_SEG / 4) {
                    WBS_OPSA((s_ + 2) & 3, s_ + 2, ab)
                } else {
                    // every read of this unit has been issued: on to the next unit's addresses and
                    // first operands (read again above when it is a top)
                    ab = abase + (par ^ 1) * GSZ;
                    WBS_OPSA(0, 0, ab)
#pragma unroll
                    for (int q = 0; q < KTW; ++q) kofd[q] = (int)tab[(p1 * KTW + q) * BLOCK];
                }
                if (s_ == 6) put(p == 0 ? NS - 1 : p - 1, par ^ 1, mz1, has1 ? 1.0f : 0.0f);
                if (s_ == 12) {                            // the rows of unit + 2 set out
                    advance(cf, has2);
                    fetch(cf);
                    mz2 = cf.mz;
                    dl2 = cf.ow0 - cf.xs0;
                    top2 = cf.oh == 0;
                }
                WBS_MMA(s_ & 3, b0)
                if (s_ == 6) { WBS_MIX(1, 3, 1, 0, 1) }
                else if (s_ == 12) { WBS_MIX(1, 1, 5, 1, 0) }
                else if (s_ == 14) { WBS_MIX(2, 1, 0, 0, 0) }
                else { WBS_MIX(1, 0, 0, 0, 0) }
                // ---- step s_ + 1
                if (s_ == 4) __syncthreads();              // every wave is through with the unit before: its
                                                           // first row's slot and its gy buffer are free
                if (s_ == 10) __syncthreads();             // tile(unit + 1) is in LDS
                if (s_ + 2 < WBR_SEG / 4) {
                    WBS_OPSB(b0, s_ + 2, kofd)
                    WBS_OPSA((s_ + 3) & 3, s_ + 3, ab)
                } else {
#pragma unroll
                    for (int q = 0; q < KTW; ++q) kofd[q] += dl1;
                    WBS_OPSB(b0, 0, kofd)
                    WBS_OPSA(1, 1, ab)
                }
                WBS_MMA((s_ + 1) & 3, b1)
                if (s_ == 14) { WBS_MIX(1, 1, 0, 0, 0) }
                else { WBS_MIX(1, 0, 0, 0, 0) }
            }
            if (++nfl == WBR_FLUSH || !has1) {             // 1024 outputs of fp32 accumulation
                nfl = 0;
#pragma unroll
                for (int q = 0; q < KTW; ++q)
#pragma unroll
                    for (int t = 0; t < OTW; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dsum[q][t][r] += (double)acc[q][t][r];
                        acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int i = 0; i < WBS_NG; ++i) {
                    bsum[i] += (double)bacc[i][0] + (double)bacc[i][1];
                    bacc[i] = (f32x2){0.f, 0.f};
                }
            }
            top = has1 && top1;
            top1 = top2;
            mz1 = mz2;
            dl1 = dl2;
            p = p1;
            par ^= 1;
        }
#undef WBS_OPSA
#undef WBS_OPSB
#undef WBS_MMA
#undef WBS_MIX
    }
#ifdef WM_WB_STAMPS
    if ((tid & 63) == 0 && blockIdx.x < 1024 && wave < 1) {
        unsigned long long *o = wm_wb_stamps + 8 * blockIdx.x;
        o[0] = t_top; o[1] = 0; o[2] = 0; o[3] = WB_CLK() - c_begin;
        o[4] = wall_clock64() - w_begin; o[5] = w_begin; o[6] = uend - ufirst; o[7] = 0;
        wm_wb_stamps[1024 * 8 + blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    }
#endif
#pragma unroll
    for (int q = 0; q < KTW; ++q) {
        const int k = (tg * KTW + q) * 16 + j;
#pragma unroll
        for (int t = 0; t < OTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (og * OTW + t) * 16 + 4 * g + r;
                if (k < K && o < d.Cout) slab[(size_t)o * (K + 1) + k] = dsum[q][t][r];
            }
    }
    if (blockIdx.x == 0 && tid == 0) {
        unsigned long long *rec = (unsigned long long *)(partial + (size_t)gridDim.x * d.Cout * (K + 1));
        rec[0] = WBS_CLOCK_MAGIC;
        rec[1] = __builtin_readcyclecounter() - clk0;
        rec[2] = wall_clock64() - wall0;
    }
    // a channel's 16 pieces sit in 16 consecutive lanes
#pragma unroll
    for (int i = 0; i < WBS_NG; ++i) {
        double b_ = bsum[i];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) b_ += __shfl_xor(b_, m);
        const int o = (tid + BLOCK * i) >> 4;
        if ((tid & 15) == 0 && gval[i]) slab[(size_t)o * (K + 1) + K] = b_;
    }
}

// ------------------------------------------------------------------------------ backward w, b, small volumes
// The reference's own Conv3D shapes ((16,1,20^3) x 7^3 and (16,16,7,7,14) x 5^3: rows of 14 and 10
// outputs, 1e4..1e5 outputs in all) are far too small for the row kernel and made the generic
// kernel above gather-bound (two scattered 16-line gathers per MFMA: 70 us per call, the largest
// item of the model's training step).  Here a workgroup owns one output plane (b, od): its input
// slab x[b, :, od..od+kD-1, :, :] and gy[b, :, od, :, :] are copied once into LDS, coalesced, and
// both MFMA operands are LDS reads (B at tap offset + position offset, both from tiny tables).
// grid = (B*OD planes, groups of WAVES*KTW tap tiles); a wave keeps KTW tap tiles x OT channel
// tiles in registers over the plane and stores them to the plane's partial slab -- as float32: the
// sums ARE float32 MFMA accumulators of one plane, so a float64 slab holds nothing more and costs
// twice the traffic (layer 2 of the reference: 49 MB written and read back per call);
// conv3d_reduce16_kernel sums the B*OD slabs in float64 in a fixed order.
template <int OT, int KTW>
__global__ __launch_bounds__(BLOCK) void conv3d_bwd_wb_vol_kernel(const float *__restrict__ x,
                                                                  const float *__restrict__ gy,
                                                                  float *__restrict__ partial,
                                                                  Dims3 d, int K)
{
    extern __shared__ __align__(16) float smem[];
    const int HW = d.H * d.W, slabn = d.Cin * d.kD * HW, P = d.OH * d.OW, P4 = (P + 3) & ~3;
    const int GP = P4 + 4;                               // gy row pitch
    // [slabn] x slab | [HW] ones | [HW] zeros | [OT*16][GP] gy | [P4] position offsets
    const int ones = slabn, zeros = ones + HW, goff = zeros + HW, poff = goff + OT * 16 * GP;
    int *pos = (int *)(smem + poff);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int plane = blockIdx.x, b_ = plane / d.OD, od = plane - b_ * d.OD;

    {
        const int run = d.kD * HW;                        // contiguous floats per input channel
        const float *src = x + ((size_t)b_ * d.Cin * d.D + od) * HW;
        for (int e = tid; e < slabn; e += BLOCK) {
            const int c = e / run, r = e - c * run;
            smem[e] = src[(size_t)c * d.D * HW + r];
        }
        for (int e = tid; e < HW; e += BLOCK) { smem[ones + e] = 1.0f; smem[zeros + e] = 0.0f; }
        const float *gsrc = gy + ((size_t)b_ * d.Cout * d.OD + od) * P;
        for (int e = tid; e < OT * 16 * P4; e += BLOCK) {
            const int o = e / P4, n = e - o * P4;
            smem[goff + o * GP + n] = (o < d.Cout && n < P) ? gsrc[(size_t)o * d.OD * P + n] : 0.0f;
        }
        for (int n = tid; n < P4; n += BLOCK) pos[n] = n < P ? (n / d.OW) * d.W + n % d.OW : 0;
    }
    // this lane's B-operand base of tap tile q: tap k = ((group*KTW + q)*WAVES + wave)*16 + j
    int koff[KTW];
#pragma unroll
    for (int q = 0; q < KTW; ++q) {
        const int k = (((int)blockIdx.y * KTW + q) * WAVES + wave) * 16 + j;
        int o_ = zeros;
        if (k < K) {
            const int kw = k % d.kW, t1 = k / d.kW, kh = t1 % d.kH, t2 = t1 / d.kH;   // t2 = c*kD + kd
            o_ = t2 * HW + kh * d.W + kw;
        } else if (k == K) {
            o_ = ones;
        }
        koff[q] = o_;
    }
    f32x4 acc[KTW][OT];
#pragma unroll
    for (int q = 0; q < KTW; ++q)
#pragma unroll
        for (int t = 0; t < OT; ++t) acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int abase = goff + j * GP + g;
    for (int s_ = 0; s_ < P4; s_ += 4) {
        const int po = pos[s_ + g];
        float a[OT], bq[KTW];
#pragma unroll
        for (int t = 0; t < OT; ++t) a[t] = smem[abase + t * 16 * GP + s_];
#pragma unroll
        for (int q = 0; q < KTW; ++q) bq[q] = smem[koff[q] + po];
#pragma unroll
        for (int q = 0; q < KTW; ++q)
#pragma unroll
            for (int t = 0; t < OT; ++t) acc[q][t] = mma(a[t], bq[q], acc[q][t]);
    }
    const int nw = d.Cout * K, nent = nw + d.Cout;
    float *dst = partial + (size_t)plane * nent;
#pragma unroll
    for (int q = 0; q < KTW; ++q) {
        const int k_b = (((int)blockIdx.y * KTW + q) * WAVES + wave) * 16 + j;
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = t * 16 + 4 * g + r;
                if (o < d.Cout) {
                    if (k_b < K) dst[(size_t)o * K + k_b] = acc[q][t][r];
                    else if (k_b == K) dst[nw + o] = acc[q][t][r];
                }
            }
    }
}

__global__ __launch_bounds__(64) void conv3d_reduce_kernel(const double *__restrict__ partial,
                                                           int nparts, int nent, int nw,
                                                           float *__restrict__ gw,
                                                           float *__restrict__ gb)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= nent) return;
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * nent + e];
    if (e < nw) gw[e] = (float)s;
    else if (gb) gb[e - nw] = (float)s;
}

// conv3d_reduce_kernel for many slabs: 64 entries x 16 slab groups per workgroup, group partials
// combined through LDS in a fixed order
__global__ __launch_bounds__(1024) void conv3d_reduce16_kernel(const float *__restrict__ partial,
                                                               int nparts, int nent, int nw,
                                                               float *__restrict__ gw,
                                                               float *__restrict__ gb)
{
    __shared__ double part[16][64];
    const int le = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + le;
    double s = 0.0;
    if (e < nent)
        for (int p = grp; p < nparts; p += 16) s += (double)partial[(size_t)p * nent + e];
    part[grp][le] = s;
    __syncthreads();
    if (grp == 0 && e < nent) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][le];
        if (e < nw) gw[e] = (float)t;
        else if (gb) gb[e - nw] = (float)t;
    }
}

// conv3d_reduce16_kernel with 16-byte reads (nent a multiple of 4, slabs 16-byte aligned): a lane owns
// four consecutive entries, so a wave reads 1 KiB runs of a slab instead of 256-byte ones -- layer 2
// of the reference's model (48 slabs x 128 064 entries = 24.6 MB): 17 -> 8 us.  Same summation order
// per entry as the scalar kernel.
__global__ __launch_bounds__(1024) void conv3d_reduce16v_kernel(const float *__restrict__ partial,
                                                                int nparts, int nent, int nw,
                                                                float *__restrict__ gw,
                                                                float *__restrict__ gb)
{
    __shared__ double part[16][64][4];
    const int le = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = (blockIdx.x * 64 + le) * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (e < nent)
        for (int p = grp; p < nparts; p += 16) {
            const f32x4 v = __builtin_nontemporal_load(
                reinterpret_cast<const f32x4 *>(partial + (size_t)p * nent + e));
            s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
        }
    part[grp][le][0] = s0; part[grp][le][1] = s1; part[grp][le][2] = s2; part[grp][le][3] = s3;
    __syncthreads();
    if (threadIdx.x < 256) {
        const int l2 = threadIdx.x >> 2, j = threadIdx.x & 3;
        const int ee = (blockIdx.x * 64 + l2) * 4 + j;
        if (ee < nent) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += part[q][l2][j];
            if (ee < nw) gw[ee] = (float)t;
            else if (gb) gb[ee - nw] = (float)t;
        }
    }
}

// the same sum for many slabs (one per persistent workgroup): 64 entries x 16 slab groups per
// workgroup, group partials combined through LDS in a fixed order
__global__ __launch_bounds__(1024) void conv3d_reduce_slabs_kernel(const double *__restrict__ partial,
                                                                   int nparts, int Cout, int K,
                                                                   float *__restrict__ gw,
                                                                   float *__restrict__ gb)
{
    __shared__ double part[16][64];
    const int le = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int nent = Cout * (K + 1);                    // slab = [Cout][K + 1], column K = bias
    const int e = blockIdx.x * 64 + le;
    double s = 0.0;
    if (e < nent)
        for (int p = grp; p < nparts; p += 16) s += partial[(size_t)p * nent + e];
    part[grp][le] = s;
    __syncthreads();
    if (grp == 0 && e < nent) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][le];
        const int o = e / (K + 1), k = e - o * (K + 1);
        if (k < K) gw[(size_t)o * K + k] = (float)t;
        else if (gb) gb[o] = (float)t;
    }
}

// ------------------------------------------------------------------------------ max pool
// x is (B, A1, A2, A3, C) contiguous; window 2, stride 2 over A1..A3 -> (B, A1/2, A2/2, A3/2, C).
// which[out] = position (0..7, a1-major) of the first maximum of the window.
// One launch serves up to two tensors of the same shape (blockIdx.y: the model's two conv branches),
// and sample b of the output sits at y + b * y_bstride: with y1 = y0 + per-sample size and
// y_bstride = twice that, the two pools write the concatenated (B, 2 * flat) matrix the Dense layer
// reads -- no concatenation copy.  Index arithmetic in I = 32 bits whenever the tensors allow it (a
// 64-bit division by a run-time value is a ~150-instruction routine; five of them per element were
// most of these kernels' time).
struct PoolPair {
    const float *x[2];
    float *y[2];
    unsigned char *which[2];
};

template <typename I>
__global__ __launch_bounds__(BLOCK) void maxpool3d_fwd_kernel(const PoolPair pp, int A1, int A2, int A3,
                                                              int C, long long total,
                                                              long long y_bstride)
{
    const float *__restrict__ x = pp.x[blockIdx.y];
    float *__restrict__ y = pp.y[blockIdx.y];
    unsigned char *__restrict__ which = pp.which[blockIdx.y];
    const I P1 = A1 / 2, P2 = A2 / 2, P3 = A3 / 2;
    const I per = P1 * P2 * P3 * (I)C;
    for (I i = (I)blockIdx.x * BLOCK + threadIdx.x; i < (I)total; i += (I)gridDim.x * BLOCK) {
        const I c = i % (I)C;
        I t = i / (I)C;
        const I p3 = t % P3; t /= P3;
        const I p2 = t % P2; t /= P2;
        const I p1 = t % P1;
        const I b_ = t / P1;
        const float *xb = x + ((((long long)b_ * A1 + 2 * p1) * A2 + 2 * p2) * A3 + 2 * p3) * C + c;
        float best = 0.0f;
        int arg = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float v = xb[(((q >> 2) * A2 + ((q >> 1) & 1)) * A3 + (q & 1)) * C];
            if (q == 0 || v > best) { best = v; arg = q; }
        }
        y[(long long)b_ * y_bstride + (i - b_ * per)] = best;
        if (which) which[i] = (unsigned char)arg;
    }
}

struct PoolPairBwd {
    const float *gy[2];
    const unsigned char *which[2];
    float *gx[2];
};

// one thread per INPUT element: windows do not overlap, so no atomics are needed
template <typename I>
__global__ __launch_bounds__(BLOCK) void maxpool3d_bwd_kernel(const PoolPairBwd pp, int A1, int A2,
                                                              int A3, int C, long long total,
                                                              long long gy_bstride)
{
    const float *__restrict__ gy = pp.gy[blockIdx.y];
    const unsigned char *__restrict__ which = pp.which[blockIdx.y];
    float *__restrict__ gx = pp.gx[blockIdx.y];
    const I P1 = A1 / 2, P2 = A2 / 2, P3 = A3 / 2;
    for (I i = (I)blockIdx.x * BLOCK + threadIdx.x; i < (I)total; i += (I)gridDim.x * BLOCK) {
        const I c = i % (I)C;
        I t = i / (I)C;
        const I a3 = t % (I)A3; t /= (I)A3;
        const I a2 = t % (I)A2; t /= (I)A2;
        const I a1 = t % (I)A1;
        const I b_ = t / (I)A1;
        const I p1 = a1 >> 1, p2 = a2 >> 1, p3 = a3 >> 1;
        float v = 0.0f;
        if (p1 < P1 && p2 < P2 && p3 < P3) {
            const I within = ((p1 * P2 + p2) * P3 + p3) * (I)C + c;
            const int q = (int)(((a1 & 1) << 2) | ((a2 & 1) << 1) | (a3 & 1));
            if (which[b_ * (P1 * P2 * P3 * (I)C) + within] == q)
                v = gy[(long long)b_ * gy_bstride + within];
        }
        gx[i] = v;
    }
}

// the wide weight-gradient kernel: enough outputs to keep persistent workgroups busy and at most
// 8 tap tiles per wave (accumulators live in registers)
bool wb_wide_shape(long long Ntot, int K)
{
    const int cus = wm_num_cus();
    return cus > 0 && Ntot >= 4096ll * cus && (K + 1 + 15) / 16 <= 8 * WAVES;
}

// The small-volume kernel: tap tiles per wave (1, 2 or 4), groups and LDS bytes, or false.
bool wb_vol_plan(int B, int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW, int *ktw,
                 int *groups, size_t *lds)
{
    const int OD = D - kD + 1, OH = H - kH + 1, OW = W - kW + 1;
    const int K = Cin * kD * kH * kW, P4 = (OH * OW + 3) & ~3;
    const int otiles = (Cout + 15) / 16, ot = otiles == 3 ? 4 : otiles, ktiles = (K + 1 + 15) / 16;
    const size_t bytes = ((size_t)Cin * kD * H * W + 2 * (size_t)H * W + (size_t)ot * 16 * (P4 + 4) +
                          (size_t)P4) * sizeof(float);
    if (bytes > RESIDENT_LDS || (long long)B * OD > 65535 || OW >= WBR_SEG) return false;
    // as many tap tiles per wave as keep the accumulators at <= 16 tiles and the grid >= 2 per CU
    int t = ot >= 4 ? 4 : (ot == 2 ? 4 : 8);
    t = t > 4 ? 4 : t;
    while (t > 1 && (long long)B * OD * ((ktiles + WAVES * t - 1) / (WAVES * t)) < 2ll * wm_num_cus()) t >>= 1;
    *ktw = t;
    *groups = (ktiles + WAVES * t - 1) / (WAVES * t);
    *lds = bytes;
    return true;
}

typedef void (*wb_vol_fn)(const float *, const float *, float *, Dims3, int);
wb_vol_fn wb_vol_kernel(int otiles, int ktw)
{
#define WBV_K(OT_)                                                                \
    return ktw == 1 ? conv3d_bwd_wb_vol_kernel<OT_, 1>                             \
           : ktw == 2 ? conv3d_bwd_wb_vol_kernel<OT_, 2>                           \
                      : conv3d_bwd_wb_vol_kernel<OT_, 4>
    if (otiles == 1) { WBV_K(1); }
    if (otiles == 2) { WBV_K(2); }
    WBV_K(4);
#undef WBV_K
}

// Resident workgroups per CU of a kernel with `lds` bytes of dynamic LDS: asked once per (device,
// kernel, bytes) and kept -- the query is a host call of microseconds and not legal under stream
// capture.  Also raises the kernel's dynamic-LDS limit when it needs more than 64 KB.  0: failed.
int occupancy_cached(const void *fn, size_t lds)
{
    // a map, not a fixed table: the row, staged and forward kernels have more than 40 template
    // variants and the LDS part of the key depends on the shape -- a full table silently fell back
    // to the query on every call, which is not legal under stream capture
    static std::mutex mu;
    static std::map<std::tuple<int, const void *, size_t>, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(dev, fn, lds);
    const auto hit = cache.find(key);
    if (hit != cache.end()) return hit->second;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return 0;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, BLOCK, lds) != hipSuccess || occ <= 0)
        return 0;
    cache[key] = occ;
    return occ;
}

// The LDS-staged row kernel: grid (a multiple of 8, one float64 slab per workgroup), LDS bytes and
// kernel for a shape, or false when the shape takes another path.
typedef void (*wb_rows_fn)(const float *, const float *, double *, Dims3, int, int);
wb_rows_fn wb_rows_kernel(int otiles, int ktw)
{
#define WBR_K(OT_)                                                                \
    switch (ktw) {                                                                \
    case 1: return conv3d_bwd_wb_rows_kernel<OT_, 1>;                             \
    case 2: return conv3d_bwd_wb_rows_kernel<OT_, 2>;                             \
    case 3: return conv3d_bwd_wb_rows_kernel<OT_, 3>;                             \
    case 4: return conv3d_bwd_wb_rows_kernel<OT_, 4>;                             \
    case 5: return conv3d_bwd_wb_rows_kernel<OT_, 5>;                             \
    case 6: return conv3d_bwd_wb_rows_kernel<OT_, 6>;                             \
    case 7: return conv3d_bwd_wb_rows_kernel<OT_, 7>;                             \
    default: return conv3d_bwd_wb_rows_kernel<OT_, 8>;                            \
    }
    if (otiles == 1) { WBR_K(1) }
    if (otiles == 2) { WBR_K(2) }
    WBR_K(4)
#undef WBR_K
}

bool wb_rows_plan(int B, int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW, int *grid,
                  size_t *lds)
{
    const int OD = D - kD + 1, OH = H - kH + 1, OW = W - kW + 1;
    const long long Ntot = (long long)B * OD * OH * OW;
    const int K = Cin * kD * kH * kW;
    if (!wb_wide_shape(Ntot, K) || kW - 1 > WBR_XW - WBR_SEG || W < WBR_XW || OW < WBR_SEG) return false;
    const int otiles = (Cout + 15) / 16, ktiles = (K + 1 + 15) / 16;
    const size_t bytes = wbr_lds_bytes(Cin * kD, kH, otiles == 3 ? 4 : otiles);
    if (bytes > RESIDENT_LDS || Cin * kD * 18 > BLOCK * WBR_MAXS) return false;
    const void *kfn = (const void *)wb_rows_kernel(otiles, (ktiles + WAVES - 1) / WAVES);
    const int occ = occupancy_cached(kfn, bytes);
    if (occ <= 0) return false;
    *grid = wm_num_cus() * occ & ~7;
    *lds = bytes;
    return *grid > 0;
}

// The register-staged row kernel: which split of the 4 waves, its grid and LDS bytes, or false
// when the shape stays with the kernels above (more than 32 output channels, more than 45 input
// rows per slot, or float64 sums that do not fit in registers).
typedef void (*wb_staged_fn)(const float *, const float *, double *, Dims3, int);
struct WbStagedPlan { wb_staged_fn fn; int grid; size_t lds; };

wb_staged_fn wb_staged_kernel(int otw, int ktw, int wos)
{
#define WBS_K(OTW_, WOS_, K_) case K_: return conv3d_bwd_wb_staged_kernel<OTW_, K_, WOS_>;
    if (wos == 1 && otw == 1) switch (ktw) { WBS_K(1, 1, 1) WBS_K(1, 1, 2) WBS_K(1, 1, 3) WBS_K(1, 1, 4) WBS_K(1, 1, 5) WBS_K(1, 1, 6) WBS_K(1, 1, 7) WBS_K(1, 1, 8) }
    if (wos == 1 && otw == 2) switch (ktw) { WBS_K(2, 1, 1) WBS_K(2, 1, 2) WBS_K(2, 1, 3) WBS_K(2, 1, 4) WBS_K(2, 1, 5) }
    if (wos == 2 && otw == 1) switch (ktw) { WBS_K(1, 2, 5) WBS_K(1, 2, 6) WBS_K(1, 2, 7) WBS_K(1, 2, 8) WBS_K(1, 2, 9) }
#undef WBS_K
    return nullptr;
}

bool wb_staged_plan(int B, int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW,
                    WbStagedPlan *plan)
{
    const int OD = D - kD + 1, OH = H - kH + 1, OW = W - kW + 1;
    const long long Ntot = (long long)B * OD * OH * OW;
    const int K = Cin * kD * kH * kW, CD = Cin * kD;
    if (!wb_wide_shape(Ntot, K) || kW - 1 > WBR_XW - WBR_SEG || W < WBR_XW || OW < WBR_SEG) return false;
    const int otiles = (Cout + 15) / 16, ktiles = (K + 15) / 16;
#ifdef WM_NO_STAGED
    return false;
#endif
    // 32-bit byte offsets inside a sample
    if (otiles > 2 || CD * 17 > BLOCK * WBS_NX || (long long)Cin * D * H * W >= (1ll << 30) ||
        (long long)Cout * OD * OH * OW >= (1ll << 30))
        return false;
    // the split with the fewest MFMAs per wave and step
    wb_staged_fn best = nullptr;
    int best_work = 1 << 30, best_ktw = 0;
    for (int wos = 1; wos <= otiles; wos *= 2) {
        const int otw = otiles / wos, ktw = (ktiles + WAVES / wos - 1) / (WAVES / wos);
        wb_staged_fn fn = otw * ktw <= 10 ? wb_staged_kernel(otw, ktw, wos) : nullptr;
        if (fn && otw * ktw < best_work) { best = fn; best_work = otw * ktw; best_ktw = ktw; }
    }
    if (!best) return false;
    const size_t bytes = wbs_lds_bytes(CD, kH, otiles, best_ktw);
    // B addresses are 16-bit words of the ring
    if (bytes > 80 * 1024 || (kH + 1) * wbr_slot_stride(CD) >= (1 << 16)) return false;
    int occ = occupancy_cached((const void *)best, bytes);
    if (occ <= 0) return false;
#ifdef WM_PROBE_ENV
    if (const char *e = getenv("WM_WB_BPC")) occ = min(occ, atoi(e));
#endif
    plan->fn = best;
    plan->grid = wm_num_cus() * occ & ~7;
    plan->lds = bytes;
    return plan->grid > 0;
}

int fill_dims(Dims3 *d, int B, int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW)
{
    WM_REQUIRE(B > 0 && Cin > 0 && D > 0 && H > 0 && W > 0 && Cout > 0 && kD > 0 && kH > 0 && kW > 0,
               WM_ERR_INVALID, "conv3d: non-positive dimension");
    WM_REQUIRE(kD <= D && kH <= H && kW <= W, WM_ERR_INVALID,
               "conv3d: kernel (%d,%d,%d) larger than input (%d,%d,%d)", kD, kH, kW, D, H, W);
    WM_REQUIRE(kD < 256 && kH < 256 && kW < 256 && Cout <= 64 && Cin <= 64, WM_ERR_UNSUPPORTED,
               "conv3d: kernel sides < 256 and at most 64 channels per side are implemented "
               "(Cin=%d Cout=%d)", Cin, Cout);
    d->B = B; d->Cin = Cin; d->D = D; d->H = H; d->W = W; d->Cout = Cout;
    d->kD = kD; d->kH = kH; d->kW = kW;
    d->OD = D - kD + 1; d->OH = H - kH + 1; d->OW = W - kW + 1;
    // 32-bit index arithmetic inside the kernels
    const long long xin = (long long)B * Cin * D * H * W;
    const long long yout = (long long)B * Cout * d->OD * d->OH * d->OW;
    const long long wn = (long long)Cout * Cin * kD * kH * kW;
    WM_REQUIRE(xin < (1ll << 31) && yout < (1ll << 31) && wn < (1ll << 31), WM_ERR_UNSUPPORTED,
               "conv3d: tensors of 2^31 elements or more are not implemented");
    return WM_OK;
}

}  // namespace

extern "C" int wm_conv3d_fwd_f32(const float *x, const float *w, const float *b, float *y, int B,
                                 int Cin, int D, int H, int W, int Cout, int kD, int kH, int kW,
                                 int relu, void *stream)
{
    WM_REQUIRE(x && w && y, WM_ERR_INVALID, "wm_conv3d_fwd_f32: null pointer");
    Dims3 d;
    if (int rc = fill_dims(&d, B, Cin, D, H, W, Cout, kD, kH, kW)) return rc;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_conv3d_fwd_f32: no HIP device");
    const int K = Cin * kD * kH * kW;
    const int Ntot = B * d.OD * d.OH * d.OW;
    hipStream_t st = (hipStream_t)stream;
    const int ot = (Cout + 15) / 16;
    if (D == 1 && kD == 1) {      // large 2-D grids: input rows staged through an LDS ring
        const int rc = wm_conv_ring_launch(0, x, w, b, y, B, Cin, H, W, Cout, kH, kW, relu, st);
        if (rc != 0) return rc < 0 ? rc : WM_OK;
    }
    if (Ntot >= 16 * WIDE_TILES * 2 * wm_num_cus()) {          // enough tiles for two waves of blocks
        const int wblocks = (Ntot + 16 * WIDE_TILES - 1) / (16 * WIDE_TILES);
        const int otk = ot == 3 ? 4 : ot;
        const size_t lds = rows_lds_bytes(K, otk);
        // 32-bit byte offsets inside one sample of x and of y
        if (lds <= RESIDENT_LDS && d.OW >= ROWS_SEG && (long long)Cin * D * H * W < (1ll << 30) &&
            (long long)Cout * d.OD * d.OH * d.OW < (1ll << 30)) {
            // persistent: exactly the workgroups that are resident at once (registers and LDS
            // decide; a grid above that runs its surplus as a second, mostly empty round)
            const void *kfn = otk == 1   ? (const void *)conv3d_fwd_rows_kernel<1>
                              : otk == 2 ? (const void *)conv3d_fwd_rows_kernel<2>
                                         : (const void *)conv3d_fwd_rows_kernel<4>;
            const int occ = occupancy_cached(kfn, lds);
            WM_REQUIRE(occ > 0, WM_ERR_HIP, "conv3d_fwd_rows_kernel does not fit on a CU");
            const int units = B * d.OD * d.OH * ((d.OW + ROWS_SEG - 1) / ROWS_SEG);
            int grid = wm_num_cus() * occ & ~7;
            if (grid > (units + WAVES - 1) / WAVES) grid = ((units + WAVES - 1) / WAVES + 7) & ~7;
            if (otk == 1)
                hipLaunchKernelGGL(conv3d_fwd_rows_kernel<1>, dim3(grid), dim3(BLOCK), lds, st, x, w,
                                   b, y, d, K, relu);
            else if (otk == 2)
                hipLaunchKernelGGL(conv3d_fwd_rows_kernel<2>, dim3(grid), dim3(BLOCK), lds, st, x, w,
                                   b, y, d, K, relu);
            else
                hipLaunchKernelGGL(conv3d_fwd_rows_kernel<4>, dim3(grid), dim3(BLOCK), lds, st, x, w,
                                   b, y, d, K, relu);
            WM_LAUNCH_CHECK("conv3d_fwd_rows_kernel");
            return WM_OK;
        }
        if (ot == 1)
            hipLaunchKernelGGL(conv3d_fwd_wide_kernel<1>, dim3(wblocks), dim3(BLOCK), 0, st, x, w, b,
                               y, d, K, Ntot, relu);
        else if (ot == 2)
            hipLaunchKernelGGL(conv3d_fwd_wide_kernel<2>, dim3(wblocks), dim3(BLOCK), 0, st, x, w, b,
                               y, d, K, Ntot, relu);
        else
            hipLaunchKernelGGL(conv3d_fwd_wide_kernel<4>, dim3(wblocks), dim3(BLOCK), 0, st, x, w, b,
                               y, d, K, Ntot, relu);
        WM_LAUNCH_CHECK("conv3d_fwd_wide_kernel");
        return WM_OK;
    }
    {
        // small volumes whose packed weights and input slab fit in LDS together: a workgroup per
        // output plane, both operands from LDS
        const int vsteps = (((K + 3) >> 2) + 1) & ~1, otk = ot == 3 ? 4 : ot;
        const int PT = (d.OH * d.OW + 15) >> 4;
        const size_t vlds = ((size_t)Cin * kD * H * W + (size_t)vsteps * otk * 64 + (size_t)vsteps * 4 +
                             (size_t)PT * 16) * sizeof(float);
        if (vlds <= RESIDENT_LDS && d.OW < ROWS_SEG && B * d.OD >= wm_num_cus() / 2 && PT >= WAVES) {
            const int planes = B * d.OD;
            if (otk == 1)
                hipLaunchKernelGGL(conv3d_fwd_vol_kernel<1>, dim3(planes), dim3(BLOCK), vlds, st, x, w,
                                   b, y, d, K, relu);
            else if (otk == 2)
                hipLaunchKernelGGL(conv3d_fwd_vol_kernel<2>, dim3(planes), dim3(BLOCK), vlds, st, x, w,
                                   b, y, d, K, relu);
            else
                hipLaunchKernelGGL(conv3d_fwd_vol_kernel<4>, dim3(planes), dim3(BLOCK), vlds, st, x, w,
                                   b, y, d, K, relu);
            WM_LAUNCH_CHECK("conv3d_fwd_vol_kernel");
            return WM_OK;
        }
    }
    const int blocks = (Ntot + 15) / 16;            // one workgroup per 16-column tile
    // few tiles (the reference's second Conv3D layer has 90): eight waves split the contraction and
    // the channel tiles go to grid.y (the kernel is bound by its scattered weight gathers)
    const bool deep = blocks <= 4 * wm_num_cus();
#define FWD_SPLITK(OT_)                                                                          \
    if (deep)   /* one channel tile per workgroup: every workgroup gathers only its 16 rows of w */ \
        hipLaunchKernelGGL((conv3d_fwd_tap_kernel<1, 8>), dim3(blocks, ot), dim3(512), 0, st, x, w, b, \
                           y, d, K, Ntot, relu);                                                 \
    else                                                                                         \
        hipLaunchKernelGGL((conv3d_fwd_tap_kernel<OT_, 4>), dim3(blocks), dim3(256), 0, st, x, w, b, \
                           y, d, K, Ntot, relu)
    if (ot == 1) { FWD_SPLITK(1); }
    else if (ot == 2) { FWD_SPLITK(2); }
    else { FWD_SPLITK(4); }
#undef FWD_SPLITK
    WM_LAUNCH_CHECK("conv3d_fwd_tap_kernel");
    return WM_OK;
}

static int conv3d_bwd_data_impl(const float *gy, const float *w, float *gx, const float *mask,
                                int B, int Cin, int D, int H, int W, int Cout, int kD, int kH,
                                int kW, void *stream);

extern "C" int wm_conv3d_bwd_data_f32(const float *gy, const float *w, float *gx, int B, int Cin,
                                      int D, int H, int W, int Cout, int kD, int kH, int kW,
                                      void *stream)
{
    return conv3d_bwd_data_impl(gy, w, gx, nullptr, B, Cin, D, H, W, Cout, kD, kH, kW, stream);
}

extern "C" int wm_conv3d_bwd_data_relu_f32(const float *gy, const float *w, const float *mask,
                                           float *gx, int B, int Cin, int D, int H, int W, int Cout,
                                           int kD, int kH, int kW, void *stream)
{
    WM_REQUIRE(mask, WM_ERR_INVALID, "wm_conv3d_bwd_data_relu_f32: null mask");
    return conv3d_bwd_data_impl(gy, w, gx, mask, B, Cin, D, H, W, Cout, kD, kH, kW, stream);
}

// every path but the row kernel applies the mask in a second, element-wise launch
#define WM_BWD_DATA_DONE(name)                                                                 \
    do {                                                                                       \
        WM_LAUNCH_CHECK(name);                                                                 \
        if (mask) {                                                                            \
            const long long n_ = (long long)B * Cin * D * H * W;                               \
            long long g_ = (n_ + BLOCK - 1) / BLOCK;                                           \
            if (g_ > (long long)wm_num_cus() * 16) g_ = (long long)wm_num_cus() * 16;          \
            hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)g_), dim3(BLOCK), 0, st, gx,   \
                               mask, n_);                                                      \
            WM_LAUNCH_CHECK("relu_mask_kernel");                                               \
        }                                                                                      \
        return WM_OK;                                                                          \
    } while (0)

static int conv3d_bwd_data_impl(const float *gy, const float *w, float *gx, const float *mask,
                                int B, int Cin, int D, int H, int W, int Cout, int kD, int kH,
                                int kW, void *stream)
{
    WM_REQUIRE(gy && w && gx, WM_ERR_INVALID, "wm_conv3d_bwd_data_f32: null pointer");
    Dims3 d;
    if (int rc = fill_dims(&d, B, Cin, D, H, W, Cout, kD, kH, kW)) return rc;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_conv3d_bwd_data_f32: no HIP device");
    const int K2 = Cout * kD * kH * kW;
    const int Mtot = B * D * H * W;
    hipStream_t st = (hipStream_t)stream;
    const int ct = (Cin + 15) / 16;
    if (D == 1 && kD == 1) {      // large 2-D grids: gy rows staged through an LDS ring, mask fused
        const int rc = wm_conv_ring_launch(1, gy, w, mask, gx, B, Cin, H, W, Cout, kH, kW, 0, st);
        if (rc != 0) return rc < 0 ? rc : WM_OK;
    }
    if (Mtot >= 16 * WIDE_TILES * 2 * wm_num_cus()) {
        const int wblocks = (Mtot + 16 * WIDE_TILES - 1) / (16 * WIDE_TILES);
        const int ctk = ct == 3 ? 4 : ct;
        const size_t lds = rows_lds_bytes(K2, ctk) + (size_t)rows_steps(K2) * 4 * sizeof(int);
        if (lds <= RESIDENT_LDS && W >= ROWS_SEG && (long long)Cin * D * H * W < (1ll << 30) &&
            (long long)Cout * d.OD * d.OH * d.OW < (1ll << 30)) {
            const void *kfn = ctk == 1   ? (const void *)conv3d_bwd_data_rows_kernel<1>
                              : ctk == 2 ? (const void *)conv3d_bwd_data_rows_kernel<2>
                                         : (const void *)conv3d_bwd_data_rows_kernel<4>;
            const int occ = occupancy_cached(kfn, lds);
            WM_REQUIRE(occ > 0, WM_ERR_HIP, "conv3d_bwd_data_rows_kernel does not fit on a CU");
            const int units = B * D * H * ((W + ROWS_SEG - 1) / ROWS_SEG);
            int grid = wm_num_cus() * occ & ~7;
            if (grid > (units + WAVES - 1) / WAVES) grid = ((units + WAVES - 1) / WAVES + 7) & ~7;
            if (ctk == 1)
                hipLaunchKernelGGL(conv3d_bwd_data_rows_kernel<1>, dim3(grid), dim3(BLOCK), lds, st,
                                   gy, w, gx, d, K2, mask);
            else if (ctk == 2)
                hipLaunchKernelGGL(conv3d_bwd_data_rows_kernel<2>, dim3(grid), dim3(BLOCK), lds, st,
                                   gy, w, gx, d, K2, mask);
            else
                hipLaunchKernelGGL(conv3d_bwd_data_rows_kernel<4>, dim3(grid), dim3(BLOCK), lds, st,
                                   gy, w, gx, d, K2, mask);
            WM_LAUNCH_CHECK("conv3d_bwd_data_rows_kernel");
            return WM_OK;
        }
        if (ct == 1)
            hipLaunchKernelGGL(conv3d_bwd_data_wide_kernel<1>, dim3(wblocks), dim3(BLOCK), 0, st, gy,
                               w, gx, d, K2, Mtot);
        else if (ct == 2)
            hipLaunchKernelGGL(conv3d_bwd_data_wide_kernel<2>, dim3(wblocks), dim3(BLOCK), 0, st, gy,
                               w, gx, d, K2, Mtot);
        else
            hipLaunchKernelGGL(conv3d_bwd_data_wide_kernel<4>, dim3(wblocks), dim3(BLOCK), 0, st, gy,
                               w, gx, d, K2, Mtot);
        WM_BWD_DATA_DONE("conv3d_bwd_data_wide_kernel");
    }
    // rows of at most 16 positions whose taps are mostly out of range: one workgroup per row
    if (W <= 16 && 2 * d.OD * d.OH <= D * H && (long long)B * D * H <= 0x7fffffffll) {
        // two batch items per workgroup while that still leaves a workgroup per CU
        const bool two = (long long)((B + 1) / 2) * D * H >= wm_num_cus();
        const int rows = (two ? (B + 1) / 2 : B) * D * H;
#define WM_ROWTAP(CT_)                                                                           \
    if (two)                                                                                     \
        hipLaunchKernelGGL((conv3d_bwd_data_rowtap_kernel<CT_, 8, 2>), dim3(rows), dim3(512), 0, st, \
                           gy, w, gx, d);                                                        \
    else                                                                                         \
        hipLaunchKernelGGL((conv3d_bwd_data_rowtap_kernel<CT_, 8, 1>), dim3(rows), dim3(512), 0, st, \
                           gy, w, gx, d)
        if (ct == 1) { WM_ROWTAP(1); }
        else if (ct == 2) { WM_ROWTAP(2); }
        else { WM_ROWTAP(4); }
#undef WM_ROWTAP
        WM_BWD_DATA_DONE("conv3d_bwd_data_rowtap_kernel");
    }
    const int blocks = (Mtot + 15) / 16;            // one workgroup per 16-position tile
    const bool deep = blocks <= 4 * wm_num_cus();
#define BWD_SPLITK(CT_)                                                                          \
    if (kD * kH * kW <= KCHUNK) {   /* contraction in tap order: 16-byte weight loads */        \
        if (deep)                                                                                \
            hipLaunchKernelGGL((conv3d_bwd_data_tap_kernel<CT_, 8>), dim3(blocks), dim3(512), 0, st, \
                               gy, w, gx, d, Mtot);                                              \
        else                                                                                     \
            hipLaunchKernelGGL((conv3d_bwd_data_tap_kernel<CT_, 4>), dim3(blocks), dim3(256), 0, st, \
                               gy, w, gx, d, Mtot);                                              \
    } else if (deep)                                                                             \
        hipLaunchKernelGGL((conv3d_bwd_data_kernel<CT_, 8>), dim3(blocks), dim3(512), 0, st, gy, w, \
                           gx, d, K2, Mtot);                                                     \
    else                                                                                         \
        hipLaunchKernelGGL((conv3d_bwd_data_kernel<CT_, 4>), dim3(blocks), dim3(256), 0, st, gy, w, \
                           gx, d, K2, Mtot)
    if (ct == 1) { BWD_SPLITK(1); }
    else if (ct == 2) { BWD_SPLITK(2); }
    else { BWD_SPLITK(4); }
#undef BWD_SPLITK
    WM_BWD_DATA_DONE("conv3d_bwd_data_kernel");
}

#ifdef WM_WB_STAMPS
extern "C" int wm_debug_read_wb_stamps(unsigned long long *out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(wm_wb_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t wm_conv3d_bwd_wb_workspace_bytes(int B, int Cin, int D, int H, int W, int Cout,
                                                   int kD, int kH, int kW)
{
    if (B <= 0 || Cin <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || kD <= 0 || kH <= 0 ||
        kW <= 0 || kD > D || kH > H || kW > W)
        return 0;
    const long long Ntot = (long long)B * (D - kD + 1) * (H - kH + 1) * (W - kW + 1);
    size_t splits = (size_t)((Ntot + NCHUNK - 1) / NCHUNK);
    const size_t nent = (size_t)Cout * Cin * kD * kH * kW + Cout;
    if (wb_wide_shape(Ntot, Cin * kD * kH * kW)) splits = (size_t)wm_num_cus() * WB_WIDE_BPC;   // one slab per workgroup
    int rgrid = 0;
    size_t rlds = 0;
    WbStagedPlan sp;
    if (wb_staged_plan(B, Cin, D, H, W, Cout, kD, kH, kW, &sp))      // slabs + the clock record
        return wm_align_up((size_t)sp.grid * nent * sizeof(double) + 3 * sizeof(unsigned long long), 256);
    if (wb_rows_plan(B, Cin, D, H, W, Cout, kD, kH, kW, &rgrid, &rlds)) splits = (size_t)rgrid;
    int vktw = 0, vgroups = 0;
    size_t vlds = 0;
    if (!rgrid && wb_vol_plan(B, Cin, D, H, W, Cout, kD, kH, kW, &vktw, &vgroups, &vlds))
        splits = (size_t)B * (D - kD + 1);                          // one slab per output plane
    return wm_align_up(splits * nent * sizeof(double), 256);
}

extern "C" size_t wm_conv3d_bwd_wb_clock_offset(int B, int Cin, int D, int H, int W, int Cout,
                                                int kD, int kH, int kW)
{
    if (B <= 0 || Cin <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || kD <= 0 || kH <= 0 ||
        kW <= 0 || kD > D || kH > H || kW > W)
        return 0;
    WbStagedPlan sp;
    if (!wb_staged_plan(B, Cin, D, H, W, Cout, kD, kH, kW, &sp)) return 0;
    return (size_t)sp.grid * ((size_t)Cout * Cin * kD * kH * kW + Cout) * sizeof(double);
}

extern "C" int wm_conv3d_bwd_wb_f32(const float *x, const float *gy, float *gw, float *gb,
                                    void *workspace, size_t workspace_bytes, int B, int Cin, int D,
                                    int H, int W, int Cout, int kD, int kH, int kW, void *stream)
{
    WM_REQUIRE(x && gy && gw, WM_ERR_INVALID, "wm_conv3d_bwd_wb_f32: null pointer");
    Dims3 d;
    if (int rc = fill_dims(&d, B, Cin, D, H, W, Cout, kD, kH, kW)) return rc;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_conv3d_bwd_wb_f32: no HIP device");
    const size_t need = wm_conv3d_bwd_wb_workspace_bytes(B, Cin, D, H, W, Cout, kD, kH, kW);
    WM_REQUIRE(workspace && workspace_bytes >= need, WM_ERR_WORKSPACE,
               "wm_conv3d_bwd_wb_f32: workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
    const int K = Cin * kD * kH * kW;
    const int Ntot = B * d.OD * d.OH * d.OW;
    const int splits = (Ntot + NCHUNK - 1) / NCHUNK;
    WM_REQUIRE(splits <= 65535, WM_ERR_UNSUPPORTED, "wm_conv3d_bwd_wb_f32: too many outputs");
    const int ktiles = (K + 1 + 15) / 16, otiles = (Cout + 15) / 16;
    const int tiles = ktiles * otiles;
    const int nw = Cout * K, nent = nw + Cout;
    hipStream_t st = (hipStream_t)stream;
    WbStagedPlan sp;
    if (wb_staged_plan(B, Cin, D, H, W, Cout, kD, kH, kW, &sp)) {
        // every workgroup writes its whole slab: no memset
        hipLaunchKernelGGL(sp.fn, dim3(sp.grid), dim3(BLOCK), sp.lds, st, x, gy, (double *)workspace,
                           d, K);
        WM_LAUNCH_CHECK("conv3d_bwd_wb_staged_kernel");
        hipLaunchKernelGGL(conv3d_reduce_slabs_kernel, dim3((nent + 63) / 64), dim3(1024), 0, st,
                           (const double *)workspace, sp.grid, Cout, K, gw, gb);
        WM_LAUNCH_CHECK("conv3d_reduce_slabs_kernel");
        return WM_OK;
    }
    int rgrid = 0;
    size_t rlds = 0;
    if (wb_rows_plan(B, Cin, D, H, W, Cout, kD, kH, kW, &rgrid, &rlds)) {
        // every workgroup writes its whole slab when the sums live in registers
        if (!wbr_regsum(otiles == 3 ? 4 : otiles, min((ktiles + WAVES - 1) / WAVES, 8)))
            WM_HIP(hipMemsetAsync(workspace, 0, (size_t)rgrid * nent * sizeof(double), st));
        hipLaunchKernelGGL(wb_rows_kernel(otiles, (ktiles + WAVES - 1) / WAVES), dim3(rgrid),
                           dim3(BLOCK), rlds, st, x, gy, (double *)workspace, d, K, ktiles);
        WM_LAUNCH_CHECK("conv3d_bwd_wb_rows_kernel");
        hipLaunchKernelGGL(conv3d_reduce_slabs_kernel, dim3((nent + 63) / 64), dim3(1024), 0, st,
                           (const double *)workspace, rgrid, Cout, K, gw, gb);
        WM_LAUNCH_CHECK("conv3d_reduce_slabs_kernel");
        return WM_OK;
    }
    int vktw = 0, vgroups = 0;
    size_t vlds = 0;
    if (wb_vol_plan(B, Cin, D, H, W, Cout, kD, kH, kW, &vktw, &vgroups, &vlds)) {
        const int planes = B * d.OD;
        hipLaunchKernelGGL(wb_vol_kernel(otiles, vktw), dim3(planes, vgroups), dim3(BLOCK), vlds, st,
                           x, gy, (float *)workspace, d, K);
        WM_LAUNCH_CHECK("conv3d_bwd_wb_vol_kernel");
        // (the 16-byte form needs enough entries to fill the chip with 256-entry workgroups:
        // layer 1's 11 008 entries x 224 slabs took 9-12 us on 43 of them, 6.9 with the scalar form)
        if (nent % 4 == 0 && (((uintptr_t)workspace) & 15) == 0 && nent >= 256 * wm_num_cus())
            hipLaunchKernelGGL(conv3d_reduce16v_kernel, dim3((nent + 255) / 256), dim3(1024), 0, st,
                               (const float *)workspace, planes, nent, nw, gw, gb);
        else
            hipLaunchKernelGGL(conv3d_reduce16_kernel, dim3((nent + 63) / 64), dim3(1024), 0, st,
                               (const float *)workspace, planes, nent, nw, gw, gb);
        WM_LAUNCH_CHECK("conv3d_reduce16_kernel");
        return WM_OK;
    }
    if (wb_wide_shape(Ntot, K)) {
        const int grid = wm_num_cus() * WB_WIDE_BPC;
        WM_HIP(hipMemsetAsync(workspace, 0, (size_t)grid * nent * sizeof(double), st));
        const int ktw = (ktiles + WAVES - 1) / WAVES;          // tap tiles per wave, 1..8
#define WB_WIDE(OT_, KTW_)                                                                     \
    hipLaunchKernelGGL((conv3d_bwd_wb_wide_kernel<OT_, KTW_>), dim3(grid), dim3(BLOCK), 0, st, \
                       x, gy, (double *)workspace, d, K, Ntot, ktiles)
#define WB_WIDE_K(OT_)                                \
    if (ktw <= 2) WB_WIDE(OT_, 2);                    \
    else if (ktw <= 4) WB_WIDE(OT_, 4);               \
    else if (ktw <= 6) WB_WIDE(OT_, 6);               \
    else WB_WIDE(OT_, 8)
        if (otiles == 1) { WB_WIDE_K(1); }
        else if (otiles == 2) { WB_WIDE_K(2); }
        else { WB_WIDE_K(4); }
#undef WB_WIDE_K
#undef WB_WIDE
        WM_LAUNCH_CHECK("conv3d_bwd_wb_wide_kernel");
        hipLaunchKernelGGL(conv3d_reduce_kernel, dim3((nent + 63) / 64), dim3(64), 0, st,
                           (const double *)workspace, grid, nent, nw, gw, gb);
        WM_LAUNCH_CHECK("conv3d_reduce_kernel");
        return WM_OK;
    }
    hipLaunchKernelGGL(conv3d_bwd_wb_kernel, dim3((tiles + WAVES - 1) / WAVES, splits), dim3(BLOCK),
                       0, st, x, gy, (double *)workspace, d, K, Ntot, ktiles, tiles);
    WM_LAUNCH_CHECK("conv3d_bwd_wb_kernel");
    hipLaunchKernelGGL(conv3d_reduce_kernel, dim3((nent + 63) / 64), dim3(64), 0, st,
                       (const double *)workspace, splits, nent, nw, gw, gb);
    WM_LAUNCH_CHECK("conv3d_reduce_kernel");
    return WM_OK;
}

namespace {

int pool_fwd_launch(const float *x0, const float *x1, float *y0, float *y1, unsigned char *w0,
                    unsigned char *w1, int B, int A1, int A2, int A3, int C, long long y_bstride,
                    void *stream, const char *who)
{
    WM_REQUIRE(x0 && y0 && (!x1 || y1), WM_ERR_INVALID, "%s: null pointer", who);
    WM_REQUIRE(B > 0 && A1 >= 2 && A2 >= 2 && A3 >= 2 && C > 0, WM_ERR_INVALID,
               "%s: pooled axes need at least 2 entries (%d,%d,%d)", who, A1, A2, A3);
    const long long per = (long long)(A1 / 2) * (A2 / 2) * (A3 / 2) * C;
    WM_REQUIRE(y_bstride >= per, WM_ERR_INVALID, "%s: output sample stride %lld below the sample size %lld",
               who, y_bstride, per);
    const int cus = wm_num_cus();
    WM_REQUIRE(cus > 0, WM_ERR_NO_DEVICE, "%s: no HIP device", who);
    const long long total = (long long)B * per;
    long long blocks = (total + BLOCK - 1) / BLOCK;
    if (blocks > (long long)cus * 16) blocks = (long long)cus * 16;
    const PoolPair pp = {{x0, x1}, {y0, y1}, {w0, w1}};
    const dim3 grid((unsigned)blocks, x1 ? 2 : 1);
    // 32-bit indices: the input (8 x the output) must stay below 2^31 elements
    if ((long long)B * A1 * A2 * A3 * C + (long long)cus * 16 * BLOCK < (1ll << 31))
        hipLaunchKernelGGL(maxpool3d_fwd_kernel<unsigned>, grid, dim3(BLOCK), 0, (hipStream_t)stream, pp,
                           A1, A2, A3, C, total, y_bstride);
    else
        hipLaunchKernelGGL(maxpool3d_fwd_kernel<long long>, grid, dim3(BLOCK), 0, (hipStream_t)stream,
                           pp, A1, A2, A3, C, total, y_bstride);
    WM_LAUNCH_CHECK("maxpool3d_fwd_kernel");
    return WM_OK;
}

int pool_bwd_launch(const float *gy0, const float *gy1, const unsigned char *w0,
                    const unsigned char *w1, float *gx0, float *gx1, int B, int A1, int A2, int A3,
                    int C, long long gy_bstride, void *stream, const char *who)
{
    WM_REQUIRE(gy0 && w0 && gx0 && (!gy1 || (w1 && gx1)), WM_ERR_INVALID, "%s: null pointer", who);
    WM_REQUIRE(B > 0 && A1 >= 2 && A2 >= 2 && A3 >= 2 && C > 0, WM_ERR_INVALID,
               "%s: pooled axes need at least 2 entries (%d,%d,%d)", who, A1, A2, A3);
    const long long per = (long long)(A1 / 2) * (A2 / 2) * (A3 / 2) * C;
    WM_REQUIRE(gy_bstride >= per, WM_ERR_INVALID, "%s: gradient sample stride %lld below the sample size %lld",
               who, gy_bstride, per);
    const int cus = wm_num_cus();
    WM_REQUIRE(cus > 0, WM_ERR_NO_DEVICE, "%s: no HIP device", who);
    const long long total = (long long)B * A1 * A2 * A3 * C;
    long long blocks = (total + BLOCK - 1) / BLOCK;
    if (blocks > (long long)cus * 16) blocks = (long long)cus * 16;
    const PoolPairBwd pp = {{gy0, gy1}, {w0, w1}, {gx0, gx1}};
    const dim3 grid((unsigned)blocks, gy1 ? 2 : 1);
    if (total + (long long)cus * 16 * BLOCK < (1ll << 31))
        hipLaunchKernelGGL(maxpool3d_bwd_kernel<unsigned>, grid, dim3(BLOCK), 0, (hipStream_t)stream, pp,
                           A1, A2, A3, C, total, gy_bstride);
    else
        hipLaunchKernelGGL(maxpool3d_bwd_kernel<long long>, grid, dim3(BLOCK), 0, (hipStream_t)stream,
                           pp, A1, A2, A3, C, total, gy_bstride);
    WM_LAUNCH_CHECK("maxpool3d_bwd_kernel");
    return WM_OK;
}

}  // namespace

extern "C" int wm_maxpool3d_fwd_f32(const float *x, float *y, unsigned char *which, int B, int A1,
                                    int A2, int A3, int C, void *stream)
{
    const long long per = (long long)(A1 / 2) * (A2 / 2) * (A3 / 2) * C;
    return pool_fwd_launch(x, nullptr, y, nullptr, which, nullptr, B, A1, A2, A3, C, per, stream,
                           "wm_maxpool3d_fwd_f32");
}

extern "C" int wm_maxpool3d_bwd_f32(const float *gy, const unsigned char *which, float *gx, int B,
                                    int A1, int A2, int A3, int C, void *stream)
{
    const long long per = (long long)(A1 / 2) * (A2 / 2) * (A3 / 2) * C;
    return pool_bwd_launch(gy, nullptr, which, nullptr, gx, nullptr, B, A1, A2, A3, C, per, stream,
                           "wm_maxpool3d_bwd_f32");
}

extern "C" int wm_maxpool3d_fwd_pair_f32(const float *x0, const float *x1, float *y0, float *y1,
                                         unsigned char *which0, unsigned char *which1, int B, int A1,
                                         int A2, int A3, int C, int64_t y_batch_stride, void *stream)
{
    WM_REQUIRE(x1 && y1, WM_ERR_INVALID, "wm_maxpool3d_fwd_pair_f32: null pointer");
    return pool_fwd_launch(x0, x1, y0, y1, which0, which1, B, A1, A2, A3, C, (long long)y_batch_stride,
                           stream, "wm_maxpool3d_fwd_pair_f32");
}

extern "C" int wm_maxpool3d_bwd_pair_f32(const float *gy0, const float *gy1,
                                         const unsigned char *which0, const unsigned char *which1,
                                         float *gx0, float *gx1, int B, int A1, int A2, int A3, int C,
                                         int64_t gy_batch_stride, void *stream)
{
    WM_REQUIRE(gy1 && which1 && gx1, WM_ERR_INVALID, "wm_maxpool3d_bwd_pair_f32: null pointer");
    return pool_bwd_launch(gy0, gy1, which0, which1, gx0, gx1, B, A1, A2, A3, C,
                           (long long)gy_batch_stride, stream, "wm_maxpool3d_bwd_pair_f32");
}
