// linmetrics.hip -- what model.fit / model.evaluate report per epoch for the linear stacks: the loss
// (mean squared error) and the compiled metrics mae / msle of the current coefficients over the rows
// (multipleRegression.py:292-296: compile(loss=mean_squared_error, weighted_metrics=['mae', 'msle']);
// singleRegression.py:90-94), as ONE streaming pass: predictions are formed in registers and never
// written.  Until round 3 the drop-ins' metrics pass was wm_linear_predict_f32 (one thread per output,
// F float64 FMAs on strided loads) followed by eight float64 framework passes over the predictions --
// an epoch's metrics cost ~25x its Gram pass.
//   linear_metrics_kernel<PT, VEC4>   thread per row; beta (float64, (F+1) x P) in LDS, read as
//                                     broadcasts; float64 accumulation of the dot products exactly as
//                                     wm_linear_predict_f32 forms them; the prediction is rounded to
//                                     float32 (the model's output type) before the residual
//   residual_sums_kernel              the same sums from predictions that already exist (large F x P:
//                                     the caller forms them with a library GEMM)
//   partials are per workgroup, float64, combined in a fixed order by metrics_finish_kernel and ADDED
//   to sums[4] = {sum d^2, sum |d|, sum (log1p(max(pred,eps)) - log1p(max(y,eps)))^2, count}.
#include <type_traits>

#include "common.h"
#include "cube_walk.h"
#include "small_stream.h"

namespace {

constexpr int BLOCK = 256;
constexpr int LM_MAX_BETA = 4096;          // (F + 1) * PT doubles in LDS: 32 KB
constexpr int LM_MAX_P = 32;
constexpr float LM_CLAMP_HI = 1.0e37f;     // msle operands are clamped to [eps, 1e37]: 1 / (2 + a + b) stays a normal number

// log1p(a) - log1p(b) = log((1 + a) / (1 + b)) = 2 atanh(s), s = (a - b) / (2 + a + b), for
// a, b in [eps, 1e37]: ONE reciprocal and the odd series (|s| <= 1/5: the s^11 term is below 1e-8 of
// the result).  Beyond that -- the two values differ by a factor of 1.5 and more, which no lane of a
// wave sees on a fitted model's rows -- the difference of the two hardware logarithms, which is then
// at least 0.4 and takes the rounding of its terms (~1e-7 of each), under a branch the wave skips.
// (Round 3, first version: log(1 + (a - b) / (1 + b)) computed beside the series for every output and
// selected -- two reciprocals and a logarithm, 12 quarter-rate issue slots of ~45 -- and that form
// returned 0 for b beyond 2^24 a.)  Within ~3e-7 relative of the float64 value for operands of order
// one (keras' own float32 log(x + 1) is not).
__device__ __forceinline__ float lm_log_ratio(float a, float b)
{
    const float u = a - b;
    const float s = u * __builtin_amdgcn_rcpf((2.0f + a) + b);
    const float s2 = s * s;
    float poly = fmaf(s2, 1.0f / 9.0f, 1.0f / 7.0f);
    poly = fmaf(s2, poly, 1.0f / 5.0f);
    poly = fmaf(s2, poly, 1.0f / 3.0f);
    poly = fmaf(s2, poly, 1.0f);
    float r = (s + s) * poly;
    if (__builtin_expect(!(fabsf(s) <= 0.2f), 0))
        r = (__builtin_amdgcn_logf(1.0f + a) - __builtin_amdgcn_logf(1.0f + b)) * 0.69314718055994530942f;
    return r;
}

__device__ __forceinline__ void lm_residual(float pred, float y, bool msle, float eps, double &sse,
                                            double &sae, double &sle)
{
    const double d = (double)pred - (double)y;
    sse = fma(d, d, sse);
    sae += fabs(d);
    if (msle) {
        const float e = lm_log_ratio(__builtin_amdgcn_fmed3f(pred, eps, LM_CLAMP_HI),
                                     __builtin_amdgcn_fmed3f(y, eps, LM_CLAMP_HI));
        sle = fma((double)e, (double)e, sle);
    }
}

// The residual sums of the N outputs of one row.  The rare far form is ONE branch behind all the
// near forms: a branch per output makes hipcc sink each output's dot product down to its use, while
// the scalar loads of beta stay at the top and are parked in VGPR lanes in between.
template <int N, bool MSLE>
__device__ __forceinline__ void lm_residuals(const float (&pred)[N], const float (&y)[N], float eps,
                                             double &sse, double &sae, double &sle)
{
#pragma unroll
    for (int p = 0; p < N; ++p) {
        const double d = (double)pred[p] - (double)y[p];
        sse = fma(d, d, sse);
        sae += fabs(d);
    }
    if (MSLE) {
        // clamped to [eps, 1e37] by one v_med3 each (fmaxf costs two instructions: it quiets its
        // operand first), so 2 + a + b and its reciprocal stay normal numbers; values beyond 1e37 count as 1e37
        float e[N];
        float smax = 0.f;                           // one v_max per output instead of a flag each
#pragma unroll
        for (int p = 0; p < N; ++p) {
            const float a = __builtin_amdgcn_fmed3f(pred[p], eps, LM_CLAMP_HI);
            const float b = __builtin_amdgcn_fmed3f(y[p], eps, LM_CLAMP_HI);
            const float u = a - b;
            const float s = u * __builtin_amdgcn_rcpf((2.0f + a) + b);
            const float s2 = s * s;
            float poly = fmaf(s2, 1.0f / 9.0f, 1.0f / 7.0f);
            poly = fmaf(s2, poly, 1.0f / 5.0f);
            poly = fmaf(s2, poly, 1.0f / 3.0f);
            poly = fmaf(s2, poly, 1.0f);
            e[p] = (s + s) * poly;
            smax = __builtin_fmaxf(smax, fabsf(s));
        }
        if (__builtin_expect(!(smax <= 0.2f), 0)) {
#pragma unroll
            for (int p = 0; p < N; ++p)
                e[p] = lm_log_ratio(__builtin_amdgcn_fmed3f(pred[p], eps, LM_CLAMP_HI),
                                    __builtin_amdgcn_fmed3f(y[p], eps, LM_CLAMP_HI));
        }
        // the squares of a group (N <= 16 terms of similar size) in float32, one float64 add per group
        float q = 0.f;
#pragma unroll
        for (int p = 0; p < N; ++p) q = fmaf(e[p], e[p], q);
        sle += (double)q;
    }
}

// Four outputs of one MFMA accumulator: the near form summed into sq for every output, smax = the
// largest |s| met (beyond 1/5 one of them needs the other form; lm_far4 then returns the sum of the
// four squares in full).
template <bool MSLE>
__device__ __forceinline__ void lm_residuals_near4(const f32x4 pred, const f32x4 y, float eps,
                                                   double &sse, double &sae, float &sq, float &smax)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double d = (double)pred[p] - (double)y[p];
        sse = fma(d, d, sse);
        sae += fabs(d);
    }
    if (MSLE) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float a = __builtin_amdgcn_fmed3f(pred[p], eps, LM_CLAMP_HI);
            const float b = __builtin_amdgcn_fmed3f(y[p], eps, LM_CLAMP_HI);
            const float s = (a - b) * __builtin_amdgcn_rcpf((2.0f + a) + b);
            const float s2 = s * s;
            float poly = fmaf(s2, 1.0f / 9.0f, 1.0f / 7.0f);
            poly = fmaf(s2, poly, 1.0f / 5.0f);
            poly = fmaf(s2, poly, 1.0f / 3.0f);
            poly = fmaf(s2, poly, 1.0f);
            const float e = (s + s) * poly;
            sq = fmaf(e, e, sq);
            smax = __builtin_fmaxf(smax, fabsf(s));          // one v_max per output; a flag per output
        }                                                    // became compare + select + 16-bit or
    }
}

// Both forms computed and selected, no branch: for the repeat pass of the wide kernel.
__device__ __forceinline__ float lm_log_ratio_select(float a, float b)
{
    const float s = (a - b) * __builtin_amdgcn_rcpf((2.0f + a) + b);
    const float s2 = s * s;
    float poly = fmaf(s2, 1.0f / 9.0f, 1.0f / 7.0f);
    poly = fmaf(s2, poly, 1.0f / 5.0f);
    poly = fmaf(s2, poly, 1.0f / 3.0f);
    poly = fmaf(s2, poly, 1.0f);
    const float near = (s + s) * poly;
    const float far = (__builtin_amdgcn_logf(1.0f + a) - __builtin_amdgcn_logf(1.0f + b)) * 0.69314718055994530942f;
    return fabsf(s) <= 0.2f ? near : far;
}

__device__ __forceinline__ void lm_block_out(double sse, double sae, double sle, double *partial,
                                             bool only_sle = false)
{
    __shared__ double red[BLOCK / 64][3];
    sse = wave_sum(sse);
    sae = wave_sum(sae);
    sle = wave_sum(sle);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = sse; red[wv][1] = sae; red[wv][2] = sle; }
    __syncthreads();
    if (threadIdx.x < 3 && (!only_sle || threadIdx.x == 2)) {
        double s = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) s += red[w][threadIdx.x];
        partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;       // entry-major
    }
}

template <int PT, int VEC>          // VEC: floats per load of a row of X (4: F % 4 == 0, 2: F % 2 == 0)
__global__ __launch_bounds__(BLOCK) void linear_metrics_kernel(const float *__restrict__ X,
                                                              const float *__restrict__ Y,
                                                              const double *__restrict__ beta,
                                                              long long n, int F, int P, int msle,
                                                              float eps, double *__restrict__ partial)
{
    extern __shared__ __align__(16) double sb[];               // [F + 1][PT], columns >= P zero
    for (int e = threadIdx.x; e < (F + 1) * PT; e += BLOCK) {
        const int f = e / PT, p = e - f * PT;
        sb[e] = p < P ? beta[f * P + p] : 0.0;
    }
    __syncthreads();
    double sse = 0.0, sae = 0.0, sle = 0.0;
    for (long long row = (long long)blockIdx.x * BLOCK + threadIdx.x; row < n;
         row += (long long)gridDim.x * BLOCK) {
        double s[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) s[p] = sb[F * PT + p];
        const float *xr = X + row * F;
        if (VEC == 4) {
            for (int f0 = 0; f0 < F; f0 += 4) {
                const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(xr + f0));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double xd = (double)x[j];
                    const double *b = sb + (f0 + j) * PT;
#pragma unroll
                    for (int p = 0; p < PT; ++p) s[p] = fma(xd, b[p], s[p]);
                }
            }
        } else if (VEC == 2) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            for (int f0 = 0; f0 < F; f0 += 2) {
                const f32x2 x = __builtin_nontemporal_load(reinterpret_cast<const f32x2 *>(xr + f0));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double xd = (double)x[j];
                    const double *b = sb + (f0 + j) * PT;
#pragma unroll
                    for (int p = 0; p < PT; ++p) s[p] = fma(xd, b[p], s[p]);
                }
            }
        } else {
            for (int f = 0; f < F; ++f) {
                const double xd = (double)xr[f];
                const double *b = sb + f * PT;
#pragma unroll
                for (int p = 0; p < PT; ++p) s[p] = fma(xd, b[p], s[p]);
            }
        }
        const float *yr = Y + row * P;
#pragma unroll
        for (int p = 0; p < PT; ++p)
            if (p < P) lm_residual((float)s[p], yr[p], msle != 0, eps, sse, sae, sle);
    }
    lm_block_out(sse, sae, sle, partial);
}

// Narrow rows (F + P <= 15: the reference's own multipleRegression fit is F = 2, P = 1).  The direct
// kernel above keeps ONE row per lane in flight and waits for X, then for Y (F=2, P=1: 4.2 TB/s).
// This one is gram_small_kernel's memory side (small_stream.h): a lane owns R consecutive rows of each
// of U steps, 16-byte non-temporal loads, batches dealt round-robin to the waves, the next batch's
// loads issued before the current batch's arithmetic.  beta sits in SGPRs (scalar loads at
// uniform addresses); the float64 dot products run in the direct kernel's order (intercept first,
// then features in order), so the predictions are the same bits.
// Metrics only: wider rows of the family F = 2p, P = p that no other kernel here reads well (F not a
// multiple of 4, or rows of beta that fragment the SGPR file in the tile kernel) -- without the Gram
// kernel's F + P <= 15 limit, because this kernel keeps no accumulator per product.  F=14,P=7: 4.3
// (lane-per-row kernel) -> 1.6 ms per 1e8 rows.
#define WM_LM_EXTRA_SHAPES(X_) X_(14, 7) X_(18, 9) X_(22, 11) X_(20, 10)   /* all but (14,7): behind the float64-MFMA kernel */

template <int F, int P>
struct LmSmallCfg {
    static constexpr int M = F + P;
    static constexpr int R0 = (F % 4 == 0) ? 1 : ((F % 2 == 0) ? 2 : 4);
    static constexpr int U0 = 48 / (R0 * M);
    static constexpr int U = U0 < 1 ? 1 : (U0 > 8 ? 8 : U0);
    using S = wm_small::Stream<F, P, U>;
};

// float32 prediction of one row: float64 dot product, intercept first, then the features in order
template <int F, int P>
__device__ __forceinline__ void lm_small_predict(const float (&x)[F], const double *__restrict__ bt,
                                                 float *__restrict__ pred)
{
    double s[P];
#pragma unroll
    for (int p = 0; p < P; ++p) s[p] = bt[F * P + p];
#pragma unroll
    for (int m = 0; m < F; ++m) {
        const double xd = (double)x[m];
#pragma unroll
        for (int p = 0; p < P; ++p) s[p] = fma(xd, bt[m * P + p], s[p]);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) pred[p] = (float)s[p];
}

template <int F, int P, bool MSLE>
__device__ __forceinline__ void lm_small_row(const float (&x)[F], const float (&y)[P],
                                             const double *__restrict__ bt, float eps, double &sse,
                                             double &sae, double &sle)
{
    float pred[P];
    lm_small_predict<F, P>(x, bt, pred);
    lm_residuals<P, MSLE>(pred, y, eps, sse, sae, sle);
}

// residual sums of outputs [O0, TOT) in groups of up to 8 (one rare far branch per group)
template <int O0, int TOT, bool MSLE>
__device__ __forceinline__ void lm_resid_groups(const float (&pred)[TOT], const float (&y)[TOT], float eps,
                                                double &sse, double &sae, double &sle)
{
    if constexpr (O0 < TOT) {
        constexpr int N = TOT - O0 < 8 ? TOT - O0 : 8;
        lm_residuals<N, MSLE>(reinterpret_cast<const float(&)[N]>(pred[O0]),
                              reinterpret_cast<const float(&)[N]>(y[O0]), eps, sse, sae, sle);
        lm_resid_groups<O0 + N, TOT, MSLE>(pred, y, eps, sse, sae, sle);
    }
}

// A batch: the predictions of ALL its rows first -- so that the scalar loads of beta are shared by the
// rows and dead before the first branch (row by row, beta stayed live across every row's far branch:
// 550-1554 spilled SGPRs at F=14,P=7 .. F=22,P=11) -- then the residual sums in groups of up to 8
// outputs, one (rare) far branch per group instead of one per output, so that the outputs' dependent
// chains (reciprocal, series, float64 square) overlap.
template <int F, int P, bool MSLE>
__device__ __forceinline__ void lm_small_consume(const wm_small::Buf<typename LmSmallCfg<F, P>::S> &b,
                                                 const double *__restrict__ bt, float eps,
                                                 double &sse, double &sae, double &sle)
{
    using S = typename LmSmallCfg<F, P>::S;
    // beta beyond ~40 doubles does not fit the SGPR file beside the rest: an opaque offset per batch
    // (the pointer must stay the kernel argument, or the loads turn into flat vector loads) keeps
    // the scalar loads next to their use instead of hoisted and spilled
    if constexpr ((F + 1) * P > 40) {
        int zero = 0;
        asm volatile("" : "+s"(zero));
        bt += zero;
    }
    constexpr int ROWS = S::U * S::R;                       // rows of the lane in a batch
    float pred[ROWS * P], y[ROWS * P];
    if constexpr ((F + 1) * P > 40) {
        // beta streams through the SGPR file one row (P doubles) ahead of its use, features outermost
        // and the batch's rows inside: left to itself hipcc issues all (F + 1) P scalar loads at the
        // top of the block and parks them in VGPR lanes (550-1554 spilled SGPRs at F=14,P=7 ..
        // F=22,P=11); the fences keep one row in flight while the previous one is used
        double sacc[ROWS * P], bc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) bc[p] = bt[F * P + p];
#pragma unroll
        for (int q = 0; q < ROWS; ++q)
#pragma unroll
            for (int p = 0; p < P; ++p) sacc[q * P + p] = bc[p];
#pragma unroll
        for (int p = 0; p < P; ++p) bc[p] = bt[p];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            double bn[P];
#pragma unroll
            for (int p = 0; p < P; ++p) bn[p] = bt[(f + 1 < F ? f + 1 : f) * P + p];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < ROWS; ++q) {
                const double xd = (double)b.xe(q / S::R, q % S::R, f);
#pragma unroll
                for (int p = 0; p < P; ++p) sacc[q * P + p] = fma(xd, bc[p], sacc[q * P + p]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < P; ++p) bc[p] = bn[p];
        }
#pragma unroll
        for (int e = 0; e < ROWS * P; ++e) pred[e] = (float)sacc[e];
    } else {
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            float x[F];
#pragma unroll
            for (int m = 0; m < F; ++m) x[m] = b.xe(q / S::R, q % S::R, m);
            lm_small_predict<F, P>(x, bt, pred + q * P);
        }
    }
#pragma unroll
    for (int q = 0; q < ROWS; ++q)
#pragma unroll
        for (int m = 0; m < P; ++m) y[q * P + m] = b.ye(q / S::R, q % S::R, m);
    lm_resid_groups<0, ROWS * P, MSLE>(pred, y, eps, sse, sae, sle);
}

template <int F, int P, bool MSLE>
__global__ __launch_bounds__(BLOCK, 2) void linear_metrics_small_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const double *__restrict__ beta, long long n,
    float eps, double *__restrict__ partial)
{
    using S = typename LmSmallCfg<F, P>::S;
    using DN = wm_small::Dense<S>;
    __shared__ __align__(16) char turn[BLOCK / 64][DN::BYTES];     // wave-private (small_stream.h)
    char *tw = turn[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const long long nwaves = (long long)gridDim.x * (BLOCK / 64);
    const long long wave_id = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const double *__restrict__ bt = beta;      // uniform addresses: scalar loads, hoisted when they fit
    double sse = 0.0, sae = 0.0, sle = 0.0;
    const long long nb = n / S::ROWS_BATCH;
    if (wave_id < nb) {
        const long long spare = wave_id % nb;          // the clamped batch of an out-of-range fetch
        wm_small::Buf<S> ba, bb;
        wm_small::fetch_dense<S>(ba, X, Y, wave_id, lane);
        long long b = wave_id;
        for (; b + nwaves < nb; b += 2 * nwaves) {
            wm_small::fetch_dense<S>(bb, X, Y, b + nwaves, lane);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DN::ANY) wm_small::untangle<S>(ba, tw, lane);
            lm_small_consume<F, P, MSLE>(ba, bt, eps, sse, sae, sle);
            __builtin_amdgcn_sched_barrier(0);
            wm_small::fetch_dense<S>(ba, X, Y, b + 2 * nwaves < nb ? b + 2 * nwaves : spare, lane);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DN::ANY) wm_small::untangle<S>(bb, tw, lane);
            lm_small_consume<F, P, MSLE>(bb, bt, eps, sse, sae, sle);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (b < nb) {
            if constexpr (DN::ANY) wm_small::untangle<S>(ba, tw, lane);
            lm_small_consume<F, P, MSLE>(ba, bt, eps, sse, sae, sle);
        }
    }
    // the rows behind the last full batch (fewer than ROWS_BATCH): one row per lane
    for (long long r0 = nb * S::ROWS_BATCH + wave_id * 64; r0 < n; r0 += nwaves * 64) {
        const long long row = r0 + lane;
        if (row < n) {
            float x[F], y[P];
#pragma unroll
            for (int m = 0; m < F; ++m) x[m] = X[row * F + m];
#pragma unroll
            for (int m = 0; m < P; ++m) y[m] = Y[row * P + m];
            lm_small_row<F, P, MSLE>(x, y, bt, eps, sse, sae, sle);
        }
    }
    lm_block_out(sse, sae, sle, partial);
}

// Rows of 32 bytes and more: a lane that walks its own row makes every load instruction of the wave
// touch 64 different cache lines (the direct kernel ran F=16,P=8 at 1.6 TB/s).  Here each wave reads
// its tile of 64 rows as ONE contiguous run, 16 bytes per lane per instruction, and turns it by 90
// degrees through a private LDS region: piece q of the run belongs to row q / (F/4); rows are stored
// 4F + 16 bytes apart, so the per-lane row reads that follow (ds_read_b128, lane = row) hit all 64
// banks.  Wave-private, so no workgroup barrier in the loop: LDS executes a wave's instructions in
// order.  Y goes the same way when its rows are whole 16-byte pieces (VECY).
// SB (P == PT): beta is read straight from global memory at wave-uniform addresses -- scalar loads
// into SGPRs, which the float64 FMAs take as an operand -- instead of 4 PT/2 broadcast LDS reads per
// feature: the LDS pipe (76 16-byte operations per 64 rows at F=16, P=8, of which 64 were beta) was
// the tightest of the kernel's three limits.
template <int PT, bool VECY, bool SB>
__global__ __launch_bounds__(BLOCK) void linear_metrics_staged_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const double *__restrict__ beta, long long n,
    int F, int P, int msle, float eps, int beta_bytes, double *__restrict__ partial)
{
    extern __shared__ __align__(16) double sb[];               // [F + 1][PT], then 4 wave regions
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int QX = F >> 2, QY = P >> 2;
    const int xstride = 4 * F + 16, ystride = 4 * P + 16;
    const int wave_bytes = 64 * xstride + (VECY ? 64 * ystride : 0);
    char *xs = reinterpret_cast<char *>(sb) + beta_bytes + wv * wave_bytes;
    char *ys = xs + 64 * xstride;
    if (!SB) {
        for (int e = threadIdx.x; e < (F + 1) * PT; e += BLOCK) {
            const int f = e / PT, p = e - f * PT;
            sb[e] = p < P ? beta[f * P + p] : 0.0;
        }
        __syncthreads();
    }
    // piece q = lane + 64 k of a tile lives in row q / Q at piece q % Q: walked incrementally
    const int dx = 64 / QX, mx = 64 % QX, r0x = lane / QX, p0x = lane % QX;
    const int dy = VECY ? 64 / QY : 0, my = VECY ? 64 % QY : 0, r0y = VECY ? lane / QY : 0,
              p0y = VECY ? lane % QY : 0;
    const char *xrow = xs + lane * xstride;
    const char *yrow = ys + lane * ystride;
    double sse = 0.0, sae = 0.0, sle = 0.0;
    const long long tiles = (n + 63) >> 6;
    for (long long t = (long long)blockIdx.x * (BLOCK / 64) + wv; t < tiles;
         t += (long long)gridDim.x * (BLOCK / 64)) {
        const long long row0 = t << 6;
        const int rows = (int)(n - row0 < 64 ? n - row0 : 64);
        {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(X + row0 * F);
            const int last = rows * QX - 1;
            int r = r0x, pc = p0x;
            for (int k0 = 0; k0 < QX; k0 += 4) {
                f32x4 v[4];
                int off[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = lane + 64 * (k0 + u);
                    v[u] = __builtin_nontemporal_load(src + (q < last ? q : last));
                    off[u] = r * xstride + pc * 16;
                    r += dx;
                    pc += mx;
                    if (pc >= QX) { pc -= QX; ++r; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k0 + u < QX) *reinterpret_cast<f32x4 *>(xs + off[u]) = v[u];
            }
        }
        if (VECY) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(Y + row0 * P);
            const int last = rows * QY - 1;
            int r = r0y, pc = p0y;
            for (int k0 = 0; k0 < QY; k0 += 2) {
                f32x4 v[2];
                int off[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = lane + 64 * (k0 + u);
                    v[u] = __builtin_nontemporal_load(src + (q < last ? q : last));
                    off[u] = r * ystride + pc * 16;
                    r += dy;
                    pc += my;
                    if (pc >= QY) { pc -= QY; ++r; }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (k0 + u < QY) *reinterpret_cast<f32x4 *>(ys + off[u]) = v[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
        double s[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) s[p] = SB ? beta[F * PT + p] : sb[F * PT + p];
        for (int f0 = 0; f0 < F; f0 += 4) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(xrow + f0 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double xd = (double)x[j];
                const double *b = (SB ? beta : sb) + (f0 + j) * PT;
#pragma unroll
                for (int p = 0; p < PT; ++p) s[p] = fma(xd, b[p], s[p]);
            }
        }
        if (lane < rows) {
            if (VECY) {
#pragma unroll
                for (int p4 = 0; p4 < PT / 4; ++p4)
                    if (p4 * 4 < P) {
                        const f32x4 y = *reinterpret_cast<const f32x4 *>(yrow + p4 * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            lm_residual((float)s[p4 * 4 + j], y[j], msle != 0, eps, sse, sae, sle);
                    }
            } else {
                const float *yr = Y + (row0 + lane) * P;
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    if (p < P) lm_residual((float)s[p], yr[p], msle != 0, eps, sse, sae, sle);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    lm_block_out(sse, sae, sle, partial);
}

// The staged kernel with F = 4 QX and P compile-time constants and the NEXT tile's global loads in
// flight while this tile is computed (the staged kernel above loads, waits, computes: with five
// waves per SIMD and ~1300 cycles of arithmetic per tile the loads' latency was not covered).  A
// lane's pieces of a tile (16 bytes at q = lane + 64 k) always land at the same LDS offsets, so those
// are computed once.  beta comes by scalar loads at compile-time offsets.  Y rows that are not whole
// 16-byte pieces are read by their own lane (8-byte loads for even P).
// (P = 10 and 12 were tried: rows of beta that are not 8 or 16 doubles fragment the SGPR file, 285-425
// spilled SGPRs, 5.1 ms against the staged kernel's 2.75 at F=20, P=10 -- those stay there.)
#define WM_LM_TILE_SHAPES(X_) X_(4, 8) X_(3, 6) X_(8, 16) X_(4, 4) X_(2, 8)

template <int QX, int P>
struct LmTile {
    static constexpr int F = 4 * QX;
    static constexpr bool VECY = P % 4 == 0;
    static constexpr int QY = VECY ? P / 4 : 0;
    static constexpr int XSTRIDE = 4 * F + 16, YSTRIDE = 4 * P + 16;          // bytes between LDS rows
    static constexpr int WAVE_BYTES = 64 * XSTRIDE + (VECY ? 64 * YSTRIDE : 0);
    static constexpr int YN = VECY ? 1 : P;
    f32x4 x[QX];
    f32x4 y4[VECY ? QY : 1];
    float y1[YN];
};

template <int QX, int P>
__device__ __forceinline__ void lm_tile_load(LmTile<QX, P> &tl, const float *__restrict__ X,
                                             const float *__restrict__ Y, long long t, long long n,
                                             int lane)
{
    using T = LmTile<QX, P>;
    const long long row0 = t << 6;
    const int rows = (int)(n - row0 < 64 ? n - row0 : 64);
    const f32x4 *src = reinterpret_cast<const f32x4 *>(X + row0 * T::F);
    const int last = rows * QX - 1;
#pragma unroll
    for (int k = 0; k < QX; ++k) {
        const int q = lane + 64 * k;
        tl.x[k] = __builtin_nontemporal_load(src + (q < last ? q : last));
    }
    if constexpr (T::VECY) {
        const f32x4 *ysrc = reinterpret_cast<const f32x4 *>(Y + row0 * P);
        const int ylast = rows * T::QY - 1;
#pragma unroll
        for (int k = 0; k < T::QY; ++k) {
            const int q = lane + 64 * k;
            tl.y4[k] = __builtin_nontemporal_load(ysrc + (q < ylast ? q : ylast));
        }
    } else {
        const float *yr = Y + (row0 + (lane < rows ? lane : rows - 1)) * P;
        if constexpr (P % 2 == 0) {
#pragma unroll
            for (int j = 0; j < P / 2; ++j) {
                const f2u v = __builtin_nontemporal_load(reinterpret_cast<const f2u *>(yr) + j);
                tl.y1[2 * j] = v[0];
                tl.y1[2 * j + 1] = v[1];
            }
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) tl.y1[j] = __builtin_nontemporal_load(yr + j);
        }
    }
}

template <int QX, int P, bool MSLE>
__global__ __launch_bounds__(BLOCK) void linear_metrics_tile_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const double *__restrict__ beta, long long n,
    float eps, double *__restrict__ partial)
{
    using T = LmTile<QX, P>;
    constexpr int F = T::F;
    extern __shared__ __align__(16) char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char *xs = lds + wv * T::WAVE_BYTES;
    char *ys = xs + 64 * T::XSTRIDE;
    int offx[QX], offy[T::VECY ? T::QY : 1];
#pragma unroll
    for (int k = 0; k < QX; ++k) {
        const int q = lane + 64 * k;
        offx[k] = (q / QX) * T::XSTRIDE + (q % QX) * 16;
    }
    if constexpr (T::VECY) {
#pragma unroll
        for (int k = 0; k < T::QY; ++k) {
            const int q = lane + 64 * k;
            offy[k] = (q / T::QY) * T::YSTRIDE + (q % T::QY) * 16;
        }
    }
    const char *xrow = xs + lane * T::XSTRIDE;
    const char *yrow = ys + lane * T::YSTRIDE;
    double sse = 0.0, sae = 0.0, sle = 0.0;
    const long long tiles = (n + 63) >> 6;
    const long long stride = (long long)gridDim.x * (BLOCK / 64);
    long long t = (long long)blockIdx.x * (BLOCK / 64) + wv;
    T cur;
    if (t < tiles) lm_tile_load<QX, P>(cur, X, Y, t, n, lane);
    for (; t < tiles; t += stride) {
        const long long row0 = t << 6;
        const int rows = (int)(n - row0 < 64 ? n - row0 : 64);
#pragma unroll
        for (int k = 0; k < QX; ++k) *reinterpret_cast<f32x4 *>(xs + offx[k]) = cur.x[k];
        float yv[P];
        if constexpr (T::VECY) {
#pragma unroll
            for (int k = 0; k < T::QY; ++k) *reinterpret_cast<f32x4 *>(ys + offy[k]) = cur.y4[k];
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) yv[j] = cur.y1[j];
        }
        __builtin_amdgcn_wave_barrier();
        // the next tile of this wave (its own again behind the last: in bounds, never used)
        lm_tile_load<QX, P>(cur, X, Y, t + stride < tiles ? t + stride : t, n, lane);
        __builtin_amdgcn_sched_barrier(0);
        // (F + 1) P doubles do not fit the SGPR file: left to itself hipcc hoists all the loads out
        // of the tile loop and parks them in VGPR lanes (477 v_readlane per tile at F=16, P=8); an opaque
        // offset (the pointer itself must stay the kernel argument, or the loads turn into flat
        // vector loads) per tile keeps them scalar loads next to their use
        int zero = 0;
        asm volatile("" : "+s"(zero));
        const double *__restrict__ bp = beta + zero;
        double s[P];
#pragma unroll
        for (int p = 0; p < P; ++p) s[p] = bp[F * P + p];
#pragma unroll
        for (int f0 = 0; f0 < F; f0 += 4) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(xrow + f0 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double xd = (double)x[j];
#pragma unroll
                for (int p = 0; p < P; ++p) s[p] = fma(xd, bp[(f0 + j) * P + p], s[p]);
            }
        }
        if constexpr (T::VECY) {
#pragma unroll
            for (int k = 0; k < T::QY; ++k) {
                const f32x4 y = *reinterpret_cast<const f32x4 *>(yrow + k * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) yv[4 * k + j] = y[j];
            }
        }
        float pred[P];
#pragma unroll
        for (int p = 0; p < P; ++p) pred[p] = (float)s[p];
        if (lane < rows) lm_residuals<P, MSLE>(pred, yv, eps, sse, sae, sle);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_wave_barrier();
    }
    lm_block_out(sse, sae, sle, partial);
}

// ---- F = 2p, P = p with 6 < p <= 16 (rows of beta that are neither short enough for the SGPR file
// nor 8 / 16 doubles long): the float64 dot products on the float64 MFMA.  v_mfma_f64_16x16x4_f64
// computes D[16 rows][16 outputs] += A[16 rows][4 features] B[4 features][16 outputs] in float64, so
// the predictions keep the narrow kernels' arithmetic (float64 sums of float32 x float64 products,
// rounded to float32 once) while beta sits in VGPRs -- lane (i, g) holds beta[f][n = i] for its own
// four features of every 16-feature block, 2 ceil(F / 16) x 4 registers, loaded once per kernel -- and
// no scalar register is involved: the scalar-beta kernels above lose 500-1500 SGPRs to VGPR lanes at
// P = 7, 9, 10, 11, 12 (F=24,P=12 ran at 0.37 of HBM, F=20,P=10 at 0.48).
//   A wave's tile of 64 rows is one contiguous run of X and one of Y: copied flat into a wave-private
//   LDS region by 16-byte pieces (the next tile's pieces in flight while this one is computed).  Lane
//   (i, g) reads 16 bytes -- features 16 kb + 4 g .. + 3 of row 16 m + i -- per 16-feature block and
//   row block m, and MFMA step r uses element r of them and register r of beta (the contraction order
//   inside a block is free as long as both operands agree); the last block takes only as many steps
//   as its features need.  D: lane (i, g) holds output n = i of rows
//   16 m + g + 4 r; lanes with i >= P carry zeros.  Targets come back from the LDS copy of Y.
typedef double f64x4 __attribute__((ext_vector_type(4)));
// Measured (6e7 rows): F=24,P=12 0.37 -> 0.65 of HBM, 20,10 0.48 -> 0.65, 22,11 0.30 -> 0.62, 18,9 0.46 ->
// 0.62; without the msle term 0.76-0.81.  Three waves per SIMD (138-148 VGPRs; four spill).  F=14,P=7
// stays on the narrow-row kernel (0.65 there, 0.58 here: 9 of 16 output columns are padding), and
// F=16,P=8 / 32,16 on the scalar-beta tile kernel (equal or faster).
#define WM_LM_MFMA64_SHAPES(X_) X_(24, 12) X_(20, 10) X_(22, 11) X_(18, 9)

template <int F, int P>
struct LmM64 {
    static constexpr int KB = (F + 15) / 16;                // 16-feature blocks
    static constexpr int XP = 64 * F / 4, YP = 64 * P / 4;  // 16-byte pieces of a tile (64 F, 64 P: whole)
    static constexpr int XL = (XP + 63) / 64, YL = (YP + 63) / 64;     // per lane
    static constexpr int WAVE_BYTES = 64 * (F + P) * 4;
    static constexpr bool VEC = F % 4 == 0;                 // vector LDS reads of a row's features
    // MFMA steps of block kb: four, but the last block takes only as many as its features need --
    // lane group g then owns `steps` consecutive features from 16 kb + g * steps (F = 20: 4 + 1 steps
    // instead of 8, F = 24: 4 + 2)
    static constexpr int REM = F - 16 * (KB - 1);
    static constexpr int steps(int kb) { return kb < KB - 1 ? 4 : (REM + 3) / 4; }
};

template <int F, int P, bool MSLE>
__global__ __launch_bounds__(BLOCK, 3) void linear_metrics_mfma64_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const double *__restrict__ beta, long long n,
    float eps, double *__restrict__ partial)
{
    using C = LmM64<F, P>;
    extern __shared__ __align__(16) char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    float *xs = reinterpret_cast<float *>(lds + wv * C::WAVE_BYTES);
    float *ys = xs + 64 * F;
    // beta operands: step (kb, r) contracts features 16 kb + 4 g + r; intercept as the accumulators' start
    double bop[C::KB][4];
#pragma unroll
    for (int kb = 0; kb < C::KB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * kb + C::steps(kb) * g + r;
            const bool ok = r < C::steps(kb) && f < F && i < P;
            bop[kb][r] = ok ? beta[(size_t)(f < F ? f : 0) * P + (i < P ? i : 0)] : 0.0;
        }
    const double b0 = i < P ? beta[(size_t)F * P + i] : 0.0;
    double sse = 0.0, sae = 0.0, sle = 0.0;
    const long long tiles = (n + 63) >> 6;
    const long long stride = (long long)gridDim.x * (BLOCK / 64);
    long long t = (long long)blockIdx.x * (BLOCK / 64) + wv;
    f32x4 px[C::XL], py[C::YL];
    auto tile_load = [&](long long tl) __attribute__((always_inline)) {
        const long long row0 = tl << 6;
        const int rows = (int)(n - row0 < 64 ? n - row0 : 64);
        const f32x4 *xsrc = reinterpret_cast<const f32x4 *>(X + row0 * F);
        const f32x4 *ysrc = reinterpret_cast<const f32x4 *>(Y + row0 * P);
        // pieces that lie wholly inside the rows that exist (a partial last tile: the copy's tail is
        // stale LDS, which only rows >= `rows` read)
        const int xlast = rows * F / 4 - 1, ylast = rows * P / 4 - 1;
#pragma unroll
        for (int k = 0; k < C::XL; ++k) {
            const int q = lane + 64 * k;
            px[k] = __builtin_nontemporal_load(xsrc + (q <= xlast ? q : (xlast < 0 ? 0 : xlast)));
        }
#pragma unroll
        for (int k = 0; k < C::YL; ++k) {
            const int q = lane + 64 * k;
            py[k] = __builtin_nontemporal_load(ysrc + (q <= ylast ? q : (ylast < 0 ? 0 : ylast)));
        }
    };
    if (t < tiles) tile_load(t);
    for (; t < tiles; t += stride) {
        const long long row0 = t << 6;
        const int rows = (int)(n - row0 < 64 ? n - row0 : 64);
#pragma unroll
        for (int k = 0; k < C::XL; ++k) {
            const int q = lane + 64 * k;
            if (C::XP % 64 == 0 || q < C::XP) *reinterpret_cast<f32x4 *>(xs + 4 * q) = px[k];
        }
#pragma unroll
        for (int k = 0; k < C::YL; ++k) {
            const int q = lane + 64 * k;
            if (C::YP % 64 == 0 || q < C::YP) *reinterpret_cast<f32x4 *>(ys + 4 * q) = py[k];
        }
        __builtin_amdgcn_wave_barrier();
        tile_load(t + stride < tiles ? t + stride : t);
        __builtin_amdgcn_sched_barrier(0);
        if (rows < 64) {
            // a partial tile: the rows that do not exist are read as features too -- make them finite
            // (the tail of rows * F floats that were not copied may hold anything, NaN included)
            for (int e = rows * F + lane; e < 64 * F; e += 64) xs[e] = 0.0f;
            for (int e = rows * P + lane; e < 64 * P; e += 64) ys[e] = 0.0f;
            // and the up to three floats of the last row behind the last whole piece (rows * F not a
            // multiple of 4): straight from global memory
            const int xe = rows * F / 4 * 4 + lane, ye = rows * P / 4 * 4 + lane;
            if (xe < rows * F) xs[xe] = X[row0 * F + xe];
            if (ye < rows * P) ys[ye] = Y[row0 * P + ye];
            __builtin_amdgcn_wave_barrier();
        }
        f64x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (f64x4){b0, b0, b0, b0};
        auto block = [&](auto kbc) __attribute__((always_inline)) {
            constexpr int kb = decltype(kbc)::value;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                constexpr int ST = C::steps(kb);
                const int f0 = 16 * kb + ST * g;
                const float *xr = xs + (16 * m + i) * F + f0;
                float xv[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (C::VEC && ST == 4) {
                    // the piece exists whole or not at all (then beta is zero, and the address is
                    // pulled back inside the row so that no NaN of a neighbour leaks in)
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(f0 < F ? xr : xr - f0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xv[r] = v[r];
                } else {
#pragma unroll
                    for (int r = 0; r < ST; ++r) xv[r] = f0 + r < F ? xr[r] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < ST; ++r)
                    acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xv[r], bop[kb][r], acc[m], 0, 0, 0);
            }
        };
        static_assert(C::KB <= 4, "linear_metrics_mfma64_kernel: F <= 64");
        block(std::integral_constant<int, 0>{});
        if constexpr (C::KB > 1) block(std::integral_constant<int, 1>{});
        if constexpr (C::KB > 2) block(std::integral_constant<int, 2>{});
        if constexpr (C::KB > 3) block(std::integral_constant<int, 3>{});
        // lane (i, g) holds output i of rows 16 m + g + 4 r -- and the lanes with i >= P nothing.
        // Scored there, every lane would sit through 16 outputs (at P = 7 more than half of them
        // padding); instead the predictions go back through the X region of the LDS tile as
        // [row][P], like the copy of Y beside it, and lane l scores the P outputs of row l.
        __builtin_amdgcn_wave_barrier();               // every lane's operand reads of xs are done
        if (i < P) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[(16 * m + g + 4 * r) * P + i] = (float)acc[m][r];
        }
        __builtin_amdgcn_wave_barrier();
        {
            float pred[P], yv[P];
            const bool ok = lane < rows;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const float pp = xs[lane * P + p], yy = ys[lane * P + p];
                pred[p] = ok ? pp : 0.0f;
                yv[p] = ok ? yy : 0.0f;
            }
            lm_resid_groups<0, P, MSLE>(pred, yv, eps, sse, sae, sle);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_wave_barrier();
    }
    lm_block_out(sse, sae, sle, partial);
}

// ---- wide shapes (the reference's singleRegression fit is F = 432, P = 216: singleRegression.py:21,33,
// 74-85): (F + 1) P coefficients do not fit registers or the scalar cache, and 2 F P flop per row is
// MFMA work (186 624 flop against 2 592 bytes).  Until now the predictions were a library GEMM into a
// buffer that wm_residual_sums_f32 then read back.  Here they are formed on v_mfma_f32_16x16x4_f32 and
// scored (and / or written) from the accumulators:
//   D[n][m] = sum_k beta[k][n] X[m][k]: A = beta^T from LDS, B = X straight from global memory.  An
//   MFMA contracts over the lane group g and nothing fixes which four indices a step takes: lane
//   (i, g) reads 16 bytes -- indices 16 kb + 4 g .. + 3 -- of X row m0 + i and of beta^T row n0 + i, and
//   step r uses element r of both.  beta is packed once per call as float32 [kb][n][16] (zero beyond
//   F and P), one 16-index stage (NT x 1 KiB) per barrier, double-buffered; a wave owns 32 rows and
//   ALL NT output tiles (NT x 2 accumulators), so X is read exactly once and a lane ends with four
//   consecutive outputs of a row per accumulator: 16-byte loads of Y / stores of the predictions.
// Predictions on this path are float32 sums of float32 products (beta rounded to float32), like the
// library GEMM's and the reference's own Dense layers.
constexpr int LW_MT = 2;                               // 16-row tiles per wave
constexpr int LW_ROWS = (BLOCK / 64) * LW_MT * 16;     // rows per workgroup pass

__global__ __launch_bounds__(BLOCK) void lw_pack_kernel(const double *__restrict__ beta, int F, int P,
                                                        int NT, int KBn, float *__restrict__ packed,
                                                        int *__restrict__ far_flag)
{
    const int N = NT * 16;
    const long long total = (long long)KBn * N * 16;
    if (blockIdx.x == 0 && threadIdx.x == 0) *far_flag = 0;
    for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total + N;
         e += (long long)gridDim.x * BLOCK) {
        if (e < total) {
            const int kk = (int)(e & 15);
            const int n = (int)((e >> 4) % N);
            const int k = (int)((e >> 4) / N) * 16 + kk;
            packed[e] = (k < F && n < P) ? (float)beta[(size_t)k * P + n] : 0.0f;
        } else {
            const int n = (int)(e - total);
            packed[e] = n < P ? (float)beta[(size_t)F * P + n] : 0.0f;        // the intercept row
        }
    }
}

// REDO: the repeat pass.  The first pass sums the near form of the msle term for every output, no
// branch, and raises *far_flag when an output needed the far form (a per-step branch or call for
// the rare case cost 100-340 spilled VGPRs at NT = 14 and 0.5 of 2.26 ms per 1e6 rows); the repeat
// pass -- same grid, launched behind it, every workgroup leaving at once unless the flag is up --
// forms the predictions again, computes both forms for every output and replaces the msle partials.
// (Starting the second workgroup of every CU half a pass late, so that one's epilogue runs beside the
// other's MFMA phase, measured 1.91-1.97 against 1.96-2.05 ms: within the run-to-run spread, left out.)
template <int NT, bool MSLE, bool RESID, bool WRITE, bool REDO>
__global__ __launch_bounds__(BLOCK, 2) void linear_metrics_wide_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const float *__restrict__ packed,
    long long n, int F, int P, int KBn, float eps, float *__restrict__ Yhat, double *__restrict__ partial,
    int *__restrict__ far_flag)
{
    if (REDO && *far_flag == 0) return;
    constexpr int STAGE = NT * 16 * 16;                 // floats of one 16-index stage of beta^T
    constexpr int FILL = STAGE / 4;                     // 16-byte pieces of a stage
    constexpr int FPT = (FILL + BLOCK - 1) / BLOCK;
    __shared__ __align__(16) float bs[2][STAGE];
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const f32x4 *__restrict__ pk4 = reinterpret_cast<const f32x4 *>(packed);
    const float *__restrict__ bias = packed + (size_t)KBn * STAGE;
    double sse = 0.0, sae = 0.0, sle = 0.0;
    float smax = 0.f;
    const long long nblocks = (n + LW_ROWS - 1) / LW_ROWS;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long m0 = blk * LW_ROWS + wv * (LW_MT * 16);
        const float *xrow[LW_MT];
#pragma unroll
        for (int mt = 0; mt < LW_MT; ++mt) {
            const long long m = m0 + 16 * mt + i;
            xrow[mt] = X + (m < n ? m : n - 1) * F;
        }
        f32x4 acc[NT][LW_MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < LW_MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 fill[FPT], xb[LW_MT];
#pragma unroll
        for (int u = 0; u < FPT; ++u) {
            const int e = threadIdx.x + u * BLOCK;
            fill[u] = pk4[e < FILL ? e : FILL - 1];
        }
#pragma unroll
        for (int mt = 0; mt < LW_MT; ++mt) {
            const int c = 4 * g;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xrow[mt] + (c < F ? c : 0));
            xb[mt] = c < F ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < FPT; ++u) {
            const int e = threadIdx.x + u * BLOCK;
            if (e < FILL) *reinterpret_cast<f32x4 *>(&bs[0][e * 4]) = fill[u];
        }
        __syncthreads();
        for (int kb = 0; kb < KBn; ++kb) {
            // the next stage's operands (the last stage again behind the end: in bounds, unused)
            const int kn = kb + 1 < KBn ? kb + 1 : kb;
            f32x4 xn[LW_MT];
#pragma unroll
            for (int u = 0; u < FPT; ++u) {
                const int e = threadIdx.x + u * BLOCK;
                fill[u] = pk4[(size_t)kn * FILL + (e < FILL ? e : FILL - 1)];
            }
#pragma unroll
            for (int mt = 0; mt < LW_MT; ++mt) {
                const int c = 16 * kn + 4 * g;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xrow[mt] + (c < F ? c : 0));
                xn[mt] = c < F ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const float *st = &bs[kb & 1][i * 16 + 4 * g];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(st + nt * 256);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mt = 0; mt < LW_MT; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], xb[mt][r], acc[nt][mt], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < FPT; ++u) {
                const int e = threadIdx.x + u * BLOCK;
                if (e < FILL) *reinterpret_cast<f32x4 *>(&bs[(kb + 1) & 1][e * 4]) = fill[u];
            }
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < LW_MT; ++mt) xb[mt] = xn[mt];
        }
        // lane (i, g) holds outputs 16 nt + 4 g .. + 3 of row m0 + 16 mt + i.  The accumulators are
        // scored one at a time (four outputs) with the targets of the next LW_PD in flight: left
        // alone, hipcc hoists all 2 NT loads of Y above the first use (228 VGPRs spilled at NT = 14).
        // No branch in here (see REDO above).
        constexpr int NACC = LW_MT * NT;
        constexpr int LW_PD = 4;
        // bias values and column offsets are the same in every pass: behind an opaque offset, or
        // hipcc computes all of them ahead of the pass loop and parks them in scratch (48 stores)
        int zero = 0;
        asm volatile("" : "+s"(zero));
        const float *__restrict__ bias_ = bias + zero;
        const float *__restrict__ Y_ = Y + zero;
        f32x4 yq[LW_PD];
        auto y_load = [&](int q) {
            const int mt = q / NT, nt = q % NT;
            const long long m = m0 + 16 * mt + i;
            const int c = 16 * nt + 4 * g;
            return *reinterpret_cast<const f32x4 *>(Y_ + (m < n ? m : n - 1) * P + (c < P ? c : P - 4));
        };
        if (RESID) {
#pragma unroll
            for (int q = 0; q < LW_PD && q < NACC; ++q) yq[q] = y_load(q);
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            const int mt = q / NT, nt = q % NT;
            const long long m = m0 + 16 * mt + i;
            const int c = 16 * nt + 4 * g;
            const bool ok = m < n && c < P;
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias_ + c);
            const f32x4 o = acc[nt][mt] + b4;
            if (WRITE && ok) *reinterpret_cast<f32x4 *>(Yhat + m * P + c) = o;
            if (RESID) {
                const f32x4 y4 = yq[q % LW_PD];
                f32x4 yv;
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[r] = ok ? y4[r] : o[r];      // a zero residual where there is no output
                if (q + LW_PD < NACC) yq[q % LW_PD] = y_load(q + LW_PD);
                float sq = 0.f;
                if (!REDO) {
                    lm_residuals_near4<MSLE>(o, yv, eps, sse, sae, sq, smax);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = lm_log_ratio_select(__builtin_amdgcn_fmed3f(o[r], eps, LM_CLAMP_HI),
                                                            __builtin_amdgcn_fmed3f(yv[r], eps, LM_CLAMP_HI));
                        sq = fmaf(e, e, sq);
                    }
                }
                if (MSLE) sle += (double)sq;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (RESID) lm_block_out(sse, sae, sle, partial, REDO);
    if (RESID && MSLE && !REDO && !(smax <= 0.2f)) *far_flag = 1;
}

// ---- rows straight from the resident volume (wm_linear_metrics_walk_f32): the metrics pass beside
// gram_walk_kernel (gram_small.hip, which has the reasoning).  lane = row of the region walk; the row's
// 3 V values come from the volume, the predictions and residual sums are lm_small_row's (float64 dot
// product, intercept first, then [x(t) | x(t + d_end)] in order): the same bits per row as gathering
// the rows and calling wm_linear_metrics_f32.
template <int V>
struct LmWalkCfg {                         // latency-bound loop: rows per lane and trip, waves per SIMD
    static constexpr int UR = V == 1 ? 8 : 4;
    static constexpr int WPS = V <= 3 ? 4 : 2;      // 66-96 VGPRs; V = 4 measured faster at 2
};

template <int V, bool MSLE, bool VEC>
__global__ __launch_bounds__(BLOCK, LmWalkCfg<V>::WPS) void linear_metrics_walk_kernel(
    const float *__restrict__ vol, int files, int Z, int Y, int X, int d_mid, int d_end, long long rows,
    wm_walk::CubeWalk w, wm_walk::CubeStrides st, wm_walk::CubeDivs dv, const double *__restrict__ beta,
    float eps, double *__restrict__ partial)
{
    constexpr int F = 2 * V, P = V, LM_WALK_UR = LmWalkCfg<V>::UR, CHUNK = 64 * LM_WALK_UR;
    const int lane = threadIdx.x & 63;
    const long long nwaves = (long long)gridDim.x * (BLOCK / 64);
    const long long wave_id = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const int dmax = d_mid > d_end ? d_mid : d_end;
    double sse = 0.0, sae = 0.0, sle = 0.0;
    const long long nchunks = (rows + CHUNK - 1) / CHUNK;
    for (long long ch = wave_id; ch < nchunks; ch += nwaves) {
        const long long c0 = ch * CHUNK;
        const wm_walk::WalkChunk wc = wm_walk::walk_chunk((unsigned)c0, CHUNK, rows, Z, Y, X, w, st, dv);
        float x[LM_WALK_UR][F], y[LM_WALK_UR * P], pred[LM_WALK_UR * P];
        bool ok[LM_WALK_UR];
        long long off[LM_WALK_UR];
        wm_walk::walk_chunk_rows<LM_WALK_UR>(wc, c0, lane, rows, files, Z, Y, X, dmax, w, st, dv, off);
#pragma unroll
        for (int u = 0; u < LM_WALK_UR; ++u) {
            float a[V], c[V], m[V];
            wm_walk::walk_load<V, VEC>(vol + off[u], st, d_mid, d_end, a, c, m);
            ok[u] = c0 + u * 64 + lane < rows;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                x[u][v] = a[v];
                x[u][V + v] = c[v];
                y[u * P + v] = m[v];
            }
        }
#pragma unroll
        for (int u = 0; u < LM_WALK_UR; ++u) {
            lm_small_predict<F, P>(x[u], beta, pred + u * P);
            // a row past the end scores prediction 0 against target 0: nothing in any of the sums
#pragma unroll
            for (int v = 0; v < P; ++v) {
                pred[u * P + v] = ok[u] ? pred[u * P + v] : 0.f;
                y[u * P + v] = ok[u] ? y[u * P + v] : 0.f;
            }
        }
        lm_resid_groups<0, LM_WALK_UR * P, MSLE>(pred, y, eps, sse, sae, sle);
    }
    lm_block_out(sse, sae, sle, partial);
}

int lw_tiles(int P) { return P <= 64 ? 4 : P <= 128 ? 8 : 14; }

__global__ __launch_bounds__(BLOCK) void residual_sums_kernel(const float *__restrict__ pred,
                                                             const float *__restrict__ Y,
                                                             long long count, int msle, float eps,
                                                             double *__restrict__ partial)
{
    double sse = 0.0, sae = 0.0, sle = 0.0;
    for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < count;
         e += (long long)gridDim.x * BLOCK)
        lm_residual(pred[e], Y[e], msle != 0, eps, sse, sae, sle);
    lm_block_out(sse, sae, sle, partial);
}

// sums[k] += fixed-order sum of the entry-major partials [3][nparts]; sums[3] += count
__global__ __launch_bounds__(BLOCK) void metrics_finish_kernel(const double *__restrict__ partial,
                                                              int nparts, double count,
                                                              double *__restrict__ sums)
{
    __shared__ double wsum[BLOCK / 64];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += BLOCK) s += partial[(size_t)k * nparts + p];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[k] += ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
        if (k == 0) sums[3] += count;
    }
}

int lm_pt(int P) { return P <= 1 ? 1 : P <= 2 ? 2 : P <= 4 ? 4 : P <= 6 ? 6 : P <= 8 ? 8 : P <= 16 ? 16 : 32; }

bool lm_direct_forced()                    // probe switch: WM_LINMETRICS_DIRECT=1 keeps the lane-per-row loads
{
    static const bool v = [] { const char *e = getenv("WM_LINMETRICS_DIRECT"); return e && e[0] == '1'; }();
    return v;
}

bool lm_untiled_forced()                   // probe switch: WM_LINMETRICS_TILE=0 keeps the round-3a staged kernel
{
    static const bool v = [] { const char *e = getenv("WM_LINMETRICS_TILE"); return e && e[0] == '0'; }();
    return v;
}

bool lm_mfma64_off()                       // probe switch: WM_LINMETRICS_MFMA64=0 keeps the scalar-beta routes
{
    static const bool v = [] { const char *e = getenv("WM_LINMETRICS_MFMA64"); return e && e[0] == '0'; }();
    return v;
}

int lm_cap_per_cu(int dflt)                // probe switch: WM_LINMETRICS_CAP=<workgroups per CU> for the tile kernels
{
    static const int v = [] { const char *e = getenv("WM_LINMETRICS_CAP"); return e ? atoi(e) : 0; }();
    return v > 0 ? v : dflt;
}

int lm_grid(long long units)
{
    long long g = (units + BLOCK - 1) / BLOCK;
    const long long cap = (long long)wm_num_cus() * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" size_t wm_linear_metrics_workspace_bytes(void)
{
    return wm_num_cus() > 0 ? wm_align_up((size_t)wm_num_cus() * 8 * 3 * sizeof(double), 256) : 0;
}

extern "C" int wm_linear_metrics_supported(int F, int P)
{
    return F >= 1 && P >= 1 && P <= LM_MAX_P && (long long)(F + 1) * lm_pt(P) <= LM_MAX_BETA;
}

extern "C" int wm_linear_metrics_f32(const float *X, const float *Y, const double *beta, int64_t n,
                                     int F, int P, int want_msle, float msle_eps, double *sums,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    WM_REQUIRE(n >= 0 && sums && beta && (n == 0 || (X && Y)), WM_ERR_INVALID,
               "wm_linear_metrics_f32: bad arguments");
    WM_REQUIRE(wm_linear_metrics_supported(F, P), WM_ERR_UNSUPPORTED,
               "wm_linear_metrics_f32: (F + 1) * P' <= %d (P' = P rounded up to 1,2,4,6,8,16,32) and P <= %d are implemented (F=%d P=%d): form "
               "the predictions with a GEMM and call wm_residual_sums_f32", LM_MAX_BETA, LM_MAX_P, F, P);
    if (n == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_linear_metrics_f32: no HIP device");
    WM_REQUIRE(workspace && workspace_bytes >= wm_linear_metrics_workspace_bytes(), WM_ERR_WORKSPACE,
               "wm_linear_metrics_f32: workspace of %zu bytes needed, %zu given",
               wm_linear_metrics_workspace_bytes(), workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const int PT = lm_pt(P);
    const bool vec = F % 4 == 0 && (((uintptr_t)X) & 15) == 0;
    const bool vecy = P % 4 == 0 && (((uintptr_t)Y) & 15) == 0;
    const size_t beta_bytes = wm_align_up((size_t)(F + 1) * PT * sizeof(double), 16);
    const size_t staged_lds =
        beta_bytes + (size_t)(BLOCK / 64) * 64 * ((4 * F + 16) + (vecy ? 4 * P + 16 : 0));
    const bool staged = vec && F >= 8 && staged_lds <= 64 * 1024 && !lm_direct_forced();
    double *partial = (double *)workspace;
    int grid = 0;
    bool m64 = false;
    if (!lm_direct_forced() && !lm_mfma64_off() && ((((uintptr_t)X) | ((uintptr_t)Y)) & 15) == 0) {
#define X_(F_, P_)                                                                             \
    if (F == F_ && P == P_) {                                                                  \
        const long long want = ((n + 63) / 64 + BLOCK / 64 - 1) / (BLOCK / 64);                \
        const long long cap = (long long)wm_num_cus() * lm_cap_per_cu(8);                      \
        grid = (int)(want < cap ? want : cap);                                                 \
        const size_t lds = (size_t)(BLOCK / 64) * LmM64<F_, P_>::WAVE_BYTES;                   \
        if (want_msle)                                                                         \
            hipLaunchKernelGGL((linear_metrics_mfma64_kernel<F_, P_, true>), dim3(grid), dim3(BLOCK), \
                               lds, st, X, Y, beta, (long long)n, msle_eps, partial);          \
        else                                                                                   \
            hipLaunchKernelGGL((linear_metrics_mfma64_kernel<F_, P_, false>), dim3(grid), dim3(BLOCK), \
                               lds, st, X, Y, beta, (long long)n, msle_eps, partial);          \
        m64 = true;                                                                            \
    }
        WM_LM_MFMA64_SHAPES(X_)
#undef X_
    }
    if (m64) {
        WM_LAUNCH_CHECK("linear_metrics_mfma64_kernel");
        hipLaunchKernelGGL(metrics_finish_kernel, dim3(3), dim3(BLOCK), 0, st, (const double *)partial, grid,
                           (double)n * (double)P, sums);
        WM_LAUNCH_CHECK("metrics_finish_kernel");
        return WM_OK;
    }
    bool small = false;
    if (!lm_direct_forced()) {
#define X_(F_, P_)                                                                             \
    if (F == F_ && P == P_ && wm_small::aligned<LmSmallCfg<F_, P_>::S>(X, Y)) {                \
        grid = wm_num_cus() * 2;             /* 2 waves per SIMD (launch bounds): all resident */ \
        if (want_msle)                                                                         \
            hipLaunchKernelGGL((linear_metrics_small_kernel<F_, P_, true>), dim3(grid), dim3(BLOCK), 0, \
                               st, X, Y, beta, (long long)n, msle_eps, partial);               \
        else                                                                                   \
            hipLaunchKernelGGL((linear_metrics_small_kernel<F_, P_, false>), dim3(grid), dim3(BLOCK), 0, \
                               st, X, Y, beta, (long long)n, msle_eps, partial);               \
        small = true;                                                                          \
    }
        WM_SMALL_SHAPES(X_)
        WM_LM_EXTRA_SHAPES(X_)
#undef X_
    }
    bool tiled = false;
    if (!small && !lm_direct_forced() && !lm_untiled_forced() && (((uintptr_t)X) & 15) == 0) {
#define X_(QX_, P_)                                                                            \
    if (F == 4 * QX_ && P == P_ && (!LmTile<QX_, P_>::VECY || (((uintptr_t)Y) & 15) == 0)) {   \
        const long long want = ((n + 63) / 64 + BLOCK / 64 - 1) / (BLOCK / 64);                \
        const long long cap = (long long)wm_num_cus() * lm_cap_per_cu(8);                      \
        grid = (int)(want < cap ? want : cap);                                                 \
        const size_t lds = (size_t)(BLOCK / 64) * LmTile<QX_, P_>::WAVE_BYTES;                 \
        if (want_msle)                                                                         \
            hipLaunchKernelGGL((linear_metrics_tile_kernel<QX_, P_, true>), dim3(grid), dim3(BLOCK), \
                               lds, st, X, Y, beta, (long long)n, msle_eps, partial);          \
        else                                                                                   \
            hipLaunchKernelGGL((linear_metrics_tile_kernel<QX_, P_, false>), dim3(grid), dim3(BLOCK), \
                               lds, st, X, Y, beta, (long long)n, msle_eps, partial);          \
        tiled = true;                                                                          \
    }
        WM_LM_TILE_SHAPES(X_)
#undef X_
    }
    if (small) {
        WM_LAUNCH_CHECK("linear_metrics_small_kernel");
    } else if (tiled) {
        WM_LAUNCH_CHECK("linear_metrics_tile_kernel");
    } else if (staged) {
        const long long tiles = (n + 63) / 64;
        const long long want = (tiles + BLOCK / 64 - 1) / (BLOCK / 64);
        const long long cap = (long long)wm_num_cus() * 8;
        grid = (int)(want < cap ? want : cap);
        if (grid < 1) grid = 1;
#define LM_STAGED(PT_, VY_)                                                                      \
    {                                                                                            \
        if (P == PT_ && (PT_ == 8 || PT_ == 1))  /* measured: 6 columns lose (odd row pieces) */  \
            hipLaunchKernelGGL((linear_metrics_staged_kernel<PT_, VY_, true>), dim3(grid), dim3(BLOCK), \
                               staged_lds, st, X, Y, beta, (long long)n, F, P, want_msle, msle_eps,   \
                               (int)beta_bytes, partial);                                        \
        else                                                                                     \
            hipLaunchKernelGGL((linear_metrics_staged_kernel<PT_, VY_, false>), dim3(grid),      \
                               dim3(BLOCK), staged_lds, st, X, Y, beta, (long long)n, F, P,      \
                               want_msle, msle_eps, (int)beta_bytes, partial);                   \
    }
        if (vecy) {
            switch (PT) {
            case 4: LM_STAGED(4, true); break;
            case 8: LM_STAGED(8, true); break;
            case 16: LM_STAGED(16, true); break;
            default: LM_STAGED(32, true); break;
            }
        } else {
            switch (PT) {
            case 1: LM_STAGED(1, false); break;
            case 2: LM_STAGED(2, false); break;
            case 4: LM_STAGED(4, false); break;
            case 6: LM_STAGED(6, false); break;
            case 8: LM_STAGED(8, false); break;
            case 16: LM_STAGED(16, false); break;
            default: LM_STAGED(32, false); break;
            }
        }
#undef LM_STAGED
        WM_LAUNCH_CHECK("linear_metrics_staged_kernel");
    } else {
        grid = lm_grid(n);
        const size_t lds = (size_t)(F + 1) * PT * sizeof(double);
#define LM_LAUNCH(PT_)                                                                           \
    {                                                                                            \
        if (vec)                                                                                 \
            hipLaunchKernelGGL((linear_metrics_kernel<PT_, 4>), dim3(grid), dim3(BLOCK), lds, st, X, Y, \
                               beta, (long long)n, F, P, want_msle, msle_eps, partial);          \
        else if (F % 2 == 0 && (((uintptr_t)X) & 7) == 0)                                        \
            hipLaunchKernelGGL((linear_metrics_kernel<PT_, 2>), dim3(grid), dim3(BLOCK), lds, st, X, Y, \
                               beta, (long long)n, F, P, want_msle, msle_eps, partial);          \
        else                                                                                     \
            hipLaunchKernelGGL((linear_metrics_kernel<PT_, 1>), dim3(grid), dim3(BLOCK), lds, st, X, Y, \
                               beta, (long long)n, F, P, want_msle, msle_eps, partial);          \
    }
        switch (PT) {
        case 1: LM_LAUNCH(1) break;
        case 2: LM_LAUNCH(2) break;
        case 4: LM_LAUNCH(4) break;
        case 6: LM_LAUNCH(6) break;
        case 8: LM_LAUNCH(8) break;
        case 16: LM_LAUNCH(16) break;
        default: LM_LAUNCH(32) break;
        }
#undef LM_LAUNCH
    }
    WM_LAUNCH_CHECK("linear_metrics_kernel");
    hipLaunchKernelGGL(metrics_finish_kernel, dim3(3), dim3(BLOCK), 0, st, (const double *)partial, grid,
                       (double)n * (double)P, sums);
    WM_LAUNCH_CHECK("metrics_finish_kernel");
    return WM_OK;
}

extern "C" int wm_linear_metrics_walk_f32(const float *vol, int files, int nvars, int Z, int Y, int X,
                                          const int64_t *strides, int d_mid, int d_end, int64_t rows,
                                          int batch, int64_t length, int64_t index0, int64_t r0,
                                          int64_t r1, int64_t r2, int dim_z, int dim_y, int dim_x,
                                          const double *beta, int want_msle, float msle_eps, double *sums,
                                          void *workspace, size_t workspace_bytes, void *stream)
{
    wm_walk::CubeWalk w;
    wm_walk::CubeStrides cs;
    wm_walk::CubeDivs dv;
    const int rc = wm_walk_setup("wm_linear_metrics_walk_f32", vol, files, nvars, Z, Y, X, strides, 1, 1, 1,
                                 d_mid, d_end, rows, batch, length, index0, r0, r1, r2, dim_z, dim_y,
                                 dim_x, &w, &cs, &dv);
    if (rc) return rc;
    WM_REQUIRE(beta && sums, WM_ERR_INVALID, "wm_linear_metrics_walk_f32: null pointer");
    if (rows == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_linear_metrics_walk_f32: no HIP device");
    WM_REQUIRE(workspace && workspace_bytes >= wm_linear_metrics_workspace_bytes(), WM_ERR_WORKSPACE,
               "wm_linear_metrics_walk_f32: workspace of %zu bytes needed, %zu given",
               wm_linear_metrics_workspace_bytes(), workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    double *partial = (double *)workspace;
    int grid = 0;                            // every workgroup resident (launch bounds); <= 4 per CU
    const bool vec = wm_walk::walk_vec_ok(nvars, cs);
#define LM_WALK_GO(V_, M_)                                                                       \
    do {                                                                                         \
    if (vec)                                                                                     \
        hipLaunchKernelGGL((linear_metrics_walk_kernel<V_, M_, true>), dim3(grid), dim3(BLOCK), 0, st, vol, \
                           files, Z, Y, X, d_mid, d_end, (long long)rows, w, cs, dv, beta, msle_eps,  \
                           partial);                                                             \
    else                                                                                         \
        hipLaunchKernelGGL((linear_metrics_walk_kernel<V_, M_, false>), dim3(grid), dim3(BLOCK), 0, st, vol, \
                           files, Z, Y, X, d_mid, d_end, (long long)rows, w, cs, dv, beta, msle_eps,  \
                           partial);                                                             \
    } while (0)
#define LM_WALK(V_)                                                                              \
    grid = wm_num_cus() * LmWalkCfg<V_>::WPS;                                                    \
    if (want_msle)                                                                               \
        LM_WALK_GO(V_, true);                                                                    \
    else                                                                                         \
        LM_WALK_GO(V_, false)
    switch (nvars) {
    case 1: LM_WALK(1); break;
    case 2: LM_WALK(2); break;
    case 3: LM_WALK(3); break;
    default: LM_WALK(4); break;
    }
#undef LM_WALK
#undef LM_WALK_GO
    WM_LAUNCH_CHECK("linear_metrics_walk_kernel");
    hipLaunchKernelGGL(metrics_finish_kernel, dim3(3), dim3(BLOCK), 0, st, (const double *)partial, grid,
                       (double)rows * (double)nvars, sums);
    WM_LAUNCH_CHECK("metrics_finish_kernel");
    return WM_OK;
}

extern "C" int wm_residual_sums_f32(const float *pred, const float *Y, int64_t count, int want_msle,
                                    float msle_eps, double *sums, void *workspace,
                                    size_t workspace_bytes, void *stream)
{
    WM_REQUIRE(count >= 0 && sums && (count == 0 || (pred && Y)), WM_ERR_INVALID,
               "wm_residual_sums_f32: bad arguments");
    if (count == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_residual_sums_f32: no HIP device");
    WM_REQUIRE(workspace && workspace_bytes >= wm_linear_metrics_workspace_bytes(), WM_ERR_WORKSPACE,
               "wm_residual_sums_f32: workspace of %zu bytes needed, %zu given",
               wm_linear_metrics_workspace_bytes(), workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const int grid = lm_grid(count);
    hipLaunchKernelGGL(residual_sums_kernel, dim3(grid), dim3(BLOCK), 0, st, pred, Y, (long long)count,
                       want_msle, msle_eps, (double *)workspace);
    WM_LAUNCH_CHECK("residual_sums_kernel");
    hipLaunchKernelGGL(metrics_finish_kernel, dim3(3), dim3(BLOCK), 0, st, (const double *)workspace,
                       grid, (double)count, sums);
    WM_LAUNCH_CHECK("metrics_finish_kernel");
    return WM_OK;
}

extern "C" int wm_linear_metrics_wide_supported(int F, int P)
{
    return F >= 4 && F % 4 == 0 && P >= 4 && P % 4 == 0 && P <= 224 && F <= 8192;
}

extern "C" size_t wm_linear_metrics_wide_workspace_bytes(int F, int P)
{
    if (!wm_linear_metrics_wide_supported(F, P)) return 0;
    const size_t NT = lw_tiles(P), KBn = (F + 15) / 16;
    return wm_align_up((size_t)wm_num_cus() * 8 * 3 * sizeof(double), 256) +
           wm_align_up((KBn * NT * 256 + NT * 16) * sizeof(float), 256) + 256;
}

extern "C" int wm_linear_metrics_wide_f32(const float *X, const float *Y, const double *beta, int64_t n,
                                          int F, int P, int want_msle, float msle_eps, double *sums,
                                          float *Yhat, void *workspace, size_t workspace_bytes,
                                          void *stream)
{
    WM_REQUIRE(n >= 0 && beta && (sums || Yhat) && (n == 0 || (X && (Y || !sums))), WM_ERR_INVALID,
               "wm_linear_metrics_wide_f32: bad arguments (sums and / or Yhat must be given; Y with sums)");
    WM_REQUIRE(wm_linear_metrics_wide_supported(F, P), WM_ERR_UNSUPPORTED,
               "wm_linear_metrics_wide_f32: F %% 4 == 0, P %% 4 == 0, P <= 224, F <= 8192 are implemented "
               "(F=%d P=%d)", F, P);
    if (n == 0) return WM_OK;
    WM_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & 15) == 0 && (((uintptr_t)Yhat) & 15) == 0,
               WM_ERR_INVALID, "wm_linear_metrics_wide_f32: X, Y and Yhat must be 16-byte aligned");
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_linear_metrics_wide_f32: no HIP device");
    WM_REQUIRE(workspace && (((uintptr_t)workspace) & 15) == 0 &&
                   workspace_bytes >= wm_linear_metrics_wide_workspace_bytes(F, P),
               WM_ERR_WORKSPACE, "wm_linear_metrics_wide_f32: 16-byte aligned workspace of %zu bytes needed, %zu given",
               wm_linear_metrics_wide_workspace_bytes(F, P), workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const int NT = lw_tiles(P), KBn = (F + 15) / 16;
    double *partial = (double *)workspace;
    float *packed = (float *)((char *)workspace + wm_align_up((size_t)wm_num_cus() * 8 * 3 * sizeof(double), 256));
    const long long pk = (long long)KBn * NT * 256 + NT * 16;
    int *far_flag = (int *)((char *)packed + wm_align_up((size_t)pk * sizeof(float), 256));
    hipLaunchKernelGGL(lw_pack_kernel, dim3((unsigned)((pk + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, beta,
                       F, P, NT, KBn, packed, far_flag);
    WM_LAUNCH_CHECK("lw_pack_kernel");
    const long long nblocks = ((long long)n + LW_ROWS - 1) / LW_ROWS;
    const long long cap = (long long)wm_num_cus() * 2;          // two workgroups per CU (launch bounds)
    const int grid = (int)(nblocks < cap ? nblocks : cap);
    const bool resid = sums != nullptr, write = Yhat != nullptr, msle = resid && want_msle;
#define LW_GO(NT_, M_, R_, W_, REDO_)                                                            \
    hipLaunchKernelGGL((linear_metrics_wide_kernel<NT_, M_, R_, W_, REDO_>), dim3(grid), dim3(BLOCK), 0, st, X, \
                       Y, (const float *)packed, (long long)n, F, P, KBn, msle_eps, Yhat, partial, far_flag)
#define LW_NT(NT_)                                                                               \
    {                                                                                            \
        if (resid && write) { if (msle) LW_GO(NT_, true, true, true, false); else LW_GO(NT_, false, true, true, false); } \
        else if (resid) { if (msle) LW_GO(NT_, true, true, false, false); else LW_GO(NT_, false, true, false, false); }   \
        else LW_GO(NT_, false, false, true, false);                                              \
        if (msle) LW_GO(NT_, true, true, false, true);                                           \
    }
    switch (NT) {
    case 4: LW_NT(4) break;
    case 8: LW_NT(8) break;
    default: LW_NT(14) break;
    }
#undef LW_NT
#undef LW_GO
    WM_LAUNCH_CHECK("linear_metrics_wide_kernel");
    if (resid) {
        hipLaunchKernelGGL(metrics_finish_kernel, dim3(3), dim3(BLOCK), 0, st, (const double *)partial, grid,
                           (double)n * (double)P, sums);
        WM_LAUNCH_CHECK("metrics_finish_kernel");
    }
    return WM_OK;
}
