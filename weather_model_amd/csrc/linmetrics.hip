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
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int LM_MAX_BETA = 4096;          // (F + 1) * PT doubles in LDS: 32 KB
constexpr int LM_MAX_P = 32;

// log1p(a) - log1p(b) = log1p(t), t = (a - b) / (1 + b), for a, b >= eps > 0 (so t > -1).  Near zero
// the odd series 2 atanh(t / (2 + t)) (|t| <= 1/2: s <= 1/5, the s^11 term is below 2e-9 of the
// result); elsewhere the hardware logarithm of 1 + t, which is then away from its zero.  ~20
// instructions against ~150 for two library log1pf calls, which were 3/4 of this kernel's time;
// the result is within ~3e-7 relative of the float64 value (keras' own float32 log(x + 1) is not).
__device__ __forceinline__ float lm_log_ratio(float a, float b)
{
    const float t = (a - b) * __builtin_amdgcn_rcpf(1.0f + b);
    const float s = t * __builtin_amdgcn_rcpf(2.0f + t);
    const float s2 = s * s;
    float poly = fmaf(s2, 1.0f / 9.0f, 1.0f / 7.0f);
    poly = fmaf(s2, poly, 1.0f / 5.0f);
    poly = fmaf(s2, poly, 1.0f / 3.0f);
    poly = fmaf(s2, poly, 1.0f);
    const float near = 2.0f * s * poly;
    const float far = __builtin_amdgcn_logf(1.0f + t) * 0.69314718055994530942f;
    return fabsf(t) <= 0.5f ? near : far;
}

__device__ __forceinline__ void lm_residual(float pred, float y, bool msle, float eps, double &sse,
                                            double &sae, double &sle)
{
    const double d = (double)pred - (double)y;
    sse = fma(d, d, sse);
    sae += fabs(d);
    if (msle) {
        const float e = lm_log_ratio(fmaxf(pred, eps), fmaxf(y, eps));
        sle = fma((double)e, (double)e, sle);
    }
}

__device__ __forceinline__ void lm_block_out(double sse, double sae, double sle, double *partial)
{
    __shared__ double red[BLOCK / 64][3];
    sse = wave_sum(sse);
    sae = wave_sum(sae);
    sle = wave_sum(sle);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = sse; red[wv][1] = sae; red[wv][2] = sle; }
    __syncthreads();
    if (threadIdx.x < 3) {
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
    int grid;
    if (staged) {
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
