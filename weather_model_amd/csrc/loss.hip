// loss.hip -- the loss and metrics of the Conv3D interpolation model's train step in one pass
// (convolution.py:176-180: compile(loss='mean_squared_error', metrics=[mae, msle])).
//
// keras.losses.mean_squared_error as the reference reaches it scores predictions (B, P) against
// targets (B, 1, P): the subtraction broadcasts to (B, B, P), every prediction against every target
// of the batch.  With `broadcast` = 0 the targets are (B, P) and pairs are (i, i).
//   loss = mean (y_j - p_i)^2,  mae = mean |y_j - p_i|,
//   msle = mean (log1p(max(p_i, eps)) - log1p(max(y_j, eps)))^2,
//   grad[i] = d loss / d p_i = 2 / (pairs * P) * sum_j (p_i - y_j)
// Framework glue takes ~20 element-wise launches over the (B, B, P) differences for this (90 us of a
// 560 us step at batch 16); here one pass.  A thread sums its (non-negative) pair terms in float32
// over blocks of at most 64 targets and folds the blocks in float64; the target sum of the
// gradient, B*p - sum(y), which cancels when p is near the batch mean of y, is float64 term by
// term (float32 blocks of 64 left 2e-5 of the gradient at B = 4096); float64 from there on,
// combined in a fixed order (bitwise reproducible).
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAXB = 4096;
constexpr int JBLOCK = 64;        // float32 pair sums per thread stay this short

// One thread per prediction (i, p): it walks the targets y[j][p] it is scored against (all B rows
// with `broadcast`, row i otherwise) -- the p of a wave are consecutive, so every load is
// coalesced, and B * P threads fill the chip where one thread per column (P = 8000: 125 waves)
// took 22 us on its serial chain of loads and logs.  partial: [3][gridDim.x] float64.
template <typename I>
__global__ __launch_bounds__(BLOCK) void mse_metrics_kernel(const float *__restrict__ pred,
                                                            const float *__restrict__ y, int B,
                                                            long long P, int broadcast, float eps,
                                                            float gscale, float *__restrict__ grad,
                                                            double *__restrict__ partial)
{
    __shared__ double red[BLOCK / 64][3];
    double se = 0.0, ae = 0.0, le = 0.0;
    const long long total = (long long)B * P;
    for (I e = (I)blockIdx.x * BLOCK + threadIdx.x; e < (I)total; e += (I)gridDim.x * BLOCK) {
        const I i = e / (I)P, p = e - i * (I)P;     // (a 64-bit division is a ~150-instruction routine)
        const float pv = pred[e];
        const float lp = wm_log1p_pos(fmaxf(pv, eps));
        const int j0 = broadcast ? 0 : (int)i, j1 = broadcast ? B : (int)i + 1;
        double ysum = 0.0;
        for (int jb = j0; jb < j1; jb += JBLOCK) {
            const int je = jb + JBLOCK < j1 ? jb + JBLOCK : j1;
            float s2 = 0.0f, s1 = 0.0f, sl = 0.0f;
#pragma unroll 8
            for (int j = jb; j < je; ++j) {
                const float yv = y[(long long)j * P + p];
                const float d = yv - pv, q = lp - wm_log1p_pos(fmaxf(yv, eps));
                s2 = fmaf(d, d, s2);
                s1 += fabsf(d);
                sl = fmaf(q, q, sl);
                ysum += (double)yv;          // every term in float64: B*p - sum(y) cancels
            }
            se += (double)s2;
            ae += (double)s1;
            le += (double)sl;
        }
        if (grad) grad[e] = (float)((double)gscale * ((double)(j1 - j0) * (double)pv - ysum));
    }
    se = wave_sum(se);
    ae = wave_sum(ae);
    le = wave_sum(le);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        red[wv][0] = se;
        red[wv][1] = ae;
        red[wv][2] = le;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) s += red[w][threadIdx.x];
        partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
}

// out[k] = (fixed-order sum of partial[k][0..nparts)) / count;  one wave per output
__global__ __launch_bounds__(64) void mse_metrics_finalize_kernel(const double *__restrict__ partial,
                                                                  int nparts, double inv_count,
                                                                  float *__restrict__ out)
{
    const int k = blockIdx.x;
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) s += partial[(size_t)k * nparts + p];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[k] = (float)(s * inv_count);
}

int loss_grid(long long BP)
{
    long long g = (BP + BLOCK - 1) / BLOCK;
    const long long cap = (long long)wm_num_cus() * 4;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" size_t wm_mse_metrics_workspace_bytes(int64_t P)
{
    // upper bound for any batch size: the grid is capped at 4 workgroups per CU
    if (P <= 0 || wm_num_cus() <= 0) return 0;
    return wm_align_up((size_t)wm_num_cus() * 4 * 3 * sizeof(double), 256);
}

extern "C" int wm_mse_metrics_f32(const float *pred, const float *y, int B, int64_t P,
                                  int broadcast, float eps, float *grad, float *out3,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    WM_REQUIRE(pred && y && out3, WM_ERR_INVALID, "wm_mse_metrics_f32: null pointer");
    WM_REQUIRE(B > 0 && B <= MAXB && P > 0, WM_ERR_INVALID,
               "wm_mse_metrics_f32: batch %d (1..%d) or row length %lld out of range", B, MAXB,
               (long long)P);
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_mse_metrics_f32: no HIP device");
    const size_t need = wm_mse_metrics_workspace_bytes(P);
    WM_REQUIRE(workspace && workspace_bytes >= need, WM_ERR_WORKSPACE,
               "wm_mse_metrics_f32: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
    const double pairs = broadcast ? (double)B * B : (double)B;
    const double count = pairs * (double)P;
    const float gscale = (float)(2.0 / count);
    const int grid = loss_grid((long long)B * P);
    hipStream_t st = (hipStream_t)stream;
    double *partial = (double *)workspace;
    WM_REQUIRE(eps > 0.0f, WM_ERR_INVALID, "wm_mse_metrics_f32: eps must be positive (the clip of msle)");
    if ((long long)B * P + (long long)grid * BLOCK < (1ll << 31))
        hipLaunchKernelGGL(mse_metrics_kernel<unsigned>, dim3(grid), dim3(BLOCK), 0, st, pred, y, B,
                           (long long)P, broadcast, eps, gscale, grad, partial);
    else
        hipLaunchKernelGGL(mse_metrics_kernel<long long>, dim3(grid), dim3(BLOCK), 0, st, pred, y, B,
                           (long long)P, broadcast, eps, gscale, grad, partial);
    WM_LAUNCH_CHECK("mse_metrics_kernel");
    hipLaunchKernelGGL(mse_metrics_finalize_kernel, dim3(3), dim3(64), 0, st, partial, grid,
                       1.0 / count, out3);
    WM_LAUNCH_CHECK("mse_metrics_finalize_kernel");
    return WM_OK;
}
