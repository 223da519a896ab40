// dense.hip -- the two Dense layers of the Conv3D interpolation model (convolution.py:175-176:
// Dense(flat, relu) and Dense(prod(input_shape))) for the batch sizes the reference trains with
// (16 rows): "skinny" products whose cost is streaming the weight matrix once.
//   forward      y[m][n]  = act(b[n] + sum_k x[m][k] W[n][k])          W is (N, K) = (out, in)
//   backward     gx[m][k] = mask(sum_n gy[m][n] W[n][k])               mask = ReLU of the layer below
//                gW[n][k] = sum_m gy[m][n] x[m][k],  gb[n] = sum_m gy[m][n]
// All on v_mfma_f32_16x16x4_f32 with the batch as one 16-wide tile side, operands straight from
// global memory.  An MFMA contracts over the lane group g = lane >> 4 and nothing fixes WHICH four
// contraction indices a step takes: a lane loads 16 bytes (indices c0 + 4g .. 4g+3) and step r uses
// element r, so four steps cover 16 indices with one vector load per operand.
// Requirements: K % 4 == 0 and N % 4 == 0 (16-byte rows), pointers 16-byte aligned.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;

__device__ __forceinline__ f32x4 mma(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 load4_or_zero(const float *row, int c, int C, bool row_ok)
{
    // 16 bytes at row[c .. c+3] when the row exists and c < C (C % 4 == 0: all or nothing)
    const f32x4 v = *reinterpret_cast<const f32x4 *>(row + (c < C ? c : 0));
    const bool ok = row_ok && c < C;
    return (f32x4){ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f};
}

// ---- forward: one workgroup per (16 outputs n, 16 rows m), its four waves split the contraction and
// are summed through LDS in wave order (one wave per tile took 12.5 us on its chain of five exposed
// load latencies); D[n][m], lane (i, g) holds y[m0+i][n0+4g+r]
__global__ __launch_bounds__(BLOCK) void dense_fwd_kernel(const float *__restrict__ x,
                                                          const float *__restrict__ W,
                                                          const float *__restrict__ b,
                                                          float *__restrict__ y, int M, int N, int K,
                                                          int relu)
{
    __shared__ f32x4 part[WAVES - 1][64];
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int n = n0 + i, m = m0 + i;
    const float *wrow = W + (size_t)(n < N ? n : N - 1) * K;
    const float *xrow = x + (size_t)(m < M ? m : M - 1) * K;
    const int kq = ((K + WAVES - 1) / WAVES + 15) / 16 * 16;      // this wave's share of K
    const int kbeg = wv * kq, kend = min(K, kbeg + kq);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    // 128 contraction indices per iteration: sixteen 16-byte loads issued together (one exposed
    // memory latency per 32 MFMAs instead of per 8), two independent accumulation chains
    constexpr int KB = 8;
    for (int k0 = kbeg; k0 < kend; k0 += 16 * KB) {
        f32x4 wr[KB], xr[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            wr[u] = load4_or_zero(wrow, k0 + 16 * u + 4 * g, kend, n < N);
            xr[u] = load4_or_zero(xrow, k0 + 16 * u + 4 * g, kend, m < M);
        }
#pragma unroll
        for (int u = 0; u < KB; u += 2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0 = mma(wr[u][r], xr[u][r], acc0);
                acc1 = mma(wr[u + 1][r], xr[u + 1][r], acc1);
            }
    }
    acc0 += acc1;
    if (wv > 0) part[wv - 1][lane] = acc0;
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int w = 0; w < WAVES - 1; ++w) acc0 += part[w][lane];
    acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (m < M) {
        const int nb = n0 + 4 * g;                 // N % 4 == 0: the four outputs exist together
        if (nb < N) {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc0[r] + acc1[r] + (b ? b[nb + r] : 0.0f);
                o[r] = relu ? fmaxf(v, 0.0f) : v;
            }
            *reinterpret_cast<f32x4 *>(y + (size_t)m * N + nb) = o;
        }
    }
}

// ---- backward data, split over n: partial[s][m][k]; D[k][m], lane (i, g) holds gx[m0+i][k0+4g+r]
__global__ __launch_bounds__(BLOCK) void dense_bwd_data_kernel(const float *__restrict__ gy,
                                                               const float *__restrict__ W,
                                                               float *__restrict__ partial, int M,
                                                               int N, int K, int nblk_per_split)
{
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int k0 = (blockIdx.x * WAVES + (threadIdx.x >> 6)) * 16, m0 = blockIdx.y * 16;
    const int split = blockIdx.z;
    if (k0 >= K) return;
    const int m = m0 + i, k = k0 + i;
    const float *grow = gy + (size_t)(m < M ? m : M - 1) * N;
    const int kc = k < K ? k : K - 1;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const int nbeg = split * nblk_per_split * 16;
    const int nend = min(N, nbeg + nblk_per_split * 16);
    constexpr int NBLK = 4;                        // 64 contraction indices per iteration
    for (int n0 = nbeg; n0 < nend; n0 += 16 * NBLK) {
        f32x4 gv[NBLK];
        float wv[NBLK][4];
#pragma unroll
        for (int u = 0; u < NBLK; ++u) {
            gv[u] = load4_or_zero(grow, n0 + 16 * u + 4 * g, nend, m < M);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int na = n0 + 16 * u + 4 * g + r;
                wv[u][r] = (na < nend && k < K) ? W[(size_t)(na < nend ? na : 0) * K + kc] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < NBLK; u += 2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0 = mma(wv[u][r], gv[u][r], acc0);
                acc1 = mma(wv[u + 1][r], gv[u + 1][r], acc1);
            }
    }
    if (m < M) {
        const int kb = k0 + 4 * g;
        if (kb < K) {
            const f32x4 o = acc0 + acc1;
            *reinterpret_cast<f32x4 *>(partial + ((size_t)split * M + m) * K + kb) = o;
        }
    }
}

// gx = mask(sum over splits, in order); mask (may be NULL) is the post-ReLU activation below
__global__ __launch_bounds__(BLOCK) void dense_bwd_data_reduce_kernel(
    const float *__restrict__ partial, int splits, long long total4, const float *__restrict__ mask,
    float *__restrict__ gx)
{
    const long long e = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (e >= total4) return;
    f32x4 s = reinterpret_cast<const f32x4 *>(partial)[e];
    for (int p = 1; p < splits; ++p) s += reinterpret_cast<const f32x4 *>(partial)[(long long)p * total4 + e];
    if (mask) {
        const f32x4 mv = reinterpret_cast<const f32x4 *>(mask)[e];
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = mv[r] > 0.0f ? s[r] : 0.0f;
    }
    reinterpret_cast<f32x4 *>(gx)[e] = s;
}

// ---- backward weights + bias: one wave per 16 outputs n, all k-tiles; D[k][n], lane (i, g) holds
// gW[n0+i][k0+4g+r]; contraction over the batch rows
__global__ __launch_bounds__(BLOCK) void dense_bwd_wb_kernel(const float *__restrict__ gy,
                                                             const float *__restrict__ x,
                                                             float *__restrict__ gW,
                                                             float *__restrict__ gb, int M, int N,
                                                             int K, int ktiles_per_block)
{
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int n0 = (blockIdx.x * WAVES + (threadIdx.x >> 6)) * 16;
    if (n0 >= N) return;
    const int n = n0 + i, nc = n < N ? n : N - 1;
    const int kt0 = blockIdx.y * ktiles_per_block;
    const int kt1 = min((K + 15) / 16, kt0 + ktiles_per_block);
    if (blockIdx.y == 0 && gb) {
        float s = 0.0f;
        for (int m = g; m < M; m += 4) s += gy[(size_t)m * N + nc];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0 && n < N) gb[n] = s;
    }
    for (int kt = kt0; kt < kt1; kt += 2) {        // two k-tiles per iteration: their loads go out together
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int mb = 0; mb < M; mb += 16) {
            float a[2][4], b[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int m = mb + 4 * s + g;
                const bool ok = m < M;
                const int mc = ok ? m : M - 1;
                b[s] = (ok && n < N) ? gy[(size_t)mc * N + nc] : 0.0f;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int k = (kt + t) * 16 + i;
                    a[t][s] = (ok && k < K && kt + t < kt1) ? x[(size_t)mc * K + (k < K ? k : K - 1)] : 0.0f;
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = mma(a[0][s], b[s], acc[0]);
                acc[1] = mma(a[1][s], b[s], acc[1]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kb = (kt + t) * 16 + 4 * g;
            if (n < N && kb < K && kt + t < kt1)
                *reinterpret_cast<f32x4 *>(gW + (size_t)n * K + kb) = acc[t];
        }
    }
}

int bwd_splits(int N)
{
    const int nblk = (N + 15) / 16;
    int s = (nblk + 3) / 4;            // one 64-wide step per wave where that gives <= 32 splits
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return s;
}

int check_dense(const char *who, int M, int N, int K)
{
    WM_REQUIRE(M > 0 && N > 0 && K > 0, WM_ERR_INVALID, "%s: bad sizes M=%d N=%d K=%d", who, M, N, K);
    WM_REQUIRE(N % 4 == 0 && K % 4 == 0, WM_ERR_UNSUPPORTED,
               "%s: N=%d and K=%d must be multiples of 4 (16-byte rows)", who, N, K);
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "%s: no HIP device", who);
    return WM_OK;
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int wm_dense_fwd_f32(const float *x, const float *W, const float *b, float *y, int M,
                                int N, int K, int relu, void *stream)
{
    WM_REQUIRE(x && W && y, WM_ERR_INVALID, "wm_dense_fwd_f32: null pointer");
    if (int rc = check_dense("wm_dense_fwd_f32", M, N, K)) return rc;
    WM_REQUIRE(aligned16(x) && aligned16(W) && aligned16(y), WM_ERR_INVALID,
               "wm_dense_fwd_f32: x, W and y must be 16-byte aligned");
    const dim3 grid((N + 15) / 16, (M + 15) / 16);
    hipLaunchKernelGGL(dense_fwd_kernel, grid, dim3(BLOCK), 0, (hipStream_t)stream, x, W, b, y, M, N,
                       K, relu);
    WM_LAUNCH_CHECK("dense_fwd_kernel");
    return WM_OK;
}

extern "C" size_t wm_dense_bwd_workspace_bytes(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return wm_align_up((size_t)bwd_splits(N) * M * K * sizeof(float), 256);
}

extern "C" int wm_dense_bwd_f32(const float *x, const float *W, const float *gy,
                                const float *mask_below, float *gx, float *gW, float *gb,
                                void *workspace, size_t workspace_bytes, int M, int N, int K,
                                void *stream)
{
    WM_REQUIRE(x && W && gy, WM_ERR_INVALID, "wm_dense_bwd_f32: null pointer");
    WM_REQUIRE(!(gb && !gW), WM_ERR_INVALID,
               "wm_dense_bwd_f32: gb is formed by the weight-gradient pass and needs gW");
    WM_REQUIRE(gx || gW, WM_ERR_INVALID, "wm_dense_bwd_f32: neither gx nor gW requested");
    if (int rc = check_dense("wm_dense_bwd_f32", M, N, K)) return rc;
    WM_REQUIRE(aligned16(x) && aligned16(W) && aligned16(gy) && aligned16(gx) && aligned16(gW) &&
                   aligned16(mask_below) && aligned16(workspace),
               WM_ERR_INVALID, "wm_dense_bwd_f32: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (gx) {
        const size_t need = wm_dense_bwd_workspace_bytes(M, N, K);
        WM_REQUIRE(workspace && workspace_bytes >= need, WM_ERR_WORKSPACE,
                   "wm_dense_bwd_f32: workspace of %zu bytes needed, %zu given", need,
                   workspace_bytes);
        const int splits = bwd_splits(N);
        const int nblk = (N + 15) / 16;
        int per = (nblk + splits - 1) / splits;
        per = (per + 3) / 4 * 4;                       // whole 64-wide steps per split
        const dim3 grid(((K + 15) / 16 + WAVES - 1) / WAVES, (M + 15) / 16, splits);
        hipLaunchKernelGGL(dense_bwd_data_kernel, grid, dim3(BLOCK), 0, st, gy, W,
                           (float *)workspace, M, N, K, per);
        WM_LAUNCH_CHECK("dense_bwd_data_kernel");
        const long long total4 = (long long)M * K / 4;
        hipLaunchKernelGGL(dense_bwd_data_reduce_kernel, dim3((unsigned)((total4 + BLOCK - 1) / BLOCK)),
                           dim3(BLOCK), 0, st, (const float *)workspace, splits, total4, mask_below,
                           gx);
        WM_LAUNCH_CHECK("dense_bwd_data_reduce_kernel");
    }
    if (gW) {
        const int ktiles = (K + 15) / 16;
        const int nwaves = (N + 15) / 16;
        // few output tiles (N small): split the k-tiles over grid.y so that the chip has work
        int ysplit = 1;
        while (nwaves * ysplit < 8 * wm_num_cus() && ysplit < ktiles) ysplit *= 2;
        const int per = (ktiles + ysplit - 1) / ysplit;
        const dim3 grid((nwaves + WAVES - 1) / WAVES, (ktiles + per - 1) / per);
        hipLaunchKernelGGL(dense_bwd_wb_kernel, grid, dim3(BLOCK), 0, st, gy, x, gW, gb, M, N, K, per);
        WM_LAUNCH_CHECK("dense_bwd_wb_kernel");
    }
    return WM_OK;
}
