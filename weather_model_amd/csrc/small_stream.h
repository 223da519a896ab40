// small_stream.h -- the memory side shared by the narrow-row streaming kernels (F + P <= 15):
// gram_small_kernel (gram_small.hip) and linear_metrics_small_kernel (linmetrics.hip).
//   * a lane owns R consecutive rows per step (R = 1, 2 or 4: the smallest with R*F % 4 == 0), so X
//     arrives as 16-byte non-temporal loads of whole rows and Y as the widest vector that fits;
//   * a batch is U steps of a wave (64 R U rows, one contiguous run of X and one of Y); batches are
//     dealt round-robin to the waves and the next batch's loads are issued before the current
//     batch's arithmetic (the callers' software pipeline).
#pragma once

#include "common.h"

// the instantiated shapes: the reference family F = 2p, P = p (p = 1..5), one target with up to
// eight predictors, and two odd shapes that take the R = 4 path with a multi-column Y
#define WM_SMALL_SHAPES(X_)                                                                    \
    X_(1, 1) X_(2, 1) X_(3, 1) X_(4, 1) X_(5, 1) X_(6, 1) X_(7, 1) X_(8, 1)                    \
    X_(4, 2) X_(6, 3) X_(8, 4) X_(10, 5) X_(5, 2) X_(5, 3)

namespace wm_small {

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int W> struct VecOf;
template <> struct VecOf<1> { using T = float; };
template <> struct VecOf<2> { using T = f32x2; };
template <> struct VecOf<4> { using T = f32x4; };

template <int F_, int P_, int U_>
struct Stream {
    static constexpr int F = F_, P = P_, U = U_;
    static constexpr int M = F + P;
    static constexpr int R = (F % 4 == 0) ? 1 : ((F % 2 == 0) ? 2 : 4);
    static constexpr int XV = R * F / 4;                 // dwordx4 loads of X per lane and step
    static constexpr int YN = R * P;                     // floats of Y per lane and step
    static constexpr int YW = (YN % 4 == 0) ? 4 : ((YN % 2 == 0) ? 2 : 1);
    static constexpr int YV = YN / YW;
    static constexpr int ROWS_STEP = 64 * R;             // rows a wave takes per step
    static constexpr int ROWS_BATCH = ROWS_STEP * U;
};

template <class S>
struct Buf {
    f32x4 x[S::U][S::XV];
    float y[S::U][S::YN];
    // element m of the lane's row r of step u
    __device__ __forceinline__ float xe(int u, int r, int m) const
    {
        return x[u][(r * S::F + m) >> 2][(r * S::F + m) & 3];
    }
    __device__ __forceinline__ float ye(int u, int r, int m) const { return y[u][r * S::P + m]; }
};

// Loads of batch t (rows t * ROWS_BATCH ...): lane owns R consecutive rows of every step.
template <class S>
__device__ __forceinline__ void fetch(Buf<S> &b, const float *__restrict__ X,
                                      const float *__restrict__ Y, long long t, int lane)
{
    using YT = typename VecOf<S::YW>::T;
    const f32x4 *__restrict__ Xv = reinterpret_cast<const f32x4 *>(X);
    const long long row0 = t * S::ROWS_BATCH;
#pragma unroll
    for (int u = 0; u < S::U; ++u) {
        const long long rb = row0 + (long long)u * S::ROWS_STEP;
        const f32x4 *xs = Xv + (rb * S::F) / 4 + lane * S::XV;
#pragma unroll
        for (int j = 0; j < S::XV; ++j) b.x[u][j] = __builtin_nontemporal_load(xs + j);
        const YT *ys = reinterpret_cast<const YT *>(Y + rb * S::P) + lane * S::YV;
#pragma unroll
        for (int v = 0; v < S::YV; ++v) {
            const YT val = __builtin_nontemporal_load(ys + v);
            if constexpr (S::YW == 1) {
                b.y[u][v] = val;
            } else {
#pragma unroll
                for (int e = 0; e < S::YW; ++e) b.y[u][v * S::YW + e] = val[e];
            }
        }
    }
}

// Dense loads for shapes whose lane-rows are three and more vectors long.  fetch() above gives lane l
// the XV consecutive vectors of its own rows, so each of the XV load instructions of a step touches
// every cache line of the step's run at a stride of XV vectors: XV requests per line -- F=6,P=3
// (XV = 3) ran at 4.8 TB/s and F=10,P=5 / F=5,P=3 (XV = 5) at 3.4-3.5 where XV <= 2 reaches 6.5-6.8.
// fetch_dense() loads piece q = j * 64 + lane with instruction j (one contiguous KiB each), and
// untangle() turns a batch back into lane-rows through a wave-private LDS region: piece q belongs to
// lane q / XV, slot q % XV.  Lane-rows are XV * 16 bytes apart, plus 16 when XV is even, so that the
// 16-byte reads of 16 consecutive lanes fall into 16 different bank groups.  Wave-private: the LDS
// executes a wave's instructions in order, a wave barrier is all the fence there is.
template <class S>
struct Dense {
    using YT = typename VecOf<S::YW>::T;
    static constexpr bool X = S::XV >= 3;
    static constexpr bool Y = S::YV >= 3;
    static constexpr int XROW = S::XV * 16 + ((S::XV & 1) ? 0 : 16);            // bytes
    static constexpr int YB = S::YW * 4;                                         // bytes of a Y vector
    static constexpr int YROW = S::YV * YB + ((S::YV & 1) ? 0 : YB);
    static constexpr int XBYTES = X ? S::U * 64 * XROW : 0;
    static constexpr int YBYTES = Y ? S::U * 64 * YROW : 0;
    static constexpr int BYTES = (XBYTES + YBYTES) > 0 ? (XBYTES + YBYTES) : 16;  // per wave
    static constexpr bool ANY = X || Y;
};

template <class S>
__device__ __forceinline__ void fetch_dense(Buf<S> &b, const float *__restrict__ X,
                                            const float *__restrict__ Y, long long t, int lane)
{
    using D = Dense<S>;
    using YT = typename D::YT;
    const f32x4 *__restrict__ Xv = reinterpret_cast<const f32x4 *>(X);
    const long long row0 = t * S::ROWS_BATCH;
#pragma unroll
    for (int u = 0; u < S::U; ++u) {
        const long long rb = row0 + (long long)u * S::ROWS_STEP;
        const f32x4 *xs = Xv + (rb * S::F) / 4 + (D::X ? lane : lane * S::XV);
#pragma unroll
        for (int j = 0; j < S::XV; ++j) b.x[u][j] = __builtin_nontemporal_load(xs + (D::X ? j * 64 : j));
        const YT *ys = reinterpret_cast<const YT *>(Y + rb * S::P) + (D::Y ? lane : lane * S::YV);
#pragma unroll
        for (int v = 0; v < S::YV; ++v) {
            const YT val = __builtin_nontemporal_load(ys + (D::Y ? v * 64 : v));
            if constexpr (S::YW == 1) {
                b.y[u][v] = val;
            } else {
#pragma unroll
                for (int e = 0; e < S::YW; ++e) b.y[u][v * S::YW + e] = val[e];
            }
        }
    }
}

// lds: this wave's Dense<S>::BYTES region (16-byte aligned)
template <class S>
__device__ __forceinline__ void untangle(Buf<S> &b, char *lds, int lane)
{
    using D = Dense<S>;
    using YT = typename D::YT;
    if constexpr (D::X) {
#pragma unroll
        for (int u = 0; u < S::U; ++u)
#pragma unroll
            for (int j = 0; j < S::XV; ++j) {
                const int q = j * 64 + lane;
                *reinterpret_cast<f32x4 *>(lds + u * 64 * D::XROW + (q / S::XV) * D::XROW + (q % S::XV) * 16) = b.x[u][j];
            }
    }
    if constexpr (D::Y) {
        char *ly = lds + D::XBYTES;
#pragma unroll
        for (int u = 0; u < S::U; ++u)
#pragma unroll
            for (int v = 0; v < S::YV; ++v) {
                const int q = v * 64 + lane;
                YT val;
                if constexpr (S::YW == 1) {
                    val = b.y[u][v];
                } else {
#pragma unroll
                    for (int e = 0; e < S::YW; ++e) val[e] = b.y[u][v * S::YW + e];
                }
                *reinterpret_cast<YT *>(ly + u * 64 * D::YROW + (q / S::YV) * D::YROW + (q % S::YV) * D::YB) = val;
            }
    }
    __builtin_amdgcn_wave_barrier();
    if constexpr (D::X) {
#pragma unroll
        for (int u = 0; u < S::U; ++u)
#pragma unroll
            for (int j = 0; j < S::XV; ++j)
                b.x[u][j] = *reinterpret_cast<const f32x4 *>(lds + u * 64 * D::XROW + lane * D::XROW + j * 16);
    }
    if constexpr (D::Y) {
        const char *ly = lds + D::XBYTES;
#pragma unroll
        for (int u = 0; u < S::U; ++u)
#pragma unroll
            for (int v = 0; v < S::YV; ++v) {
                const YT val = *reinterpret_cast<const YT *>(ly + u * 64 * D::YROW + lane * D::YROW + v * D::YB);
                if constexpr (S::YW == 1) {
                    b.y[u][v] = val;
                } else {
#pragma unroll
                    for (int e = 0; e < S::YW; ++e) b.y[u][v * S::YW + e] = val[e];
                }
            }
    }
    __builtin_amdgcn_wave_barrier();
}

// X 16-byte aligned, Y aligned to its vector width
template <class S>
inline bool aligned(const float *X, const float *Y)
{
    return (((uintptr_t)X) & 15) == 0 && (((uintptr_t)Y) & (S::YW * 4 - 1)) == 0;
}

}  // namespace wm_small
