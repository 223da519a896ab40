// gram_mid.hip -- Gram accumulation for D = F + P + 1 <= 64 that none of the special kernels take:
// the multi-variable fits of the drop-ins (multipleRegression.py:273-286 with several variables or
// cubes: F = 2p, P = p -- eight variables at CUBE_SIZE 1 are F = 16, P = 8, D = 25).
//
// Such rows are 100-250 bytes: the pass is HBM-bound (96 B/row at F = 16, P = 8: 83 G rows/s at
// 8 TB/s), and the blocked SYRK (gram_syrk.hip) -- whose unit is a 64 x 64 tile of Z'Z fed from
// 128-column LDS panels -- spends its time on MFMAs over padding: 14.6 G rows/s at D = 25, and 2-8
// G rows/s where X or Y rows are not 16-byte pieces.  Here the operands go straight from global
// memory into MFMA registers, for any F, P and alignment:
//   * Z = [X | Y | 1] is padded to 16*NQ virtual columns (NQ = 2: D <= 32, NQ = 4: D <= 64); lane
//     (i, g) of a wave holds, for row 4s + g of k-step s, the NQ virtual columns NQ*i + q, one per
//     register q.  Register q is then both the A and the B operand shape of
//     v_mfma_f32_16x16x4_f32 (k = the four rows) for the column set {NQ*i + q}, and the
//     NQ(NQ+1)/2 register pairs q1 <= q2 give the upper blocks of the Gram: 3 MFMAs per four rows
//     at NQ = 2, 10 at NQ = 4 -- no LDS, no transposition;
//   * every lane's columns are classified once (a column of X, of Y, the ones column, padding) into
//     a pointer and a row advance; the ones / padding lanes read their constants from mid_consts
//     through the same load with an advance of zero, so the main loop is one load (an 8- or 16-byte
//     piece of an X or Y row where F, P and the pointers allow: VEC; NQ dword loads otherwise), a
//     pointer increment and NP MFMAs per k-step -- no address multiply, no select, no bounds test
//     (k-steps with rows past n and the last partial batch of a wave take a checked loop);
//   * a wave owns a contiguous range of k-steps, U of them in flight; float32 sums of <= 1024 rows
//     are folded into float64 registers; per-wave float64 slabs; a fixed-order finalize (one wave
//     per Gram entry) adds them to acc -- bitwise reproducible like the other Gram kernels.
// Measured (tools/probe/gram_ref_time.py, 1-4e7 rows): F=16,P=8 51-53 G rows/s = 4.9-5.1 TB/s (SYRK:
// 14.6 G rows/s), F=12,P=6 45-46 (2.6), F=24,P=12 21.6 (13.3), F=40,P=20 18.3 = 4.4 TB/s; unaligned:
// F=32,P=1 17.7 (2.2; 10.7 with the SYRK's hand-filled Y), F=14,P=7 28, F=33,P=7 15.6, F=60,P=3 13.8.
// At NQ = 2 the loop is bound by its 3 MFMAs per k-step (96 cycles per wave and four rows: 51 G
// rows/s), at NQ = 4 by its 10.  The first version computed each address with a 64-bit multiply,
// selected constants and tested bounds per k-step: 9.6 G rows/s at F=16,P=8.
#include "gram_internal.h"

namespace {

constexpr int BLOCK = 256, WAVES = 4;
constexpr int MID_FLUSH = 256;            // k-steps between float64 folds: 1024 rows
__constant__ float mid_consts[8] = {1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // the ones column, padding

template <int NQ>
struct MidCfg {
    static constexpr int NP = NQ * (NQ + 1) / 2;      // register pairs = 16 x 16 blocks per k-step
#ifndef WM_MID_U2
#define WM_MID_U2 16
#endif
#ifndef WM_MID_U4
#define WM_MID_U4 8
#endif
    static constexpr int U = NQ == 2 ? WM_MID_U2 : WM_MID_U4;     // k-steps in flight
    static constexpr int UT = NQ == 2 ? 8 : 4;        // the same for the dword-load loop
    static constexpr int SLAB = NP * 256;             // doubles per wave
};

__device__ __forceinline__ int mid_pair(int NQ, int q1, int q2)      // q1 <= q2
{
    return q1 * NQ - q1 * (q1 - 1) / 2 + (q2 - q1);
}

template <int NQ, bool VEC>
__global__ __launch_bounds__(BLOCK, 2) void gram_mid_kernel(const float *__restrict__ X,
                                                            const float *__restrict__ Y, long long n,
                                                            int F, int P, double *__restrict__ slabs)
{
    using C = MidCfg<NQ>;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const long long nwaves = (long long)gridDim.x * WAVES;
    const long long wave_id = (long long)blockIdx.x * WAVES + (threadIdx.x >> 6);
    const long long steps = (n + 3) >> 2;
    const long long s0 = steps * wave_id / nwaves, s1 = steps * (wave_id + 1) / nwaves;
    const int M = F + P;

    // this lane's NQ virtual columns: where they come from
    const float *base[NQ];
    long long stride[NQ];
    float cst[NQ];           // value when the column is not loaded
    bool ld[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int v = NQ * i + q;
        if (v < F) { base[q] = X + v; stride[q] = F; ld[q] = true; cst[q] = 0.f; }
        else if (v < M) { base[q] = Y + (v - F); stride[q] = P; ld[q] = true; cst[q] = 0.f; }
        else { base[q] = X; stride[q] = 0; ld[q] = false; cst[q] = v == M ? 1.f : 0.f; }
    }

    f32x4 c[C::NP];
    double d[C::NP * 4];
#pragma unroll
    for (int p = 0; p < C::NP; ++p) c[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < C::NP * 4; ++k) d[k] = 0.0;
    int since = 0;

    // VEC: F, P multiples of NQ and both arrays aligned to NQ floats -- a lane's NQ columns are one
    // 8- or 16-byte piece of an X or a Y row.  The main loop is then one load and NP MFMAs per
    // k-step and nothing else: the lanes of the ones / padding columns read their constants from
    // mid_consts through the same load (stride 0), and the row pointer just advances.
    typedef float vecq __attribute__((ext_vector_type(NQ)));
#define MID_FOLD()                                                       \
    {                                                                    \
        _Pragma("unroll") for (int p = 0; p < C::NP; ++p)                \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {              \
                d[p * 4 + r] += (double)c[p][r];                         \
                c[p][r] = 0.f;                                           \
            }                                                            \
        since = 0;                                                       \
    }
    long long s = s0;
    if (VEC) {
        const long long full_end = min(s1, n >> 2);       // k-steps whose four rows all exist
        const float *pz = ld[0] ? base[0] + (s0 * 4 + g) * stride[0]
                                : mid_consts + (NQ * i == M ? 0 : 4);
        const long long adv = ld[0] ? 4 * stride[0] : 0;
        for (; s + C::U <= full_end; s += C::U) {
            vecq t[C::U];
#pragma unroll
            for (int u = 0; u < C::U; ++u) {
                t[u] = __builtin_nontemporal_load(reinterpret_cast<const vecq *>(pz));
                pz += adv;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < C::U; ++u) {
                int p = 0;
#pragma unroll
                for (int q1 = 0; q1 < NQ; ++q1)
#pragma unroll
                    for (int q2 = q1; q2 < NQ; ++q2, ++p)
                        c[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(t[u][q1], t[u][q2], c[p], 0, 0, 0);
            }
            since += C::U;
            if (since >= MID_FLUSH) MID_FOLD()
        }
    }
    if (!VEC) {       // any alignment: the same lean loop with NQ dword loads per k-step
        const long long full_end = min(s1, n >> 2);
        const float *pq[NQ];
        long long advq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            pq[q] = ld[q] ? base[q] + (s0 * 4 + g) * stride[q] : mid_consts + (NQ * i + q == M ? 0 : 4);
            advq[q] = ld[q] ? 4 * stride[q] : 0;
        }
        for (; s + C::UT <= full_end; s += C::UT) {
            float z[C::UT][NQ];
#pragma unroll
            for (int u = 0; u < C::UT; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    z[u][q] = *pq[q];
                    pq[q] += advq[q];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < C::UT; ++u) {
                int p = 0;
#pragma unroll
                for (int q1 = 0; q1 < NQ; ++q1)
#pragma unroll
                    for (int q2 = q1; q2 < NQ; ++q2, ++p)
                        c[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[u][q1], z[u][q2], c[p], 0, 0, 0);
            }
            since += C::UT;
            if (since >= MID_FLUSH) MID_FOLD()
        }
    }
    // the last k-steps of a range (a partial batch, rows past n): checked loads
    for (; s < s1; s += C::UT) {
        float z[C::UT][NQ];
#pragma unroll
        for (int u = 0; u < C::UT; ++u) {
            const long long row = (s + u) * 4 + g;
            const bool live = s + u < s1 && row < n;
            const long long rc = live ? row : 0;                  // in range: unconditional loads
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float t = base[q][rc * stride[q]];
                z[u][q] = live ? (ld[q] ? t : cst[q]) : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < C::UT; ++u) {
            int p = 0;
#pragma unroll
            for (int q1 = 0; q1 < NQ; ++q1)
#pragma unroll
                for (int q2 = q1; q2 < NQ; ++q2, ++p)
                    c[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(z[u][q1], z[u][q2], c[p], 0, 0, 0);
        }
        since += C::UT;
        if (since >= MID_FLUSH) MID_FOLD()
    }
#undef MID_FOLD
    double *dst = slabs + (size_t)wave_id * C::SLAB;
#pragma unroll
    for (int p = 0; p < C::NP; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[p * 256 + r * 64 + lane] = d[p * 4 + r] + (double)c[p][r];
}

// acc (D x D, order [X | 1 | Y]) += fixed-order sum over the per-wave slabs.  One wave per entry.
template <int NQ>
__global__ __launch_bounds__(64) void gram_mid_finalize_kernel(const double *__restrict__ slabs,
                                                               int nparts, int F, int P,
                                                               double *__restrict__ acc)
{
    using C = MidCfg<NQ>;
    const int M = F + P, D = M + 1;
    const int e = blockIdx.x, a = e / D, b = e - a * D;
    // acc order [X | 1 | Y] -> virtual order [X | Y | 1]
    int va = a < F ? a : (a == F ? M : a - 1), vb = b < F ? b : (b == F ? M : b - 1);
    int qa = va % NQ, qb = vb % NQ;
    if (qa > qb) {
        const int t = va; va = vb; vb = t;
        qa = va % NQ; qb = vb % NQ;
    }
    // pair (qa <= qb): A index m = va / NQ, B index nn = vb / NQ: register m & 3 of lane 16 (m >> 2) + nn
    const int m = va / NQ, nn = vb / NQ;
    const int idx = mid_pair(NQ, qa, qb) * 256 + (m & 3) * 64 + (m >> 2) * 16 + nn;
    double s = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 64) s += slabs[(size_t)k * C::SLAB + idx];
    s = wave_sum(s);
    if (threadIdx.x == 0) acc[(size_t)a * D + b] += s;
}

int mid_grid() { return wm_num_cus() * 2; }

}  // namespace

bool wm_gram_mid_supported(int F, int P) { return F >= 1 && P >= 1 && F + P + 1 <= 64; }

size_t wm_gram_mid_workspace_bytes(int F, int P)
{
    if (!wm_gram_mid_supported(F, P) || wm_num_cus() <= 0) return 0;
    const int slab = F + P + 1 <= 32 ? MidCfg<2>::SLAB : MidCfg<4>::SLAB;
    return (size_t)mid_grid() * WAVES * slab * sizeof(double);
}

int wm_gram_mid_launch(const float *X, const float *Y, long long n, int F, int P, double *acc,
                       double *slabs, hipStream_t st)
{
    const int grid = mid_grid(), D = F + P + 1;
#define MID_LAUNCH(NQ_)                                                                           \
    {                                                                                             \
        const bool vec = F % NQ_ == 0 && P % NQ_ == 0 &&                                          \
                         (((uintptr_t)X | (uintptr_t)Y) & (NQ_ * sizeof(float) - 1)) == 0;        \
        if (vec)                                                                                  \
            hipLaunchKernelGGL((gram_mid_kernel<NQ_, true>), dim3(grid), dim3(BLOCK), 0, st, X, Y, n, \
                               F, P, slabs);                                                      \
        else                                                                                      \
            hipLaunchKernelGGL((gram_mid_kernel<NQ_, false>), dim3(grid), dim3(BLOCK), 0, st, X, Y, n, \
                               F, P, slabs);                                                      \
        WM_LAUNCH_CHECK("gram_mid_kernel");                                                       \
        hipLaunchKernelGGL(gram_mid_finalize_kernel<NQ_>, dim3(D * D), dim3(64), 0, st,           \
                           (const double *)slabs, grid * WAVES, F, P, acc);                       \
    }
    if (D <= 32) MID_LAUNCH(2) else MID_LAUNCH(4)
#undef MID_LAUNCH
    WM_LAUNCH_CHECK("gram_mid_finalize_kernel");
    return WM_OK;
}
