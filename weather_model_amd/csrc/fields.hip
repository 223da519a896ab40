// fields.hip -- the steps either side of the hot path, all HBM-bound:
//   * one-pass field statistics (min, max, sum, sum of squares, count) replacing the per-file
//     xarray reductions of data.py:52-85 / multipleRegression.py:381-425,
//   * element-wise normalise / standardise  (newConvolution.py:47-49, multipleRegression.py:190-197,
//     convolution.py:80-82),
//   * the sample gather of CustomDataset (newConvolution.py:86-106): flat index -> (time, height,
//     row, column) decode and the (t, t+d_end ; t+d_mid) triplet gather from a resident volume.
// Algorithmic bytes: stats 4 B/element (read once); normalise 8 B/element; gather 72 B read +
// 72 B written per sample (6 variables).
#include "common.h"
#include "cube_walk.h"

namespace {

constexpr int BLOCK = 256;
constexpr int NSTAT = 5;      // min, max, sum, sumsq, count
// workgroups per CU of the read+write streams (tools/probe/fields_time.py, mask_time.py): normalise
// 92.4-93.3 us at 8, 89.6-90.2 at 4 (1 GB: 395 -> 375); triplet gather 31.5-34 at 8, 28.5 at 2
#ifndef WM_AFFINE_CAP
#define WM_AFFINE_CAP 4
#endif
#ifndef WM_GATHER_CAP
#define WM_GATHER_CAP 2
#endif
#ifndef WM_STAT_U
#define WM_STAT_U 4
#endif
constexpr int STAT_U = WM_STAT_U;   // KiB chunks a wave of field_stats_kernel keeps in flight

__device__ __forceinline__ void stat_add(float v, float &mn, float &mx, double &s, double &ss,
                                         double &cnt)
{
    // NaNs are skipped, as xarray's reductions do by default (skipna for float data); branch-free:
    // fminf/fmaxf return the other operand for a NaN, the sums take a zero
    const bool ok = v == v;
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
    const double d = ok ? (double)v : 0.0;
    s += d;
    ss = fma(d, d, ss);
    cnt += ok ? 1.0 : 0.0;
}

// partial: [gridDim.x][NSTAT] float64
__global__ __launch_bounds__(BLOCK) void field_stats_kernel(const float *__restrict__ x,
                                                            long long n,
                                                            double *__restrict__ partial)
{
    __shared__ double red[BLOCK / 64][NSTAT];
    float mn = __builtin_inff(), mx = -__builtin_inff();
    double s = 0.0, ss = 0.0, cnt = 0.0;
    const long long n4 = n / 4;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
    // Batches of STAT_U KiB chunks (64 lanes x 16 B each) are dealt to the waves round-robin and
    // read as non-temporal loads, all STAT_U in flight: the active waves read one compact window.
    // 266 MB with the finalize launch (tools/probe/fields_time.py, one box): grid-stride pairs of
    // loads 50.2 us, a contiguous range per wave 48.4, this 46.0 (batches of 2 or 8 chunks 46.7).
    const int lane_ = threadIdx.x & 63;
    const long long nw = (long long)gridDim.x * (BLOCK / 64);
    const long long wv_ = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const long long nchunk = n4 / 64;
    const long long nbat = nchunk / STAT_U;
    for (long long b = wv_; b < nbat; b += nw) {
        const long long t = b * STAT_U;
        f32x4 a[STAT_U];
#pragma unroll
        for (int u = 0; u < STAT_U; ++u) a[u] = __builtin_nontemporal_load(x4 + (t + u) * 64 + lane_);
#pragma unroll
        for (int u = 0; u < STAT_U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) stat_add(a[u][e], mn, mx, s, ss, cnt);
    }
    for (long long t = nbat * STAT_U + wv_; t < nchunk; t += nw) {
        const f32x4 a = __builtin_nontemporal_load(x4 + t * 64 + lane_);
#pragma unroll
        for (int e = 0; e < 4; ++e) stat_add(a[e], mn, mx, s, ss, cnt);
    }
    if (blockIdx.x == 0) {
        // what is left of the last KiB: fewer than 64 16-byte pieces, then fewer than 4 floats
        const long long r4 = nchunk * 64 + threadIdx.x;
        if (threadIdx.x < 64 && r4 < n4) {
            const f32x4 a = x4[r4];
#pragma unroll
            for (int e = 0; e < 4; ++e) stat_add(a[e], mn, mx, s, ss, cnt);
        }
        if (threadIdx.x < (int)(n - n4 * 4)) stat_add(x[n4 * 4 + threadIdx.x], mn, mx, s, ss, cnt);
    }

    double v[NSTAT] = {(double)mn, (double)mx, s, ss, cnt};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v[0] = fmin(v[0], __shfl_xor(v[0], o, 64));
        v[1] = fmax(v[1], __shfl_xor(v[1], o, 64));
        v[2] += __shfl_xor(v[2], o, 64);
        v[3] += __shfl_xor(v[3], o, 64);
        v[4] += __shfl_xor(v[4], o, 64);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NSTAT; ++k) red[wv][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        double r[NSTAT] = {red[0][0], red[0][1], red[0][2], red[0][3], red[0][4]};
        for (int w = 1; w < BLOCK / 64; ++w) {
            r[0] = fmin(r[0], red[w][0]);
            r[1] = fmax(r[1], red[w][1]);
            r[2] += red[w][2];
            r[3] += red[w][3];
            r[4] += red[w][4];
        }
#pragma unroll
        for (int k = 0; k < NSTAT; ++k) partial[(size_t)k * gridDim.x + blockIdx.x] = r[k];
    }
}

// fixed-order combine of the block partials, stored entry-major [NSTAT][nparts] so that the 256
// threads read them coalesced (thread t takes parts t, t+256, ... in order, then a shuffle tree and
// the four wave results in wave order: the same association every run); merges into out when
// accumulate != 0.  (The first version -- one wave, stride-5 loads -- took 10.5 us.)
__global__ __launch_bounds__(BLOCK) void field_stats_finalize_kernel(
    const double *__restrict__ partial, int nparts, double *__restrict__ out, int accumulate)
{
    __shared__ double red[BLOCK / 64][NSTAT];
    double r[NSTAT] = {__builtin_inf(), -__builtin_inf(), 0.0, 0.0, 0.0};
    for (int p = threadIdx.x; p < nparts; p += BLOCK) {
        r[0] = fmin(r[0], partial[p]);
        r[1] = fmax(r[1], partial[(size_t)nparts + p]);
        r[2] += partial[(size_t)2 * nparts + p];
        r[3] += partial[(size_t)3 * nparts + p];
        r[4] += partial[(size_t)4 * nparts + p];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        r[0] = fmin(r[0], __shfl_xor(r[0], o, 64));
        r[1] = fmax(r[1], __shfl_xor(r[1], o, 64));
        r[2] += __shfl_xor(r[2], o, 64);
        r[3] += __shfl_xor(r[3], o, 64);
        r[4] += __shfl_xor(r[4], o, 64);
    }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NSTAT; ++k) red[threadIdx.x >> 6][k] = r[k];
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int w = 1; w < BLOCK / 64; ++w) {
        r[0] = fmin(r[0], red[w][0]);
        r[1] = fmax(r[1], red[w][1]);
        r[2] += red[w][2];
        r[3] += red[w][3];
        r[4] += red[w][4];
    }
    if (accumulate) {
        r[0] = fmin(r[0], out[0]);
        r[1] = fmax(r[1], out[1]);
        r[2] += out[2];
        r[3] += out[3];
        r[4] += out[4];
    }
#pragma unroll
    for (int k = 0; k < NSTAT; ++k) out[k] = r[k];
}

// y = (x - a) / d in float32, the arithmetic numpy performs on a float32 array with Python-float
// scalars ((data - lo) / (hi - lo), newConvolution.py:48): true division, so results are bitwise
// those of the reference.
__global__ __launch_bounds__(BLOCK) void affine_scale_kernel(const float *__restrict__ x,
                                                             float *__restrict__ y, long long n,
                                                             float a, float d)
{
    const long long n4 = n / 4;
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
    f32x4 *y4 = reinterpret_cast<f32x4 *>(y);
    const long long stride = (long long)gridDim.x * BLOCK;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) {
        f32x4 v = __builtin_nontemporal_load(x4 + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __fdiv_rn(v[e] - a, d);
        __builtin_nontemporal_store(v, y4 + i);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
        const long long i = n4 * 4 + threadIdx.x;
        y[i] = __fdiv_rn(x[i] - a, d);
    }
}

// vol: (T, Z, Y, X, V) float32, the V variables of a grid point contiguous.  Sample idx decodes as
// CustomDataset.__getitem__ does (newConvolution.py:93-99) with shape = (St, Sz, Sy, Sx):
//   column = idx % Sx, row = idx / Sx % Sy, height = idx / (Sx Sy) % Sz,
//   time = idx / (Sx Sy Sz) % St + time_offset
// xout[b] = [vol[time], vol[time + d_end]] (2 x V), yout[b] = vol[time + d_mid] (V).
__global__ __launch_bounds__(BLOCK) void gather_triplets_kernel(
    const float *__restrict__ vol, const long long *__restrict__ idx, long long idx0,
    float *__restrict__ xout, float *__restrict__ yout, long long B, int V, int Z, int Y, int X,
    int St, int Sz, int Sy, int Sx, int time_offset, int d_mid, int d_end)
{
    const long long total = B * V;
    const long long plane = (long long)Z * Y * X * V;     // one time slice
    const long long tot = (long long)St * Sz * Sy * Sx;
    for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total;
         e += (long long)gridDim.x * BLOCK) {
        const long long b = e / V;
        const int v = (int)(e - b * V);
        long long id = idx ? idx[b] : idx0 + b;
        id %= tot;                                        // Python-style modulo: any index decodes
        if (id < 0) id += tot;                            // inside the sample shape
        const int column = (int)(id % Sx);
        const int row = (int)((id / Sx) % Sy);
        const int height = (int)((id / ((long long)Sx * Sy)) % Sz);
        const int time = (int)((id / ((long long)Sx * Sy * Sz)) % St) + time_offset;
        const long long pos = (((long long)height * Y + row) * X + column) * V + v;
        const float *p = vol + (long long)time * plane + pos;
        xout[b * 2 * V + v] = p[0];
        xout[b * 2 * V + V + v] = p[(long long)d_end * plane];
        yout[b * V + v] = p[(long long)d_mid * plane];
    }
}

// The same gather in 32-bit arithmetic with VEC floats per thread (the common case: fewer than 2^31
// units, sample shape and volume below 2^31 elements, V a multiple of VEC).  The kernel above
// spends five 64-bit divisions per 4-byte element -- 41 us for one 721 x 1440 slice of samples,
// 3.0 TB/s; here a thread moves VEC floats from each of the three time slices after four 32-bit
// divisions.
template <int VEC>
__global__ __launch_bounds__(BLOCK) void gather_triplets_u32_kernel(
    const float *__restrict__ vol, const long long *__restrict__ idx, unsigned idx0,
    float *__restrict__ xout, float *__restrict__ yout, unsigned units, unsigned U, unsigned V,
    unsigned Y, unsigned X, unsigned plane, unsigned St, unsigned Sz, unsigned Sy, unsigned Sx,
    unsigned tot, unsigned time_offset, unsigned d_mid, unsigned d_end)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    for (unsigned e = blockIdx.x * BLOCK + threadIdx.x; e < units; e += gridDim.x * BLOCK) {
        const unsigned b = e / U, u = e - b * U;
        unsigned id;
        if (idx) {
            long long raw = idx[b] % (long long)tot;
            if (raw < 0) raw += tot;
            id = (unsigned)raw;
        } else {
            id = (idx0 + b) % tot;
        }
        const unsigned q1 = id / Sx, column = id - q1 * Sx;
        const unsigned q2 = q1 / Sy, row = q1 - q2 * Sy;
        const unsigned q3 = q2 / Sz, height = q2 - q3 * Sz;
        const unsigned time = q3 % St + time_offset;
        const unsigned pos = ((height * Y + row) * X + column) * V + u * VEC;
        const float *p = vol + (size_t)time * plane + pos;
        const vec_t a = *reinterpret_cast<const vec_t *>(p);
        const vec_t c = *reinterpret_cast<const vec_t *>(p + (size_t)d_end * plane);
        const vec_t m = *reinterpret_cast<const vec_t *>(p + (size_t)d_mid * plane);
        float *xo = xout + (size_t)b * 2 * V + u * VEC;
        __builtin_nontemporal_store(a, reinterpret_cast<vec_t *>(xo));
        __builtin_nontemporal_store(c, reinterpret_cast<vec_t *>(xo + V));
        __builtin_nontemporal_store(m, reinterpret_cast<vec_t *>(yout + (size_t)b * V + u * VEC));
    }
}

// ------------------------------------------------------------------------------------------
// Row construction of the regression / Conv3D generators from a RESIDENT volume
// vol (files, nvars, Z, Y, X): row r takes the (cz,cy,cx) cube of every variable at origin
// (z0,y0,x0) from the three files f, f + d_end and f + d_mid:
//     x1[r] = cube(vol[f]),  x2[r] = cube(vol[f + d_end]),  y[r] = cube(vol[f + d_mid])
// each flattened (var, dz, dy, dx) -- multipleRegression.py:217-257 (x(t), x(t+TIME_SPAN);
// x(t+TIME_SPAN/2)), singleRegression.py:251-299, convolution.py:118-147.
// (f, z0, y0, x0) comes from `origins` (int32 x 4 per row: the random cubes of convolution.py,
// drawn on the host) or, when origins == NULL, from the region walk of WeatherDataGenerator
// (multipleRegression.py:259-270 / singleRegression.py:330-340) in closed form: batch b = r / bs
// of the walk that starts at (index0; R0, R1, R2) has
//     index_b = index0 + b,            f = (index_b % length) * bs + r % bs,
//     R0_b = index_b / length,  R1_b = R0_{b-1} / dim_z,  R2_b = R1_{b-1} / dim_y   (b >= 1),
//     origin = (R0_b % dim_z, R1_b % dim_y, R2_b % dim_x)
// -- the reference's update rule, including that R1 and R2 lag one and two batches behind.
using namespace wm_walk;          // CubeWalk, CubeStrides, FastDiv, CubeDivs, fdiv, walk_r*: cube_walk.h

// VEC = 4: cube 1 with the variables of a point contiguous and 16-byte aligned (the point-major
// volume): a thread moves four variables per time slice; p, the element count and dv.p are then in
// units of VEC floats
template <bool SMALL, int VEC>
__global__ __launch_bounds__(BLOCK) void gather_cubes_kernel(
    const float *__restrict__ vol, int files, int nvars, int Z, int Y, int X, int cz, int cy,
    int cx, int d_mid, int d_end, const int *__restrict__ origins, long long rows, CubeWalk w,
    CubeStrides st, CubeDivs dv, float *__restrict__ x1out, float *__restrict__ x2out,
    long long xstride, float *__restrict__ yout)
{
    const int p = nvars * cz * cy * cx / VEC;
    const long long total = rows * p;
    const int dmax = d_mid > d_end ? d_mid : d_end;
    for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total;
         e += (long long)gridDim.x * BLOCK) {
        long long r, f;
        int q, z0, y0, x0;
        if constexpr (SMALL) {
            const unsigned eu = (unsigned)e, ru = fdiv(eu, dv.p);
            r = ru;
            q = (int)(eu - ru * (unsigned)p);
            if (!origins) {
                decode_small(ru, w, dv, f, z0, y0, x0);
            }
        } else {
            r = e / p;
            q = (int)(e - r * p);
            if (!origins) {
                const long long b = r / w.bs;
                f = ((w.index0 + b) % w.length) * w.bs + (r - b * w.bs);
                z0 = (int)(walk_r0(w, b) % w.dim_z);
                y0 = (int)(walk_r1(w, b) % w.dim_y);
                x0 = (int)(walk_r2(w, b) % w.dim_x);
            }
        }
        if (origins) {
            f = origins[4 * r];
            z0 = origins[4 * r + 1];
            y0 = origins[4 * r + 2];
            x0 = origins[4 * r + 3];
        }
        // the host validated the ranges; clamping keeps a bad origin from faulting the GPU
        f = max(0ll, min(f, (long long)files - 1 - dmax));
        z0 = max(0, min(z0, Z - cz));
        y0 = max(0, min(y0, Y - cy));
        x0 = max(0, min(x0, X - cx));
        const int o = q * VEC;
        int ix, iy, iz, v;
        if constexpr (VEC > 1) {
            ix = iy = iz = 0;
            v = o;
        } else if constexpr (SMALL) {
            unsigned t = (unsigned)q, u = fdiv(t, dv.cx);
            ix = (int)(t - u * (unsigned)cx); t = u; u = fdiv(t, dv.cy);
            iy = (int)(t - u * (unsigned)cy); t = u; u = fdiv(t, dv.cz);
            iz = (int)(t - u * (unsigned)cz);
            v = (int)u;
        } else {
            ix = q % cx; q /= cx;
            iy = q % cy; q /= cy;
            iz = q % cz;
            v = q / cz;
        }
        const float *src = vol + f * st.f + v * st.v + (z0 + iz) * st.z + (y0 + iy) * st.y +
                           (x0 + ix) * st.x;
        if constexpr (VEC == 4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(src);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(src + d_end * st.f);
            const f32x4 m = *reinterpret_cast<const f32x4 *>(src + d_mid * st.f);
            __builtin_nontemporal_store(a, reinterpret_cast<f32x4 *>(x1out + r * xstride + o));
            __builtin_nontemporal_store(c, reinterpret_cast<f32x4 *>(x2out + r * xstride + o));
            __builtin_nontemporal_store(m, reinterpret_cast<f32x4 *>(yout + r * (long long)p * VEC + o));
        } else {
            x1out[r * xstride + o] = src[0];
            x2out[r * xstride + o] = src[d_end * st.f];
            yout[r * p + o] = src[d_mid * st.f];
        }
    }
}

// Rows of 64 ... 256 elements (singleRegression's 6^3 cubes of one variable: p = 216).  The kernel
// above decodes (row, element) -> (file, origin, variable, dz, dy, dx) for EVERY element: ~170 vector
// instructions for three loads and three stores, and it ran singleRegression's rows at 2.1 ms per 1e6
// (2.4 TB/s of useful bytes; the Gram pass they feed takes 4.7 ms).  Here a wave owns a row: the row's
// (file, origin) is wave-uniform -- scalar arithmetic, or scalar loads of `origins` -- and the offset
// of element q inside a cube does not depend on the row at all, so every lane decodes its NQ elements
// (q = 64 j + lane) once, before the row loop.  Per element that leaves an add, three loads and three
// stores; the stores of a row are contiguous.  Four rows per trip keep 12 NQ loads in flight.
constexpr int GATHER_ROWS_UR = 4;

template <int NQ>
__global__ __launch_bounds__(BLOCK) void gather_cubes_rows_kernel(
    const float *__restrict__ vol, int files, int nvars, int Z, int Y, int X, int cz, int cy,
    int cx, int d_mid, int d_end, const int *__restrict__ origins, long long rows, CubeWalk w,
    CubeStrides st, CubeDivs dv, float *__restrict__ x1out, float *__restrict__ x2out,
    long long xstride, float *__restrict__ yout)
{
    constexpr int UR = GATHER_ROWS_UR;
    const int p = nvars * cz * cy * cx;
    const int dmax = d_mid > d_end ? d_mid : d_end;
    const int lane = threadIdx.x & 63;
    const long long nwaves = (long long)gridDim.x * (BLOCK / 64);
    const long long wave_id = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    long long off[NQ];
    bool live[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int q = j * 64 + lane;
        live[j] = q < p;
        unsigned t = (unsigned)(live[j] ? q : 0), u = fdiv(t, dv.cx);
        const int ix = (int)(t - u * (unsigned)cx); t = u; u = fdiv(t, dv.cy);
        const int iy = (int)(t - u * (unsigned)cy); t = u; u = fdiv(t, dv.cz);
        const int iz = (int)(t - u * (unsigned)cz);
        off[j] = (long long)u * st.v + iz * st.z + iy * st.y + ix * st.x;
    }
    for (long long r0 = wave_id * UR; r0 < rows; r0 += nwaves * UR) {
        long long base[UR];
        bool ok[UR];
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            const unsigned ru = (unsigned)__builtin_amdgcn_readfirstlane((int)(r0 + k < rows ? r0 + k : rows - 1));
            ok[k] = r0 + k < rows;
            long long f;
            int z0, y0, x0;
            if (origins) {
                f = origins[4 * (long long)ru];
                z0 = origins[4 * (long long)ru + 1];
                y0 = origins[4 * (long long)ru + 2];
                x0 = origins[4 * (long long)ru + 3];
            } else {
                decode_small(ru, w, dv, f, z0, y0, x0);
            }
            // the host validated the ranges; clamping keeps a bad origin from faulting the GPU
            f = max(0ll, min(f, (long long)files - 1 - dmax));
            z0 = max(0, min(z0, Z - cz));
            y0 = max(0, min(y0, Y - cy));
            x0 = max(0, min(x0, X - cx));
            base[k] = f * st.f + z0 * st.z + y0 * st.y + x0 * st.x;
        }
        float a[UR][NQ], c[UR][NQ], m[UR][NQ];
#pragma unroll
        for (int k = 0; k < UR; ++k)
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float *src = vol + base[k] + off[j];
                a[k][j] = src[0];
                c[k][j] = src[d_end * st.f];
                m[k][j] = src[d_mid * st.f];
            }
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            const long long r = r0 + k;
#pragma unroll
            for (int j = 0; j < NQ; ++j)
                if (ok[k] && live[j]) {
                    const int q = j * 64 + lane;
                    __builtin_nontemporal_store(a[k][j], x1out + r * xstride + q);
                    __builtin_nontemporal_store(c[k][j], x2out + r * xstride + q);
                    __builtin_nontemporal_store(m[k][j], yout + r * (long long)p + q);
                }
        }
    }
}

bool gather_rows_off()                     // probe switch: WM_GATHER_ROWS=0 keeps the per-element kernel
{
    static const bool v = [] { const char *e = getenv("WM_GATHER_ROWS"); return e && e[0] == '0'; }();
    return v;
}

int stream_grid(long long work_items, int per_cu = 8)
{
    long long g = (work_items + BLOCK - 1) / BLOCK;
    const long long cap = (long long)wm_num_cus() * per_cu;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" size_t wm_field_stats_workspace_bytes(int64_t n)
{
    if (n < 0 || wm_num_cus() <= 0) return 0;
    return wm_align_up((size_t)wm_num_cus() * 8 * NSTAT * sizeof(double), 256);
}

extern "C" int wm_field_stats_f32(const float *x, int64_t n, double *out, int accumulate,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    WM_REQUIRE(x && out, WM_ERR_INVALID, "wm_field_stats_f32: null pointer");
    WM_REQUIRE(n > 0, WM_ERR_INVALID, "wm_field_stats_f32: need at least one element (n=%lld)",
               (long long)n);
    WM_REQUIRE(((uintptr_t)x & 15) == 0, WM_ERR_INVALID,
               "wm_field_stats_f32: x must be 16-byte aligned");
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_field_stats_f32: no HIP device");
    const size_t need = wm_field_stats_workspace_bytes(n);
    WM_REQUIRE(workspace && workspace_bytes >= need, WM_ERR_WORKSPACE,
               "wm_field_stats_f32: workspace of %zu bytes needed, %zu given", need,
               workspace_bytes);
    const int grid = stream_grid((n / 4 + 1) / 2 + 1);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(field_stats_kernel, dim3(grid), dim3(BLOCK), 0, st, x, (long long)n,
                       (double *)workspace);
    WM_LAUNCH_CHECK("field_stats_kernel");
    hipLaunchKernelGGL(field_stats_finalize_kernel, dim3(1), dim3(BLOCK), 0, st,
                       (const double *)workspace, grid, out, accumulate);
    WM_LAUNCH_CHECK("field_stats_finalize_kernel");
    return WM_OK;
}

extern "C" int wm_affine_scale_f32(const float *x, float *y, int64_t n, float offset,
                                   float divisor, void *stream)
{
    WM_REQUIRE(x && y, WM_ERR_INVALID, "wm_affine_scale_f32: null pointer");
    WM_REQUIRE(n >= 0, WM_ERR_INVALID, "wm_affine_scale_f32: negative size");
    WM_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, WM_ERR_INVALID,
               "wm_affine_scale_f32: x and y must be 16-byte aligned");
    if (n == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_affine_scale_f32: no HIP device");
    hipLaunchKernelGGL(affine_scale_kernel, dim3(stream_grid(n / 4 + 1, WM_AFFINE_CAP)), dim3(BLOCK), 0,
                       (hipStream_t)stream, x, y, (long long)n, offset, divisor);
    WM_LAUNCH_CHECK("affine_scale_kernel");
    return WM_OK;
}

extern "C" int wm_gather_triplets_f32(const float *vol, const int64_t *idx, int64_t idx0,
                                      float *xout, float *yout, int64_t B, int V, int T, int Z,
                                      int Y, int X, int St, int Sz, int Sy, int Sx,
                                      int time_offset, int d_mid, int d_end, void *stream)
{
    WM_REQUIRE(vol && xout && yout, WM_ERR_INVALID, "wm_gather_triplets_f32: null pointer");
    WM_REQUIRE(B >= 0 && V > 0 && T > 0 && Z > 0 && Y > 0 && X > 0, WM_ERR_INVALID,
               "wm_gather_triplets_f32: bad volume shape");
    WM_REQUIRE(St > 0 && Sz > 0 && Sy > 0 && Sx > 0 && Sz <= Z && Sy <= Y && Sx <= X,
               WM_ERR_INVALID, "wm_gather_triplets_f32: sample shape outside the volume");
    WM_REQUIRE(time_offset >= 0 && d_mid >= 0 && d_end >= 0 &&
                   time_offset + St - 1 + (d_end > d_mid ? d_end : d_mid) < T,
               WM_ERR_INVALID,
               "wm_gather_triplets_f32: time %d + %d steps + offset %d leaves the %d slices",
               time_offset, St, d_end > d_mid ? d_end : d_mid, T);
    WM_REQUIRE(idx || idx0 >= 0, WM_ERR_INVALID, "wm_gather_triplets_f32: negative first index");
    if (B == 0) return WM_OK;
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_gather_triplets_f32: no HIP device");
    {
        const long long plane = (long long)Z * Y * X * V, tot = (long long)St * Sz * Sy * Sx;
        const int vec = (V % 2 == 0 && (((uintptr_t)vol | (uintptr_t)xout | (uintptr_t)yout) & 7) == 0)
                            ? 2 : 1;
        const long long units = B * (V / vec);
        const long long last = idx ? tot : idx0 + B;          // largest sample id formed in 32 bits
        if (units < (1ll << 31) && plane < (1ll << 31) && tot < (1ll << 31) &&
            last < (1ll << 32)) {
            const int grid = stream_grid(units, WM_GATHER_CAP);
            if (vec == 2)
                hipLaunchKernelGGL(gather_triplets_u32_kernel<2>, dim3(grid), dim3(BLOCK), 0,
                                   (hipStream_t)stream, vol, (const long long *)idx,
                                   (unsigned)idx0, xout, yout, (unsigned)units, (unsigned)(V / 2),
                                   (unsigned)V, (unsigned)Y, (unsigned)X, (unsigned)plane,
                                   (unsigned)St, (unsigned)Sz, (unsigned)Sy, (unsigned)Sx,
                                   (unsigned)tot, (unsigned)time_offset, (unsigned)d_mid,
                                   (unsigned)d_end);
            else
                hipLaunchKernelGGL(gather_triplets_u32_kernel<1>, dim3(grid), dim3(BLOCK), 0,
                                   (hipStream_t)stream, vol, (const long long *)idx,
                                   (unsigned)idx0, xout, yout, (unsigned)units, (unsigned)V,
                                   (unsigned)V, (unsigned)Y, (unsigned)X, (unsigned)plane,
                                   (unsigned)St, (unsigned)Sz, (unsigned)Sy, (unsigned)Sx,
                                   (unsigned)tot, (unsigned)time_offset, (unsigned)d_mid,
                                   (unsigned)d_end);
            WM_LAUNCH_CHECK("gather_triplets_u32_kernel");
            return WM_OK;
        }
    }
    hipLaunchKernelGGL(gather_triplets_kernel, dim3(stream_grid(B * V)), dim3(BLOCK), 0,
                       (hipStream_t)stream, vol, (const long long *)idx, (long long)idx0, xout,
                       yout, (long long)B, V, Z, Y, X, St, Sz, Sy, Sx, time_offset, d_mid, d_end);
    WM_LAUNCH_CHECK("gather_triplets_kernel");
    return WM_OK;
}

extern "C" int wm_gather_cubes_f32(const float *vol, int files, int nvars, int Z, int Y, int X,
                                   int cz, int cy, int cx, int d_mid, int d_end,
                                   const int32_t *origins, int64_t rows, int batch, int64_t length,
                                   int64_t index0, int64_t r0, int64_t r1, int64_t r2, int dim_z,
                                   int dim_y, int dim_x, float *x1out, float *x2out,
                                   int64_t xstride, float *yout, void *stream)
{
    const int64_t strides[5] = {(int64_t)nvars * Z * Y * X, (int64_t)Z * Y * X, (int64_t)Y * X, X, 1};
    return wm_gather_cubes_strided_f32(vol, files, nvars, Z, Y, X, strides, cz, cy, cx, d_mid, d_end,
                                       origins, rows, batch, length, index0, r0, r1, r2, dim_z,
                                       dim_y, dim_x, x1out, x2out, xstride, yout, stream);
}

extern "C" int wm_gather_cubes_strided_f32(const float *vol, int files, int nvars, int Z, int Y,
                                           int X, const int64_t *strides, int cz, int cy, int cx,
                                           int d_mid, int d_end, const int32_t *origins,
                                           int64_t rows, int batch, int64_t length, int64_t index0,
                                           int64_t r0, int64_t r1, int64_t r2, int dim_z, int dim_y,
                                           int dim_x, float *x1out, float *x2out, int64_t xstride,
                                           float *yout, void *stream)
{
    WM_REQUIRE(vol && x1out && x2out && yout, WM_ERR_INVALID, "wm_gather_cubes_f32: null pointer");
    WM_REQUIRE(strides, WM_ERR_INVALID, "wm_gather_cubes_f32: null strides");
    for (int a = 0; a < 5; ++a)
        WM_REQUIRE(strides[a] >= 0, WM_ERR_INVALID, "wm_gather_cubes_f32: negative stride %lld",
                   (long long)strides[a]);
    const CubeStrides cs = {strides[0], strides[1], strides[2], strides[3], strides[4]};
    WM_REQUIRE(files > 0 && nvars > 0 && Z > 0 && Y > 0 && X > 0, WM_ERR_INVALID,
               "wm_gather_cubes_f32: bad volume shape");
    WM_REQUIRE(cz > 0 && cy > 0 && cx > 0 && cz <= Z && cy <= Y && cx <= X, WM_ERR_INVALID,
               "wm_gather_cubes_f32: cube (%d,%d,%d) outside the (%d,%d,%d) grid", cz, cy, cx, Z,
               Y, X);
    WM_REQUIRE(d_mid >= 0 && d_end >= 0 && d_mid < files && d_end < files, WM_ERR_INVALID,
               "wm_gather_cubes_f32: file offsets (%d, %d) outside %d files", d_mid, d_end, files);
    const long long p = (long long)nvars * cz * cy * cx;
    WM_REQUIRE(rows >= 0 && p < (1ll << 31) && xstride >= p, WM_ERR_INVALID,
               "wm_gather_cubes_f32: bad row count or row stride");
    if (rows == 0) return WM_OK;
    CubeWalk w = {length, index0, r0, r1, r2, batch, dim_z, dim_y, dim_x};
    if (!origins) {
        WM_REQUIRE(batch > 0 && length > 0 && index0 >= 0 && r0 >= 0 && r1 >= 0 && r2 >= 0,
                   WM_ERR_INVALID, "wm_gather_cubes_f32: bad walk state");
        WM_REQUIRE(dim_z > 0 && dim_y > 0 && dim_x > 0 && dim_z - 1 + cz <= Z &&
                       dim_y - 1 + cy <= Y && dim_x - 1 + cx <= X,
                   WM_ERR_INVALID, "wm_gather_cubes_f32: walk region (%d,%d,%d) + cube leaves the grid",
                   dim_z, dim_y, dim_x);
        const int dmax = d_mid > d_end ? d_mid : d_end;
        WM_REQUIRE((length - 1) * batch + batch - 1 + dmax < files, WM_ERR_INVALID,
                   "wm_gather_cubes_f32: walk of %lld batches of %d files + offset %d needs more "
                   "than the %d resident files", (long long)length, batch, dmax, files);
    }
    WM_REQUIRE(wm_num_cus() > 0, WM_ERR_NO_DEVICE, "wm_gather_cubes_f32: no HIP device");
    // 31-bit fast path: element count, the batch index reached by the walk and its state
    const long long lim = 1ll << 31;
    const long long last_b = origins ? 0 : index0 + rows / (batch > 0 ? batch : 1) + 2;
    const bool small = rows * p < lim && rows < lim &&
                       (origins || (last_b < lim && length < lim && r0 < lim && r1 < lim && r2 < lim &&
                                    (length + 1) * batch < lim));
    const unsigned ub = origins ? 1u : (unsigned)batch, ul = origins ? 1u : (unsigned)length;
    // four variables per thread: cube 1, variables contiguous, everything 16-byte aligned
    const bool vec4 = small && cz == 1 && cy == 1 && cx == 1 && nvars % 4 == 0 && cs.v == 1 &&
                      cs.f % 4 == 0 && cs.z % 4 == 0 && cs.y % 4 == 0 && cs.x % 4 == 0 &&
                      xstride % 4 == 0 &&
                      (((uintptr_t)vol | (uintptr_t)x1out | (uintptr_t)x2out | (uintptr_t)yout) & 15) == 0;
    const long long units = vec4 ? rows * (p / 4) : rows * p;
    const CubeDivs dv = {fast_div((unsigned)(vec4 ? p / 4 : p)), fast_div(ub), fast_div(ul),
                         fast_div((unsigned)(origins ? 1 : dim_z)),
                         fast_div((unsigned)(origins ? 1 : dim_y)),
                         fast_div((unsigned)(origins ? 1 : dim_x)), fast_div((unsigned)cx),
                         fast_div((unsigned)cy), fast_div((unsigned)cz)};
    if (small && !vec4 && p >= 64 && p <= 256 && !gather_rows_off()) {
        // a wave per row, GATHER_ROWS_UR rows per trip; every workgroup resident or the rows run out first
        const long long want = (rows + GATHER_ROWS_UR * (BLOCK / 64) - 1) / (GATHER_ROWS_UR * (BLOCK / 64));
        const long long cap = (long long)wm_num_cus() * 8;
        hipLaunchKernelGGL(gather_cubes_rows_kernel<4>, dim3((unsigned)(want < cap ? want : cap)), dim3(BLOCK),
                           0, (hipStream_t)stream, vol, files, nvars, Z, Y, X, cz, cy, cx, d_mid, d_end,
                           origins, (long long)rows, w, cs, dv, x1out, x2out, (long long)xstride, yout);
        WM_LAUNCH_CHECK("gather_cubes_rows_kernel");
        return WM_OK;
    }
    auto kern = vec4 ? gather_cubes_kernel<true, 4>
                     : (small ? gather_cubes_kernel<true, 1> : gather_cubes_kernel<false, 1>);
    hipLaunchKernelGGL(kern, dim3(stream_grid(units, 4)), dim3(BLOCK), 0, (hipStream_t)stream, vol,
                       files, nvars, Z, Y, X, cz, cy, cx, d_mid, d_end, origins, (long long)rows, w,
                       cs, dv, x1out, x2out, (long long)xstride, yout);
    WM_LAUNCH_CHECK("gather_cubes_kernel");
    return WM_OK;
}

// ------------------------------------------------------------------------------ batched copy
// Up to 8 device-to-device copies in ONE launch (blockIdx.y = the copy): a training step replayed from
// a HIP graph stages its batch (x1, x2, y) into the graph's fixed input buffers first, and three
// copy launches -- or the framework's multi-tensor copy, 11 us for 1.5 MB -- are then a visible part
// of a 0.4 ms step.  The records travel as launch arguments (graph-capturable, no device table).
struct WmCopyRecord {
    void *dst;
    const void *src;
    long long bytes;
};

namespace {

constexpr int COPY_MAX = 8;
struct CopyTable {
    WmCopyRecord r[COPY_MAX];
};

template <typename T>
__global__ __launch_bounds__(BLOCK) void copy_multi_kernel(const CopyTable tab)
{
    const WmCopyRecord r = tab.r[blockIdx.y];
    const long long n = r.bytes / (long long)sizeof(T);
    T *__restrict__ d = static_cast<T *>(r.dst);
    const T *__restrict__ s = static_cast<const T *>(r.src);
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n;
         i += (long long)gridDim.x * BLOCK)
        d[i] = s[i];
}

}  // namespace

extern "C" int wm_copy_multi(const void *records, int ncopies, void *stream)
{
    WM_REQUIRE(records && ncopies > 0, WM_ERR_INVALID, "wm_copy_multi: bad arguments");
    const WmCopyRecord *rec = (const WmCopyRecord *)records;
    const int cus = wm_num_cus();
    WM_REQUIRE(cus > 0, WM_ERR_NO_DEVICE, "wm_copy_multi: no HIP device");
    for (int k0 = 0; k0 < ncopies; k0 += COPY_MAX) {
        const int nk = ncopies - k0 < COPY_MAX ? ncopies - k0 : COPY_MAX;
        CopyTable tab = {};
        long long longest = 0;
        unsigned long long align = 0;
        for (int k = 0; k < nk; ++k) {
            const WmCopyRecord &r = rec[k0 + k];
            WM_REQUIRE(r.bytes >= 0 && (r.bytes == 0 || (r.dst && r.src)), WM_ERR_INVALID,
                       "wm_copy_multi: record %d has a null pointer or a negative size", k0 + k);
            tab.r[k] = r;
            if (r.bytes > longest) longest = r.bytes;
            align |= (unsigned long long)(uintptr_t)r.dst | (unsigned long long)(uintptr_t)r.src |
                     (unsigned long long)r.bytes;
        }
        if (longest == 0) continue;
        const int unit = (align & 15) == 0 ? 16 : (align & 3) == 0 ? 4 : 1;
        long long blocks = (longest / unit + BLOCK - 1) / BLOCK;
        if (blocks > (long long)cus * 8) blocks = (long long)cus * 8;
        if (blocks < 1) blocks = 1;
        const dim3 grid((unsigned)blocks, (unsigned)nk);
        if (unit == 16)
            hipLaunchKernelGGL(copy_multi_kernel<f32x4>, grid, dim3(BLOCK), 0, (hipStream_t)stream, tab);
        else if (unit == 4)
            hipLaunchKernelGGL(copy_multi_kernel<float>, grid, dim3(BLOCK), 0, (hipStream_t)stream, tab);
        else
            hipLaunchKernelGGL(copy_multi_kernel<unsigned char>, grid, dim3(BLOCK), 0,
                               (hipStream_t)stream, tab);
        WM_LAUNCH_CHECK("copy_multi_kernel");
    }
    return WM_OK;
}
