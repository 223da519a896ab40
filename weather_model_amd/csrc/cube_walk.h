// cube_walk.h -- the region walk of the regression generators in closed form (see the comment above
// wm_gather_cubes in fields.hip: multipleRegression.py:259-270 / singleRegression.py:330-340), shared
// by the row gather (fields.hip) and by the kernels that read their rows straight from the resident
// volume (gram_small.hip: gram_walk_kernel, linmetrics.hip: linear_metrics_walk_kernel).
#pragma once

#include "common.h"

namespace wm_walk {

struct CubeWalk {
    long long length, index0, r0, r1, r2;
    int bs, dim_z, dim_y, dim_x;
};
// element strides of the volume's (file, variable, z, y, x) axes: the row-major layout of the
// reference, or any other resident layout (wm_gather_cubes_strided_f32)
struct CubeStrides {
    long long f, v, z, y, x;
};

// Division by a launch-invariant divisor (Granlund-Montgomery, round-up form): exact for every
// 32-bit unsigned dividend.  A hardware 32-bit division is ~30 instructions and a 64-bit one ~100;
// the walk below takes nine per element and the kernel was bound by them (point-major volume,
// 1 M rows of 8: 95 us with 64-bit divisions, 78 with 32-bit ones; see DESIGN.md for this form).
struct FastDiv {
    unsigned d, m, s1, s2;
};
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv &f)
{
    const unsigned t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}
struct CubeDivs {
    FastDiv p, bs, length, dim_z, dim_y, dim_x, cx, cy, cz;
};

// walk state of batch b (see the comment above CubeWalk); I = long long, or unsigned-with-FastDiv
// when every count fits 31 bits
__device__ __forceinline__ long long walk_r0(const CubeWalk &w, long long b)
{
    return b <= 0 ? w.r0 : (w.index0 + b) / w.length;
}
__device__ __forceinline__ long long walk_r1(const CubeWalk &w, long long b)
{
    return b <= 0 ? w.r1 : walk_r0(w, b - 1) / w.dim_z;
}
__device__ __forceinline__ long long walk_r2(const CubeWalk &w, long long b)
{
    return b <= 0 ? w.r2 : walk_r1(w, b - 1) / w.dim_y;
}


// (file, origin) of row `ru` of a walk whose counts all fit 31 bits -- the SMALL path of
// gather_cubes_kernel, which this restates: batch b = ru / bs,
//   f = ((index0 + b) % length) * bs + ru % bs,  origin = (R0_b % dim_z, R1_b % dim_y, R2_b % dim_x)
__device__ __forceinline__ void decode_small(unsigned ru, const CubeWalk &w, const CubeDivs &dv,
                                             long long &f, int &z0, int &y0, int &x0)
{
    const unsigned b = fdiv(ru, dv.bs), ib = (unsigned)w.index0 + b;
    const unsigned R0 = fdiv(ib, dv.length);
    f = (long long)((ib - R0 * (unsigned)w.length) * (unsigned)w.bs + (ru - b * (unsigned)w.bs));
    // R1_b = R0_{b-1} / dim_z, R2_b = (R0_{b-2} / dim_z) / dim_y; raw state for b <= 0
    const unsigned R0m1 = b >= 1 ? fdiv(ib - 1, dv.length) : 0u;
    const unsigned R0m2 = b >= 2 ? fdiv(ib - 2, dv.length) : 0u;
    const unsigned r0 = b == 0 ? (unsigned)w.r0 : R0;
    const unsigned r1 = b == 0 ? (unsigned)w.r1 : fdiv(b == 1 ? (unsigned)w.r0 : R0m1, dv.dim_z);
    const unsigned r1m1 = b <= 1 ? (unsigned)w.r1 : fdiv(b == 2 ? (unsigned)w.r0 : R0m2, dv.dim_z);
    const unsigned r2 = b == 0 ? (unsigned)w.r2 : fdiv(r1m1, dv.dim_y);
    z0 = (int)(r0 - fdiv(r0, dv.dim_z) * (unsigned)w.dim_z);
    y0 = (int)(r1 - fdiv(r1, dv.dim_y) * (unsigned)w.dim_y);
    x0 = (int)(r2 - fdiv(r2, dv.dim_x) * (unsigned)w.dim_x);
}

// Kernels that read their rows straight from the volume give a wave a CHUNK of consecutive rows
// (row = chunk start + k, k = 64 u + lane).  decode_small costs ~130 vector instructions per row and
// bound the first version of those kernels (6.8e7 rows of one variable: 0.30 ms where the loads need
// 0.06); decoding once per 64 rows in the scalar unit still left ~50 scalar instructions per 64 rows
// on the CU's one scalar ALU (0.20 ms).  Per chunk, in wave-uniform (scalar) arithmetic:
//   * the pass R over the files that batch b0 - 2 belongs to (b0: the chunk's first batch; the region
//     state reaches y and x one and two batches late, hence the lag), the row `rb` at which pass R + 1
//     starts, and the z / y / x offsets that the states R and R + 1 select;
//   * mode 1: the chunk ends before rb -- one cell for all its rows, a row adds its file;
//   * mode 2: the chunk crosses into pass R + 1 (and not beyond) -- a row picks each of its three
//     offsets by comparing k with rb, rb + batch, rb + 2 batch;
//   * mode 0: the walk's first three batches (raw start state) or passes shorter than a chunk --
//     decode_small per row.
struct WalkChunk {
    int mode;
    int f_cur, f_next;      // file of chunk row k: k + f_cur in pass R, k + f_next in pass R + 1
    int kb;                 // chunk row at which pass R + 1 starts (mode 2)
    int bs;
    long long zc[2], yc[2], xc[2];      // element offsets selected by the states R, R + 1
};

// base: first row of the chunk (wave-uniform, < rows); count: rows of a chunk
__device__ __forceinline__ WalkChunk walk_chunk(unsigned base, unsigned count, long long rows, int Z, int Y,
                                                int X, const CubeWalk &w, const CubeStrides &st,
                                                const CubeDivs &dv)
{
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    const unsigned last = base + count - 1u < (unsigned)rows ? base + count - 1u : (unsigned)rows - 1u;
    const unsigned bs = (unsigned)w.bs, len = (unsigned)w.length;
    const unsigned b0 = fdiv(base, dv.bs);
    WalkChunk c;
    c.mode = 0;
    c.bs = w.bs;
    if (b0 >= 3u) {
        const unsigned R = fdiv((unsigned)w.index0 + b0 - 2u, dv.length);
        const long long span = (long long)len * bs;                       // rows of a pass
        const long long rb = ((long long)(R + 1u) * len - w.index0) * bs;
        c.mode = (long long)last < rb ? 1 : ((long long)last < rb + span ? 2 : 0);
        c.f_cur = (int)((long long)base - (rb - span));
        c.f_next = (int)((long long)base - rb);
        c.kb = (int)(rb - (long long)base);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned Re = R + (unsigned)e;
            const unsigned r1 = fdiv(Re, dv.dim_z), r2 = fdiv(r1, dv.dim_y);
            int z0 = (int)(Re - r1 * (unsigned)w.dim_z);
            int y0 = (int)(r1 - r2 * (unsigned)w.dim_y);
            int x0 = (int)(r2 - fdiv(r2, dv.dim_x) * (unsigned)w.dim_x);
            // the host validated the ranges; clamping keeps a bad walk from faulting the GPU
            c.zc[e] = max(0, min(z0, Z - 1)) * st.z;
            c.yc[e] = max(0, min(y0, Y - 1)) * st.y;
            c.xc[e] = max(0, min(x0, X - 1)) * st.x;
        }
    }
    return c;
}

// Element offsets at which the chunk's rows r = c0 + k, k = 64 u + lane, start in the volume: the value
// of variable 0 at x(t).  Rows past the end get a clamped, valid offset.  Offsets, not pointers: a
// pointer merged from three branches loses its address space and the loads turn into flat loads.  The
// mode branch sits outside the row loop, so the common mode is straight-line code.
template <int UR>
__device__ __forceinline__ void walk_chunk_rows(const WalkChunk &c, long long c0,
                                                int lane, long long rows, int files, int Z, int Y, int X,
                                                int dmax, const CubeWalk &w, const CubeStrides &st,
                                                const CubeDivs &dv, long long (&off)[UR])
{
    const unsigned fmax = (unsigned)(files - 1 - dmax);
    if (c.mode == 1) {                                                    // wave-uniform branches
        const long long cell = c.zc[0] + c.yc[0] + c.xc[0];
        const unsigned f0 = (unsigned)c.f_cur + (unsigned)lane;
#pragma unroll
        for (int u = 0; u < UR; ++u) off[u] = cell + (long long)min(f0 + 64u * u, fmax) * st.f;
    } else if (c.mode == 2) {
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int kk = u * 64 + lane;
            const bool i0 = kk >= c.kb, i1 = kk >= c.kb + c.bs, i2 = kk >= c.kb + 2 * c.bs;
            const unsigned f = min((unsigned)kk + (unsigned)(i0 ? c.f_next : c.f_cur), fmax);
            const long long cell = (i0 ? c.zc[1] : c.zc[0]) + (i1 ? c.yc[1] : c.yc[0]) + (i2 ? c.xc[1] : c.xc[0]);
            off[u] = cell + (long long)f * st.f;
        }
    } else {
#pragma unroll 1
        for (int u = 0; u < UR; ++u) {
            const long long r = c0 + u * 64 + lane;
            long long f;
            int z0, y0, x0;
            decode_small((unsigned)(r < rows ? r : rows - 1), w, dv, f, z0, y0, x0);
            f = max(0ll, min(f, (long long)files - 1 - dmax));
            z0 = max(0, min(z0, Z - 1));
            y0 = max(0, min(y0, Y - 1));
            x0 = max(0, min(x0, X - 1));
            off[u] = f * st.f + z0 * st.z + y0 * st.y + x0 * st.x;
        }
    }
}

// The same for ONE row, as a pointer (gram_walk_kernel<4> schedules better around it: 0.32 against
// 0.42 ms per 6.8e7 rows).  Where row r = chunk start + k starts in the volume: the value of variable 0 at x(t).  Rows past the
// end get a clamped, valid address and ok = false.
__device__ __forceinline__ const float *walk_chunk_row(const WalkChunk &c, const float *__restrict__ vol,
                                                       long long r, unsigned k, long long rows, int files,
                                                       int Z, int Y, int X, int dmax, const CubeWalk &w,
                                                       const CubeStrides &st, const CubeDivs &dv, bool &ok)
{
    ok = r < rows;
    const unsigned fmax = (unsigned)(files - 1 - dmax);
    if (c.mode == 1) {                                                    // wave-uniform branches
        const unsigned f = min(k + (unsigned)c.f_cur, fmax);
        return vol + (c.zc[0] + c.yc[0] + c.xc[0]) + (long long)f * st.f;
    }
    if (c.mode == 2) {
        const int kk = (int)k;
        const bool i0 = kk >= c.kb, i1 = kk >= c.kb + c.bs, i2 = kk >= c.kb + 2 * c.bs;
        const unsigned f = min(k + (unsigned)(i0 ? c.f_next : c.f_cur), fmax);
        const long long cell = (i0 ? c.zc[1] : c.zc[0]) + (i1 ? c.yc[1] : c.yc[0]) + (i2 ? c.xc[1] : c.xc[0]);
        return vol + cell + (long long)f * st.f;
    }
    long long f;
    int z0, y0, x0;
    decode_small((unsigned)(ok ? r : rows - 1), w, dv, f, z0, y0, x0);
    f = max(0ll, min(f, (long long)files - 1 - dmax));
    z0 = max(0, min(z0, Z - 1));
    y0 = max(0, min(y0, Y - 1));
    x0 = max(0, min(x0, X - 1));
    return vol + (f * st.f + z0 * st.z + y0 * st.y + x0 * st.x);
}

// One row's 3 V values: a = x(t), c = x(t + d_end), m = x(t + d_mid).  VEC: the V variables of a point
// are adjacent (st.v == 1) -- one dword-aligned vector load per time instead of V loads.
template <int V, bool VEC>
__device__ __forceinline__ void walk_load(const float *__restrict__ src, const CubeStrides &st, int d_mid,
                                          int d_end, float (&a)[V], float (&c)[V], float (&m)[V])
{
    if constexpr (VEC && V >= 2) {
        typedef float vec __attribute__((ext_vector_type(V), aligned(4)));
        const vec va = *reinterpret_cast<const vec *>(src);
        const vec vc = *reinterpret_cast<const vec *>(src + d_end * st.f);
        const vec vm = *reinterpret_cast<const vec *>(src + d_mid * st.f);
#pragma unroll
        for (int v = 0; v < V; ++v) { a[v] = va[v]; c[v] = vc[v]; m[v] = vm[v]; }
    } else {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            a[v] = src[v * st.v];
            c[v] = src[d_end * st.f + v * st.v];
            m[v] = src[d_mid * st.f + v * st.v];
        }
    }
}

// host: may walk_load<V, true> be used?
inline bool walk_vec_ok(int V, const CubeStrides &st)
{
    return V >= 2 && st.v == 1;
}

// l = ceil(log2 d); m = floor(2^32 (2^l - d) / d) + 1; q = (t + ((n - t) >> min(l,1))) >> max(l-1,0)
inline FastDiv fast_div(unsigned d)
{
    unsigned l = 0;
    while (l < 32 && (1ull << l) < d) ++l;
    const unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    return {d, (unsigned)m, l < 1 ? l : 1u, l > 0 ? l - 1 : 0u};
}

}  // namespace wm_walk

// Validates a CUBE_SIZE-1 walk over a resident volume with 1..4 variables and fills the launch
// structs (gram_small.hip); returns WM_OK or a negative WM_ERR_* with the thread's error text set.
int wm_walk_setup(const char *who, const float *vol, int files, int nvars, int Z, int Y, int X,
                  const int64_t *strides, int cz, int cy, int cx, int d_mid, int d_end, int64_t rows,
                  int batch, int64_t length, int64_t index0, int64_t r0, int64_t r1, int64_t r2,
                  int dim_z, int dim_y, int dim_x, wm_walk::CubeWalk *w, wm_walk::CubeStrides *cs,
                  wm_walk::CubeDivs *dv);
