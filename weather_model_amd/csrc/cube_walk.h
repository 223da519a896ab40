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

// l = ceil(log2 d); m = floor(2^32 (2^l - d) / d) + 1; q = (t + ((n - t) >> min(l,1))) >> max(l-1,0)
inline FastDiv fast_div(unsigned d)
{
    unsigned l = 0;
    while (l < 32 && (1ull << l) < d) ++l;
    const unsigned long long m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
    return {d, (unsigned)m, l < 1 ? l : 1u, l > 0 ? l - 1 : 0u};
}

}  // namespace wm_walk
