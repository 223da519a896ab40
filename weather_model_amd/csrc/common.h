// common.h -- shared host/device helpers for libwm.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "wm.h"

#define WM_VERSION_NUM 100   // 0.1.0

// ---- thread-local last-error text -------------------------------------------------------
void wm_set_error(const char *fmt, ...);

#define WM_REQUIRE(cond, code, ...)                  \
    do {                                             \
        if (!(cond)) {                               \
            wm_set_error(__VA_ARGS__);               \
            return (code);                           \
        }                                            \
    } while (0)

#define WM_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            wm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),        \
                         __FILE__, __LINE__);                                          \
            return WM_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

#define WM_LAUNCH_CHECK(name)                                                          \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            wm_set_error("launch of %s failed: %s", name, hipGetErrorString(e_));      \
            return WM_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

// Compute units of the current device (cached per device); <= 0 on failure.
int wm_num_cus();

static inline size_t wm_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- device helpers ---------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// 4-byte-aligned vector views: rows of a (H-kH+1)-wide tensor start at any dword, the hardware
// takes dword-aligned dwordx2/x4 accesses, and hipcc emits them for these types.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// log1p(x) for x >= 0 in ~15 instructions (the library log1pf is ~75 and was most of the loss kernels'
// time): the odd series 2 atanh(x / (2 + x)) up to x = 1/2 (s <= 1/5: the first omitted term is below
// 2e-9 of the result), the hardware logarithm of 1 + x beyond, where it is away from its zero.  Within
// ~2e-7 relative of the float64 value.
__device__ __forceinline__ float wm_log1p_pos(float x)
{
    const float s = x * __builtin_amdgcn_rcpf(2.0f + x);
    const float s2 = s * s;
    float poly = fmaf(s2, 1.0f / 9.0f, 1.0f / 7.0f);
    poly = fmaf(s2, poly, 1.0f / 5.0f);
    poly = fmaf(s2, poly, 1.0f / 3.0f);
    poly = fmaf(s2, poly, 1.0f);
    const float far = __builtin_amdgcn_logf(1.0f + x) * 0.69314718055994530942f;
    return x <= 0.5f ? 2.0f * s * poly : far;
}

// 16-byte LDS (or global) accesses through a float pointer.  `*(f32x4 *)(array + i)` is NOT that:
// clang takes the alignment of such an lvalue from the array's element type (4), and the backend
// then splits the access into two ds_write2_b32 / ds_read2_b32.
__device__ inline void wm_store4(float *p, f32x4 v) { *(f32x4 *)__builtin_assume_aligned(p, 16) = v; }
__device__ inline f32x4 wm_load4(const float *p) { return *(const f32x4 *)__builtin_assume_aligned(p, 16); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One Adam update (torch.optim.Adam single-tensor order, see adam_kernel in model.hip); shared by
// wm_adam_step_f32 and the deferred-finalize form wm_adam_step_partials_f32 so the two agree bit
// for bit.
struct AdamScalars {
    float one_minus_b1, b2, one_minus_b2, eps, step_size, bc2_sqrt, gscale;
};
__device__ __forceinline__ void adam_update(float &p, float &m, float &v, float g, const AdamScalars &a)
{
    const float gv = g * a.gscale;
    const float mn = m + a.one_minus_b1 * (gv - m);
    const float vn = a.b2 * v + (a.one_minus_b2 * gv) * gv;
    m = mn;
    v = vn;
    const float denom = sqrtf(vn) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (mn / denom);
}
