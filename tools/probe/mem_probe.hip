// mem_probe.hip -- access-pattern micro-benchmarks behind the stencil kernel design (MI355X).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/mem_probe tools/probe/mem_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void flat_copy(const f4* __restrict__ in, f4* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void flat_copy_nt(const f4* __restrict__ in, f4* __restrict__ out, long long n4, int ntload) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f4 v = ntload ? __builtin_nontemporal_load(&in[i]) : in[i];
        __builtin_nontemporal_store(v, &out[i]);
    }
}
// chunked copy: each wave copies a contiguous range, 4 KiB per iteration
__global__ __launch_bounds__(256) void chunk_copy(const f4* __restrict__ in, f4* __restrict__ out, long long n4, int nt) {
    const long long nw = (long long)gridDim.x * 4, w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long t0 = (n4 / 64) * w / nw, t1 = (n4 / 64) * (w + 1) / nw;
    const int lane = threadIdx.x & 63;
    for (long long t = t0; t + 4 <= t1; t += 4) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = in[(t + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) { if (nt) __builtin_nontemporal_store(v[u], &out[(t + u) * 64 + lane]); else out[(t + u) * 64 + lane] = v[u]; }
    }
}
__global__ __launch_bounds__(256) void flat_read(const f4* __restrict__ in, float* __restrict__ out, long long n4) {
    f4 s = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) s += in[i];
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}
// chunked read: each wave reads contiguous CH KiB per iteration, waves own contiguous ranges (gram-like)
__global__ __launch_bounds__(256) void chunk_read(const f4* __restrict__ in, float* __restrict__ out, long long n4) {
    const long long nw = (long long)gridDim.x * 4, w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long t0 = (n4 / 64) * w / nw, t1 = (n4 / 64) * (w + 1) / nw;
    const int lane = threadIdx.x & 63;
    f4 s = {0, 0, 0, 0};
    for (long long t = t0; t + 8 <= t1; t += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = in[(t + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;
}
// row-walk: thread owns a 4-column strip and walks `rows` consecutive rows of a W-wide image stream.
// MODE bit0: extra float2 load; bit1: store (else read-only accumulate); bit2: second input stream (gy-like, OW stride)
template <int MODE, int PFD>
__global__ __launch_bounds__(256) void row_walk(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ y,
                                                int W, int OW, int strips, long long total_rows, int segs) {
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tid >= (long long)strips * segs) return;
    const int strip = tid % strips, seg = tid / strips;
    const long long r0 = total_rows * seg / segs, r1 = total_rows * (seg + 1) / segs;
    const int j0 = strip * 4;
    const float* xp = x + r0 * W + j0;
    const float* gp = g + r0 * OW + j0;
    float* yp = y + r0 * OW + j0;
    const bool full = j0 + 6 <= W, ofull = j0 + 4 <= OW;
    f4u pre[PFD]; f2u pre2[PFD]; f4u gpre[PFD];
    const long long steps = r1 - r0;
#pragma unroll
    for (int u = 0; u < PFD; ++u) {
        pre[u] = (f4u){0,0,0,0}; pre2[u] = (f2u){0,0}; gpre[u] = (f4u){0,0,0,0};
        if (u < steps) {
            pre[u] = *(const f4u*)(xp + (size_t)u * W);
            if ((MODE & 1) && full) pre2[u] = *(const f2u*)(xp + (size_t)u * W + 4);
            if ((MODE & 4) && ofull) gpre[u] = *(const f4u*)(gp + (size_t)u * OW);
        }
    }
    f4u acc = {0, 0, 0, 0};
    for (long long s = 0; s < steps; s += PFD) {
#pragma unroll
        for (int u = 0; u < PFD; ++u) {
            if (s + u < steps) {
                f4u v = pre[u]; f2u v2 = pre2[u]; f4u gv = gpre[u];
                if (s + u + PFD < steps) {
                    pre[u] = *(const f4u*)(xp + (size_t)(s + u + PFD) * W);
                    if ((MODE & 1) && full) pre2[u] = *(const f2u*)(xp + (size_t)(s + u + PFD) * W + 4);
                    if ((MODE & 4) && ofull) gpre[u] = *(const f4u*)(gp + (size_t)(s + u + PFD) * OW);
                }
                v.x += v2.x; v.y += v2.y;
                if (MODE & 4) v += gv;
                if (MODE & 2) { if (ofull) { if (MODE & 8) __builtin_nontemporal_store(v, (f4u*)(yp + (size_t)(s + u) * OW)); else *(f4u*)(yp + (size_t)(s + u) * OW) = v; } }
                else acc += v;
            }
        }
    }
    if (!(MODE & 2) && acc.x + acc.y + acc.z + acc.w == 12345.678f) y[0] = acc.x;
}

// band sweep: item = (band of R output rows, 4-column strip), items taken grid-stride in band-major
// order so the blocks in flight cover a compact window of rows.  Loads R+2 rows (f4+f2), stores R.
template <int R, int NT>
__global__ __launch_bounds__(256) void band_copy(const float* __restrict__ x, float* __restrict__ y, int W, int OW, int strips, int total_rows) {
    const int bands = (total_rows + R - 1) / R;
    const long long items = (long long)bands * strips;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
        const int band = it / strips, strip = it - (long long)band * strips;
        const int j0 = min(strip * 4, OW - 4);
        const int r0 = band * R;
        const int cnt = min(R, total_rows - r0 - 2);
        if (cnt <= 0) continue;
        const float* xp = x + (size_t)r0 * W + j0;
        float* yp = y + (size_t)r0 * OW + j0;
        f4u v[R + 2]; f2u v2[R + 2];
#pragma unroll
        for (int u = 0; u < R + 2; ++u) { const int uu = min(u, cnt + 1); v[u] = *(const f4u*)(xp + (size_t)uu * W); v2[u] = *(const f2u*)(xp + (size_t)uu * W + 4); }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (u < cnt) {
                f4u o = v[u] + v[u + 1] + v[u + 2]; o.x += v2[u].x + v2[u + 1].y + v2[u + 2].x;
                if (NT) __builtin_nontemporal_store(o, (f4u*)(yp + (size_t)u * OW)); else *(f4u*)(yp + (size_t)u * OW) = o;
            }
        }
    }
}

template <typename F>
float time_ms(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(r); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const int N = 64, H = 721, W = 1440, OW = 1438;
    const long long n = (long long)N * H * W;
    const int NB = 3;
    float *x[NB], *g[NB], *y[NB];
    for (int k = 0; k < NB; ++k) {
        CK(hipMalloc(&x[k], n * 4)); CK(hipMalloc(&g[k], n * 4)); CK(hipMalloc(&y[k], n * 4));
        CK(hipMemset(x[k], 0x3c, n * 4)); CK(hipMemset(g[k], 0x3c, n * 4)); CK(hipMemset(y[k], 0, n * 4));
    }
    CK(hipDeviceSynchronize());
    const double MB = n * 4 / 1e6;
    const int reps = 15;
    for (int grid : {1024, 2048, 4096}) {
        float ms = time_ms([&](int r) { hipLaunchKernelGGL(flat_copy, dim3(grid), dim3(256), 0, 0, (const f4*)x[r % NB], (f4*)y[r % NB], n / 4); }, reps);
        printf("flat_copy   grid %5d: %.1f us  %.0f GB/s (r+w)\n", grid, ms * 1e3, 2 * MB / ms);
        ms = time_ms([&](int r) { hipLaunchKernelGGL(flat_read, dim3(grid), dim3(256), 0, 0, (const f4*)x[r % NB], y[0], n / 4); }, reps);
        printf("flat_read   grid %5d: %.1f us  %.0f GB/s\n", grid, ms * 1e3, MB / ms);
    }
    for (int grid : {256, 512, 768, 1024, 1536, 2048}) {
        float ms = time_ms([&](int r) { hipLaunchKernelGGL(flat_copy, dim3(grid), dim3(256), 0, 0, (const f4*)x[r % NB], (f4*)y[r % NB], n / 4); }, reps);
        printf("flat_copy      grid %5d: %.1f us  %.0f GB/s (r+w)\n", grid, ms * 1e3, 2 * MB / ms);
        ms = time_ms([&](int r) { hipLaunchKernelGGL(flat_copy_nt, dim3(grid), dim3(256), 0, 0, (const f4*)x[r % NB], (f4*)y[r % NB], n / 4, 0); }, reps);
        printf("flat_copy_nt   grid %5d: %.1f us  %.0f GB/s (r+w)\n", grid, ms * 1e3, 2 * MB / ms);
        ms = time_ms([&](int r) { hipLaunchKernelGGL(flat_copy_nt, dim3(grid), dim3(256), 0, 0, (const f4*)x[r % NB], (f4*)y[r % NB], n / 4, 1); }, reps);
        printf("flat_copy_ntnt grid %5d: %.1f us  %.0f GB/s (r+w)\n", grid, ms * 1e3, 2 * MB / ms);
        ms = time_ms([&](int r) { hipLaunchKernelGGL(chunk_copy, dim3(grid), dim3(256), 0, 0, (const f4*)x[r % NB], (f4*)y[r % NB], n / 4, 0); }, reps);
        printf("chunk_copy     grid %5d: %.1f us  %.0f GB/s (r+w)\n", grid, ms * 1e3, 2 * MB / ms);
        ms = time_ms([&](int r) { hipLaunchKernelGGL(chunk_copy, dim3(grid), dim3(256), 0, 0, (const f4*)x[r % NB], (f4*)y[r % NB], n / 4, 1); }, reps);
        printf("chunk_copy_nt  grid %5d: %.1f us  %.0f GB/s (r+w)\n", grid, ms * 1e3, 2 * MB / ms);
    }
    for (int grid : {512, 1024, 2048}) {
        float ms = time_ms([&](int r) { hipLaunchKernelGGL(chunk_read, dim3(grid), dim3(256), 0, 0, (const f4*)x[r % NB], y[0], n / 4); }, reps);
        printf("chunk_read  grid %5d: %.1f us  %.0f GB/s\n", grid, ms * 1e3, MB / ms);
    }
    const int strips = W / 4;
    const long long rows = (long long)N * H;
#define RW(MODE, PFD, bpc, label, bytes_mb) { \
        const int segs = 256 * bpc * 256 / strips; const int blocks = (strips * segs + 255) / 256; \
        float ms = time_ms([&](int r) { hipLaunchKernelGGL((row_walk<MODE, PFD>), dim3(blocks), dim3(256), 0, 0, x[r % NB], g[r % NB], y[r % NB], W, OW, strips, rows, segs); }, reps); \
        printf("row_walk %-28s pf %d bpc %d: %.1f us  %.0f GB/s\n", label, PFD, bpc, ms * 1e3, (bytes_mb) / ms); }
#define BC(R, NT, grid, ow) { float ms = time_ms([&](int r) { hipLaunchKernelGGL((band_copy<R, NT>), dim3(grid), dim3(256), 0, 0, x[r % NB], y[r % NB], W, ow, strips, (int)rows); }, reps); \
        printf("band_copy R %2d nt %d grid %5d ow %d: %.1f us  %.0f GB/s\n", R, NT, grid, ow, ms * 1e3, 2 * MB / ms); }
    for (int grid : {768, 1024, 2048}) {
        BC(6, 0, grid, 1438) BC(6, 0, grid, 1440) BC(6, 0, grid, 1436) BC(6, 1, grid, 1440) BC(14, 0, grid, 1440) BC(14, 1, grid, 1440)
    }
    for (int bpc : {4}) {
        RW(3, 1, bpc, "copy x + f2 -> y(1438)", 2 * MB)
        RW(3, 4, bpc, "copy x + f2 -> y(1438)", 2 * MB)
        RW(11, 1, bpc, "copy x + f2 -> y(1438) nt", 2 * MB)
        RW(11, 4, bpc, "copy x + f2 -> y(1438) nt", 2 * MB)
    }
    return 0;
}
