// Standalone timing harness for the 32-channel 3x3 implicit-GEMM conv passes (BASELINE config 4b
// shape): links conv3d.hip + common.hip directly, so kernel variants can be built with -D flags.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iweather_model_amd/csrc \
//         tools/probe/conv32_native.cpp weather_model_amd/csrc/conv3d.hip \
//         weather_model_amd/csrc/common.hip -o tools/probe/variants/conv32_native
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "wm.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 2, C = 32, H = 721, W = 1440, OH = H - 2, OW = W - 2;
    const size_t nx = (size_t)N * C * H * W, ny = (size_t)N * C * OH * OW, nw = (size_t)C * C * 9;
    std::vector<float> hx(nx), hw(nw), hb(C), hy(ny);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (auto &v : hx) v = rnd();
    for (auto &v : hw) v = (rnd() - 0.5f) / 17.0f;
    for (auto &v : hb) v = rnd();
    for (auto &v : hy) v = rnd() - 0.5f;
    float *x, *w, *b, *y, *gx, *gw, *gb;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&b, C * 4));
    CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&gx, nx * 4)); CK(hipMalloc(&gw, nw * 4)); CK(hipMalloc(&gb, C * 4));
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice));
    const size_t wsb = wm_conv3d_bwd_wb_workspace_bytes(N, C, 1, H, W, C, 1, 3, 3);
    void *ws; CK(hipMalloc(&ws, wsb ? wsb : 8));
    const double flops = 2.0 * C * C * 9 * N * OH * OW;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 3; ++pass) {
        auto run = [&]() -> int {
            if (pass == 0) return wm_conv3d_fwd_f32(x, w, b, y, N, C, 1, H, W, C, 1, 3, 3, 0, nullptr);
            if (pass == 1) return wm_conv3d_bwd_data_f32(y, w, gx, N, C, 1, H, W, C, 1, 3, 3, nullptr);
            return wm_conv3d_bwd_wb_f32(x, y, gw, gb, ws, wsb, N, C, 1, H, W, C, 1, 3, 3, nullptr);
        };
        if (pass == 1) CK(hipMemcpy(y, hy.data(), ny * 4, hipMemcpyHostToDevice));
        if (run()) { printf("error: %s\n", wm_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        const int reps = 5;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) run();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        // checksum of the output for variant-to-variant comparison
        const size_t no = pass == 0 ? ny : pass == 1 ? nx : nw;
        float *src = pass == 0 ? y : pass == 1 ? gx : gw;
        std::vector<float> ho(no);
        CK(hipMemcpy(ho.data(), src, no * 4, hipMemcpyDeviceToHost));
        double cs = 0; for (size_t i = 0; i < no; i += 7) cs += ho[i];
        printf("%s N=%d: %.3f ms %.1f TFLOP/s checksum %.6e\n", pass == 0 ? "fwd" : pass == 1 ? "bwd_data" : "bwd_wb", N, ms, flops / ms / 1e9, cs);
    }
    return 0;
}
