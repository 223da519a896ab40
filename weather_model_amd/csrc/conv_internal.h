// Internal interface between conv3d.hip's dispatch and conv_ring.hip (not part of include/wm.h).
#pragma once
#include <hip/hip_runtime.h>

// 2-D (depth 1) multi-channel convolution through the LDS-ring kernel: forward (bwd = 0: in = x,
// aux = bias or null, out = y) or input gradient (bwd = 1: in = gy, aux = ReLU mask or null,
// out = gx); w is the layer's (Cout, Cin, kH, kW) weight in both cases.
// Returns 1 = launched, 0 = shape not taken, < 0 = WM_ERR_*.
int wm_conv_ring_launch(int bwd, const float *in, const float *w, const float *aux, float *out, int B,
                        int Cin, int H, int W, int Cout, int kH, int kW, int relu, hipStream_t st);
