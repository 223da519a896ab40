#!/usr/bin/env python3
"""32-channel 3x3 conv on a 721x1440 grid (BASELINE config 4b shape, batch 2): the generic direct
conv2d kernels against the MFMA implicit-GEMM conv3d kernels fed with D = kD = 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weather_model_amd import ops
N, C, H, W = 2, 32, 721, 1440
x = torch.rand(N, C, H, W, device="cuda")
w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) / 17
b = torch.rand(C, device="cuda")
gy = torch.rand(N, C, H - 2, W - 2, device="cuda")
flops = 2 * C * C * 9 * N * (H - 2) * (W - 2)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(reps): fn()
    e[1].record(); torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / reps
for name, fn in [
    ("conv2d generic fwd", lambda: ops.conv2d_forward(x, w, b)),
    ("conv3d mfma    fwd", lambda: ops.conv3d_forward(x.unsqueeze(2), w.unsqueeze(2), b)),
    ("conv2d generic bwd_data", lambda: ops.conv2d_backward_data(gy, w, x.shape)),
    ("conv3d mfma    bwd_data", lambda: ops.conv3d_backward_data(gy.unsqueeze(2), w.unsqueeze(2), x.unsqueeze(2).shape)),
    ("conv2d generic bwd_wb", lambda: ops.conv2d_backward_wb(x, gy, w.shape)),
    ("conv3d mfma    bwd_wb", lambda: ops.conv3d_backward_wb(x.unsqueeze(2), gy.unsqueeze(2), w.unsqueeze(2).shape)),
]:
    ms = timeit(fn)
    print(f"{name:26s} {ms:9.3f} ms  {flops / ms / 1e9:8.2f} TFLOP/s", flush=True)
