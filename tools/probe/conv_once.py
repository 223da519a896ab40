#!/usr/bin/env python3
"""Run the config-3 conv forward + backward(weight,bias) a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weather_model_amd import ops
N, H, W = 64, 721, 1440
x = ops.synth_uniform((N, 1, H, W), seed=2)
gy = ops.synth_uniform((N, 1, H - 2, W - 2), seed=4, lo=-1, hi=1)
w = ops.synth_uniform((1, 1, 3, 3), seed=3, lo=-1 / 3, hi=1 / 3)
b = ops.synth_uniform((1,), seed=5)
for _ in range(3):
    y = ops.conv2d_forward(x, w, b)
    gw, gb = ops.conv2d_backward_wb(x, gy, (1, 1, 3, 3))
torch.cuda.synchronize()
print("done", float(gw.sum()))
