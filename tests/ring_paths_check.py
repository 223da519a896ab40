"""Child process of tests/test_gpu_conv.py::test_conv_ring_forward_and_masked_paths_in_a_subprocess:
run with WM_CONV_RING=all so that forward, input gradient and masked input gradient of ring-eligible
shapes all take conv_ring.hip; compared with the float64 oracle at 1e-5 (||a-b||inf / ||b||inf)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from weather_model_amd import ops, synth      # noqa: E402
from oracle import cpu as ocpu                # noqa: E402

assert os.environ.get("WM_CONV_RING") == "all"
TOL = 1e-5


def relinf(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


for shape in [(4, 32, 140, 330, 32, 3, 3), (5, 20, 100, 300, 14, 3, 3), (20, 32, 130, 66, 32, 3, 3),
              (4, 13, 151, 257, 24, 3, 3)]:
    N, Cin, H, W, Cout, kH, kW = shape
    x = synth.uniform(90, (N, Cin, H, W), -1, 1)
    w = synth.uniform(91, (Cout, Cin, kH, kW), -1, 1)
    b = synth.uniform(92, (Cout,), -1, 1)
    gy = synth.uniform(93, (N, Cout, H - kH + 1, W - kW + 1), -1, 1)
    below = np.maximum(synth.uniform(94, (N, Cin, H, W), -1, 1), 0.0).astype(np.float32)
    ref = ocpu.conv2d_fwd(x, w, b)
    y = ops.conv2d_forward(dev(x), dev(w), dev(b)).cpu().numpy()
    assert relinf(y, ref) < TOL, ("fwd", shape)
    yr = ops.conv2d_forward(dev(x), dev(w), dev(b), relu=True).cpu().numpy()
    assert relinf(yr, np.maximum(ref, 0.0)) < TOL, ("fwd relu", shape)
    refg = ocpu.conv2d_bwd_data(gy, w, x.shape)
    gx = ops.conv2d_backward_data(dev(gy), dev(w), x.shape).cpu().numpy()
    assert relinf(gx, refg) < TOL, ("bwd", shape)
    gm = ops.conv2d_backward_data(dev(gy), dev(w), x.shape, relu_mask=dev(below)).cpu().numpy()
    assert relinf(gm, refg * (below > 0)) < TOL, ("bwd masked", shape)
    assert (gm[below <= 0] == 0).all()
    # one kernel for both: the fused store is bit for bit gradient + mask
    assert np.array_equal(gm, np.where(below > 0, gx, np.float32(0.0))), ("fused == two launches", shape)
print("ring paths ok")
