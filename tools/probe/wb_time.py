"""Time of the conv32 weight-gradient call (events around 10 calls, best and median of 7 rounds) for
a library variant: WB_LIB=<file under tools/probe/variants> or unset for the built library."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
from weather_model_amd import ops
N, C, H, W = 8, 32, 721, 1440
x = ops.synth_uniform((N, C, H, W), seed=30)
gy = ops.synth_uniform((N, C, H - 2, W - 2), seed=33, lo=-1.0, hi=1.0)
ws = ops.conv2d_bwd_wb_workspace(x.shape, (C, C, 3, 3))
gw = torch.empty((C, C, 3, 3), device="cuda"); gb = torch.empty((C,), device="cuda")
for _ in range(3):
    ops.conv2d_backward_wb(x, gy, (C, C, 3, 3), ws=ws, out_w=gw, out_b=gb)
torch.cuda.synchronize()
res = []
for r in range(7):
    time.sleep(0.05)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(10):
        ops.conv2d_backward_wb(x, gy, (C, C, 3, 3), ws=ws, out_w=gw, out_b=gb)
    e[1].record()
    torch.cuda.synchronize()
    res.append(e[0].elapsed_time(e[1]) / 10)
fl = 2.0 * C * C * 9 * N * (H - 2) * (W - 2)
print("%-28s BPC=%s  ms/call best %.4f median %.4f  -> %.1f TFLOP/s (median), %.3f of 157.3" % (
    os.environ.get("WB_LIB", "libwm.so"), os.environ.get("WM_WB_BPC", "-"), min(res), float(np.median(res)),
    fl / np.median(res) / 1e9, fl / np.median(res) / 1e9 / 157.3))
