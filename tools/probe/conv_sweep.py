#!/usr/bin/env python3
"""Sweep the forward stencil's band height / resident blocks (env knobs) in one process; steady
state per kernel (a kernel also pays for its predecessor's dirty lines)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weather_model_amd import _lib
L = _lib.lib()
N, H, W = 64, 721, 1440
nset = 3
xs = [torch.rand(N, 1, H, W, device="cuda") for _ in range(nset)]
gys = [torch.rand(N, 1, H - 2, W - 2, device="cuda") * 2 - 1 for _ in range(nset)]
ys = [torch.empty(N, 1, H - 2, W - 2, device="cuda") for _ in range(nset)]
w = torch.rand(1, 1, 3, 3, device="cuda"); b = torch.rand(1, device="cuda")
gw = torch.zeros(10, device="cuda")
ws = torch.empty(1 << 20, device="cuda", dtype=torch.uint8)
st = torch.cuda.current_stream().cuda_stream
mb = 4 * (N * H * W + N * (H - 2) * (W - 2)) / 1e6
def fwd(i): _lib.check(L.wm_conv2d_fwd_f32(xs[i].data_ptr(), w.data_ptr(), b.data_ptr(), ys[i].data_ptr(), N, 1, H, W, 1, 3, 3, 0, st))
def bwd(i): _lib.check(L.wm_conv2d_bwd_wb_f32(xs[i].data_ptr(), gys[i].data_ptr(), gw.data_ptr(), gw[9:].data_ptr(), ws.data_ptr(), ws.numel(), N, 1, H, W, 1, 3, 3, st))
def timeit(fns, reps=12):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(reps + 2):
        if rep == 2: ev[0].record()
        for f in fns: f(rep % nset)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3
for R, bpc in itertools.product([2, 4, 6, 10, 14, 22, 46], [2, 3, 4, 6]):
    os.environ["WM_STENCIL_R"] = str(R); os.environ["WM_STENCIL_GRID_BPC"] = str(bpc)
    f = timeit([fwd]); fb = timeit([fwd, bwd])
    print(f"R {R:3d} bpc {bpc}: fwd solo {f:6.1f} us {mb/f*1e3:6.0f} GB/s | fwd+bwd {fb:6.1f} us {2*mb/fb*1e3:6.0f} GB/s")
os.environ.pop("WM_STENCIL_R"); os.environ.pop("WM_STENCIL_GRID_BPC")
print("bwd solo", timeit([bwd]))
