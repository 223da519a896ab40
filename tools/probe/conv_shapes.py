#!/usr/bin/env python3
"""Forward / backward stencil time by kernel shape on the 64 x 721 x 1440 grid (steady state per
kernel, 3 rotating buffer sets): separates the cost of output-row misalignment and column overlap
((3,3): OW = 1438) from the read+write pattern itself ((3,1), (1,1): OW = 1440, aligned)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weather_model_amd import _lib
L = _lib.lib()
N, H, W = 64, 721, 1440
nset = 3
st = torch.cuda.current_stream().cuda_stream
xs = [torch.rand(N, 1, H, W, device="cuda") for _ in range(nset)]
ws = torch.empty(1 << 20, device="cuda", dtype=torch.uint8)
import os
SHAPES = [(3, 3), (1, 1)] if os.environ.get('WM_EXP_ROWS') else [(3, 3), (3, 1), (1, 3), (1, 1), (5, 5), (2, 2)]
for kh, kw in SHAPES:
    OH, OW = H - kh + 1, W - kw + 1
    ys = [torch.empty(N, 1, OH, OW, device="cuda") for _ in range(nset)]
    gys = [torch.rand(N, 1, OH, OW, device="cuda") for _ in range(nset)]
    w = torch.rand(1, 1, kh, kw, device="cuda"); b = torch.rand(1, device="cuda")
    gw = torch.zeros(kh * kw + 1, device="cuda")
    def fwd(i): _lib.check(L.wm_conv2d_fwd_f32(xs[i].data_ptr(), w.data_ptr(), b.data_ptr(), ys[i].data_ptr(), N, 1, H, W, 1, kh, kw, 0, st))
    def bwd(i): _lib.check(L.wm_conv2d_bwd_wb_f32(xs[i].data_ptr(), gys[i].data_ptr(), gw.data_ptr(), gw[kh * kw:].data_ptr(), ws.data_ptr(), ws.numel(), N, 1, H, W, 1, kh, kw, st))
    def timeit(fns, reps=15):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(reps + 3):
            if rep == 3: ev[0].record()
            for f in fns: f(rep % nset)
        ev[1].record(); torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps * 1e3
    mb = 4 * (N * H * W + N * OH * OW) / 1e6
    f, bw, fb = timeit([fwd]), timeit([bwd]), timeit([fwd, bwd])
    print(f"k=({kh},{kw}) OW={OW}: fwd {f:6.1f} us {mb/f*1e3:5.0f} GB/s | bwd {bw:6.1f} us {mb/bw*1e3:5.0f} GB/s | fwd+bwd {fb:6.1f} us {2*mb/fb*1e3:5.0f} GB/s", flush=True)
    del ys, gys
