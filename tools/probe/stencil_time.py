"""Per-kernel times of the three flat stencil kernels (events around 20 back-to-back launches each,
rotating buffer sets), for a library variant: WB_LIB=<file under tools/probe/variants>."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
from weather_model_amd import ops
L = _lib.lib()
N, H, W, K, nset = 64, 721, 1440, 20, 3
w = ops.synth_uniform((1, 1, 3, 3), seed=3, lo=-1 / 3, hi=1 / 3)
b = ops.synth_uniform((1,), seed=5, lo=-1 / 3, hi=1 / 3)
xs = [ops.synth_uniform((N, 1, H, W), seed=2, offset=k * N * H * W) for k in range(nset)]
gys = [ops.synth_uniform((N, 1, H - 2, W - 2), seed=4, lo=-1.0, hi=1.0, offset=k * N * H * W) for k in range(nset)]
ys = [torch.empty((N, 1, H - 2, W - 2), device="cuda") for _ in range(nset)]
gxs = [torch.empty((N, 1, H, W), device="cuda") for _ in range(nset)]
ws = torch.empty((L.wm_conv2d_bwd_wb_workspace_bytes(N, 1, H, W, 1, 3, 3),), device="cuda", dtype=torch.uint8)
grads = torch.zeros((10,), device="cuda")
st = 0
def fwd(k): _lib.check(L.wm_conv2d_fwd_f32(xs[k].data_ptr(), w.data_ptr(), b.data_ptr(), ys[k].data_ptr(), N, 1, H, W, 1, 3, 3, 0, st))
def bww(k): _lib.check(L.wm_conv2d_bwd_wb_f32(xs[k].data_ptr(), gys[k].data_ptr(), grads.data_ptr(), grads[9:].data_ptr(), ws.data_ptr(), ws.numel(), N, 1, H, W, 1, 3, 3, st))
def bwd(k): _lib.check(L.wm_conv2d_bwd_data_f32(gys[k].data_ptr(), w.data_ptr(), gxs[k].data_ptr(), N, 1, H, W, 1, 3, 3, st))
def pair(k): fwd(k); bww(k)
def trio(k): fwd(k); bww(k); bwd(k)
out = []
for name, fn in (("fwd", fwd), ("bwd_wb", bww), ("bwd_data", bwd), ("pair", pair), ("trio", trio)):
    for i in range(4): fn(i % nset)
    torch.cuda.synchronize()
    res = []
    for r in range(5):
        time.sleep(0.03)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for i in range(K): fn(i % nset)
        e[1].record(); torch.cuda.synchronize()
        res.append(e[0].elapsed_time(e[1]) / K * 1e3)
    out.append("%s %.1f" % (name, float(np.median(res))))
print("%-22s us: " % os.environ.get("WB_LIB", "libwm.so") + "  ".join(out))
