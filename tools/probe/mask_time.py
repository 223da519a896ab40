"""relu_mask (out of place, 8x32x721x1440) and affine_scale (64x721x1440) times; WB_LIB selects the library."""
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
n = 8 * 32 * 721 * 1440
g = ops.synth_uniform((n,), seed=1, lo=-1.0, hi=1.0); y = ops.synth_uniform((n,), seed=2, lo=-1.0, hi=1.0); o = torch.empty_like(g)
n2 = 64 * 721 * 1440
def mask(): _lib.check(L.wm_relu_mask_to_f32(g.data_ptr(), y.data_ptr(), o.data_ptr(), n, 0))
def norm(): _lib.check(L.wm_affine_scale_f32(g.data_ptr(), o.data_ptr(), n2, 0.25, 1.5, 0))
def norm_big(): _lib.check(L.wm_affine_scale_f32(g.data_ptr(), o.data_ptr(), n, 0.25, 1.5, 0))
out = []
for name, fn, K in (("relu_mask", mask, 10), ("normalise", norm, 20), ("normalise_1GB", norm_big, 10)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    res = []
    for r in range(5):
        time.sleep(0.03)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(K): fn()
        e[1].record(); torch.cuda.synchronize()
        res.append(e[0].elapsed_time(e[1]) / K * 1e3)
    out.append("%s %.1f" % (name, float(np.median(res))))
ref = torch.where(y > 0, g, torch.zeros_like(g)); mask(); torch.cuda.synchronize()
assert torch.equal(o, ref)
print("%-22s us: " % os.environ.get("WB_LIB", "libwm.so") + "  ".join(out))
