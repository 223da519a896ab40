"""Times of the field kernels (stats incl. its finalize launch, normalise) over 64x721x1440 fp32,
rotating three buffers, for a library variant: WB_LIB=<file under tools/probe/variants>.  Also
checks the statistics against torch on the device."""
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
n, K, nset = 64 * 721 * 1440, 20, 3
xs = [ops.synth_uniform((n,), seed=2, offset=k * n) for k in range(nset)]
ys = [torch.empty_like(xs[0]) for _ in range(nset)]
ws = torch.empty((L.wm_field_stats_workspace_bytes(n),), device="cuda", dtype=torch.uint8)
out = torch.zeros((5,), device="cuda", dtype=torch.float64)
st = 0
def stats(k): _lib.check(L.wm_field_stats_f32(xs[k].data_ptr(), n, out.data_ptr(), 0, ws.data_ptr(), ws.numel(), st))
def norm(k): _lib.check(L.wm_affine_scale_f32(xs[k].data_ptr(), ys[k].data_ptr(), n, 0.25, 1.5, st))
T, Z, Y, X, V = 9, 1, 721, 1440, 6
vol = ops.synth_uniform((T, Z, Y, X, V), seed=80)
def gath(k): ops.gather_triplets(vol, (T - 6, Z, Y, X), first=0, count=Y * X)
res_s = []
for name, fn in (("stats", stats), ("normalise", norm), ("gather", gath)):
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
    res_s.append("%s %.1f" % (name, float(np.median(res))))
for m in (n, n - 1, n - 255, 1000003, 63, 5):     # ragged sizes against torch
    stats_x = xs[1][:m]
    _lib.check(L.wm_field_stats_f32(stats_x.data_ptr(), m, out.data_ptr(), 0, ws.data_ptr(), ws.numel(), st))
    torch.cuda.synchronize()
    d = stats_x.double()
    ref = torch.stack([d.min(), d.max(), d.sum(), (d * d).sum(), torch.tensor(float(m), device="cuda", dtype=torch.float64)])
    err = ((out - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
    assert err < 1e-12, (m, out, ref)
print("%-22s us: " % os.environ.get("WB_LIB", "libwm.so") + "  ".join(res_s) + "  stats ok")
