"""Shader clock, cycles per unit and MFMA-pipe use of the conv32 weight-gradient kernel, from the
cycle/wall-clock stamps each workgroup leaves in a debug build:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWM_WB_STAMPS -DWM_PROBE_ENV -Iinclude \
        -Iweather_model_amd/csrc weather_model_amd/csrc/*.hip -o tools/probe/variants/libwm_wbs.so
WM_WB_BPC=1 caps the launch at one workgroup per CU.  (The fill / MFMA-step split quoted in DESIGN.md
came from the same stamps placed around the two barriers of the round-1 row kernel.)"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ.get("WB_LIB", "libwm_wbs.so"))
from weather_model_amd import ops
N, C, H, W = 8, 32, 721, 1440
x = ops.synth_uniform((N, C, H, W), seed=30)
gy = ops.synth_uniform((N, C, H - 2, W - 2), seed=33, lo=-1.0, hi=1.0)
for _ in range(3):
    ops.conv2d_backward_wb(x, gy, (C, C, 3, 3))
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
e[0].record()
ops.conv2d_backward_wb(x, gy, (C, C, 3, 3))
e[1].record()
torch.cuda.synchronize()
print("call ms %.4f" % e[0].elapsed_time(e[1]))
L = _lib.lib()
n = 1024 * 9
buf = (ctypes.c_ulonglong * n)()
L.wm_debug_read_wb_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.wm_debug_read_wb_stamps(ctypes.byref(buf), n) == 0
allraw = np.array(list(buf), dtype=np.uint64)
raw, hwid = allraw[:8192].reshape(1024, 8), allraw[8192:]
keep = raw[:, 6] > 0
a, hw = raw[keep].astype(np.float64), hwid[keep].astype(np.int64)
units, wall = a[:, 6], a[:, 4] / 100
mhz = a[:, 3] / a[:, 4] * 100.0
print("workgroups %d, units each %.1f" % (len(a), units.mean()))
print("shader clock MHz (cycles / 100 MHz wall clock): mean %.0f min %.0f max %.0f" % (mhz.mean(), mhz.min(), mhz.max()))
print("wall us per workgroup: mean %.1f min %.1f max %.1f" % (wall.mean(), wall.min(), wall.max()))
per_wg = (a[:, 3] / units).mean()
print("cycles per unit and workgroup: %.0f   (top-of-stripe fills: %.0f of them)" % (per_wg, (a[:, 0] / units).mean()))
idx = np.flatnonzero(keep)
key = (idx & 7) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15)
per_cu = {}
for i, k in enumerate(key):
    per_cu.setdefault(int(k), []).append(i)
bpc = np.mean([len(v) for v in per_cu.values()])
mf = 9 * 16 * 32
print("workgroups per CU %.2f; MFMA issue floor %d clk per unit and wave -> pipe busy %.1f %% while all are resident" % (
    bpc, mf, 100.0 * mf * bpc / per_wg))
if bpc > 1.5:
    fast = np.mean([wall[v].min() for v in per_cu.values() if len(v) == 2])
    slow = np.mean([wall[v].max() for v in per_cu.values() if len(v) == 2])
    print("the two workgroups of a CU: first-dispatched done at %.0f us, the other at %.0f us" % (fast, slow))
