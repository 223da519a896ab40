"""Where the conv32 weight-gradient kernel's time goes, from cycle stamps (debug build:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWM_WB_STAMPS -Iinclude \
        -Iweather_model_amd/csrc weather_model_amd/csrc/*.hip -o tools/probe/variants/libwm_wbs.so)"""
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
print("call ms", e[0].elapsed_time(e[1]))
L = _lib.lib()
n = 1024 * 9
buf = (ctypes.c_ulonglong * n)()
L.wm_debug_read_wb_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.wm_debug_read_wb_stamps(ctypes.byref(buf), n) == 0
allraw = np.array(list(buf), dtype=np.uint64)
raw = allraw[:8192].reshape(1024, 8)
hwid = allraw[8192:]
a = raw.astype(np.float64)
a = a[a[:, 6] > 0]
print("workgroups", len(a))
units = a[:, 6]
mhz = a[:, 3] / a[:, 4] * 100.0
print("shader clock MHz (cycles / 100 MHz wall): mean %.0f min %.0f max %.0f" % (mhz.mean(), mhz.min(), mhz.max()))
print("wall us per workgroup: mean %.1f min %.1f max %.1f" % ((a[:, 4] / 100).mean(), (a[:, 4] / 100).min(), (a[:, 4] / 100).max()))
start = (a[:, 5] - a[:, 5].min()) / 100
print("start skew us: max %.1f" % start.max())
if os.environ.get("WB_LIB", "").find("old") < 0:
    sel = raw[:, 6] > 0
    wr = (raw[sel, 7] >> np.uint64(32)).astype(np.float64)
    b2 = (raw[sel, 7] & np.uint64(0xffffffff)).astype(np.float64)
    print("  wait at barrier 1 %.0f, LDS writes (+vmcnt) %.0f, wait at barrier 2 %.0f clk/unit" % (
        (a[:, 2] / units).mean(), (wr / units).mean(), (b2 / units).mean()))
for k, nm in ((0, "fill (barrier..barrier)"), (1, "mfma steps")):
    print("%-26s %8.0f clk/unit   %5.1f %% of the workgroup's cycles" % (nm, (a[:, k] / units).mean(), 100 * (a[:, k] / a[:, 3]).mean()))
print("mfma issue floor: 160 MFMAs x 32 clk = 5120 clk/unit/wave; 3 waves per SIMD -> 15360 clk per round")
print("cycles per unit per workgroup: %.0f" % (a[:, 3] / units).mean())

if os.environ.get("WB_DETAIL"):
    nb = len(a)
    wall = a[:, 4] / 100
    idx = np.arange(nb)
    hw = hwid[:nb].astype(np.int64)
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3; wv = hw & 15
    xcd = idx & 7
    print("per XCD mean wall:", [round(float(wall[xcd == c].mean()), 0) for c in range(8)])
    order = np.argsort(a[:, 5])
    print("first-started half mean wall %.0f, second half %.0f" % (wall[order[:nb // 2]].mean(), wall[order[nb // 2:]].mean()))
    key = xcd * 4096 + se * 512 + sh * 256 + cu
    pairs = {}
    for i in range(nb):
        pairs.setdefault(int(key[i]), []).append(i)
    sizes = np.bincount([len(v) for v in pairs.values()])
    print("workgroups per (xcd,se,sh,cu):", sizes)
    fast = [min(wall[v]) for v in pairs.values() if len(v) == 2]
    slow = [max(wall[v]) for v in pairs.values() if len(v) == 2]
    print("pairs on one CU: faster %.0f, slower %.0f us (means)" % (np.mean(fast), np.mean(slow)))
    print("wave slot ids of the faster / slower of a pair:", np.bincount([int(wv[v[int(np.argmin(wall[v]))]]) for v in pairs.values() if len(v) == 2], minlength=8), np.bincount([int(wv[v[int(np.argmax(wall[v]))]]) for v in pairs.values() if len(v) == 2], minlength=8))
    print("wall histogram:", np.histogram(wall, bins=10))

    fastset = np.array([v[int(np.argmin(wall[v]))] for v in pairs.values() if len(v) == 2])
    slowset = np.array([v[int(np.argmax(wall[v]))] for v in pairs.values() if len(v) == 2])
    sel = raw[:, 6] > 0
    wr = (raw[sel, 7] >> np.uint64(32)).astype(np.float64)
    b2 = (raw[sel, 7] & np.uint64(0xffffffff)).astype(np.float64)
    for nm, st in (("older (slot 0)", fastset), ("younger (slot 1)", slowset)):
        u = units[st]
        print("%-18s clk/unit: barrier1 %.0f  writes %.0f  barrier2 %.0f  fill total %.0f  mfma steps %.0f  all %.0f" % (
            nm, (a[st, 2] / u).mean(), (wr[st] / u).mean(), (b2[st] / u).mean(), (a[st, 0] / u).mean(),
            (a[st, 1] / u).mean(), (a[st, 3] / u).mean()))
