"""Phase timing of the fused model kernel from cycle stamps (debug build with -DWM_MODEL_STAMPS:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWM_MODEL_STAMPS -Iinclude \
        -Iweather_model_amd/csrc weather_model_amd/csrc/*.hip -o tools/probe/variants/libwm_dbg.so)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", "libwm_dbg.so")
from weather_model_amd import ops
B = 721 * 1440
p = (torch.rand(ops.MODEL_PARAM_COUNT, device="cuda") - 0.5) * 0.2
x = torch.rand(B, 2, 6, device="cuda")
y = torch.rand(B, 6, device="cuda")
for _ in range(2):
    ops.model_train(p, x, y=y)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * 32)()
L.wm_debug_read_stamps.argtypes = [ctypes.c_void_p]
assert L.wm_debug_read_stamps(ctypes.byref(buf)) == 0
names = {1: "x->LDS", 2: "conv1-4 fwd", 3: "linear1", 4: "tanh", 6: "linear2+loss", 7: "dT",
         10: "dH4 + conv4-2 bwd", 11: "weight grads"}
st = list(buf)[:12]
prev = st[0]
for k in sorted(names):
    print(f"{names[k]:20s} {st[k] - prev:8d} clk")
    prev = st[k]
print(f"{'tile total':20s} {st[11] - st[0]:8d} clk   (a stamp's own store can add a vmcnt wait to the phase after it)")
