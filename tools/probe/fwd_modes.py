import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weather_model_amd import _lib
lib = _lib.lib()
N, C, H, W = 2, 32, 721, 1440
x = torch.rand(N, C, H, W, device="cuda")
w = (torch.rand(C, C, 3, 3, device="cuda") - 0.5) / 17
b = torch.rand(C, device="cuda")
y = torch.empty(N, C, H - 2, W - 2, device="cuda")
flops = 2 * C * C * 9 * N * (H - 2) * (W - 2)
st = torch.cuda.current_stream().cuda_stream
for mode in (0, 2, 4, 6):
    def fn():
        rc = lib.wm_conv3d_fwd_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, 1, H, W, C, 1, 3, 3, mode, st)
        assert rc == 0
    fn(); torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(5): fn()
    e[1].record(); torch.cuda.synchronize()
    ms = e[0].elapsed_time(e[1]) / 5
    print(f"mode {mode}: {ms:.3f} ms {flops/ms/1e9:.1f} TF", flush=True)
