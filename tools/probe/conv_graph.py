"""conv3x3 fwd + bwd(w,b) step on the config-3 grid: eager launches vs one HIP graph of three
steps (one per rotating buffer set)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weather_model_amd import ops, _lib
L = _lib.lib()
N, H, W, nset = 64, 721, 1440, 3
w = ops.synth_uniform((1, 1, 3, 3), seed=3, lo=-1 / 3, hi=1 / 3)
b = ops.synth_uniform((1,), seed=5, lo=-1 / 3, hi=1 / 3)
xs = [ops.synth_uniform((N, 1, H, W), seed=2, offset=k * N * H * W) for k in range(nset)]
gys = [ops.synth_uniform((N, 1, H - 2, W - 2), seed=4, lo=-1.0, hi=1.0, offset=k * N * H * W) for k in range(nset)]
ys = [torch.empty((N, 1, H - 2, W - 2), device="cuda") for _ in range(nset)]
ws = torch.empty((L.wm_conv2d_bwd_wb_workspace_bytes(N, 1, H, W, 1, 3, 3),), device="cuda", dtype=torch.uint8)
grads = torch.zeros((10,), device="cuda")
def step(k):
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.wm_conv2d_fwd_f32(xs[k].data_ptr(), w.data_ptr(), b.data_ptr(), ys[k].data_ptr(), N, 1, H, W, 1, 3, 3, 0, st))
    _lib.check(L.wm_conv2d_bwd_wb_f32(xs[k].data_ptr(), gys[k].data_ptr(), grads.data_ptr(), grads[9:].data_ptr(), ws.data_ptr(), ws.numel(), N, 1, H, W, 1, 3, 3, st))
for i in range(6): step(i % nset)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(60): step(i % nset)
torch.cuda.synchronize()
print(f"eager  {1e6 * (time.perf_counter() - t0) / 60:.1f} us/step", flush=True)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step(0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for k in range(nset): step(k)
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): g.replay()
torch.cuda.synchronize()
print(f"graph  {1e6 * (time.perf_counter() - t0) / 60:.1f} us/step", flush=True)
