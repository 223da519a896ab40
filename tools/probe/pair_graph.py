"""Stencil pair (forward + weight gradient) launched eagerly vs replayed from one HIP graph of K steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
from weather_model_amd import ops
L = _lib.lib()
N, H, W, K = 64, 721, 1440, 20
nset = 3
w = ops.synth_uniform((1, 1, 3, 3), seed=3, lo=-1 / 3, hi=1 / 3)
b = ops.synth_uniform((1,), seed=5, lo=-1 / 3, hi=1 / 3)
xs = [ops.synth_uniform((N, 1, H, W), seed=2, offset=k * N * H * W) for k in range(nset)]
gys = [ops.synth_uniform((N, 1, H - 2, W - 2), seed=4, lo=-1.0, hi=1.0, offset=k * N * H * W) for k in range(nset)]
ys = [torch.empty((N, 1, H - 2, W - 2), device="cuda") for _ in range(nset)]
ws = torch.empty((L.wm_conv2d_bwd_wb_workspace_bytes(N, 1, H, W, 1, 3, 3),), device="cuda", dtype=torch.uint8)
grads = torch.zeros((10,), device="cuda")


def pair(k, st):
    _lib.check(L.wm_conv2d_fwd_f32(xs[k].data_ptr(), w.data_ptr(), b.data_ptr(), ys[k].data_ptr(), N, 1, H, W, 1, 3, 3, 0, st))
    _lib.check(L.wm_conv2d_bwd_wb_f32(xs[k].data_ptr(), gys[k].data_ptr(), grads.data_ptr(), grads[9:].data_ptr(),
                                      ws.data_ptr(), ws.numel(), N, 1, H, W, 1, 3, 3, st))


def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        time.sleep(0.05)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record(); fn(); e[1].record(); torch.cuda.synchronize()
        out.append(e[0].elapsed_time(e[1]) / K * 1e3)
    return out


def eager():
    st = torch.cuda.current_stream().cuda_stream
    for i in range(K):
        pair(i % nset, st)


for _ in range(2):
    eager()
torch.cuda.synchronize()
print("eager  us/pair:", ["%.1f" % v for v in timed(eager)], "grads", grads[:3].tolist())
if os.environ.get("NO_GRAPH"):
    sys.exit(0)
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        st = torch.cuda.current_stream().cuda_stream
        for i in range(K):
            pair(i % nset, st)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph  us/pair:", ["%.1f" % v for v in timed(g.replay)])
print("eager  us/pair:", ["%.1f" % v for v in timed(eager)])
