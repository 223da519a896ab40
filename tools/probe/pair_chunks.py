"""Stencil pair (forward + weight gradient over 64 images) in one piece vs in chunks of images --
forward(chunk) then weight gradient(chunk) -- so that the chunk's x is still in the 256 MB
Infinity Cache when the weight gradient reads it.  HIP-graph replays of K steps (no launch gaps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
from weather_model_amd import ops
L = _lib.lib()
N, H, W, K = 64, 721, 1440, 12
nset = 3
w = ops.synth_uniform((1, 1, 3, 3), seed=3, lo=-1 / 3, hi=1 / 3)
b = ops.synth_uniform((1,), seed=5, lo=-1 / 3, hi=1 / 3)
xs = [ops.synth_uniform((N, 1, H, W), seed=2, offset=k * N * H * W) for k in range(nset)]
gys = [ops.synth_uniform((N, 1, H - 2, W - 2), seed=4, lo=-1.0, hi=1.0, offset=k * N * H * W) for k in range(nset)]
ys = [torch.empty((N, 1, H - 2, W - 2), device="cuda") for _ in range(nset)]
grads = torch.zeros((64, 10), device="cuda")


def make(chunks):
    n = N // chunks
    wss = [torch.empty((L.wm_conv2d_bwd_wb_workspace_bytes(n, 1, H, W, 1, 3, 3),), device="cuda", dtype=torch.uint8)
           for _ in range(chunks)]

    def step(k, st):
        for c in range(chunks):
            x, y, gy = xs[k][c * n:], ys[k][c * n:], gys[k][c * n:]
            _lib.check(L.wm_conv2d_fwd_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), n, 1, H, W, 1, 3, 3, 0, st))
            _lib.check(L.wm_conv2d_bwd_wb_f32(x.data_ptr(), gy.data_ptr(), grads[c].data_ptr(), grads[c][9:].data_ptr(),
                                              wss[c].data_ptr(), wss[c].numel(), n, 1, H, W, 1, 3, 3, st))
    st = torch.cuda.current_stream().cuda_stream
    step(0, st); torch.cuda.synchronize()         # warm the occupancy caches before capture
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            st = torch.cuda.current_stream().cuda_stream
            for i in range(K):
                step(i % nset, st)
    torch.cuda.synchronize()
    return g, wss


def timed(g, reps=5):
    out = []
    g.replay(); torch.cuda.synchronize()
    for _ in range(reps):
        time.sleep(0.05)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record(); g.replay(); e[1].record(); torch.cuda.synchronize()
        out.append(e[0].elapsed_time(e[1]) / K * 1e3)
    return out


graphs = {c: make(c) for c in (1, 2, 4, 8)}
for rnd in range(2):
    for c, (g, _) in graphs.items():
        t = timed(g)
        print("chunks %d: us/step %s  median %.1f -> %.3f of 8 TB/s at 16 B/pt" % (
            c, ["%.1f" % v for v in t], sorted(t)[len(t) // 2], 16.0 * N * H * W / (sorted(t)[len(t) // 2] * 1e-6) / 8e12), flush=True)
