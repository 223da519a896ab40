"""wm_gather_cubes_f32 timings: (a) the regression generators' region walk (cube 1, 8 variables, 16
rows per batch: every element a 4-byte gather from its own (file, variable) plane), (b) random
20^3 cubes of one variable (convolution.py's generator: runs of 20 floats)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
from weather_model_amd import ops

def timed(fn, K=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    res = []
    for r in range(5):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(K): fn()
        e[1].record(); torch.cuda.synchronize()
        res.append(e[0].elapsed_time(e[1]) / K * 1e3)
    return float(np.median(res))

files, nv, Z, Y, X = 70, 8, 2, 721, 1440
vol = ops.synth_uniform((files, nv, Z, Y, X), seed=3)
bs, length = 16, 4
count = 65536
walk = {"batch": bs, "length": length, "index": 0, "region": (0, 0, 0), "dims": (Z, Y, X)}
us = timed(lambda: ops.gather_cubes(vol, (1, 1, 1), 3, 6, count * bs, walk=walk))
rows = count * bs
print("walk  : %d rows x p=%d  %.1f us  %.2f G rows/s  %.0f GB/s algorithmic (24 B x p per row)" % (rows, nv, us, rows / us / 1e3, rows * nv * 24 / us / 1e3))
pm = vol.permute(2, 3, 4, 0, 1).contiguous().permute(3, 4, 0, 1, 2)       # stored (Z, Y, X, files, vars)
ref = ops.gather_cubes(vol, (1, 1, 1), 3, 6, count * bs, walk=walk)
got = ops.gather_cubes(pm, (1, 1, 1), 3, 6, count * bs, walk=walk)
assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
us = timed(lambda: ops.gather_cubes(pm, (1, 1, 1), 3, 6, count * bs, walk=walk))
print("walk, point-major volume: %.1f us  %.2f G rows/s  %.0f GB/s algorithmic" % (us, rows / us / 1e3, rows * nv * 24 / us / 1e3))
del pm, ref, got
del vol
files, nv, Z, Y, X = 12, 1, 40, 721, 1440
vol = ops.synth_uniform((files, nv, Z, Y, X), seed=4)
rng = np.random.default_rng(1)
rows = 2048
org = np.stack([rng.integers(0, files - 4, rows), rng.integers(0, Z - 19, rows), rng.integers(0, Y - 19, rows), rng.integers(0, X - 19, rows)], axis=1).astype(np.int32)
orgd = torch.from_numpy(org).cuda()
us = timed(lambda: ops.gather_cubes(vol, (20, 20, 20), 2, 4, rows, origins=orgd, separate=True))
print("cubes : %d rows x p=8000  %.1f us  %.0f GB/s algorithmic" % (rows, us, rows * 8000 * 24 / us / 1e3))
