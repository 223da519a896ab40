"""wm_gather_cubes_f32 on singleRegression's own configuration (6^3 cubes of one variable, BATCH_SIZE 4,
x2 at +2 files, y at +1: singleRegression.py:22,33) beside the Gram and metrics passes it feeds."""
import time
import torch
from weather_model_amd import ops


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


files, Z, Y, X = 2000, 12, 40, 40
vol = torch.rand((files, 1, Z, Y, X), device="cuda")
bs = 4
length = (files - 2) // bs - 1
walk = {"batch": bs, "length": length, "index": 0, "region": (0, 0, 0), "dims": (Z - 5, Y - 5, X - 5)}
rows = 1_000_000
tg = timed(lambda: ops.gather_cubes(vol, (6, 6, 6), 1, 2, rows, walk=walk))
Xr, Yr = ops.gather_cubes(vol, (6, 6, 6), 1, 2, rows, walk=walk)
acc = torch.zeros((649, 649), dtype=torch.float64, device="cuda")
tgram = timed(lambda: ops.gram(Xr, Yr, acc))
gb = rows * 216 * 24 / 1e9
print(f"rows={rows} p=216: gather {tg:.3f} ms ({gb / tg * 1e3:.0f} GB/s of 24 B per element), gram {tgram:.3f} ms")
