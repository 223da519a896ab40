"""gram_f16 kernel time on the 1e8 resident rows of configs[1] (events around 10 launches) and a
check of the accumulated Gram against torch float64 on the first 2e6 rows; WB_LIB selects the library."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
from weather_model_amd import ops
n, F = 100_000_000, 16
X = ops.synth_uniform((n, F), seed=1); Y = ops.synth_uniform((n, 1), seed=2)
m = 2_000_003
small = ops.GramPlan(m, F, 1, "cuda")
small.accumulate(X[:m], Y[:m])
Z = torch.cat([X[:m].double(), torch.ones((m, 1), device="cuda", dtype=torch.float64), Y[:m].double()], 1)
ref = Z.T @ Z
err = ((small.acc - ref).abs() / ref.abs()).max().item()
assert err < 1e-7, err   # float32 products, float32 sums of up to 1024 rows
del Z
plan = ops.GramPlan(n, F, 1, "cuda")
for _ in range(2): plan.accumulate(X, Y)
torch.cuda.synchronize()
res = []
for r in range(5):
    time.sleep(0.05)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(10): plan.accumulate(X, Y)
    e[1].record(); torch.cuda.synchronize()
    res.append(e[0].elapsed_time(e[1]) / 10)
print("%-18s gram_f16 + finalize ms: median %.4f best %.4f -> %.2f TB/s  (Gram err %.1e)" % (os.environ.get("WB_LIB", "libwm.so"), np.median(res), min(res), n * 17 * 4 / np.median(res) / 1e9, err))
