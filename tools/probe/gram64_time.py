"""gram_x64 kernel time on one resident 5e7-row chunk (events around 10 launches), WB_LIB selects the library."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
from weather_model_amd import ops
n, F = 50_000_000, 64
X = ops.synth_uniform((n, F), seed=1); Y = ops.synth_uniform((n, 1), seed=2)
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
print("%-16s gram_x64 + finalize ms/chunk: median %.4f best %.4f -> %.2f TB/s" % (os.environ.get("WB_LIB", "libwm.so"), np.median(res), min(res), n * 65 * 4 / np.median(res) / 1e9))
