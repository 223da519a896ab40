"""Gram kernels on the reference's own regression shapes: F=2,P=1 (multipleRegression) and
F=432,P=216 (singleRegression), plus the solve.  WB_LIB selects a variant library."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from weather_model_amd import _lib
if os.environ.get("WB_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "probe", "variants", os.environ["WB_LIB"])
from weather_model_amd import ops

def timeit(fn, reps=10, rounds=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    res = []
    for r in range(rounds):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(reps): fn()
        e[1].record(); torch.cuda.synchronize()
        res.append(e[0].elapsed_time(e[1]) / reps)
    return np.median(res), min(res)

shapes = [(2, 1, 400_000_000), (1, 1, 400_000_000), (4, 2, 200_000_000), (8, 4, 100_000_000), (10, 5, 80_000_000),
          (432, 216, 1_000_000), (64, 8, 8_000_000), (128, 64, 4_000_000)]
if len(sys.argv) > 1 and ":" in sys.argv[1]:           # F:P:n triples
    shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]]
elif len(sys.argv) > 1:
    shapes = [s for s in shapes if str(s[0]) in sys.argv[1:]]
for F, P, n in shapes:
    X = ops.synth_uniform((n, F), seed=1); Y = ops.synth_uniform((n, P), seed=2)
    plan = ops.GramPlan(n, F, P, "cuda")
    med, best = timeit(lambda: plan.accumulate(X, Y), reps=5 if F > 100 else 10)
    byts = 4.0 * (F + P) * n
    flops = 2.0 * F * (F + P) * n
    print("F=%d P=%d n=%d: gram median %.3f ms best %.3f -> %.2f TB/s, %.1f TFLOP/s (2F(F+P)), %.2f G rows/s" % (
        F, P, n, med, best, byts / med / 1e9, flops / med / 1e9, n / med / 1e6), flush=True)
    plan.acc.zero_(); plan.accumulate(X, Y)
    smed, sbest = timeit(lambda: plan.solve(), reps=3, rounds=3)
    print("    solve median %.3f ms best %.3f; status %d" % (smed, sbest, int(plan.status.item())), flush=True)
    del X, Y, plan
    torch.cuda.empty_cache()
